"""Numerical claims made in kernel comments, pinned on the CPU (numpy restatements of the device formulas)."""
import numpy as np
from scipy.special import erf


def gelu_exact(x):
    return 0.5 * x * (1 + erf(x / np.sqrt(2)))


def test_gelu_erf_approximation_fp32_path():
    """kernels_gemm.hip gelu_f: erf by Abramowitz-Stegun 7.1.26 -> |GELU error| <= 4e-7 * max(1,|x|)."""
    x = np.linspace(-8, 8, 200001).astype(np.float32)
    z = np.abs(x) * np.float32(0.70710678118654752440)
    t = 1 / (1 + np.float32(0.3275911) * z)
    poly = t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429))))
    erf_v = np.sign(x) * (1 - poly * np.exp(-z * z))
    got = 0.5 * x * (1 + erf_v)
    assert np.abs(got - gelu_exact(x.astype(np.float64))).max() < 1e-6


def test_gelu_bf16_output_form():
    """kernels_gemm.hip gelu_bf16_f (used only when the result is rounded to bf16):
    x * sigmoid(1.5957691 x (1 + 0.044715 x^2)); |error| <= 5e-4 absolute and below half a bf16 ulp for |y| > 0.2."""
    x = np.linspace(-8, 8, 200001)
    t = x * (x * x * -0.10294324 - 2.30220819)
    got = x / (1 + np.exp2(t))
    ref = gelu_exact(x)
    err = np.abs(got - ref)
    assert err.max() < 5e-4
    big = np.abs(ref) > 0.2
    assert np.all(err[big] < np.abs(ref[big]) * 2.0 ** -9)
