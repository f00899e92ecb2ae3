"""The host-side parsers under AddressSanitizer + UndefinedBehaviorSanitizer (make host-asan; SURVEY.md section 5 "sanitizers";
CPU only — no GPU sanitizer exists on this pool): (1) the host test files run against the instrumented library, (2) a seeded
corpus of malformed asset directories and strings goes through every entry point that reads caller-supplied bytes and must end in a
result or an STN_ERR_* code, never in a sanitizer report.  The reference delegates this to ONNX Runtime and nlohmann/json
(/root/reference/cpp/helper.cpp:784-823, 829-897, 1054-1064) and has no sanitizer build (cpp/CMakeLists.txt:13-14)."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_LIB = os.path.join(ROOT, "build_asan", "libstn_host_asan.so")
FUZZ = os.path.join(ROOT, "build_asan", "host_fuzz")


@pytest.fixture(scope="module")
def asan_build():
    p = subprocess.run(["make", "-C", ROOT, "-s", "host-asan"], capture_output=True, text=True, timeout=900)
    if p.returncode != 0 and ("libasan" in p.stderr or "libubsan" in p.stderr or "sanitize" in p.stderr):
        pytest.skip("this toolchain has no sanitizer runtime: " + p.stderr[-300:])
    assert p.returncode == 0, p.stderr[-2000:]
    # python links neither runtime: libasan first, and libstdc++ with it (the interceptor of __cxa_throw resolves the real one at
    # start-up; without libstdc++ in the process by then, the first C++ exception inside the library aborts)
    rt = " ".join(subprocess.run(["g++", f"-print-file-name={n}"], capture_output=True, text=True).stdout.strip() for n in ("libasan.so", "libstdc++.so"))
    assert os.path.exists(ASAN_LIB) and os.path.exists(FUZZ)
    return rt


def _no_report(p):
    txt = p.stdout + p.stderr
    assert "AddressSanitizer" not in txt and "runtime error:" not in txt and "LeakSanitizer" not in txt, txt[-3000:]


def test_host_test_files_pass_on_the_sanitizer_build(asan_build):
    env = dict(os.environ, STN_LIB=ASAN_LIB, STN_HOST_ONLY="1", LD_PRELOAD=asan_build, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1")
    files = ["tests/test_host_cpp.py", "tests/test_graph_bind_cpu.py", "tests/test_onnx_reader_cpu.py", "tests/test_voice_style_cpu.py"]
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + files, capture_output=True, text=True, cwd=ROOT, env=env,
                       timeout=1500)
    _no_report(p)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-2000:])
    assert " passed" in p.stdout


# ---- the corpus -----------------------------------------------------------------------------------------------------------------
def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | 0x80 if v else b)
        if not v:
            return bytes(out)


def _mutations(blob, rng, n):
    """n seeded variants of one file: truncations, byte flips, oversized varints, length fields that lie, huge dims, nesting."""
    out = []
    L = len(blob)
    for k in range(n):
        kind = k % 7
        b = bytearray(blob)
        if kind == 0:  # truncation
            b = b[: int(rng.integers(0, L))]
        elif kind == 1:  # a few byte flips
            for _ in range(int(rng.integers(1, 8))):
                b[int(rng.integers(0, L))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 2:  # an over-long varint (11+ continuation bytes) in the middle of the stream
            pos = int(rng.integers(0, L))
            b[pos:pos] = b"\xff" * int(rng.integers(10, 20)) + b"\x01"
        elif kind == 3:  # a length-delimited field whose length runs past the end of the file / wraps 64 bits
            pos = int(rng.integers(0, L))
            b[pos:pos] = bytes([0x3A]) + _varint(int(rng.choice([L * 4, 2 ** 31, 2 ** 63 - 1, 2 ** 64 - 1])))
        elif kind == 4:  # a tensor whose dims promise far more than its payload holds (dims = field 1 varint, raw_data = field 9)
            t = b"".join(bytes([0x08]) + _varint(int(d)) for d in (int(rng.choice([2 ** 31, 2 ** 40, 65536])), 65536)) + bytes([0x10, 0x01]) + \
                bytes([0x42, 0x01, 0x77]) + bytes([0x4A, 0x04]) + b"\0\0\0\0"
            g = bytes([0x2A]) + _varint(len(t)) + t  # GraphProto.initializer
            b += bytes([0x3A]) + _varint(len(g)) + g  # ModelProto.graph (a second one: concatenates)
        elif kind == 5:  # an external-data tensor (data_location = EXTERNAL, external_data entries) with a path outside the directory
            kv = bytes([0x0A, 0x08]) + b"location" + bytes([0x12, 0x0B]) + b"/etc/passwd"
            t = bytes([0x08, 0x04, 0x10, 0x01, 0x42, 0x01, 0x65]) + bytes([0x6A]) + _varint(len(kv)) + kv + bytes([0x70, 0x01])
            g = bytes([0x2A]) + _varint(len(t)) + t
            b += bytes([0x3A]) + _varint(len(g)) + g
        else:  # deep nesting: a graph inside a graph inside ... (attribute g = field 6 of AttributeProto is not followed; nodes = field 1)
            inner = b"\x08\x01"
            for _ in range(int(rng.integers(50, 400))):
                inner = bytes([0x0A]) + _varint(len(inner)) + inner
            b += bytes([0x3A]) + _varint(len(inner)) + inner
        out.append(bytes(b))
    return out


def _json_mutations(text, rng, n):
    out = []
    for k in range(n):
        kind = k % 6
        if kind == 0:
            out.append(text[: int(rng.integers(0, len(text)))])
        elif kind == 1:
            out.append("[" * int(rng.integers(1000, 20000)))  # nesting without end
        elif kind == 2:
            out.append("{\"a\":" * 3000 + "1" + "}" * 3000)
        elif kind == 3:
            out.append(text.replace("1", "1e999999", 3).replace("[", "[[", 1))
        elif kind == 4:
            out.append(text.replace("\"dims\"", "\"dims\": [1, 100000, 100000], \"x\"", 1))  # dims that promise 10^10 floats
        else:
            b = bytearray(text.encode())
            for _ in range(8):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            out.append(b.decode("latin-1"))
    return out


def test_malformed_inputs_end_in_error_codes_not_in_sanitizer_reports(asan_build, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.neural_ref import RefModel
    from supertonic_amd.arch import tiny_arch
    from onnx_graphs import build_graph_dir
    a = tiny_arch()
    good = tmp_path / "good"
    good.mkdir()
    build_graph_dir(good, a, RefModel(a, 7).tensor)
    style = {"style_ttl": {"data": np.zeros((1, a.n_style_ttl, a.d_style_ttl)).tolist(), "dims": [1, a.n_style_ttl, a.d_style_ttl]},
             "style_dp": {"data": np.zeros((1, a.n_style_dp, a.d_style_dp)).tolist(), "dims": [1, a.n_style_dp, a.d_style_dp]}}
    (good / "F1.json").write_text(json.dumps(style))
    (good / "texts.txt").write_bytes("Hello, world.  Dr. Smith paid $5 @ 3 p.m.!\n한국어 문장입니다... 두번째 문장?\n".encode() + b"\xff\xfe\xed\xa0\x80 broken utf-8 \xf4\x90\x80\x80\n" +
                                     ("a" * 5000 + "\n").encode() + ("." * 700 + "\n").encode() + "é".encode() * 999 + b"\n")
    rng = np.random.default_rng(20261005)
    dirs = [str(good)]
    files = sorted(os.listdir(good))
    n_cases = 0
    for f in files:
        if f.endswith(".txt"):
            continue
        blob = (good / f).read_bytes()
        muts = _mutations(blob, rng, 28) if f.endswith(".onnx") else [m.encode("latin-1", "replace") if isinstance(m, str) else m for m in _json_mutations(blob.decode(), rng, 12)]
        for k, m in enumerate(muts):
            d = tmp_path / f"case_{f}_{k}"
            shutil.copytree(good, d)
            (d / f).write_bytes(m)
            dirs.append(str(d))
            n_cases += 1
    # text-only cases: random bytes as text lines
    d = tmp_path / "texts_only"
    d.mkdir()
    (d / "t.txt").write_bytes(b"\n".join(bytes(rng.integers(1, 256, int(rng.integers(1, 400))).astype(np.uint8)).replace(b"\n", b" ") for _ in range(200)))
    dirs.append(str(d))
    assert n_cases >= 140
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([FUZZ] + dirs, capture_output=True, text=True, timeout=1500, env=env)
    _no_report(p)
    assert p.returncode == 0, (p.returncode, p.stderr[-3000:])
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    assert rec["dirs"] == len(dirs) and rec["errors_returned"] > n_cases // 2, rec  # most mutants are refused; none crashed
    print("host fuzz:", rec)
