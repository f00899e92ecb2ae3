"""Measure the engine-vs-oracle error of every named parity case on an MI355X and derive the test bounds from it.

  python tools/parity_record.py [--out-dir gpurun_out/parity]

Runs the oracle-parity tests (tests/test_gpu_stages.py, tests/test_gpu_configs.py, tests/test_gpu_load_dir.py, tests/test_gpu_ffn.py) once with STN_PARITY_RECORD set — the tests'
own inputs, engines and comparison code, so what is recorded is exactly what the tests assert on — and writes
  parity_r04.json     {"measured": {case: {dtype: {"max", "rms", "kind", "n"}}}, "_source_sha": ...}   -> commit as profiles/parity_r04.json
  parity_bounds.json  {"bounds": {case: {dtype: {"max", "rms"}}}}: 2 x measured, rounded DOWN to two digits, not below
                      FLOOR (fp32 summation-order noise)                                                      -> tests/golden/
(max, rms) are relative to the rms of the oracle's output (tests/gpu_util.rel_err).  tests/test_parity_bounds_cpu.py keeps the
two files consistent: no bound looser than 2 x its measurement."""
import argparse
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.src_hash import source_sha  # noqa: E402


FLOOR = {"max": 2e-6, "rms": 5e-7}


def floor2(x):
    """x rounded down to two significant digits (so that 2 x measured stays an upper limit of the bound)."""
    if x <= 0:
        return 0.0
    e = math.floor(math.log10(x)) - 1
    return math.floor(x / 10 ** e) * 10 ** e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out-dir", default=os.path.join(ROOT, "gpurun_out", "parity"))
    args = ap.parse_args()
    os.makedirs(args.out_dir, exist_ok=True)
    log = os.path.join(args.out_dir, "records.jsonl")
    if os.path.exists(log):
        os.remove(log)
    env = dict(os.environ, STN_PARITY_RECORD=log, STN_PARITY_RECORD_ONLY="1")
    rc = subprocess.call([sys.executable, "-m", "pytest", "tests/test_gpu_stages.py", "tests/test_gpu_configs.py", "tests/test_gpu_load_dir.py", "tests/test_gpu_ffn.py", "-q", "-m", "gpu", "-x"],
                         cwd=ROOT, env=env)
    if rc != 0:
        sys.exit(f"parity tests failed (rc={rc}): nothing recorded")
    measured = {}
    with open(log) as f:
        for line in f:
            r = json.loads(line)
            m = measured.setdefault(r["case"], {}).setdefault(r["dtype"], {"max": 0.0, "rms": 0.0, "kind": r["kind"], "n": r["n"]})
            m["max"], m["rms"] = max(m["max"], r["max"]), max(m["rms"], r["rms"])
    # an error of a few fp32 ulps is summation-order noise: its bound does not go below FLOOR (16 / 4 ulps of the output's rms)
    bounds = {c: {d: {"max": max(floor2(2 * v["max"]), FLOOR["max"]), "rms": max(floor2(2 * v["rms"]), FLOOR["rms"])} for d, v in per.items()}
              for c, per in measured.items()}
    note = ("relative to the rms of the oracle output (tests/gpu_util.rel_err); tiny = tests' small descriptor, c1/c3/c4/c5 = the 66 M stack on "
            "BASELINE.json's configs; engine through the C ABI vs oracle/stn_ref.c on identical synthetic weights and inputs")
    with open(os.path.join(args.out_dir, "parity_r04.json"), "w") as f:
        json.dump({"_source_sha": source_sha(), "note": note, "measured": measured}, f, indent=1, sort_keys=True)
    with open(os.path.join(args.out_dir, "parity_bounds.json"), "w") as f:
        json.dump({"source": "profiles/parity_r04.json (tools/parity_record.py): bound = max(2 x measured rounded down to two digits, floor)",
                   "floor": FLOOR, "bounds": bounds},
                  f, indent=1, sort_keys=True)
    for c in sorted(measured):
        for d, v in sorted(measured[c].items()):
            print(f"{c:36s} {d:5s} max {v['max']:.3e} rms {v['rms']:.3e}   bound max {bounds[c][d]['max']:.2g} rms {bounds[c][d]['rms']:.2g}")


if __name__ == "__main__":
    main()
