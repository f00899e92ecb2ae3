#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes of `bench.py` into profiles/pmc_traffic.json (HBM bytes per launch of one kernel family).

  tools/pmc_summary.py --fetch <..counter_collection.csv> --write <..counter_collection.csv> \
      --kernel-substr 'gemm_bf16_tiled_kernel<0, 256, 256' --grid 119808 --family ve.gemm_pw1_gelu

Corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly half the bytes of a wide (16 B/lane) coalesced streaming read (buffer_load ... lds included) -> doubled; WRITE_SIZE is
exact for 16-B-per-lane stores."""
import argparse, csv, json, os

ap = argparse.ArgumentParser()
ap.add_argument("--fetch"); ap.add_argument("--write"); ap.add_argument("--kernel-substr"); ap.add_argument("--grid", type=int)
ap.add_argument("--family"); ap.add_argument("--trace"); ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "profiles", "pmc_traffic.json"))
a = ap.parse_args()


def mean_counter(path, name):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == name and a.kernel_substr in r["Kernel_Name"] and (a.grid is None or int(r["Grid_Size"]) == a.grid)]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


fetch, nf = mean_counter(a.fetch, "FETCH_SIZE")
if fetch is None:
    raise SystemExit("no FETCH_SIZE rows match (check --kernel-substr / --grid)")
write, nw = mean_counter(a.write, "WRITE_SIZE")
out = {}
if os.path.exists(a.out):
    out = json.load(open(a.out))
out[a.family] = {"hbm_bytes_per_launch": (2 * fetch + write) * 1024.0, "fetch_size_kib_raw": fetch, "write_size_kib": write,
                 "fetch_correction": "x2 (gfx950 wide-read under-count)", "launches_fetch_pass": nf, "launches_write_pass": nw,
                 "kernel": a.kernel_substr, "grid": a.grid}
if a.trace:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(a.trace))
         if a.kernel_substr in r["Kernel_Name"] and (a.grid is None or int(r["Grid_Size_X"]) == a.grid)]
    out[a.family]["rocprof_avg_us"] = sum(d) / len(d)
    out[a.family]["rocprof_launches"] = len(d)
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out[a.family], indent=1))
