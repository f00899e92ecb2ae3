#!/usr/bin/env python3
"""MFMA utilisation per kernel family from one rocprofv3 --pmc pass of bench.py:

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <dir> -o pmc_mfma -- python3 bench.py ...
  tools/pmc_mfma.py <dir>/.../pmc_mfma_counter_collection.csv [--trace <kernel_trace.csv>] > profiles/<tag>_mfma_util.json

Units (/opt/skills/guides/MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over the chip's
1024 SIMDs (32 per v_mfma_f32_32x32x16); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so GUI/8 is the dispatch's length in
shader cycles (it reads high on dispatches shorter than ~0.3 ms: utilisation of the short GEMMs is a LOWER bound).
  mfma_util = MFMA_BUSY / (1024 * GUI / 8);   with --trace: clock_ghz = (GUI / 8) / duration."""
import argparse, collections, csv, json, re

ap = argparse.ArgumentParser(); ap.add_argument("csv"); ap.add_argument("--trace"); a = ap.parse_args()
FAM = [("ve.gemm_pw1_gelu", "gemm_tiled_kernel<0, 192, 256, 3, 4, 4, 32, 2", 179712), ("ve.gemm_pw2_resid", "gemm_tiled_kernel<1, 128, 128, 2, 4, 4, 64, 2", 90624),
       ("vo.gemm_pw1_gelu", "gemm_tiled_kernel<0, 256, 128, 4, 2, 3, 32, 2", 1916928), ("vo.gemm_pw2_resid", "gemm_tiled_kernel<1, 256, 256, 4, 4, 4, 32, 2", None),
       ("ve.attention", "attn_mfma_kernel<96", None), ("te.attention", "attn_mfma_kernel<64", None)]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(a.csv)):
    for fam, sub, grid in FAM:
        if sub in r["Kernel_Name"] and (grid is None or int(r["Grid_Size"]) == grid):
            vals[fam][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
if a.trace:
    for r in csv.DictReader(open(a.trace)):
        for fam, sub, grid in FAM:
            if sub in r["Kernel_Name"] and (grid is None or int(r["Grid_Size_X"]) == grid):
                dur[fam].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
out = {}
for fam, _, _ in FAM:
    v = vals.get(fam)
    if not v or "SQ_VALU_MFMA_BUSY_CYCLES" not in v or "GRBM_GUI_ACTIVE" not in v:
        continue
    busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(v["SQ_VALU_MFMA_BUSY_CYCLES"])
    gui = sum(v["GRBM_GUI_ACTIVE"]) / len(v["GRBM_GUI_ACTIVE"])
    o = {"launches": len(v["GRBM_GUI_ACTIVE"]), "mfma_busy_cycles_per_launch": busy, "gui_active_per_launch": gui, "dispatch_cycles": gui / 8,
         "mfma_util": busy / (1024 * gui / 8)}
    if dur.get(fam):
        d = sum(dur[fam]) / len(dur[fam])
        o["trace_avg_us"] = d * 1e6
        o["clock_ghz_from_gui"] = gui / 8 / d / 1e9
        o["mfma_util_vs_2p4ghz_wall"] = busy / (1024 * 2.4e9 * d)
    out[fam] = o
print(json.dumps(out, indent=1))
