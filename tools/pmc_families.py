#!/usr/bin/env python3
"""Per-family kernel time, HBM traffic and matrix-pipe utilisation from rocprofv3 runs of `bench.py`, attributed by POSITION:

bench.py (with STN_LAUNCH_LOG=<file>) ends every profiler run with one fully tagged step of the bench batch and writes that
step's launch sequence [(family, kernel), ...].  The same step is the tail of rocprofv3's per-dispatch rows; the two sequences
are aligned from the end (kernels the runtime adds itself — copies, fills — are skipped by name), which gives every dispatch
its family without guessing from template arguments and grid sizes (two families can share both).

  tools/pmc_families.py --dir gpurun_out/<tag> --tag <tag>
expects in --dir:  trace_*kernel_trace.csv + log_trace.json, pmc_fetch_*counter_collection.csv + log_fetch.json,
                   pmc_write_* + log_write.json, pmc_mfma_* + log_mfma.json   (any pass may be missing)
writes profiles/pmc_traffic.json (per family: rocprof_avg_us, launches, FETCH_SIZE / WRITE_SIZE KiB raw, hbm_bytes_per_launch),
profiles/mfma_util.json and profiles/<tag>_families.csv; all carry `_source_sha` (tools/src_hash.py) so bench.py can tell
whether they belong to the sources it runs.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB; on
gfx950 FETCH_SIZE reports half the bytes of wide (16 B per lane) streaming reads (buffer_load ... lds included) -> doubled;
WRITE_SIZE is exact for 16-byte stores.  Counters come from separate --pmc passes (no trace domains beside them)."""
import argparse, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.src_hash import source_sha  # noqa: E402


def find(d, pat):
    m = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    return m[0] if m else None


def base(kname):
    """'void stn::v_x::gemm_tiled_kernel<1, 128>(...)' or 'stn::gemm_tiled_kernel' -> 'gemm_tiled_kernel'"""
    k = kname.replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].strip()
    return k.split(" ")[-1].split("::")[-1]


def align(dispatches, log):
    """dispatches: [(id, kernel_name)] in dispatch order; log: [(family, kernel)].  Returns {dispatch id: family} for the tail."""
    out, i = {}, len(dispatches) - 1
    for fam, kern in reversed(log):
        want = base(kern)
        while i >= 0 and base(dispatches[i][1]) != want:
            i -= 1  # a runtime kernel (copy / fill) or an earlier phase of the process
        if i < 0:
            raise SystemExit(f"launch log does not fit the dispatch rows (looking for {want})")
        out[dispatches[i][0]] = fam
        i -= 1
    return out


def load_pass(d, csv_pat, log_name, id_col, name_col):
    path, logp = find(d, csv_pat), os.path.join(d, log_name)
    if not path or not os.path.exists(logp):
        return None, None
    rows = list(csv.DictReader(open(path)))
    log = [tuple(e) for e in json.load(open(logp))["entries"]]
    seen, disp = set(), []
    for r in rows:
        k = int(r[id_col])
        if k not in seen:
            seen.add(k)
            disp.append((k, r[name_col]))
    disp.sort()
    return rows, align(disp, log)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", required=True)
    ap.add_argument("--tag", required=True)
    a = ap.parse_args()
    fam = {}

    def ent(f):
        return fam.setdefault(f, {})

    rows, m = load_pass(a.dir, "trace*kernel_trace.csv", "log_trace.json", "Dispatch_Id", "Kernel_Name")
    if rows:
        acc = {}
        for r in rows:
            f = m.get(int(r["Dispatch_Id"]))
            if f and f != "-":
                acc.setdefault(f, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for f, v in acc.items():
            ent(f).update(rocprof_avg_us=round(sum(v) / len(v), 3), rocprof_launches=len(v), rocprof_total_us=round(sum(v), 1))
    for pat, logn, cname, key in (("pmc_fetch*counter_collection.csv", "log_fetch.json", "FETCH_SIZE", "fetch_size_kib_raw"),
                                  ("pmc_write*counter_collection.csv", "log_write.json", "WRITE_SIZE", "write_size_kib")):
        rows, m = load_pass(a.dir, pat, logn, "Dispatch_Id", "Kernel_Name")
        if not rows:
            continue
        acc = {}
        for r in rows:
            f = m.get(int(r["Dispatch_Id"]))
            if f and f != "-" and r["Counter_Name"] == cname:
                acc.setdefault(f, []).append(float(r["Counter_Value"]))
        for f, v in acc.items():
            ent(f)[key] = round(sum(v) / len(v), 2)
            ent(f)[key + "_launches"] = len(v)
    for f, e in fam.items():
        if "fetch_size_kib_raw" in e and "write_size_kib" in e:
            e["hbm_bytes_per_launch"] = (2 * e["fetch_size_kib_raw"] + e["write_size_kib"]) * 1024.0
            e["fetch_correction"] = "x2 (gfx950 wide-read under-count)"
    sha = source_sha()
    traffic = {"_source_sha": sha, "_tag": a.tag}
    traffic.update({f: e for f, e in sorted(fam.items())})
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)

    rows, m = load_pass(a.dir, "pmc_mfma*counter_collection.csv", "log_mfma.json", "Dispatch_Id", "Kernel_Name")
    if rows:
        busy, act = {}, {}
        for r in rows:
            f = m.get(int(r["Dispatch_Id"]))
            if not f or f == "-":
                continue
            if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                busy.setdefault(f, []).append(float(r["Counter_Value"]))
            elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                act.setdefault(f, []).append(float(r["Counter_Value"]))
        mu = {"_source_sha": sha, "_tag": a.tag,
              "_note": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): busy is summed over the chip's 1024 SIMDs, "
                       "GUI over the 8 XCDs (MI355X_MICROARCH.md); GUI reads high on dispatches shorter than ~0.3 ms, so the figure is a lower bound "
                       "for the short GEMMs; averaged over the family's launches of one step"}
        for f in sorted(busy):
            if f in act and len(act[f]) == len(busy[f]):
                u = [b / (1024.0 * g / 8.0) for b, g in zip(busy[f], act[f]) if g > 0]
                mu[f] = {"mfma_util": round(sum(u) / len(u), 4), "launches": len(u), "mfma_busy_cycles_per_launch": round(sum(busy[f]) / len(busy[f]), 1),
                         "gui_active_per_launch": round(sum(act[f]) / len(act[f]), 1)}
        json.dump(mu, open(os.path.join(ROOT, "profiles", "mfma_util.json"), "w"), indent=1)
    # L2 hit rate per family (TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum), MI355X_MICROARCH.md): where a kernel's bytes come from — its
    # XCD's L2 or beyond it (Infinity Cache / HBM)
    rows, m = load_pass(a.dir, "pmc_tcc*counter_collection.csv", "log_tcc.json", "Dispatch_Id", "Kernel_Name")
    if rows:
        hit, miss = {}, {}
        for r in rows:
            f = m.get(int(r["Dispatch_Id"]))
            if not f or f == "-":
                continue
            if r["Counter_Name"] == "TCC_HIT_sum":
                hit.setdefault(f, []).append(float(r["Counter_Value"]))
            elif r["Counter_Name"] == "TCC_MISS_sum":
                miss.setdefault(f, []).append(float(r["Counter_Value"]))
        l2 = {"_source_sha": sha, "_tag": a.tag, "_note": "per launch, averaged over the family's launches of one fully tagged step; requests are 128-byte lines"}
        for f in sorted(hit):
            if f in miss and len(miss[f]) == len(hit[f]):
                h_, m_ = sum(hit[f]) / len(hit[f]), sum(miss[f]) / len(miss[f])
                l2[f] = {"l2_hit_rate": round(h_ / max(h_ + m_, 1.0), 4), "tcc_hit_per_launch": round(h_, 1), "tcc_miss_per_launch": round(m_, 1), "launches": len(hit[f])}
        json.dump(l2, open(os.path.join(ROOT, "profiles", "l2_hit.json"), "w"), indent=1)
    with open(os.path.join(ROOT, "profiles", f"{a.tag}_families.csv"), "w") as f:
        f.write("family,launches,avg_us,total_us,fetch_kib_raw,write_kib,hbm_bytes_per_launch\n")
        for k, e in sorted(fam.items(), key=lambda kv: -kv[1].get("rocprof_total_us", 0)):
            f.write(f"{k},{e.get('rocprof_launches','')},{e.get('rocprof_avg_us','')},{e.get('rocprof_total_us','')},{e.get('fetch_size_kib_raw','')},"
                    f"{e.get('write_size_kib','')},{e.get('hbm_bytes_per_launch','')}\n")
    print(json.dumps({k: v for k, v in traffic.items() if not k.startswith("_")}, indent=1)[:3000])


if __name__ == "__main__":
    main()
