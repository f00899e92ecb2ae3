#!/bin/bash
# Reproduce the committed rocprofv3 evidence for one round on a GPU box:
#   tools/profile_round.sh <tag>            e.g.  tools/profile_round.sh r01_e
# 1) kernel trace + stats of the default bench command, 2) FETCH_SIZE pass, 3) WRITE_SIZE pass (separate --pmc runs, no
# trace domains beside them), then the per-family HBM traffic of the dominant kernels into profiles/pmc_traffic.json
# (tools/pmc_summary.py applies the guide's gfx950 corrections).  Outputs: gpurun_out/<tag>/..., summaries in profiles/.
set -eo pipefail
TAG=${1:-r01_e}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT" "$ROOT/profiles"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- python3 "$B" --steps 10 --warmup 3 --cpu-sample 0 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "trace pass done" >&2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT" -o pmc_fetch -- python3 "$B" --steps 2 --warmup 1 --cpu-sample 0 --no-profile > /dev/null 2> "$OUT/fetch.err"
echo "fetch pass done" >&2
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT" -o pmc_write -- python3 "$B" --steps 2 --warmup 1 --cpu-sample 0 --no-profile > /dev/null 2> "$OUT/write.err"
echo "write pass done" >&2
cd "$ROOT"
F=$(find "$OUT" -name 'pmc_fetch_counter_collection.csv' | head -1)
W=$(find "$OUT" -name 'pmc_write_counter_collection.csv' | head -1)
T=$(find "$OUT" -name 'trace_kernel_trace.csv' | head -1)
S=$(find "$OUT" -name 'trace_kernel_stats.csv' | head -1)
cp "$S" "profiles/${TAG}_bench_kernel_stats.csv"
grep '^{' "$OUT/bench.json" | tail -1 > "profiles/${TAG}_bench.json"
# dominant families of config C3 (kernel template, grid = workgroups x threads)
python3 tools/pmc_summary.py --fetch "$F" --write "$W" --trace "$T" --family ve.gemm_pw1_gelu --kernel-substr 'gemm_tiled_kernel<0, 192, 256, 3, 4, 4, 32, 2, false>' --grid 179712 || echo "  (no rows matched for this family)"
python3 tools/pmc_summary.py --fetch "$F" --write "$W" --trace "$T" --family ve.gemm_pw2_resid --kernel-substr 'gemm_tiled_kernel<1, 128, 128, 2, 4, 4, 64, 2, false>' --grid 90624 || echo "  (no rows matched for this family)"
python3 tools/pmc_summary.py --fetch "$F" --write "$W" --trace "$T" --family vo.gemm_pw1_gelu --kernel-substr 'gemm_tiled_kernel<0, 256, 128, 4, 2, 3, 32, 2, false>' --grid 1916928 || echo "  (no rows matched for this family)"
python3 tools/pmc_summary.py --fetch "$F" --write "$W" --trace "$T" --family vo.dwconv_ln --kernel-substr 'dwconv_ln_v3_kernel<unsigned short, 7, 4>' || echo "  (no rows matched for this family)"
python3 tools/pmc_summary.py --fetch "$F" --write "$W" --trace "$T" --family ve.dwconv_ln --kernel-substr 'dwconv_ln_v3_kernel<unsigned short, 5, 2>' || echo "  (no rows matched for this family)"
# 4) matrix-pipe utilisation pass (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE); summarise on the host with tools/pmc_mfma.py
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT" -o pmc_mfma -- python3 "$B" --steps 2 --warmup 1 --cpu-sample 0 --no-profile > /dev/null 2> "$OUT/mfma.err"
echo "mfma pass done" >&2
