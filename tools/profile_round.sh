#!/bin/bash
# Reproduce the committed rocprofv3 evidence for one round on a GPU box:
#   tools/profile_round.sh <tag>            e.g.  tools/profile_round.sh r02_a
# Four runs of the default bench command, each ending with one fully tagged step whose launch sequence bench.py writes out
# (STN_LAUNCH_LOG): 1) kernel trace + stats, 2) FETCH_SIZE, 3) WRITE_SIZE, 4) MFMA busy / GUI active — counters in their own
# --pmc passes with no trace domain beside them.  tools/pmc_families.py then attributes every dispatch of that step to its
# kernel family by position and writes profiles/pmc_traffic.json, profiles/mfma_util.json, profiles/<tag>_families.csv.
# Raw outputs stay in gpurun_out/<tag>/ (scratch); the summaries go to profiles/ (committed).
set -eo pipefail
TAG=${1:-r02_a}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT" "$ROOT/profiles"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
export STN_LAUNCH_LOG="$OUT/log_trace.json"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- python3 "$B" --steps 10 --warmup 3 --cpu-sample 0 --no-host-loop --no-b1 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "trace pass done" >&2
export STN_LAUNCH_LOG="$OUT/log_fetch.json"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT" -o pmc_fetch -- python3 "$B" --steps 1 --warmup 1 --cpu-sample 0 --no-profile --no-host-loop --no-b1 > /dev/null 2> "$OUT/fetch.err"
echo "fetch pass done" >&2
export STN_LAUNCH_LOG="$OUT/log_write.json"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT" -o pmc_write -- python3 "$B" --steps 1 --warmup 1 --cpu-sample 0 --no-profile --no-host-loop --no-b1 > /dev/null 2> "$OUT/write.err"
echo "write pass done" >&2
export STN_LAUNCH_LOG="$OUT/log_mfma.json"
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT" -o pmc_mfma -- python3 "$B" --steps 1 --warmup 1 --cpu-sample 0 --no-profile --no-host-loop --no-b1 > /dev/null 2> "$OUT/mfma.err"
echo "mfma pass done" >&2
unset STN_LAUNCH_LOG
cd "$ROOT"
S=$(find "$OUT" -name 'trace_kernel_stats.csv' | head -1)
cp "$S" "profiles/${TAG}_bench_kernel_stats.csv"
grep '^{' "$OUT/bench.json" | tail -1 > "profiles/${TAG}_bench.json"
python3 tools/pmc_families.py --dir "$OUT" --tag "$TAG" > "$OUT/families.log"
# the summaries travel back with gpurun_out/ (profiles/ on the box is not merged): copy them beside the raw output
mkdir -p "$OUT/profiles" && cp profiles/pmc_traffic.json profiles/mfma_util.json "profiles/${TAG}_families.csv" "profiles/${TAG}_bench_kernel_stats.csv" "profiles/${TAG}_bench.json" "$OUT/profiles/"
echo "summaries in $OUT/profiles" >&2
