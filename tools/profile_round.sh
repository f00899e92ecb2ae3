#!/bin/bash
# Reproduce the committed rocprofv3 evidence for one round on a GPU box:
#   tools/profile_round.sh <tag>            e.g.  tools/profile_round.sh r02_a
# Five runs of the default bench command, each ending with one fully tagged step whose launch sequence bench.py writes out
# (STN_LAUNCH_LOG): 1) kernel trace + stats, 2) FETCH_SIZE, 3) WRITE_SIZE, 4) MFMA busy / GUI active, 5) TCC_HIT / TCC_MISS (L2 hit rate per family) — counters in their own
# --pmc passes with no trace domain beside them.  tools/pmc_families.py then attributes every dispatch of that step to its
# kernel family by position and writes profiles/pmc_traffic.json, profiles/mfma_util.json, profiles/<tag>_families.csv.
# Raw outputs stay in gpurun_out/<tag>/ (scratch); the summaries go to profiles/ (committed).
# Every pass runs with --eager (rocprofv3's tracing of many hipGraph launches crashes inside the ROCm 7.2 runtime; the kernels and their
# durations are the same as in the replayed pipeline) and with --no-profile: the engine's own event timing (events on dispatch packets) under rocprofv3's interception crashes
# inside the ROCm 7.2 runtime, and a profiler run is timed by the profiler anyway; the launch log needs family tags only.
# A sixth, un-profiled run of the default command is the round's bench line (profiles/<tag>_default_bench.json).
set -eo pipefail
TAG=${1:-r02_a}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT" "$ROOT/profiles"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
export STN_LAUNCH_LOG="$OUT/log_trace.json"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- python3 "$B" --steps 10 --warmup 3 --cpu-sample 0 --no-profile --eager --no-host-loop --no-b1 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "trace pass done" >&2
export STN_LAUNCH_LOG="$OUT/log_fetch.json"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT" -o pmc_fetch -- python3 "$B" --steps 1 --warmup 1 --cpu-sample 0 --no-profile --eager --no-host-loop --no-b1 > /dev/null 2> "$OUT/fetch.err"
echo "fetch pass done" >&2
export STN_LAUNCH_LOG="$OUT/log_write.json"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT" -o pmc_write -- python3 "$B" --steps 1 --warmup 1 --cpu-sample 0 --no-profile --eager --no-host-loop --no-b1 > /dev/null 2> "$OUT/write.err"
echo "write pass done" >&2
export STN_LAUNCH_LOG="$OUT/log_mfma.json"
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT" -o pmc_mfma -- python3 "$B" --steps 1 --warmup 1 --cpu-sample 0 --no-profile --eager --no-host-loop --no-b1 > /dev/null 2> "$OUT/mfma.err"
echo "mfma pass done" >&2
export STN_LAUNCH_LOG="$OUT/log_tcc.json"
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT" -o pmc_tcc -- python3 "$B" --steps 1 --warmup 1 --cpu-sample 0 --no-profile --eager --no-host-loop --no-b1 > /dev/null 2> "$OUT/tcc.err" || echo "tcc pass failed (counters unavailable?)" >&2
echo "tcc pass done" >&2
unset STN_LAUNCH_LOG
cd "$ROOT"
S=$(find "$OUT" -name 'trace_kernel_stats.csv' | head -1)
cp "$S" "profiles/${TAG}_bench_kernel_stats.csv"
grep '^{' "$OUT/bench.json" | tail -1 > "profiles/${TAG}_bench.json"
python3 tools/pmc_families.py --dir "$OUT" --tag "$TAG" > "$OUT/families.log"
# the un-profiled default run, AFTER the summaries exist: its roofline.traffic / mfma_util_pmc come from them (source-hash checked)
(cd /tmp && python3 "$B" > "$OUT/default_bench.json" 2> "$OUT/default_bench.err") || echo "default bench failed" >&2
grep '^{' "$OUT/default_bench.json" | tail -1 > "profiles/${TAG}_default_bench.json"
# the summaries travel back with gpurun_out/ (profiles/ on the box is not merged): copy them beside the raw output
mkdir -p "$OUT/profiles" && cp profiles/l2_hit.json "$OUT/profiles/" 2>/dev/null || true
cp profiles/pmc_traffic.json profiles/mfma_util.json "profiles/${TAG}_families.csv" "profiles/${TAG}_bench_kernel_stats.csv" "profiles/${TAG}_bench.json" "profiles/${TAG}_default_bench.json" "$OUT/profiles/"
echo "summaries in $OUT/profiles" >&2
