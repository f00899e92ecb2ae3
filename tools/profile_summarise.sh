#!/bin/bash
# Host-side half of tools/profile_round.sh: gpurun merges only gpurun_out/ back, so the summaries under profiles/ are written here.
#   tools/profile_summarise.sh <tag>
set -eo pipefail
TAG=${1:?tag}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
OUT=gpurun_out/$TAG
F=$(find "$OUT" -name 'pmc_fetch_counter_collection.csv' | head -1)
W=$(find "$OUT" -name 'pmc_write_counter_collection.csv' | head -1)
T=$(find "$OUT" -name 'trace_kernel_trace.csv' | head -1)
S=$(find "$OUT" -name 'trace_kernel_stats.csv' | head -1)
M=$(find "$OUT" -name 'pmc_mfma_counter_collection.csv' | head -1)
cp "$S" "profiles/${TAG}_bench_kernel_stats.csv"
grep '^{' "$OUT/bench.json" | tail -1 > "profiles/${TAG}_bench.json"
grep "pmc_summary.py" tools/profile_round.sh | sed "s#\"\$F\"#$F#; s#\"\$W\"#$W#; s#\"\$T\"#$T#" | bash
if [ -n "$M" ]; then python3 tools/pmc_mfma.py "$M" --trace "$T" > "profiles/${TAG}_mfma_util.json"; cp "profiles/${TAG}_mfma_util.json" profiles/mfma_util.json; fi
echo "summaries written for $TAG"
