#!/bin/bash
# rocprofv3 profile of bench.py on the GPU box; summaries land in gpurun_out/prof_<tag>/ (copy the ones to keep into profiles/)
# usage: tools/profile.sh <tag> [bench args]
set -u
TAG=${1:-r01}; shift || true
cd "$(dirname "$0")/.."
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-extras $*"
cd /tmp
rocprofv3 -f csv --kernel-trace --stats -d "$OUT/trace" -o trace -- python "$ROOT/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
rocprofv3 -f csv --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o fetch -- python "$ROOT/bench.py" $ARGS > "$OUT/bench_fetch.log" 2>&1
rocprofv3 -f csv --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o write -- python "$ROOT/bench.py" $ARGS > "$OUT/bench_write.log" 2>&1
cd "$ROOT"
python tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
