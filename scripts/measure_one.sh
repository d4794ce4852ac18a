#!/bin/bash
# One workload on the MI355X box: bench line, rocprofv3 kernel trace, the two PMC passes.
# usage: scripts/measure_one.sh <tag> <workload> [extra bench args]
set -eo pipefail
TAG=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python bench.py --workload "$WL" --steps 10 --warmup 2 "$@" > "$OUT/bench_$WL.json" 2> "$OUT/bench_$WL.err"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$WL" -- \
  python3 "$ROOT/bench.py" --workload "$WL" --steps 5 --warmup 1 --no-cpu-baseline "$@" > "$OUT/kt_$WL.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$WL" -- \
  python3 "$ROOT/bench.py" --workload "$WL" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/pmc_fetch_$WL.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$WL" -- \
  python3 "$ROOT/bench.py" --workload "$WL" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/pmc_write_$WL.log" 2>&1
cd "$ROOT"
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete || true
find "$OUT" -name "*.db" -delete || true
find "$OUT/kt_$WL" -name "*kernel_stats.csv" | head -1 | xargs -r head -12
