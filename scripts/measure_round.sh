#!/bin/bash
# Runs on the MI355X box (through gpurun): the bench lines and the rocprofv3 passes whose
# summaries are committed under profiles/.  usage: scripts/measure_round.sh <tag> <part>
#   part 1: bench lines of every workload;  part 2: kernel traces + PMC passes
set -eo pipefail
TAG=${1:-r03}; PART=${2:-1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
b() {  # b <name> <bench args...>
  local name=$1; shift
  echo "[measure] $name"; date
  python bench.py "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || { tail -5 "$OUT/bench_$name.err"; return 1; }
}
if [ "$PART" = 1 ]; then
  b config3 --steps 20 --warmup 3
  b config3_exact_sums --steps 20 --warmup 3 --float-sums exact --no-cpu-baseline
  b config2 --steps 20 --warmup 3 --workload config2 --no-cpu-baseline
  b config2_bitpacked --steps 20 --warmup 3 --workload config2 --k-bits 10 --no-cpu-baseline
  b config3_leb128 --steps 20 --warmup 3 --workload config3l
  b config4 --steps 10 --warmup 2 --workload config4
  b config4_nohint --steps 10 --warmup 2 --workload config4 --no-hint --no-cpu-baseline
  b config4_string_keys --steps 10 --warmup 2 --workload config4s
  b config5 --steps 20 --warmup 3 --workload config5
  b config5w --steps 10 --warmup 2 --workload config5w --no-cpu-baseline
  echo "[measure] device writer"; date
  python scripts/bench_writer.py 1e8 > "$OUT/bench_writer.jsonl" 2> "$OUT/bench_writer.err" || tail -5 "$OUT/bench_writer.err"
else
  cd /tmp; export TMPDIR=/tmp
  kt() {  # kt <name> <bench args...>
    local name=$1; shift
    echo "[measure] kernel trace $name"; date
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$name" -- \
      python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/kt_$name.log" 2>&1
  }
  pmc() {  # pmc <counter> <name> <bench args...>
    local ctr=$1 name=$2; shift 2
    echo "[measure] pmc $ctr $name"; date
    rocprofv3 --pmc "$ctr" --output-format csv -d "$OUT/pmc_${ctr}_$name" -- \
      python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/pmc_${ctr}_$name.log" 2>&1
  }
  kt config3 --steps 10 --warmup 2
  kt config4 --steps 5 --warmup 1 --workload config4
  kt config4s --steps 5 --warmup 1 --workload config4s
  kt config3l --steps 10 --warmup 2 --workload config3l
  kt config4_nohint --steps 5 --warmup 1 --workload config4 --no-hint
  kt config5 --steps 10 --warmup 2 --workload config5
  kt config5w --steps 5 --warmup 1 --workload config5w
  kt config2 --steps 10 --warmup 2 --workload config2
  for c in FETCH_SIZE WRITE_SIZE; do
    pmc $c config3 --steps 3 --warmup 1
    pmc $c config4 --steps 3 --warmup 1 --workload config4
    pmc $c config4s --steps 3 --warmup 1 --workload config4s
  done
  cd "$ROOT"
  find "$OUT" -name "*kernel_trace.csv" -size +2M -delete || true
  find "$OUT" -name "*.db" -delete || true
fi
echo "[measure] done"; date
