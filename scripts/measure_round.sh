#!/bin/bash
# Runs on the MI355X box (through gpurun): the bench lines and the rocprofv3 passes
# whose summaries are committed under profiles/.  Usage: scripts/measure_round.sh r01
set -eo pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
echo "[measure] bench lines"; date
python bench.py --steps 20 --warmup 3 > "$OUT/bench_config3.json" 2> "$OUT/bench_config3.err"
python bench.py --steps 20 --warmup 3 --workload config2 > "$OUT/bench_config2.json" 2> "$OUT/bench_config2.err"
python bench.py --steps 20 --warmup 3 --workload config2 --k-bits 10 --no-cpu-baseline \
  > "$OUT/bench_config2_bitpacked.json" 2> "$OUT/bench_config2_bitpacked.err"
python bench.py --steps 20 --warmup 3 --workload config3l > "$OUT/bench_config3_leb128.json" 2> "$OUT/bench_config3_leb128.err"
echo "[measure] config4 / config5"; date
python bench.py --steps 10 --warmup 2 --workload config4 > "$OUT/bench_config4.json" 2> "$OUT/bench_config4.err"
python bench.py --steps 20 --warmup 3 --workload config5 > "$OUT/bench_config5.json" 2> "$OUT/bench_config5.err"
echo "[measure] rocprofv3 kernel traces"; date
cd /tmp
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_config3" -- \
  python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/kt_config3.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_config2" -- \
  python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --workload config2 > "$OUT/kt_config2.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_config4" -- \
  python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --workload config4 > "$OUT/kt_config4.log" 2>&1
echo "[measure] rocprofv3 PMC passes (separate runs)"; date
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- \
  python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- \
  python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"
# keep only the small summaries
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete || true
find "$OUT" -name "*.db" -delete || true
echo "[measure] done"; date
ls -R "$OUT" | head -60
