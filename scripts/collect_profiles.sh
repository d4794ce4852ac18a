#!/bin/bash
# Copies what scripts/measure_round.sh left under gpurun_out/<tag>/ into profiles/ (tracked):
# bench lines, kernel summaries (foreign kernels of the table generators removed), PMC rows
# of the evql_* kernels, and refreshes profiles/traffic.json.  usage: collect_profiles.sh <tag>
set -eo pipefail
TAG=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out/$TAG
P=$ROOT/profiles
for f in "$O"/bench_*.json; do
  [ -e "$f" ] || continue
  grep '^{' "$f" > "$P/${TAG}_$(basename "$f")"
done
[ -e "$O/bench_writer.jsonl" ] && cp "$O/bench_writer.jsonl" "$P/${TAG}_bench_writer.jsonl"
for d in "$O"/kt_*/; do
  [ -d "$d" ] || continue
  w=$(basename "$d"); w=${w#kt_}
  f=$(ls "$d"/*/*_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && grep -v "at::native\|rocprim\|at::cuda" "$f" > "$P/${TAG}_${w}_kernel_stats.csv"
done
for c in FETCH_SIZE WRITE_SIZE; do
  for d in "$O"/pmc_${c}_*/; do
    [ -d "$d" ] || continue
    w=$(basename "$d"); w=${w#pmc_${c}_}
    f=$(ls "$d"/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && { head -1 "$f"; grep '"evql_' "$f" || true; } > "$P/${TAG}_${w}_pmc_$c.csv"
  done
done
[ -d "$O/pmc_FETCH_SIZE_config3" ] && python3 "$ROOT/scripts/pmc_summary.py" config3_1000000000 "$O/pmc_FETCH_SIZE_config3" "$O/pmc_WRITE_SIZE_config3" > /dev/null
[ -d "$O/pmc_FETCH_SIZE_config4" ] && python3 "$ROOT/scripts/pmc_summary.py" config4_125000000 "$O/pmc_FETCH_SIZE_config4" "$O/pmc_WRITE_SIZE_config4" evql_part_ > /dev/null
[ -d "$O/pmc_FETCH_SIZE_config4s" ] && python3 "$ROOT/scripts/pmc_summary.py" config4s_125000000 "$O/pmc_FETCH_SIZE_config4s" "$O/pmc_WRITE_SIZE_config4s" evql_part_ > /dev/null
ls "$P" | grep "^${TAG}_" | wc -l
