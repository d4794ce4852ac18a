#!/bin/bash
# Warms the in-tree kernel cache (eventql_amd/_kcache, git-ignored, travels with the tree
# like the built .so files) with every plan the GPU suite compiles.
#   on the MI355X box (through gpurun):  bash scripts/harvest_kcache.sh
#   afterwards, in the container:        tar -xJf gpurun_out/kcache.tar.xz -C eventql_amd
# The cache is keyed by the fingerprint of the generated source: stale entries are never
# used, a missing one is compiled (hiprtc) and added.  Cold suite 644 s, warm 294 s (r03).
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=40 > gpurun_out/harvest_gpu_tests.log 2>&1 || {
  tail -20 gpurun_out/harvest_gpu_tests.log; exit 1; }
tail -3 gpurun_out/harvest_gpu_tests.log
tar -cJf gpurun_out/kcache.tar.xz -C eventql_amd _kcache
ls -la gpurun_out/kcache.tar.xz
ls eventql_amd/_kcache | wc -l
