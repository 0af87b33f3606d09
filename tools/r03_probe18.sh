#!/bin/bash
# Round-3 probe 18: direct decode attention requests rows 128-191 only up to the context - batch parity, B = 33..64 timings.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p18
mkdir -p $O
cd $R
echo "== tests" | tee $O/progress.log
timeout -k 10 700 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "batch or sequences or config4 or short_cache or tiled or gqa" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -5 $O/tests.log
for a in "64 50 128 bf16" "64 50 128 bf16" "48 50 128 bf16" "64 50 100 bf16" "64 30 170 bf16"; do
  timeout -k 10 150 python tools/decode_prof.py $a graph >> $O/dp.log 2>&1 || exit 1
done
cat $O/dp.log
exit 0
