#!/bin/bash
# Round-3 probe 13: batched_reg_kernel with preloaded leading arguments - parity of the batch paths, B = 3 / 8 / 16 timings.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p13
mkdir -p $O
cd $R
echo "== tests" | tee $O/progress.log
timeout -k 10 700 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "batch or sequences or config4 or launch_sequence or tiled" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -5 $O/tests.log
for a in "8 100 128 bf16" "8 100 128 bf16" "16 100 128 bf16" "3 100 128 bf16" "8 100 128 fp8"; do
  timeout -k 10 150 python tools/decode_prof.py $a graph >> $O/dp.log 2>&1 || exit 1
done
cat $O/dp.log
exit 0
