#!/bin/bash
# Round-3 probe 11: flash attention emitting the o_proj's fp8 operand - bit-identity, config 5 timing.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p11
mkdir -p $O
cd $R
echo "== tests" | tee $O/progress.log
timeout -k 10 700 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q -m gpu -k "config5 or prefill or attention or sdpa or flash" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -5 $O/tests.log
timeout -k 10 300 python tools/config5_prefill.py 4096 8 2 > $O/c5.log 2>&1 || echo "config5 failed" | tee -a $O/progress.log
tail -6 $O/c5.log
exit 0
