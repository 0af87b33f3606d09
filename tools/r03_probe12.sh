#!/bin/bash
# Round-3 probe 12: 192-column tiles of the staggered bf16 GEMM - parity, GEMM timings, config 5.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p12
mkdir -p $O
cd $R
echo "== tests" | tee $O/progress.log
timeout -k 10 700 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu -k "gemm or config5 or prefill or linear or matmul" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -5 $O/tests.log
timeout -k 10 200 python tools/gemm_bench.py bf16 4096 6144 4096 4096 4096 4096 4096 3072 4096 8192 6144 8192 > $O/gemm.log 2>&1 || exit 1
PGK_GEMM256S=0 timeout -k 10 200 python tools/gemm_bench.py bf16 4096 6144 4096 >> $O/gemm.log 2>&1 || exit 1
cat $O/gemm.log
timeout -k 10 300 python tools/config5_prefill.py 4096 8 2 > $O/c5.log 2>&1 || echo "config5 failed" | tee -a $O/progress.log
tail -6 $O/c5.log
exit 0
