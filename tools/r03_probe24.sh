#!/bin/bash
# Round-3 probe 24: SwiGLU epilogue on 192-column tiles - bit-identity at 2048 rows, long-prompt timings.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p24
mkdir -p $O
cd $R
timeout -k 10 700 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q -m gpu -k "qkv_head or fused_prefill or prefill or gemm256 or config5" > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/progress.log
tail -4 $O/tests.log
for i in 1 2; do timeout -k 10 200 python tools/prefill_prof.py 2048 5 >> $O/pf.log 2>&1 || exit 1; done
timeout -k 10 200 python tools/prefill_prof.py 4096 3 >> $O/pf.log 2>&1 || exit 1
cat $O/pf.log
exit 0
