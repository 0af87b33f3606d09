#!/bin/bash
# Round-3 probe 15: QKV-heads epilogue on the fp8 256-tile kernel - bit-identity, config 5 timing.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p15
mkdir -p $O
cd $R
echo "== tests" | tee $O/progress.log
timeout -k 10 700 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "fused_prefill or qkv_head or config5 or prefill" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -5 $O/tests.log
timeout -k 10 300 python tools/config5_prefill.py 4096 8 2 > $O/c5.log 2>&1 || echo "config5 failed" | tee -a $O/progress.log
tail -6 $O/c5.log
exit 0
