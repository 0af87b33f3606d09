#!/bin/bash
# Round-3 probe 16: SwiGLU epilogue of the 128-tile GEMM - bit-identity, long-prompt parity, timings.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p16
mkdir -p $O
cd $R
echo "== tests" | tee $O/progress.log
timeout -k 10 700 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "qkv_head or prefill or config3 or long" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -5 $O/tests.log
for n in 2048 1024 512 300; do
  timeout -k 10 200 python tools/prefill_prof.py $n 5 >> $O/pf.log 2>&1 || exit 1
  PGK_FUSED_EPILOGUES=0 timeout -k 10 200 python tools/prefill_prof.py $n 5 >> $O/pf.log 2>&1 || exit 1
done
cat $O/pf.log
exit 0
