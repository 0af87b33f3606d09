#!/bin/bash
# Round-3 probe 20: flash KV split into runs of bounded length - parity, attention and prefill timings.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p20
mkdir -p $O
cd $R
echo "== tests" | tee $O/progress.log
timeout -k 10 700 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu -k "sdpa or attention or flash or prefill or fused_prefill or config5 or qkv_head or chunk" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -5 $O/tests.log
timeout -k 10 200 python tools/attn_bench.py 16 8 2048 128 16 8 1024 128 16 8 512 128 32 8 2048 128 32 8 4096 128 16 8 4096 128 > $O/attn.log 2>&1 || exit 1
cat $O/attn.log
for n in 2048 1024 512; do timeout -k 10 200 python tools/prefill_prof.py $n 5 >> $O/pf.log 2>&1 || exit 1; done
cat $O/pf.log
exit 0
