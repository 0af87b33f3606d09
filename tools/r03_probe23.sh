#!/bin/bash
# Round-3 probe 23: would two concurrent half-batches beat one 64-sequence step?  Two PROCESSES of 32 sequences each on one GPU
# (separate queues: their kernels may share the CUs) against one process of 64 - no engine change needed to ask the question.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p23
mkdir -p $O
cd $R
timeout -k 10 150 python tools/decode_prof.py 64 300 128 bf16 graph > $O/one64.log 2>&1 || exit 1
timeout -k 10 150 python tools/decode_prof.py 32 300 128 bf16 graph > $O/one32.log 2>&1 || exit 1
timeout -k 10 200 python tools/decode_prof.py 32 300 128 bf16 graph > $O/two32_a.log 2>&1 &
pa=$!
timeout -k 10 200 python tools/decode_prof.py 32 300 128 bf16 graph > $O/two32_b.log 2>&1 &
pb=$!
wait $pa; wait $pb
cat $O/one64.log $O/one32.log $O/two32_a.log $O/two32_b.log
exit 0
