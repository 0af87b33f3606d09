#!/bin/bash
# Round-3 probe 37: where should the short-context launch sequence end?  Batch 1 and 8, contexts 200..500, fused / whole-context
# kernels (default below 512) against the split-KV sequence (PGK_FUSED_ATTN=0) at the same contexts.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/p37; mkdir -p $O; cd $R
for b in 1 8; do for ctx in 200 300 400 480; do
  echo "-- short B=$b" >> $O/dp.log; timeout -k 10 100 python tools/decode_prof.py $b 24 $ctx bf16 graph 1024 >> $O/dp.log 2>&1
  echo "-- long  B=$b" >> $O/dp.log; PGK_FUSED_ATTN=0 timeout -k 10 100 python tools/decode_prof.py $b 24 $ctx bf16 graph 1024 >> $O/dp.log 2>&1
done; done
cat $O/dp.log
