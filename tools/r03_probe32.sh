#!/bin/bash
# Round-3 probe 32: 5..8 sequences - register-fragment MFMA kernels (default) against GEMV chunks of 8 / 4 / 2 / 1 (PGK_BATCHED_MFMA=0)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/p32; mkdir -p $O; cd $R
for b in 5 6 8; do
  echo "-- default B=$b" >> $O/dp.log; timeout -k 10 100 python tools/decode_prof.py $b 100 128 bf16 graph >> $O/dp.log 2>&1
  echo "-- PGK_BATCHED_MFMA=0 B=$b" >> $O/dp.log; PGK_BATCHED_MFMA=0 timeout -k 10 100 python tools/decode_prof.py $b 100 128 bf16 graph >> $O/dp.log 2>&1
done
cat $O/dp.log
