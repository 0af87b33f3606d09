#!/bin/bash
# Round-3 probe 28: small batches - GEMV kernels (M = 2, 4) against the register-fragment MFMA kernels (PGK_BATCHED_MFMA=2: from 3 sequences up)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/p28; mkdir -p $O; cd $R
for b in 2 3 4 5; do
  echo "-- default B=$b" >> $O/dp.log; timeout -k 10 100 python tools/decode_prof.py $b 100 128 bf16 graph >> $O/dp.log 2>&1 || exit 1
  echo "-- PGK_BATCHED_MFMA=2 B=$b" >> $O/dp.log; PGK_BATCHED_MFMA=2 timeout -k 10 100 python tools/decode_prof.py $b 100 128 bf16 graph >> $O/dp.log 2>&1 || exit 1
done
for b in 2 4; do
  echo "-- default fp8 B=$b" >> $O/dp.log; timeout -k 10 100 python tools/decode_prof.py $b 100 128 fp8 graph >> $O/dp.log 2>&1 || exit 1
  echo "-- PGK_BATCHED_MFMA=2 fp8 B=$b" >> $O/dp.log; PGK_BATCHED_MFMA=2 timeout -k 10 100 python tools/decode_prof.py $b 100 128 fp8 graph >> $O/dp.log 2>&1 || exit 1
done
cat $O/dp.log
