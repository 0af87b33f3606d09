#!/bin/bash
# Round-3 probe 39: the single-sequence hand-over (384) on big caches: contexts 300 / 400 / 480 at cache 4096 and 2164, short sequence forced
# up to 512 by the library before this change is not available any more - compare the long sequence against the 300-position short step.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/p39; mkdir -p $O; cd $R
for cap in 2164 4096; do for ctx in 300 400 480 1000; do
  timeout -k 10 100 python tools/decode_prof.py 1 24 $ctx bf16 graph $cap >> $O/dp.log 2>&1
done; done
cat $O/dp.log
