#!/bin/bash
# Round-3 probe 19: A/B of the direct decode attention's preload (library before / after) on one box, interleaved.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p19
mkdir -p $O
cd $R
for i in 1 2 3; do
  for a in "64 50 128 bf16" "48 50 128 bf16"; do
    echo "new: " >> $O/dp.log; timeout -k 10 150 python tools/decode_prof.py $a graph >> $O/dp.log 2>&1 || exit 1
    echo "old: " >> $O/dp.log; PGK_LIB=$R/tools/micro/libpgk_old.so timeout -k 10 150 python tools/decode_prof.py $a graph >> $O/dp.log 2>&1 || exit 1
  done
done
cat $O/dp.log
exit 0
