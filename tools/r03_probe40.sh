#!/bin/bash
# Round-3 probe 40: split-KV slices cut for the step's context tier instead of the cache length - full GPU suite, big-cache timings.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/p40; mkdir -p $O; cd $R
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for a in "1 24 400 bf16 graph 4096" "1 24 1000 bf16 graph 4096" "1 24 3000 bf16 graph 4096" "1 24 400 bf16 graph 8192" "1 100 2048 fp8 graph" "8 24 1000 bf16 graph 4096" "1 100 128 bf16 graph"; do
  timeout -k 10 150 python tools/decode_prof.py $a >> $O/dp.log 2>&1
done
cat $O/dp.log
