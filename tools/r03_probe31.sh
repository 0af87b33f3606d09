#!/bin/bash
# Round-3 probe 31: two sequences on the whole-context attention + o_proj GEMV path - full GPU suite, B = 2 timings.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/p31; mkdir -p $O; cd $R
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for a in "2 100 128 bf16" "2 100 128 fp8" "2 100 600 bf16" "1 100 128 bf16"; do timeout -k 10 100 python tools/decode_prof.py $a graph >> $O/dp.log 2>&1; done
cat $O/dp.log
