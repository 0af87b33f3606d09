#!/bin/bash
# Round-3 probe 38: short-context limit of 384 for a single sequence - launch-sequence tests, contexts 300..500 at batch 1.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/p38; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "launch_sequence or crosses_from or short_cache or strided or config3 or engine_greedy or decode_equals or full_size_engine" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for ctx in 300 370 400 480; do timeout -k 10 100 python tools/decode_prof.py 1 24 $ctx bf16 graph 1024 >> $O/dp.log 2>&1; done
cat $O/dp.log
