#!/bin/bash
# Round-3 probe 34: w8a16 engines' long prompts on the staged bf16 GEMMs reading the dequantised fragment-major copy - full GPU suite, timings.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/p34; mkdir -p $O; cd $R
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for n in 300 512 2048; do python tools/prefill_prof.py $n 5 fp8; done
timeout -k 10 300 python tools/config5_prefill.py 4096 8 2 > $O/c5.log 2>&1; tail -5 $O/c5.log
