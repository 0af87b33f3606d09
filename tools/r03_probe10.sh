#!/bin/bash
# Round-3 probe 10: fp32 += epilogue through LDS - parity of the long-prompt paths, config 5 timings and kernel stats.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p10
mkdir -p $O
cd $R
echo "== tests" | tee $O/progress.log
timeout -k 10 700 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q -m gpu -k "config5 or prefill or gemm or fused_prefill" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -5 $O/tests.log
echo "== timings" | tee -a $O/progress.log
timeout -k 10 300 python tools/config5_prefill.py 4096 8 2 > $O/c5.log 2>&1 || echo "config5 failed" | tee -a $O/progress.log
tail -6 $O/c5.log
timeout -k 10 200 python tools/prefill_prof.py 2048 5 > $O/pf.log 2>&1 || exit 1
cat $O/pf.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_c5 -- python3 $R/tools/config5_prefill.py 4096 4 2 > $O/st_c5.log 2>&1 || echo "stats c5 failed" | tee -a $O/progress.log
cd $R
python tools/rocprof_by_grid.py $O/st_c5 24 > $O/st_c5_grid.txt 2>&1; cat $O/st_c5_grid.txt | cut -c1-200
find $O -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
exit 0
