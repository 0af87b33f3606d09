#!/bin/bash
# Round-3 probe 8: SwiGLU in the gate/up GEMM epilogue - bit-identity tests, config 5 and long-prompt timings, S = 2048 kernel stats.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p8
mkdir -p $O
cd $R
echo "== tests" | tee $O/progress.log
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "fused_prefill_epilogues or config5 or prefill" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -5 $O/tests.log
echo "== timings" | tee -a $O/progress.log
timeout -k 10 200 python tools/prefill_prof.py 2048 5 > $O/pf.log 2>&1 || exit 1
PGK_FUSED_EPILOGUES=0 timeout -k 10 200 python tools/prefill_prof.py 2048 5 >> $O/pf.log 2>&1 || exit 1
timeout -k 10 200 python tools/prefill_prof.py 1024 5 >> $O/pf.log 2>&1 || exit 1
cat $O/pf.log
timeout -k 10 300 python tools/config5_prefill.py 4096 8 2 > $O/c5.log 2>&1 || echo "config5 failed" | tee -a $O/progress.log
PGK_FUSED_EPILOGUES=0 timeout -k 10 300 python tools/config5_prefill.py 4096 8 2 > $O/c5_sep.log 2>&1 || echo "config5 sep failed" | tee -a $O/progress.log
tail -6 $O/c5.log; tail -6 $O/c5_sep.log
echo "== rocprof stats S=2048 / config 5" | tee -a $O/progress.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_pf2048 -- python3 $R/tools/prefill_prof.py 2048 5 > $O/st_pf2048.log 2>&1 || echo "stats pf2048 failed" | tee -a $O/progress.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_c5 -- python3 $R/tools/config5_prefill.py 4096 4 2 > $O/st_c5.log 2>&1 || echo "stats c5 failed" | tee -a $O/progress.log
cd $R
python tools/rocprof_top.py $O/st_pf2048 16 > $O/st_pf2048.txt 2>&1; cat $O/st_pf2048.txt | cut -c1-200
python tools/rocprof_by_grid.py $O/st_pf2048 24 > $O/st_pf2048_grid.txt 2>&1; cat $O/st_pf2048_grid.txt | cut -c1-200
python tools/rocprof_top.py $O/st_c5 16 > $O/st_c5.txt 2>&1; cat $O/st_c5.txt | cut -c1-200
find $O -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
exit 0
