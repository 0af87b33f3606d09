#!/bin/bash
# Round-3 probe 9: the staged 128-tile GEMM - parity (whole GPU suite), long-prompt timings, split count A/B, kernel stats.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p9
mkdir -p $O
cd $R
echo "== tests" | tee $O/progress.log
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -5 $O/tests.log
echo "== timings" | tee -a $O/progress.log
for wgs in 256 512; do
  echo "-- PGK_TUNE_SPLIT_WGS=$wgs" >> $O/pf.log
  for n in 2048 1024 512 256; do
    PGK_TUNE_SPLIT_WGS=$wgs timeout -k 10 200 python tools/prefill_prof.py $n 5 >> $O/pf.log 2>&1 || exit 1
  done
done
cat $O/pf.log
timeout -k 10 200 python tools/gemm_bench.py bf16 2048 1024 2048 2048 1024 3072 2048 4096 1024 2048 6144 1024 1024 1024 2048 512 4096 1024 > $O/gemm.log 2>&1 || exit 1
cat $O/gemm.log
echo "== rocprof stats S=2048" | tee -a $O/progress.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_pf2048 -- python3 $R/tools/prefill_prof.py 2048 5 > $O/st_pf2048.log 2>&1 || echo "stats pf2048 failed" | tee -a $O/progress.log
cd $R
python tools/rocprof_by_grid.py $O/st_pf2048 24 > $O/st_pf2048_grid.txt 2>&1; cat $O/st_pf2048_grid.txt | cut -c1-200
find $O -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
exit 0
