#!/bin/bash
# Round-3 probe 7: parity with fp8 engines on the packed-weight paths, decode matrix, the full bench line, its rocprof stats.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p7
mkdir -p $O
cd $R
echo "== tests" | tee $O/progress.log
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -5 $O/tests.log
echo "== decode matrix" | tee -a $O/progress.log
for a in "1 200 128 bf16" "1 200 128 fp8" "8 100 128 bf16" "8 100 128 fp8" "16 100 128 bf16" "32 50 128 bf16" "32 50 128 fp8" "64 50 128 bf16" "64 50 128 fp8" "1 100 2048 fp8" "1 100 2048 bf16"; do
  timeout -k 10 150 python tools/decode_prof.py $a graph >> $O/dp.log 2>&1 || exit 1
done
cat $O/dp.log
timeout -k 10 200 python tools/prefill_prof.py 128 20 > $O/pf.log 2>&1 || exit 1
timeout -k 10 200 python tools/prefill_prof.py 2048 5 >> $O/pf.log 2>&1 || exit 1
cat $O/pf.log
echo "== bench" | tee -a $O/progress.log
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/progress.log
cat $O/bench.json; tail -5 $O/bench.err
echo "== timeline" | tee -a $O/progress.log
timeout -k 10 120 python tools/timeline_dump.py 1 150 bf16 $O/tl_b1.json > $O/tl_b1.txt 2>&1 || exit 1
timeout -k 10 120 python tools/timeline_dump.py 1 2048 fp8 $O/tl_c3.json > $O/tl_c3.txt 2>&1 || exit 1
echo "== rocprof stats of the bench command" | tee -a $O/progress.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_bench -- python3 $R/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-extras --long-prefill 0 > $O/st_bench.log 2>&1 || echo "stats bench failed" | tee -a $O/progress.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_c5 -- python3 $R/tools/config5_prefill.py 4096 4 2 > $O/st_c5.log 2>&1 || echo "stats c5 failed" | tee -a $O/progress.log
cd $R
python tools/rocprof_top.py $O/st_bench 12 > $O/st_bench.txt 2>&1; cat $O/st_bench.txt | cut -c1-160
python tools/rocprof_top.py $O/st_c5 14 > $O/st_c5.txt 2>&1; cat $O/st_c5.txt | cut -c1-160
find $O -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
tail -4 $O/st_c5.log
