#!/bin/bash
# Round-3 probe 3: parity after the fp8-GEMM / flash / GEMV changes, their timings, PMC summaries of the MFMA kernels.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p3
mkdir -p $O
cd $R
echo "== tests" | tee $O/progress.log
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -5 $O/tests.log
echo "== timings" | tee -a $O/progress.log
timeout -k 10 200 python tools/gemm_bench.py fp8 4096 4096 4096 8192 8192 8192 4096 6144 4096 4096 28672 4096 4096 4096 14336 > $O/gemm_fp8.log 2>&1 || exit 1
timeout -k 10 200 python tools/attn_bench.py 32 8 4096 128 16 8 2048 128 32 8 8192 128 > $O/attn.log 2>&1 || exit 1
timeout -k 10 120 python tools/decode_prof.py 1 200 128 bf16 graph > $O/dp.log 2>&1 || exit 1
timeout -k 10 200 python tools/prefill_prof.py 2048 5 > $O/pf.log 2>&1 || exit 1
timeout -k 10 300 python tools/config5_prefill.py 4096 8 2 > $O/c5.log 2>&1 || echo "config5 failed" | tee -a $O/progress.log
cat $O/gemm_fp8.log $O/attn.log $O/dp.log $O/pf.log; tail -12 $O/c5.log
echo "== kernel stats S=2048 / S=128" | tee -a $O/progress.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/st_pf2048 --output-format csv -- python3 $R/tools/prefill_prof.py 2048 5 > $O/st_pf2048.log 2>&1 || echo "stats 2048 failed" | tee -a $O/progress.log
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/st_pf128 --output-format csv -- python3 $R/tools/prefill_prof.py 128 20 > $O/st_pf128.log 2>&1 || echo "stats 128 failed" | tee -a $O/progress.log
cd $R
python tools/rocprof_by_grid.py $O/st_pf2048 40 > $O/pf2048_by_grid.txt 2>&1
python tools/rocprof_by_grid.py $O/st_pf128 20 > $O/pf128_by_grid.txt 2>&1
head -45 $O/pf2048_by_grid.txt
echo "== PMC passes" | tee -a $O/progress.log
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run_pmc() {
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P1 -d $O/pmc_${name}_1 --output-format csv -- python3 "$@" > $O/pmc_${name}_1.log 2>&1 || { echo "pmc pass 1 failed: $name" | tee -a $O/progress.log; return 1; }
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P2 -d $O/pmc_${name}_2 --output-format csv -- python3 "$@" > $O/pmc_${name}_2.log 2>&1 || { echo "pmc pass 2 failed: $name" | tee -a $O/progress.log; return 1; }
  echo "pmc done: $name" | tee -a $O/progress.log
}
run_pmc gemm_bf16 $R/tools/gemm_bench.py bf16 4096 4096 4096 4096 28672 4096 || exit 1
run_pmc gemm_fp8 $R/tools/gemm_bench.py fp8 4096 4096 4096 4096 28672 4096 || exit 1
run_pmc flash $R/tools/attn_bench.py 32 8 4096 128 || exit 1
run_pmc pf128 $R/tools/prefill_prof.py 128 10 || exit 1
run_pmc pf2048 $R/tools/prefill_prof.py 2048 3 || exit 1
cd $R
for n in gemm_bf16 gemm_fp8 flash pf128 pf2048; do
  { echo "# $n pass 1: $P1"; python tools/pmc_summary.py $O/pmc_${n}_1; echo "# $n pass 2: $P2"; python tools/pmc_summary.py $O/pmc_${n}_2; } > $O/pmc_$n.txt 2>&1
  find $O/pmc_${n}_1 $O/pmc_${n}_2 $O/st_pf2048 $O/st_pf128 -name "*.csv" -size +2M -delete 2>/dev/null
done
head -50 $O/pmc_gemm_fp8.txt
