#!/bin/bash
# Round-3 probe 4: GEMM parity after the fp8 pipeline / swizzle changes, their timings, SLP A/B on the GEMM-bound prefill.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p4
mkdir -p $O
cd $R
echo "== gemm tests" | tee $O/progress.log
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or matmul or fp8 or config5" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -4 $O/tests.log
echo "== timings" | tee -a $O/progress.log
timeout -k 10 200 python tools/gemm_bench.py fp8 4096 4096 4096 8192 8192 8192 4096 6144 4096 4096 28672 4096 4096 4096 14336 > $O/gemm_fp8.log 2>&1 || exit 1
timeout -k 10 200 python tools/gemm_bench.py bf16 4096 4096 4096 8192 8192 8192 4096 6144 4096 4096 28672 4096 4096 4096 14336 > $O/gemm_bf16.log 2>&1 || exit 1
PGK_LIB=$R/tools/micro/libpgk_slp.so timeout -k 10 200 python tools/gemm_bench.py bf16 4096 4096 4096 4096 28672 4096 4096 4096 14336 > $O/gemm_bf16_slp.log 2>&1 || exit 1
cat $O/gemm_fp8.log $O/gemm_bf16.log $O/gemm_bf16_slp.log
timeout -k 10 300 python tools/config5_prefill.py 4096 8 2 > $O/c5.log 2>&1 || echo "config5 failed" | tee -a $O/progress.log
PGK_LIB=$R/tools/micro/libpgk_slp.so timeout -k 10 300 python tools/config5_prefill.py 4096 8 2 > $O/c5_slp.log 2>&1 || echo "config5 slp failed" | tee -a $O/progress.log
tail -6 $O/c5.log; tail -6 $O/c5_slp.log
echo "== PMC fp8 / bf16" | tee -a $O/progress.log
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
for n in fp8 bf16; do
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P1 -d $O/pmc_${n}_1 --output-format csv -- python3 $R/tools/gemm_bench.py $n 4096 4096 4096 4096 28672 4096 > $O/pmc_${n}_1.log 2>&1 || echo "pmc $n 1 failed" | tee -a $O/progress.log
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P2 -d $O/pmc_${n}_2 --output-format csv -- python3 $R/tools/gemm_bench.py $n 4096 4096 4096 4096 28672 4096 > $O/pmc_${n}_2.log 2>&1 || echo "pmc $n 2 failed" | tee -a $O/progress.log
done
cd $R
for n in fp8 bf16; do
  { echo "# gemm_$n pass 1: $P1"; python tools/pmc_summary.py $O/pmc_${n}_1 gemm256; echo "# gemm_$n pass 2: $P2"; python tools/pmc_summary.py $O/pmc_${n}_2 gemm256; } > $O/pmc_gemm_$n.txt 2>&1
  find $O/pmc_${n}_1 $O/pmc_${n}_2 -name "*.csv" -size +2M -delete 2>/dev/null
done
cat $O/pmc_gemm_fp8.txt
