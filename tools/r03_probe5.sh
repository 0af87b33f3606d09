#!/bin/bash
# Round-3 probe 5: parity after the flash pipeline / KV split and the fp8 cross-tile pipeline; timings; flash PMC.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p5
mkdir -p $O
cd $R
echo "== tests" | tee $O/progress.log
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -5 $O/tests.log
echo "== timings" | tee -a $O/progress.log
timeout -k 10 200 python tools/gemm_bench.py fp8 4096 4096 4096 8192 8192 8192 4096 6144 4096 4096 28672 4096 4096 4096 14336 > $O/gemm_fp8.log 2>&1 || exit 1
timeout -k 10 200 python tools/attn_bench.py 32 8 4096 128 16 8 2048 128 16 8 1024 128 32 8 8192 128 > $O/attn.log 2>&1 || exit 1
timeout -k 10 200 python tools/prefill_prof.py 2048 5 > $O/pf.log 2>&1 || exit 1
timeout -k 10 300 python tools/config5_prefill.py 4096 8 2 > $O/c5.log 2>&1 || echo "config5 failed" | tee -a $O/progress.log
timeout -k 10 120 python tools/decode_prof.py 1 100 2048 fp8 graph >> $O/dp.log 2>&1 || exit 1
timeout -k 10 120 python tools/decode_prof.py 64 50 128 fp8 graph >> $O/dp.log 2>&1 || exit 1
timeout -k 10 120 python tools/decode_prof.py 8 100 128 bf16 graph >> $O/dp.log 2>&1 || exit 1
timeout -k 10 120 python tools/decode_prof.py 16 100 128 bf16 graph >> $O/dp.log 2>&1 || exit 1
timeout -k 10 120 python tools/decode_prof.py 1 200 128 bf16 graph >> $O/dp.log 2>&1 || exit 1
cat $O/gemm_fp8.log $O/attn.log $O/pf.log $O/dp.log; tail -6 $O/c5.log
echo "== PMC flash / fp8" | tee -a $O/progress.log
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P1 -d $O/pmc_flash_1 --output-format csv -- python3 $R/tools/attn_bench.py 32 8 4096 128 > $O/pmc_flash_1.log 2>&1 || echo "pmc flash 1 failed" | tee -a $O/progress.log
timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P2 -d $O/pmc_flash_2 --output-format csv -- python3 $R/tools/attn_bench.py 32 8 4096 128 > $O/pmc_flash_2.log 2>&1 || echo "pmc flash 2 failed" | tee -a $O/progress.log
timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P1 -d $O/pmc_fp8_1 --output-format csv -- python3 $R/tools/gemm_bench.py fp8 4096 4096 4096 4096 28672 4096 > $O/pmc_fp8_1.log 2>&1 || echo "pmc fp8 1 failed" | tee -a $O/progress.log
cd $R
{ echo "# flash pass 1: $P1"; python tools/pmc_summary.py $O/pmc_flash_1 flash_fwd; echo "# flash pass 2: $P2"; python tools/pmc_summary.py $O/pmc_flash_2 flash_fwd; } > $O/pmc_flash.txt 2>&1
{ echo "# gemm_fp8 pass 1: $P1"; python tools/pmc_summary.py $O/pmc_fp8_1 gemm256; } > $O/pmc_gemm_fp8.txt 2>&1
find $O -name "*.csv" -size +2M -delete 2>/dev/null
cat $O/pmc_flash.txt $O/pmc_gemm_fp8.txt | cut -c1-100
