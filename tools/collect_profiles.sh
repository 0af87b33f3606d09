#!/bin/bash
# Collects the evidence behind DESIGN.md / profiles/README.md into gpurun_out/final/ (copy what is to be judged into profiles/).
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh 1'     bench line, decode stats + PMC traffic, timelines, batch stats
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh 2'     prefill stats, config 5, GEMM / attention timings and PMC summaries
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
mkdir -p $O
part=${1:-1}
BENCH="$R/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-extras --long-prefill 0"
stats() {   # name, program args...
  local name=$1; shift
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$name -- python3 "$@" > $O/st_$name.log 2>&1) || { echo "stats failed: $name" | tee -a $O/progress.log; return 0; }
  cp $(ls $O/st_$name/*/*kernel_stats.csv | head -1) $O/${name}_kernel_stats.csv
  (cd $R && python tools/rocprof_by_grid.py $O/st_$name 30 > $O/${name}_by_grid.txt 2>&1)
  find $O/st_$name -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
  echo "stats done: $name" | tee -a $O/progress.log
}
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
pmc() {   # name, kernel filter, program args...
  local name=$1 filt=$2; shift 2
  (cd /tmp && TMPDIR=/tmp timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P1 -d $O/pmc_${name}_1 --output-format csv -- python3 "$@" > $O/pmc_${name}_1.log 2>&1) || { echo "pmc 1 failed: $name" | tee -a $O/progress.log; return 0; }
  (cd /tmp && TMPDIR=/tmp timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P2 -d $O/pmc_${name}_2 --output-format csv -- python3 "$@" > $O/pmc_${name}_2.log 2>&1) || { echo "pmc 2 failed: $name" | tee -a $O/progress.log; return 0; }
  (cd $R && { echo "# $* (rocprofv3 --kernel-trace --pmc, two passes; per-kernel averages by tools/pmc_summary.py)"; echo "# pass 1: $P1"; python tools/pmc_summary.py $O/pmc_${name}_1 $filt; echo "# pass 2: $P2"; python tools/pmc_summary.py $O/pmc_${name}_2 $filt; } > $O/${name}_pmc.txt 2>&1)
  find $O/pmc_${name}_1 $O/pmc_${name}_2 -name "*.csv" -size +2M -delete 2>/dev/null
  echo "pmc done: $name" | tee -a $O/progress.log
}
cd $R
if [ "$part" = 1 ]; then
  echo "== part 1" | tee $O/progress.log
  timeout -k 10 500 python bench.py > $O/bench_n1.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/progress.log
  timeout -k 10 120 python tools/timeline_dump.py 1 150 bf16 $O/decode_timeline.json > $O/decode_timeline.txt 2>&1 || echo "timeline failed" | tee -a $O/progress.log
  timeout -k 10 120 python tools/timeline_dump.py 1 2048 fp8 $O/config3_timeline.json > $O/config3_timeline.txt 2>&1 || echo "timeline c3 failed" | tee -a $O/progress.log
  stats decode $BENCH
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python3 $BENCH > $O/pmc_fetch.log 2>&1) || echo "pmc fetch failed" | tee -a $O/progress.log
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python3 $BENCH > $O/pmc_write.log 2>&1) || echo "pmc write failed" | tee -a $O/progress.log
  python tools/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/decode_pmc_hbm.json > $O/decode_pmc_hbm.txt 2>&1 || echo "pmc_to_json failed" | tee -a $O/progress.log
  find $O/pmc_fetch $O/pmc_write -name "*.csv" -size +2M -delete 2>/dev/null
  stats batch8 $R/tools/decode_prof.py 8 30
  stats batch64 $R/tools/decode_prof.py 64 30
  stats batch64_fp8 $R/tools/decode_prof.py 64 30 128 fp8
  cat $O/bench_n1.json | cut -c1-600; cat $O/decode_pmc_hbm.txt | head -60
else
  echo "== part 2" | tee -a $O/progress.log
  stats prefill128 $R/tools/prefill_prof.py 128 20
  stats prefill2048 $R/tools/prefill_prof.py 2048 5
  stats prefill512 $R/tools/prefill_prof.py 512 10
  stats config5_L4 $R/tools/config5_prefill.py 4096 4 2
  timeout -k 10 400 python tools/config5_prefill.py 4096 32 3 > $O/config5_prefill.log 2>&1 || echo "config5 L32 failed" | tee -a $O/progress.log
  timeout -k 10 200 python tools/gemm_bench.py bf16 4096 4096 4096 8192 8192 8192 4096 6144 4096 4096 28672 4096 4096 4096 14336 2048 1024 2048 2048 1024 3072 2048 4096 1024 2048 6144 1024 > $O/gemm_bench.log 2>&1 || echo "gemm bf16 failed" | tee -a $O/progress.log
  timeout -k 10 200 python tools/gemm_bench.py fp8 4096 4096 4096 8192 8192 8192 4096 6144 4096 4096 28672 4096 4096 4096 14336 >> $O/gemm_bench.log 2>&1 || echo "gemm fp8 failed" | tee -a $O/progress.log
  timeout -k 10 200 python tools/attn_bench.py 32 8 4096 128 16 8 2048 128 16 8 1024 128 32 8 8192 128 > $O/attn_bench.log 2>&1 || echo "attn failed" | tee -a $O/progress.log
  pmc gemm256s gemm256s $R/tools/gemm_bench.py bf16 4096 4096 4096 4096 28672 4096
  pmc gemm256_fp8 gemm256_fp8 $R/tools/gemm_bench.py fp8 4096 4096 4096 4096 28672 4096
  pmc flash flash_fwd $R/tools/attn_bench.py 32 8 4096 128
  pmc pkgemm pkgemm $R/tools/prefill_prof.py 128 10
  pmc gemm128s gemm128s $R/tools/gemm_bench.py bf16 2048 1024 2048 2048 4096 1024
  cat $O/config5_prefill.log $O/gemm_bench.log $O/attn_bench.log
fi
exit 0
