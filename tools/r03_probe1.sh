#!/bin/bash
# Round-3 probe 1 (one gpurun call): parity after the cleanup, then kernel-argument preloading A/B and runtime switches.
#   gpurun --timeout 1100 -- 'bash tools/r03_probe1.sh'
set -o pipefail
O=gpurun_out/p1
mkdir -p $O
NP=$PWD/tools/micro/libpgk_nopreload.so
echo "== tests" | tee $O/progress.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log
tail -3 $O/tests.log
echo "== timelines" | tee -a $O/progress.log
timeout -k 10 120 python tools/timeline_dump.py 1 150 bf16 $O/tl_preload.json > $O/tl_preload.txt 2>&1 || exit 1
PGK_LIB=$NP timeout -k 10 120 python tools/timeline_dump.py 1 150 bf16 $O/tl_nopreload.json > $O/tl_nopreload.txt 2>&1 || exit 1
echo "== decode_prof" | tee -a $O/progress.log
for i in 1 2; do
  timeout -k 10 120 python tools/decode_prof.py 1 200 128 bf16 graph >> $O/dp.log 2>&1 || exit 1
  PGK_LIB=$NP timeout -k 10 120 python tools/decode_prof.py 1 200 128 bf16 graph >> $O/dp_np.log 2>&1 || exit 1
done
# context 150 on a big cache (the launch sequence must follow the context) and past 512
timeout -k 10 120 python tools/decode_prof.py 1 200 128 bf16 graph 4096 >> $O/dp.log 2>&1 || exit 1
timeout -k 10 120 python tools/decode_prof.py 1 100 600 bf16 graph 4096 >> $O/dp.log 2>&1 || exit 1
PGK_FUSED_ATTN=0 timeout -k 10 120 python tools/decode_prof.py 1 200 128 bf16 graph 4096 >> $O/dp.log 2>&1 || exit 1
timeout -k 10 120 python tools/decode_prof.py 8 100 128 bf16 graph >> $O/dp.log 2>&1 || exit 1
timeout -k 10 120 python tools/decode_prof.py 64 50 128 bf16 graph >> $O/dp.log 2>&1 || exit 1
timeout -k 10 120 python tools/decode_prof.py 1 100 2048 fp8 graph >> $O/dp.log 2>&1 || exit 1
echo "== runtime switches" | tee -a $O/progress.log
for ev in "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=256" "AMD_DIRECT_DISPATCH=0" "GPU_MAX_HW_QUEUES=1" "ROC_USE_FGS_KERNARG=0" "DEBUG_HIP_KERNARG_COPY_OPT=0"; do
  echo "-- $ev" >> $O/env.log
  env $ev timeout -k 10 120 python tools/decode_prof.py 1 200 128 bf16 graph >> $O/env.log 2>&1 || echo "failed: $ev" >> $O/env.log
done
cat $O/dp.log $O/dp_np.log $O/env.log
python - <<'EOF'
import json
for n in ("preload", "nopreload"):
    d = json.load(open(f"gpurun_out/p1/tl_{n}.json"))
    print(n, d["step_us_first_start_to_last_end"], d["sum_of_spans_us"], d["sum_of_gaps_us"], {k: (v["mean_span_us"], v["mean_gap_after_us"]) for k, v in d["by_kernel"].items()})
EOF
