#!/bin/bash
# Rehearsal of bench.py's N = 2 control flow on a ONE-GPU box: two ranks under torch.distributed.run, both on GPU 0
# (PGK_REHEARSE_SHARED_GPU=1).  Either RCCL brings a two-rank communicator up on the shared device - then the weight
# broadcast, the timed legs and the token gather run for real - or it refuses the duplicate GPU and every rank must leave
# with the JSON error line and exit code 3.  Both outcomes exercise code no single-rank run reaches.
#   gpurun --timeout 600 -- 'bash tools/rehearse_n2.sh'          (NPROC=4 for four ranks; at most 6 processes may use the card)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/n2
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PGK_REHEARSE_SHARED_GPU=1
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${NPROC:-2} --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus ${NPROC:-2} --steps 16 --warmup 4 --no-cpu-baseline --config4-steps 8 > $O/bench_n2.json 2> $O/bench_n2.err
echo "rc=$?" | tee $O/rc.txt
head -c 3000 $O/bench_n2.json; echo; tail -15 $O/bench_n2.err | cut -c1-300
exit 0
