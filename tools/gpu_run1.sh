#!/bin/bash
# N > 1 code paths on a one-GPU box: two ranks on GPU 0 over gloo -- (1) config E tensor-parallel at world 2 on the HIP
# kernels, (2) two replicas of the default config + the tp_70b child job
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
export SEQUOIA_BENCH_ONE_DEVICE=1
echo "== TP world 2 (gloo, one device), config E" 
SEQUOIA_TS_EXCLUSIVE=1 timeout 900 python bench.py --gpus 2 --backend gloo --config E --steps 6 --warmup 2 --no-cpu-baseline --no-autoregressive --no-tuned-growmap --no-tp-extra --sync-loop > gpurun_out/r2/tp2_gloo.json 2> gpurun_out/r2/tp2_gloo.err; echo "rc=$?"; tail -c 1200 gpurun_out/r2/tp2_gloo.json; tail -5 gpurun_out/r2/tp2_gloo.err

