#!/bin/bash
# GPU box: the gpu parity suite, then the N>1 bench path rehearsed with two ranks sharing the one GPU (gloo carries the gather).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/dist_tests.log 2>&1 || { tail -30 gpurun_out/dist_tests.log; exit 1; }
tail -3 gpurun_out/dist_tests.log
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --steps 40 --warmup 5 > gpurun_out/dist_bench2.log 2>&1 || { tail -30 gpurun_out/dist_bench2.log; exit 1; }
tail -2 gpurun_out/dist_bench2.log
timeout -k 10 300 python bench.py --steps 200 --warmup 20 > gpurun_out/dist_bench1.log 2>&1 || { tail -30 gpurun_out/dist_bench1.log; exit 1; }
tail -1 gpurun_out/dist_bench1.log
