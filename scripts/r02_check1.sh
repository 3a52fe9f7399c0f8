#!/bin/bash
# round-2 first GPU pass: RCCL probe, new parity tests, bench (N=1) and the 2-rank dry run
mkdir -p gpurun_out/r02a
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export HSA_ENABLE_IPC_MODE_LEGACY=0
nproc > gpurun_out/r02a/host.txt; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/r02a/host.txt; free -g | head -2 >> gpurun_out/r02a/host.txt
timeout 180 python scripts/nccl_try.py 1 > gpurun_out/r02a/nccl_try_w1.log 2>&1
timeout 180 python scripts/nccl_try.py 2 > gpurun_out/r02a/nccl_try_w2.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_golden_wrappers.py tests/test_gpu_multirank.py tests/test_gpu_zbaseline.py -m gpu -q -s -x --durations=10 > gpurun_out/r02a/pytest_new.log 2>&1
echo "pytest rc $?" >> gpurun_out/r02a/pytest_new.log
timeout 600 python bench.py > gpurun_out/r02a/bench.log 2>&1
echo "bench rc $?" >> gpurun_out/r02a/bench.log
VSR_BENCH_DRYRUN_1GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r02a/bench_dry2.log 2>&1
echo "dry rc $?" >> gpurun_out/r02a/bench_dry2.log
tail -5 gpurun_out/r02a/pytest_new.log; tail -c 600 gpurun_out/r02a/bench.log; tail -c 400 gpurun_out/r02a/bench_dry2.log; cat gpurun_out/r02a/nccl_try_w2.log | tail -5
