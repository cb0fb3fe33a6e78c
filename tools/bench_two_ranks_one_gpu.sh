# the N = 2 bench path on a ONE-GPU box: both ranks on cuda:0, collectives over gloo (tools: not what the driver runs)
export PYSLAM_BENCH_ONE_GPU=1 PYSLAM_BENCH_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 2 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -5 | cut -c1-1500
