cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 ORBFE_BENCH_SHARE_DEVICE=1 ORBFE_BENCH_BACKEND=gloo
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --frames 16 2>&1 | tail -8 | cut -c1-900
