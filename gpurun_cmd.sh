export HF_FORCE_DIST=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --swap-triples 1 2>&1 | tail -3 | cut -c1-700
