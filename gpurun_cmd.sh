# GPU call r06bk: the driver's multi-GPU launch form at world size 1 (torch.distributed.run, RCCL) with the final bench.py
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r06bk_torchrun_bench.json 2> gpurun_out/r06bk_torchrun_bench.err; echo "rc=$?"
tail -c 500 gpurun_out/r06bk_torchrun_bench.json; tail -3 gpurun_out/r06bk_torchrun_bench.err | cut -c1-200
