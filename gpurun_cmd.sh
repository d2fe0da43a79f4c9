# GPU call r06ak: walking blocks take consecutive tiles (hip) vs strided (wc0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
for v in wc0 hip wc0 hip; do echo "== $v"; HAIRFAST_HIP_LIB=$C/libhairfast_$v.so python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 40 --warmup 5 --no-kernel-events | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done
for v in wc0 hip; do echo "== $v"; HAIRFAST_HIP_LIB=$C/libhairfast_$v.so PROBE_TUNE=0 python tools/probes/gen_layers.py 2>&1 | grep "same\|upfu\|up2p 512"; done
