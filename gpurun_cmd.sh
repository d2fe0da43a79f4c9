cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/bench_enc_layers.py > /dev/null 2>&1
(echo "# tools/bench_enc_layers.py - the encoders' / PostProcess's 3x3 layers at the batch sizes of one swap (r02f)"; python tools/bench_enc_layers.py 2>&1 | grep -v amdgpu; echo; echo "# the same layers with 8 triples per pass (HairFast.swap_batch): ENC_BATCH_MULT=8"; ENC_BATCH_MULT=8 python tools/bench_enc_layers.py 2>&1 | grep -v amdgpu) > gpurun_out/r02f_encoder_layers.txt
python tools/bench_encoders.py 2>&1 | grep -v amdgpu | tail -12 >> gpurun_out/r02f_encoder_layers.txt
tail -30 gpurun_out/r02f_encoder_layers.txt
python bench.py --steps 20 --warmup 3 > gpurun_out/r02f_bench.json 2>gpurun_out/r02f_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r02f_bench.json')); print(d['value'], d['swap_pipeline'])"
