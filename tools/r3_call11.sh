cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round3.py -m gpu -q -x -k "crosspath or stencil or discarded or bilinear or upsum" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_modules.py -m gpu -q -x -k "not b5 and not batch_consistency" 2>&1 | tail -4
python tools/crosspath_bench.py 2>&1 | grep -v amdgpu | tail -12 | tee gpurun_out/r3k/crosspath_bench.txt
python bench.py --steps 8 --warmup 3 --no-train --no-cpu-baseline > gpurun_out/r3k/bench_fwd.json 2>gpurun_out/r3k/bench_fwd.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3k/bench_fwd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('whole_path_frac'), d['roofline']['frac'])
print(d.get('without_discarded_encoder_stages'))
print({k:(round(v['achieved_GBps']),round(v['avg_launch_ms'],3)) for k,v in d.get('hbm_bound_kernels',{}).items()})
PY
