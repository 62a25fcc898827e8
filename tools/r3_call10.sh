cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
timeout 300 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "stencil" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_modules.py -m gpu -q -x -k "pair_b1 or fusion_net_in_all or fusion_blocks" 2>&1 | tail -4
python bench.py --steps 8 --warmup 3 --no-train --no-cpu-baseline > gpurun_out/r3j/bench_fwd.json 2>gpurun_out/r3j/bench_fwd.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3j/bench_fwd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('whole_path_frac'), d['roofline']['frac'])
print(json.dumps(d.get('hbm_bound_kernels'), indent=1))
PY
python tools/cpu_threads_probe.py --full 4 8 12 16 24 32 2>&1 | grep -v amdgpu | tee gpurun_out/r3j/cpu_threads_full.txt
