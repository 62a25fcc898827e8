# Measurement set of round 4's final state, one gpurun call:   gpurun --timeout 1500 -- 'bash tools/r4_final.sh'
# driver-style bench line, kernel table of the forward step, kernel tables of both training steps.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4final; mkdir -p $out
S=$(date +%s); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench wall seconds: $(( $(date +%s) - S ))" > $out/bench_seconds.txt
bash tools/kstats.sh $out/kstats_forward.txt python bench.py --steps 8 --warmup 2 --no-train --no-extras --no-cpu-baseline --no-kernel-timer --no-configs > $out/bench_prof.json 2> $out/bench_prof.err
bash tools/kstats.sh $out/kstats_segtrain.txt python tools/train_bench.py --step seg --steps 6 --warmup 2 --train-mode > $out/segtrain.json 2> $out/segtrain.err
bash tools/kstats.sh $out/kstats_fusiontrain.txt python tools/train_bench.py --step fusion --steps 6 --warmup 2 --train-mode > $out/fusiontrain.json 2> $out/fusiontrain.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4final/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['whole_path_frac'], d['f16x3_guard'])
t = d['train']
print({k: round(t[k]['ms_per_step'], 1) for k in t if isinstance(t[k], dict) and 'ms_per_step' in t[k]})
print({k: (round(v['value'], 1), round(v['ms_per_step'], 1)) for k, v in d['configs'].items()})
PY
