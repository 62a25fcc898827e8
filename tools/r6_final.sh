# Measurement set of round 6's final state, one gpurun call:   gpurun --timeout 2400 -- 'bash tools/r6_final.sh'
# GPU tests + smoke, driver-style bench line, kernel tables (forward, both training steps), SQ counters and HBM traffic of the forward.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r6final; mkdir -p $out
{ echo "# pytest tests -m gpu + smoke on this HEAD (tools/r6_final.sh)"; timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4; timeout 120 python __graft_entry__.py smoke 2>&1 | tail -2; } > $out/gpu_tests.txt
S=$(date +%s); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench wall seconds: $(( $(date +%s) - S ))" > $out/bench_seconds.txt
bash tools/kstats.sh $out/kstats_forward.txt python bench.py --steps 8 --warmup 2 --no-train --no-extras --no-cpu-baseline --no-kernel-timer --no-configs > $out/bench_prof.json 2> $out/bench_prof.err
bash tools/kstats.sh $out/kstats_segtrain.txt python tools/train_bench.py --step seg --steps 6 --warmup 2 --train-mode > $out/segtrain.json 2> $out/segtrain.err
bash tools/kstats.sh $out/kstats_fusiontrain.txt python tools/train_bench.py --step fusion --steps 6 --warmup 2 --train-mode > $out/fusiontrain.json 2> $out/fusiontrain.err
bash tools/pmc_sq.sh $out/pmc_sq_counters.txt python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-train --no-extras --no-configs > /dev/null 2>&1
bash tools/pmc_traffic.sh $out python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-train --no-extras --no-configs > /dev/null 2>&1
cat $out/gpu_tests.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r6final/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['whole_path_frac'], d['f16x3_guard'])
t = d['train']
print({k: round(t[k]['ms_per_step'], 1) for k in t if isinstance(t[k], dict) and 'ms_per_step' in t[k]})
print({k: (round(v['value'], 1), round(v['ms_per_step'], 1), v.get('f16x3_pairs_repeated_fp32conv')) for k, v in d['configs'].items()})
PY
head -12 $out/kstats_forward.txt | cut -c1-150
grep -E "gemm_pairs|conv3x3_planes|mixffn|crosspath_tail" $out/pmc_sq_counters.txt | cut -c1-220 | head -12
