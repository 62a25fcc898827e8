# last verification of the round (f16x3 default): full gpu tests + smoke + the default bench line + its kernel-trace summary
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3final3; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error|^E " | tail -6 | tee $out/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
timeout 600 python bench.py --steps 10 --warmup 3 > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3final3/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['whole_path_frac'], 'fallbacks', d['f16x3_range_fallbacks'])
print('bf16x6:', d.get('with_bf16x6_planes',{}).get('value'), 'dse:', d.get('without_discarded_encoder_stages',{}).get('value'))
t=d['train']; print({k:round(t[k]['ms_per_step'],1) for k in ('seg_train','fusion_train','seg_eval_regime','fusion_eval_regime') if k in t} if 'error' not in t else t)
PY
bash tools/kstats.sh $out/bench_ks.txt python bench.py --steps 4 --warmup 2 --no-train --no-extras --no-cpu-baseline --no-kernel-timer > /dev/null
head -8 $out/bench_ks.txt | cut -c1-150
