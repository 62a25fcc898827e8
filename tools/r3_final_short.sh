# last verification of the round: full gpu tests + smoke + the default bench line
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3final2; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee $out/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
timeout 600 python bench.py --steps 10 --warmup 3 > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3final2/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['whole_path_frac'])
t=d['train']; print({k:round(t[k]['ms_per_step'],1) for k in ('seg_train','fusion_train','seg_eval_regime','fusion_eval_regime')})
PY
