# First GPU call for the four-sub-tile f16x3 conv (written in round 3, not yet run): its opt-in parity test, the
# kernel-level A/B against the two-sub-tile kernel (separate processes: the variable is read once), the bench line with it.
#   gpurun --timeout 400 -- 'bash tools/sub4_ab.sh'
cd $GRAFT_REPO_ROOT
out=gpurun_out/sub4; mkdir -p $out
SEGMIF_PLANES_SUB=4 timeout 120 python -m pytest tests/test_gpu_planes16.py -q -x -k four_subtiles 2>&1 | grep -E "passed|failed|Error|^E " | tail -6 | tee $out/pytest.txt
timeout 90 python tools/planes_bench.py --kernel planes16 --batch 16 2>&1 | grep -E "^dcov" | tee $out/sub2.txt
SEGMIF_PLANES_SUB=4 timeout 90 python tools/planes_bench.py --kernel planes16 --batch 16 2>&1 | grep -E "^dcov" | tee $out/sub4.txt
SEGMIF_PLANES_SUB=4 timeout 150 python bench.py --steps 5 --warmup 2 --no-train --no-extras --no-cpu-baseline > $out/bench_sub4.json 2> $out/bench_sub4.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/sub4/bench_sub4.json').read().strip().splitlines()[-1])
print('SUB=4:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY
