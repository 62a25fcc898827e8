cd $GRAFT_REPO_ROOT
timeout 150 python -m pytest tests/test_gpu_planes16.py -q -x 2>&1 | grep -E "passed|failed|Error|^E " | tail -8
timeout 100 python bench.py --graph --batch 1 --steps 20 --warmup 3 --no-train --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('graph b1:', d['value'], d['ms_per_step'], d['config']['launch'], d['f16x3_range_fallbacks'])"
