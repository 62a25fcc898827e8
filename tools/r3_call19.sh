cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_train_golden.py tests/test_gpu_round3.py -m gpu -q -x -k "not batch_of_64 and not config1" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
python tools/train_bench.py --step fusion --steps 8 --warmup 2 2>&1 | tail -1 | cut -c1-220
python tools/train_bench.py --step fusion --steps 8 --warmup 2 2>&1 | tail -1 | cut -c1-220
