cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3q
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_train_golden.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
python tools/train_bench.py --step fusion --steps 6 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r3q/fus.json
SEGMIF_WGRAD3X3=split1 python tools/train_bench.py --step fusion --steps 6 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r3q/fus_split1.json
