cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3p
timeout 600 python -m pytest tests/test_gpu_backward.py -m gpu -q -x -k "conv_backward" 2>&1 | tail -2
python tools/wgrad3_bench.py 2>&1 | grep -v amdgpu | grep "fp32 \|two-team\|one-team" | tee gpurun_out/r3p/wgrad3.txt
timeout 600 python -m pytest tests/test_train_golden.py tests/test_gpu_backward.py -m gpu -q -x 2>&1 | tail -2
python tools/train_bench.py --step fusion --steps 6 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r3p/fus.json
SEGMIF_WGRAD3X3=split1 python tools/train_bench.py --step fusion --steps 6 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r3p/fus_oneteam.json
