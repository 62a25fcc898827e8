cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
python tools/wgrad3_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3f/wgrad3_bench.txt
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_train_golden.py tests/test_gpu_round3.py -m gpu -q -x -k "not batch_of_64 and not config1" 2>&1 | tail -8 > gpurun_out/r3f/pytest.txt
cat gpurun_out/r3f/pytest.txt | tail -4
python tools/train_bench.py --step seg --steps 8 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r3f/seg_eager.json
python tools/train_bench.py --step fusion --steps 6 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r3f/fus.json
