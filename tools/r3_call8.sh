cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_train_golden.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r3h/pytest.txt
cat gpurun_out/r3h/pytest.txt | tail -4
python tools/train_bench.py --step seg --steps 8 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r3h/seg.json
SEGMIF_LINEAR=fp32 python tools/train_bench.py --step seg --steps 8 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r3h/seg_linear_fp32.json
python tools/train_bench.py --step fusion --steps 6 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r3h/fus.json
SEGMIF_LINEAR=fp32 python tools/train_bench.py --step fusion --steps 6 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r3h/fus_linear_fp32.json
