cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_train_golden.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r3c/pytest.txt
cat gpurun_out/r3c/pytest.txt | tail -4
SEGMIF_WGRAD3X3=fp32 python tools/train_bench.py --step fusion --steps 6 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r3c/fus_fp32wgrad.json
bash tools/kstats.sh gpurun_out/r3c/fustrain_ks.txt python tools/train_bench.py --step fusion --steps 4 --warmup 2 > gpurun_out/r3c/fus.json
cat gpurun_out/r3c/fus_fp32wgrad.json gpurun_out/r3c/fus.json
head -24 gpurun_out/r3c/fustrain_ks.txt
bash tools/kstats.sh gpurun_out/r3c/segtrain_ks.txt python tools/train_bench.py --step seg --steps 4 --warmup 2
head -16 gpurun_out/r3c/segtrain_ks.txt
