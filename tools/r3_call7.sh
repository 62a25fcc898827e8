cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_train_golden.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r3g/pytest.txt
cat gpurun_out/r3g/pytest.txt | tail -4
python tools/wgrad3_bench.py 2>&1 | grep -v amdgpu.ids | grep "fp32 \|bf16x6\|mfma-only" | tee gpurun_out/r3g/wgrad3_bench.txt
SEGMIF_WGRAD=fp32 python tools/train_bench.py --step seg --steps 8 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r3g/seg_fp32wgrad.json
python tools/train_bench.py --step seg --steps 8 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r3g/seg.json
SEGMIF_WGRAD=fp32 python tools/train_bench.py --step fusion --steps 6 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r3g/fus_fp32wgrad.json
python tools/train_bench.py --step fusion --steps 6 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r3g/fus.json
bash tools/kstats.sh gpurun_out/r3g/segtrain_ks.txt python tools/train_bench.py --step seg --steps 4 --warmup 2 > /dev/null
head -22 gpurun_out/r3g/segtrain_ks.txt
