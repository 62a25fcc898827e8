cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_train_golden.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r3i/pytest.txt
cat gpurun_out/r3i/pytest.txt | tail -4
python tools/train_bench.py --step seg --steps 8 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r3i/seg.json
python tools/train_bench.py --step fusion --steps 6 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r3i/fus.json
bash tools/kstats.sh gpurun_out/r3i/segtrain_ks.txt python tools/train_bench.py --step seg --steps 4 --warmup 2 > /dev/null
head -26 gpurun_out/r3i/segtrain_ks.txt
bash tools/kstats.sh gpurun_out/r3i/fustrain_ks.txt python tools/train_bench.py --step fusion --steps 4 --warmup 2 > /dev/null
head -30 gpurun_out/r3i/fustrain_ks.txt
