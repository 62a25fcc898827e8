cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_train_golden.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r3d/pytest.txt
cat gpurun_out/r3d/pytest.txt | tail -4
python tools/train_bench.py --step seg --steps 8 --warmup 2 2>&1 | tail -1 > gpurun_out/r3d/seg_eager.json
python tools/train_bench.py --step seg --steps 8 --warmup 2 --graph 2>&1 | tail -3 > gpurun_out/r3d/seg_graph.json
python tools/train_bench.py --step seg --steps 8 --warmup 2 --graph --train-mode 2>&1 | tail -3 > gpurun_out/r3d/seg_graph_train.json
cat gpurun_out/r3d/seg_eager.json gpurun_out/r3d/seg_graph.json gpurun_out/r3d/seg_graph_train.json
SEGMIF_WGRAD3X3=fp32 python tools/train_bench.py --step fusion --steps 6 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r3d/fus_fp32wgrad.json
bash tools/kstats.sh gpurun_out/r3d/fustrain_ks.txt python tools/train_bench.py --step fusion --steps 4 --warmup 2 > gpurun_out/r3d/fus.json
cat gpurun_out/r3d/fus_fp32wgrad.json gpurun_out/r3d/fus.json
head -14 gpurun_out/r3d/fustrain_ks.txt
