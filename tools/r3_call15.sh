cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3o
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "cross_attention" 2>&1 | tail -2
bash tools/kstats.sh gpurun_out/r3o/segtrain_ks.txt python tools/train_bench.py --step seg --steps 4 --warmup 2 --train-mode | tail -1
bash tools/kstats.sh gpurun_out/r3o/fustrain_ks.txt python tools/train_bench.py --step fusion --steps 4 --warmup 2 | tail -1
head -5 gpurun_out/r3o/segtrain_ks.txt; head -5 gpurun_out/r3o/fustrain_ks.txt
