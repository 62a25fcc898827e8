set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r3a/pytest.txt
timeout 400 python bench.py --steps 10 --warmup 3 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
tail -c 3000 gpurun_out/r3a/bench.err
bash tools/kstats.sh gpurun_out/r3a/segtrain_ks.txt python tools/train_bench.py --step seg --steps 4 --warmup 2
bash tools/kstats.sh gpurun_out/r3a/fustrain_ks.txt python tools/train_bench.py --step fusion --steps 4 --warmup 2
cat gpurun_out/r3a/pytest.txt | tail -30
