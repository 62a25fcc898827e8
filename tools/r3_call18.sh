cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3r
bash tools/pmc_sq.sh gpurun_out/r3r/pmc_sq_fusiontrain.txt python tools/train_bench.py --step fusion --steps 2 --warmup 1 > /dev/null 2>&1
head -14 gpurun_out/r3r/pmc_sq_fusiontrain.txt | cut -c1-64,150-400
bash tools/kstats.sh gpurun_out/r3r/fustrain_ks.txt python tools/train_bench.py --step fusion --steps 4 --warmup 2 | tail -1 | cut -c1-200
head -12 gpurun_out/r3r/fustrain_ks.txt
bash tools/pmc_sq.sh gpurun_out/r3r/pmc_sq_segtrain.txt python tools/train_bench.py --step seg --steps 2 --warmup 1 --train-mode > /dev/null 2>&1
head -8 gpurun_out/r3r/pmc_sq_segtrain.txt | cut -c1-64,150-400
