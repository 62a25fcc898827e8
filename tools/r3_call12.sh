cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round3.py tests/test_gpu_backward.py -m gpu -q -x -k "dwconv" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_modules.py -m gpu -q -x -k "mit_b0 or mit_blocks or pair_b1" 2>&1 | tail -3
B="python bench.py --steps 8 --warmup 3 --no-train --no-extras --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['hbm_bound_kernels']['dwconv']; print('$name', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],1), 'ms; dwconv', round(h['avg_launch_ms'],3), 'ms', round(h['achieved_GBps']), 'GB/s')" | tee -a gpurun_out/r3l/ab.txt; }
run x1 SEGMIF_DWCONV_X1=1
run x2 X=1
run x1b SEGMIF_DWCONV_X1=1
run x2b X=1
