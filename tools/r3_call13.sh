cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
SEGMIF_DWCONV_XT=4 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "dwconv" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "dwconv" 2>&1 | tail -2
B="python bench.py --steps 8 --warmup 3 --no-train --no-extras --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['hbm_bound_kernels']['dwconv']; print('$name', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],1), 'ms; dwconv', round(h['avg_launch_ms'],3), 'ms', round(h['achieved_GBps']), 'GB/s')" | tee -a gpurun_out/r3m/ab.txt; }
run xt2 SEGMIF_DWCONV_XT=2
run xt4 SEGMIF_DWCONV_XT=4
run xt2b SEGMIF_DWCONV_XT=2
run xt4b SEGMIF_DWCONV_XT=4
