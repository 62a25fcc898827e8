cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3n
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm_split" 2>&1 | tail -2
SEGMIF_GEMM_EPI=direct timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm_split" 2>&1 | tail -2
python tools/gemm_epi_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r3n/gemm_epi.txt
B="python bench.py --steps 8 --warmup 3 --no-train --no-extras --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],1), 'ms')" | tee -a gpurun_out/r3n/ab.txt; }
run direct SEGMIF_GEMM_EPI=direct
run lds SEGMIF_GEMM_EPI=lds
run direct2 SEGMIF_GEMM_EPI=direct
run lds2 SEGMIF_GEMM_EPI=lds
