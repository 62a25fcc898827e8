cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
python tools/wgrad3_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3e/wgrad3_bench.txt
