cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_train_golden.py tests/test_gpu_round3.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r3b/pytest.txt
cat gpurun_out/r3b/pytest.txt | tail -5
python tools/train_bench.py --step seg --steps 6 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r3b/seg.json
python tools/train_bench.py --step fusion --steps 6 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r3b/fus.json
cat gpurun_out/r3b/seg.json gpurun_out/r3b/fus.json
B="python bench.py --steps 6 --warmup 2 --no-train --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],1), 'ms')" | tee -a gpurun_out/r3b/ab.txt; }
run base X=1
run mlp64 SEGMIF_MLP_GROUP_MB=64
run mlp128 SEGMIF_MLP_GROUP_MB=128
run stage_4_8 SEGMIF_STAGE_GROUPS=4,8,0,0
run stage_8_16 SEGMIF_STAGE_GROUPS=8,16,0,0
run stage_8_16_32 SEGMIF_STAGE_GROUPS=8,16,32,0
run stage_16_32 SEGMIF_STAGE_GROUPS=16,32,0,0
run base2 X=1
