# the self-scaled f16x3 GEMM of the training path (experimental switch SEGMIF_TRAIN_GEMM=f16x3): its unit test, the segmentation step with and
# without it, and the segmentation training golden under it
cd $GRAFT_REPO_ROOT
out=gpurun_out/train_gemm_ab; mkdir -p $out
timeout 60 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "self_scaled" > $out/pytest_unit.txt 2>&1; tail -2 $out/pytest_unit.txt
for m in f16x3 fp32; do
  SEGMIF_TRAIN_GEMM=$m timeout 60 python tools/train_bench.py --step seg --steps 4 --warmup 2 > $out/seg_$m.json 2> $out/seg_$m.err; echo "$m: $(python -c "import json;print(round(json.load(open('$out/seg_$m.json'))['ms_per_step'],1))" 2>/dev/null || tail -1 $out/seg_$m.err)"
done
SEGMIF_TRAIN_GEMM=f16x3 timeout 60 python -m pytest tests/test_train_golden.py -m gpu -x -q -k "seg_train_step_three" > $out/pytest_golden.txt 2>&1; tail -2 $out/pytest_golden.txt
