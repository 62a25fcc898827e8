# quick GPU check of the training convs: their tests, the micro-benchmark, the fusion step timed with both arithmetics
cd $GRAFT_REPO_ROOT
out=gpurun_out/convquick; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_round4.py tests/test_gpu_backward.py -m gpu -x -q -k "split_conv or conv_wgrad or drdb or fusion_network" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
timeout 200 python tools/split_conv_bench.py > $out/split_conv_bench.txt 2>&1; grep -- "->" $out/split_conv_bench.txt | grep -v "{"
for ab in SEGMIF_TRAIN_CONV=f16x3 SEGMIF_TRAIN_CONV=bf16x6; do
  env $ab timeout 300 python tools/train_bench.py --step fusion --steps 6 --warmup 2 --train-mode > $out/fusion_$ab.json 2>> $out/fusion.err; echo "$ab: $(cat $out/fusion_$ab.json | python -c 'import sys,json; print(json.load(sys.stdin)["ms_per_step"])')"
done
