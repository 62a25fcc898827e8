# Training-path check, one gpurun call:   gpurun --timeout 900 -- 'bash tools/train_check.sh'
# the training tests, then both steps timed (train mode) with the table of torch ops each still issues, then the same-box A/B
# of the switches named in $AB (e.g. AB="SEGMIF_TRAIN_CONV=bf16x6").
cd $GRAFT_REPO_ROOT
out=gpurun_out/traincheck; mkdir -p $out
timeout 700 python -m pytest tests/test_gpu_round4.py tests/test_gpu_backward.py tests/test_train_golden.py tests/test_gpu_round3.py -m gpu -x -q \
  -k "add_layernorm or relu_mask or gemm_epilogue or crosspath_training or split_conv or conv_wgrad or backward or train or prelu or graphed or gradients or adamw or ssim or softmax_ce or batchnorm" > $out/pytest.txt 2>&1
tail -15 $out/pytest.txt
timeout 200 python tools/split_conv_bench.py > $out/split_conv_bench.txt 2>&1; grep -- "->" $out/split_conv_bench.txt | grep -v "{"
for st in seg fusion; do
  timeout 300 python tools/train_bench.py --step $st --steps 6 --warmup 2 --train-mode --native-sites $out/sites_$st.txt > $out/$st.json 2> $out/$st.err
  tail -2 $out/$st.err; cat $out/$st.json
done
for ab in $AB; do
  echo "== $ab"
  env $ab timeout 300 python tools/train_bench.py --step fusion --steps 6 --warmup 2 --train-mode > $out/fusion_$ab.json 2>> $out/fusion.err; cat $out/fusion_$ab.json
done
