# A/B of CrossPath reading the segmentation feature at its own resolution (ops.LazySeg; profiles/r05_lazy_seg_ab.txt):
#   gpurun --timeout 700 -- 'bash tools/lazy_seg_ab.sh'
# the lazy kernels' parity tests, then the forward's kernel table and bench line with the feature resized first (0) and read lazily (1).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/lazy; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_round5.py tests/test_gpu_kernels.py -q -x -k "lazy or resized or crosspath or gram" 2>&1 | tail -3
for m in 0 1; do
  SEGMIF_LAZY_SEG=$m bash tools/kstats.sh $out/kstats_$m.txt python bench.py --steps 4 --warmup 2 --no-train --no-extras --no-cpu-baseline --no-configs --no-kernel-timer > $out/bench_$m.json 2> /dev/null
  echo "SEGMIF_LAZY_SEG=$m"
  grep -E "crosspath|bilinear_kernel<4>" $out/kstats_$m.txt | cut -c1-150
  python -c "import json; d = json.loads(open('$out/bench_$m.json').read().strip().splitlines()[-1]); print(d['value'], 'pairs/s', d['ms_per_step'], 'ms')"
done
