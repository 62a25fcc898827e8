#!/bin/bash
# (r6) where the fused tail's epilogue spends its time: conv3x3_planes.hip built with the PLANES_DBG epilogue bits (256 no out1 stores,
# 512 no residual, 1024 no 1x1 MFMAs on the conv's own channels, 2048 no epilogue), each also with the timeline probe (32).
# Build here; on the GPU box: bash tools/planes_tail_ablate.sh run
set -e
cd "$(dirname "$0")/.."
V=segmif_amd/lib/variants
MASKS="${MASKS:-0 256 512 1024 768 1792 2048}"
if [ "$1" = "run" ]; then
  for m in $MASKS; do
    echo "== PLANES_DBG=$m"
    SEGMIF_HIP_LIB=$PWD/$V/lib_tail$m.so python tools/planes_bench.py --batch 16 --kernel planes16 2>/dev/null | grep "fused tail"
    SEGMIF_HIP_LIB=$PWD/$V/lib_tl_tail$m.so python tools/planes_timeline.py 192 f16 tail 2>&1 | grep "whole epilogue\|item period" | head -2
  done
  exit 0
fi
mkdir -p $V
objs=$(ls segmif_amd/lib/obj/*.o | grep -v conv3x3_planes)
build() {  # name, flags
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Isegmif_amd/csrc $2 -c segmif_amd/csrc/conv3x3_planes.hip -o /tmp/planes_$1.o
  hipcc -shared -fPIC --offload-arch=gfx950 -o $V/lib_$1.so $objs /tmp/planes_$1.o
}
for m in $MASKS; do
  build tail$m "-DPLANES_DBG=$m" &
  build tl_tail$m "-DPLANES_DBG=$((m + 32))" &
  wait
done
ls $V
