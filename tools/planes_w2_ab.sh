#!/bin/bash
# (r6) A/B of conv3x3_planes.hip's PLANES_W2REG switch (f16x3: the weights' third plane 2^-11 W0 made in registers by v_pk_mul_f16
# instead of read from LDS).  Build here (variants travel with gpurun); on the GPU box: bash tools/planes_w2_ab.sh run [batch]
set -e
cd "$(dirname "$0")/.."
V=segmif_amd/lib/variants
if [ "$1" = "run" ]; then
  for rep in 1 2; do
    for m in ${VARIANTS:-0 1}; do
      echo "== PLANES_W2REG=$m (pass $rep)"
      SEGMIF_HIP_LIB=$PWD/$V/lib_w2reg$m.so python tools/planes_bench.py --batch ${2:-16} --kernel planes16 2>/dev/null | grep -v "from_f32"
    done
  done
  exit 0
fi
mkdir -p $V
objs=$(ls segmif_amd/lib/obj/*.o | grep -v conv3x3_planes)
for m in ${VARIANTS:-0 1}; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Isegmif_amd/csrc -DPLANES_W2REG=$m -c segmif_amd/csrc/conv3x3_planes.hip -o /tmp/planes_w2reg$m.o
  hipcc -shared -fPIC --offload-arch=gfx950 -o $V/lib_w2reg$m.so $objs /tmp/planes_w2reg$m.o
done
ls -la $V
