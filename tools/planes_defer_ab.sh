#!/bin/bash
# (r6) A/B of the round's conv3x3_planes.hip changes against the file as it stood before them (git $BASE):
#   PLANES_DEFER (plain f16x3 four-sub-tile conv: the patch's plane stores issued in the team's next LOAD phase) and the fused
#   tail's batched epilogue (1x1 weights of its own channels resident in LDS, residual pieces requested a sub-tile ahead).
# Build here (variants travel with gpurun); on the GPU box:
#   bash tools/planes_defer_ab.sh run [batch]      -> tools/planes_bench.py per variant, two interleaved passes
#   bash tools/planes_defer_ab.sh timeline         -> tools/planes_timeline.py per timeline variant
set -e
cd "$(dirname "$0")/.."
V=segmif_amd/lib/variants
BASE=${BASE:-3da927b}
VARS="${VARIANTS:-base new}"
if [ "$1" = "run" ]; then
  for rep in 1 2; do
    for m in $VARS; do
      echo "== $m (pass $rep)"
      SEGMIF_HIP_LIB=$PWD/$V/lib_$m.so python tools/planes_bench.py --batch ${2:-16} --kernel planes16 2>/dev/null | grep "^dcov\|planes16 fused\|DRDB planes16"
    done
  done
  exit 0
fi
if [ "$1" = "timeline" ]; then
  for m in $VARS; do
    for a in "128 f16" "192 f16 tail"; do
      echo "== $m: $a"
      SEGMIF_HIP_LIB=$PWD/$V/lib_tl_$m.so python tools/planes_timeline.py $a 2>&1 | grep -v amdgpu.ids
    done
  done
  exit 0
fi
mkdir -p $V
objs=$(ls segmif_amd/lib/obj/*.o | grep -v conv3x3_planes)
mkdir -p /tmp/planes_base; for f in conv3x3_planes.hip planes16.h igemm_common.h device_once.h; do git show $BASE:segmif_amd/csrc/$f > /tmp/planes_base/$f; done  # (the old file with ITS OWN headers: quote-includes resolve next to it)
build() {  # name, source, flags
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Isegmif_amd/csrc $3 -c $2 -o /tmp/planes_$1.o
  hipcc -shared -fPIC --offload-arch=gfx950 -o $V/lib_$1.so $objs /tmp/planes_$1.o
}
build base /tmp/planes_base/conv3x3_planes.hip "" &
build new segmif_amd/csrc/conv3x3_planes.hip "" &
build tl_base /tmp/planes_base/conv3x3_planes.hip "-DPLANES_DBG=32" &
build tl_new segmif_amd/csrc/conv3x3_planes.hip "-DPLANES_DBG=32" &
wait
ls -la $V
