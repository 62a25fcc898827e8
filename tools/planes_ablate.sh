#!/bin/bash
# Builds ablation variants of csrc/conv3x3_planes.hip (PLANES_DBG mask) as alternative libsegmif_hip.so files
# under segmif_amd/lib/variants/ (they travel with gpurun); run on the GPU box with `tools/planes_ablate.sh run`.
set -e
cd "$(dirname "$0")/.."
V=segmif_amd/lib/variants
MASKS="${MASKS:-0 1 2 3 4 8 12}"
if [ "$1" = "run" ]; then
  for m in $MASKS; do
    echo "== PLANES_DBG=$m"
    SEGMIF_HIP_LIB=$PWD/$V/lib_dbg$m.so python tools/planes_bench.py --batch 8 --only 128 --kernel planes "${@:2}" 2>/dev/null
  done
  exit 0
fi
mkdir -p $V
objs=$(ls segmif_amd/lib/obj/*.o | grep -v conv3x3_planes)
for m in $MASKS; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -DPLANES_DBG=$m -c segmif_amd/csrc/conv3x3_planes.hip -o /tmp/planes_dbg$m.o
  hipcc -shared -fPIC --offload-arch=gfx950 -o $V/lib_dbg$m.so $objs /tmp/planes_dbg$m.o
done
ls -la $V
