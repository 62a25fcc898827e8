#!/bin/bash
# Builds csrc/attention_split.hip with -DATTN_ABL=<mask> for each mask given (default: the study's set) as segmif_amd/lib/variants/lib_attn_abl<mask>.so
# (travels with gpurun); on the GPU box:   bash tools/attn_ablate.sh run   times the stage-3 / config[4] shapes under each variant.
set -e
cd "$(dirname "$0")/.."
V=segmif_amd/lib/variants
MASKS="0 1 17 2 4 8 12 14 32 63"
if [ "$1" == "run" ]; then
  for m in $MASKS; do
    echo "== ATTN_ABL=$m"; SEGMIF_HIP_LIB=$PWD/$V/lib_attn_abl$m.so python tools/attn_ablate.py 2>&1 | grep -v amdgpu.ids
  done
  exit 0
fi
mkdir -p $V
objs=$(ls segmif_amd/lib/obj/*.o | grep -v attention_split)
for m in $MASKS; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Isegmif_amd/csrc -DATTN_ABL=$m -c segmif_amd/csrc/attention_split.hip -o /tmp/attn_abl$m.o
  hipcc -shared -fPIC --offload-arch=gfx950 -o $V/lib_attn_abl$m.so $objs /tmp/attn_abl$m.o
done
ls -la $V/
