#!/bin/bash
# Build ablated variants of conv3x3_split.hip (-DSPLIT_DBG=<mask>) into segmif_amd/lib/dbg/ for
# bottleneck hunting: 1 = no MFMAs, 2 = no LDS fragment reads, 4 = no operand split, 8 = no global
# loads after the first chunk, 16 = no LDS stores after the first chunk (results are wrong by
# construction; only timings mean anything); 32 = s_memtime timeline probe (tools/split_timeline.py),
# 64 = pad the LDS request so one workgroup fits per CU, 128 = 16-row patches for 32 output channels.
# An argument is a mask or name:mask (e.g. tl:32, solo:64).  Usage: tools/split_ablate.sh 0 1 2 ...; then
#   SEGMIF_HIP_LIB=segmif_amd/lib/dbg/libsegmif_hip_<mask>.so python tools/kernel_bench.py --drdb-only 128
set -e
cd "$(dirname "$0")/.."
python -m segmif_amd.build >/dev/null
mkdir -p segmif_amd/lib/dbg
for spec in "$@"; do
  m=${spec%%:*}; mask=${spec##*:}; [ "$mask" = "$spec" ] && mask=$m   # "name:mask" or just "mask"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -DSPLIT_DBG=$mask ${SPLIT_EXTRA} -c segmif_amd/csrc/conv3x3_split.hip -o segmif_amd/lib/dbg/split_$m.o
  objs=$(ls segmif_amd/lib/obj/*.o | grep -v conv3x3_split)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o segmif_amd/lib/dbg/libsegmif_hip_$m.so $objs segmif_amd/lib/dbg/split_$m.o
done
