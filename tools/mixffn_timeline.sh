#!/bin/bash
# Builds csrc/mixffn.hip with -DMF_DBG=1 as segmif_amd/lib/variants/lib_mixffn_dbg.so (travels with gpurun); on the GPU box:
#   SEGMIF_HIP_LIB=$PWD/segmif_amd/lib/variants/lib_mixffn_dbg.so python tools/mixffn_timeline.py 64      (or 128)
set -e
cd "$(dirname "$0")/.."
V=segmif_amd/lib/variants
mkdir -p $V
objs=$(ls segmif_amd/lib/obj/*.o | grep -v mixffn)
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Isegmif_amd/csrc -DMF_DBG=1 -DMF_DBG_TID=${1:-0} -c segmif_amd/csrc/mixffn.hip -o /tmp/mixffn_dbg.o
hipcc -shared -fPIC --offload-arch=gfx950 -o $V/lib_mixffn_dbg.so $objs /tmp/mixffn_dbg.o
ls -la $V/lib_mixffn_dbg.so
