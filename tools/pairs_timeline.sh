#!/bin/bash
# Builds csrc/gemm_pairs.hip with -DPAIRS_DBG=1 as segmif_amd/lib/variants/lib_pairs_dbg.so (travels with gpurun; delete it afterwards);
# on the GPU box:  SEGMIF_HIP_LIB=$PWD/segmif_amd/lib/variants/lib_pairs_dbg.so python tools/pairs_timeline.py M N K
set -e
cd "$(dirname "$0")/.."
V=segmif_amd/lib/variants
mkdir -p $V
objs=$(ls segmif_amd/lib/obj/*.o | grep -v gemm_pairs)
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Isegmif_amd/csrc -DPAIRS_DBG=1 -c segmif_amd/csrc/gemm_pairs.hip -o /tmp/pairs_dbg.o
hipcc -shared -fPIC --offload-arch=gfx950 -o $V/lib_pairs_dbg.so $objs /tmp/pairs_dbg.o
ls -la $V/lib_pairs_dbg.so
