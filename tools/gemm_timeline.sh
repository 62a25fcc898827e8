#!/bin/bash
# Builds csrc/gemm_split.hip with -DGEMM_DBG=1 as segmif_amd/lib/variants/lib_gemm_dbg.so (travels with gpurun);
# on the GPU box:  SEGMIF_HIP_LIB=$PWD/segmif_amd/lib/variants/lib_gemm_dbg.so python tools/gemm_timeline.py M N K
set -e
cd "$(dirname "$0")/.."
V=segmif_amd/lib/variants
mkdir -p $V
objs=$(ls segmif_amd/lib/obj/*.o | grep -v gemm_split)
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Isegmif_amd/csrc -DGEMM_DBG=1 -c segmif_amd/csrc/gemm_split.hip -o /tmp/gemm_dbg.o
hipcc -shared -fPIC --offload-arch=gfx950 -o $V/lib_gemm_dbg.so $objs /tmp/gemm_dbg.o
ls -la $V/lib_gemm_dbg.so
