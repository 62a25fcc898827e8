#!/bin/bash
# One rocprofv3 SQ-counter pass (+ kernel trace) over a command; per-kernel table -> $1 (text).
# usage: tools/pmc_sq.sh <out.txt> <command...>      (run on the GPU box, through gpurun)
out=$1; shift
export TMPDIR=/tmp
d=/tmp/sq_$$; rm -rf $d; mkdir -p $d
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --kernel-trace --output-format rocpd -d $d -o p -- "$@" > $d/run.log 2>&1
db=$(find $d -name '*.db' | head -1)
python "$(dirname "$0")/rocpd_sq.py" "$db" "$out" > /dev/null
tail -5 $d/run.log
