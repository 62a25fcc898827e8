#!/bin/bash
# One rocprofv3 counter pass (+ kernel trace) with a caller-chosen counter list; per-kernel table -> $1 (text).
# usage: tools/pmc_any.sh <out.txt> "<CTR1 CTR2 ...>" <command...>      (run on the GPU box, through gpurun)
out=$1; ctrs=$2; shift; shift
export TMPDIR=/tmp
d=/tmp/pa_$$; rm -rf $d; mkdir -p $d
rocprofv3 --pmc $ctrs --kernel-trace --output-format rocpd -d $d -o p -- "$@" > $d/run.log 2>&1
db=$(find $d -name '*.db' | head -1)
python "$(dirname "$0")/rocpd_sq.py" "$db" "$out" > /dev/null
tail -3 $d/run.log
