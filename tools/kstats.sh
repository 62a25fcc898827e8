#!/bin/bash
# rocprofv3 kernel trace of a command -> per-kernel stats table (text) at $1.   usage: tools/kstats.sh <out.txt> <command...>
out=$1; shift
export TMPDIR=/tmp
d=/tmp/ks_$$; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --output-format rocpd -d $d -o st -- "$@" > $d/run.log 2>&1
db=$(find $d -name '*.db' | head -1)
python "$(dirname "$0")/rocpd_stats.py" "$db" "$out" > /dev/null
grep -v "^W2026\|^E2026\|simple_timer" $d/run.log | tail -3
