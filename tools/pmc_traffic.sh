#!/bin/bash
# Two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass) + kernel trace over a command;
# per-kernel averages -> <outdir>/pmc_FETCH_SIZE.txt, pmc_WRITE_SIZE.txt.   usage: tools/pmc_traffic.sh <outdir> <command...>
out=$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_$c_$$; rm -rf $d; mkdir -p $d
  rocprofv3 --pmc $c --kernel-trace --output-format rocpd -d $d -o pmc -- "$@" > $d/run.log 2>&1
  db=$(find $d -name '*.db' | head -1)
  python "$(dirname "$0")/rocpd_pmc.py" "$db" "$out/pmc_$c.txt" > /dev/null
done
head -8 "$out/pmc_FETCH_SIZE.txt" "$out/pmc_WRITE_SIZE.txt"
