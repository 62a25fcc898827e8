#!/bin/bash
# Same-box A/B of bench.py's forward leg: tools/ab_bench.sh <outdir> "<env A>" "<env B>" [rounds]   (boxes of the pool differ
# by up to ~5 % on identical code, so variants are only comparable inside one gpurun call; A and B alternate)
out=$1; A=$2; B=$3; R=${4:-2}
mkdir -p $out
for r in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then e="$A"; else e="$B"; fi
    env $e timeout 200 python bench.py --steps 6 --warmup 2 --no-train --no-extras --no-cpu-baseline --no-configs > $out/bench_${v}_$r.json 2> $out/bench_${v}_$r.err
  done
done
python - "$out" "$A" "$B" "$R" <<'PY'
import json, sys
out, A, B, R = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
for v, e in (("A", A), ("B", B)):
    vals = []
    for r in range(1, R + 1):
        try:
            d = json.loads(open(f"{out}/bench_{v}_{r}.json").read().strip().splitlines()[-1])
            vals.append((d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
        except Exception as exc:
            vals.append(("error", str(exc)))
    print(v, repr(e), vals)
PY
