run() { timeout 200 python bench.py --no-cpu-baseline --no-train --no-extras "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print(c['backbone'], c['pairs_per_gpu'], str(c['height'])+'x'+str(c['width']), c['launch'], round(d['value'],1), round(d['ms_per_step'],1), round(d['whole_path_tflops'],1), d['arithmetic_modes']['conv3x3'], d['arithmetic_modes']['linear'])"; }
run --backbone mit_b1 --batch 64 --steps 4 --warmup 1
run --backbone mit_b1 --batch 4 --steps 10 --warmup 2
run --backbone mit_b5 --height 1024 --width 1024 --batch 8 --steps 3 --warmup 1
run --backbone mit_b5 --height 1024 --width 1024 --batch 2 --steps 6 --warmup 2
run --batch 8 --steps 8 --warmup 2
run --batch 1 --steps 20 --warmup 3
run --batch 1 --steps 20 --warmup 3 --graph
SEGMIF_CONV3X3=fp32 SEGMIF_LINEAR=fp32 SEGMIF_CROSSPATH=gemm SEGMIF_ATTENTION=fp32 run --batch 32 --steps 3 --warmup 1
