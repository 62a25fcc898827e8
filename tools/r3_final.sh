# round-3 validation + measurement pass (run through gpurun): full gpu tests, default bench line, kernel stats, PMC passes
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3final; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $out/pytest.txt; tail -3 $out/pytest.txt
timeout 600 python bench.py --steps 10 --warmup 3 > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.err
bash tools/kstats.sh $out/bench_ks.txt python bench.py --steps 4 --warmup 2 --no-train --no-extras --no-cpu-baseline --no-kernel-timer > /dev/null
head -30 $out/bench_ks.txt
bash tools/pmc_traffic.sh $out/pmc python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-train --no-extras > $out/pmc.log 2>&1; head -6 $out/pmc/pmc_FETCH_SIZE.txt $out/pmc/pmc_WRITE_SIZE.txt
bash tools/pmc_sq.sh $out/pmc_sq.txt python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-train --no-extras > /dev/null 2>&1; head -14 $out/pmc_sq.txt
bash tools/other_configs.sh 2>&1 | tee $out/other_configs.txt
