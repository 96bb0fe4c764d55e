set -u
O=gpurun_out/final_r3_v3; mkdir -p $O
bash profiles/run_rocprof.sh r3_v3 > $O/rocprof.log 2>&1 || { tail -5 $O/rocprof.log; exit 1; }
echo rocprof done
timeout -k 10 300 python bench.py > $O/bench_default.json 2>$O/bench_default.err || exit 1
echo default done
for w in cat_checker cat_grid spheres; do timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>$O/bench_$w.err || exit 1; echo $w done; done
timeout -k 10 400 python bench.py --workload cats_4k --mode sample --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cats_4k.json 2>$O/bench_cats_4k.err || exit 1
echo 4k done
timeout -k 10 400 python -u tools/fuzz_parity.py 300 2 > $O/fuzz_seed2.log 2>&1; tail -1 $O/fuzz_seed2.log
