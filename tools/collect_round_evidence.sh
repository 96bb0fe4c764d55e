# Round-end evidence on the GPU box, one call: the rocprofv3 passes of the default workload (profiles/run_rocprof.sh), the default bench line
# with its CPU baseline, the other four workloads, and a 300 s parity campaign -> gpurun_out/final_r3_v3/ and gpurun_out/prof_r3_v3/
# (summarised into profiles/ by tools/summarize_prof.py).  usage: gpurun --timeout 1200 -- bash tools/collect_round_evidence.sh
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
