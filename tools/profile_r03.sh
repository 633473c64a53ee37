# rocprofv3 evidence of round 3 (run on the MI355X box through gpurun; outputs under gpurun_out/r03prof):
#   bench lines of C2 / C4 / C5; kernel-trace --stats of bench.py (C2) and of the basis-stage calls (rom_pod, rom_greedy);
#   PMC passes (FETCH_SIZE, WRITE_SIZE, one counter group per pass, kernel-trace only: gpurun's rule) of the C2 sweep and
#   of the basis stage
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03prof
if [ "$1" = "c45only" ]; then mkdir -p $O; set -- c45; else
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 500 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout -k 10 400 python bench.py --config c4 > $O/bench_c4.json 2> $O/bench_c4.err
timeout -k 10 500 python bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/bench_c2_under_rocprof.json 2> $O/stats.err
find $O/stats -name "*kernel_stats.csv" | tail -1 | xargs -I{} cp {} $O/bench_c2_kernel_stats.csv
rm -rf $O/stats
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_$c.err
  find $O/pmc_$c -name "*counter_collection.csv" | tail -1 | xargs -I{} cp {} $O/bench_c2_$c.csv
  rm -rf $O/pmc_$c
done
python3 $R/tools/pmc_summary.py $O/bench_c2_FETCH_SIZE.csv $O/bench_c2_WRITE_SIZE.csv $O/pmc_traffic.json > /dev/null
# the basis stage: rom_pod (C2 and C3 size) and rom_greedy (C4 size, both modes)
timeout -k 10 300 python3 $R/tools/gpu_basis_profile.py all > $O/basis_stage_hip_events.txt 2> $O/basis.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bstats -- python3 $R/tools/gpu_basis_profile.py all > /dev/null 2> $O/bstats.err
find $O/bstats -name "*kernel_stats.csv" | tail -1 | xargs -I{} cp {} $O/basis_stage_kernel_stats.csv
rm -rf $O/bstats
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/bp_$c -- python3 $R/tools/gpu_basis_profile.py greedy > /dev/null 2> $O/bp_$c.err
  find $O/bp_$c -name "*counter_collection.csv" | tail -1 | xargs -I{} cp {} $O/basis_greedy_$c.csv
  rm -rf $O/bp_$c
done
python3 $R/tools/pmc_summary.py $O/basis_greedy_FETCH_SIZE.csv $O/basis_greedy_WRITE_SIZE.csv $O/basis_greedy_pmc_traffic.json > /dev/null
rm -f $O/*.err
ls -la $O
fi
# PMC traffic of the C4 / C5 sweeps (one launch = the whole sweep: ROMHC_STREAMS=1, as in the bench's per-kernel pass)
if [ "$1" = "c45" ] || [ "$1" = "all" ]; then
cd /tmp
for cfg in c4 c5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ROMHC_STREAMS=1 timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p45_$c -- python3 $R/bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/p45.err
    find $O/p45_$c -name "*counter_collection.csv" | tail -1 | xargs -I{} cp {} $O/p45_$c.csv
    rm -rf $O/p45_$c
  done
  python3 $R/tools/pmc_summary.py $O/p45_FETCH_SIZE.csv $O/p45_WRITE_SIZE.csv $O/pmc_traffic_$cfg.json > /dev/null
  rm -f $O/p45_*.csv $O/p45.err
done
fi
