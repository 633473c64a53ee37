# rocprofv3 evidence of round 5 (run on the MI355X box through gpurun; outputs under gpurun_out/r05prof):
#   the default bench line (C2 + C4 / C5 legs with the factored and api_* records); kernel-trace --stats of bench.py (C2, C4);
#   PMC passes (FETCH_SIZE, WRITE_SIZE: one counter per pass, kernel-trace only) of the C2 / C4 / C5 sweeps, summarised with the
#   hash of the kernel sources (bench.py withholds `traffic` taken on another build); HIP events of the basis-stage calls;
#   kernel timeline of rom_pod at the C2 geometry (rows and factored)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05prof
rm -rf $O; mkdir -p $O
cd $R
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --preroll 0 --no-cpu-baseline --no-extras --no-other-configs > /dev/null 2> $O/pmc_$c.err
  find $O/pmc_$c -name "*counter_collection.csv" | tail -1 | xargs -I{} cp {} $O/bench_c2_$c.csv
  rm -rf $O/pmc_$c
done
python3 $R/tools/pmc_summary.py $O/bench_c2_FETCH_SIZE.csv $O/bench_c2_WRITE_SIZE.csv $O/pmc_traffic.json > /dev/null
for cfg in c4 c5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p45_$c -- python3 $R/bench.py --config $cfg --steps 1 --warmup 1 --preroll 0 --no-cpu-baseline --no-extras > /dev/null 2> $O/p45.err
    find $O/p45_$c -name "*counter_collection.csv" | tail -1 | xargs -I{} cp {} $O/p45_$c.csv
    rm -rf $O/p45_$c
  done
  python3 $R/tools/pmc_summary.py $O/p45_FETCH_SIZE.csv $O/p45_WRITE_SIZE.csv $O/pmc_traffic_$cfg.json > /dev/null
  rm -f $O/p45_*.csv $O/p45.err
done
# the traffic files go where bench.py looks for them, so that the lines below carry roofline.traffic of THIS build
cp $O/pmc_traffic.json $R/profiles/r05_pmc_traffic.json
cp $O/pmc_traffic_c4.json $R/profiles/r05_pmc_traffic_c4.json
cp $O/pmc_traffic_c5.json $R/profiles/r05_pmc_traffic_c5.json
cd $R
timeout -k 10 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout -k 10 400 python bench.py --config c4 --no-other-configs > $O/bench_c4.json 2> $O/bench_c4.err
timeout -k 10 500 python bench.py --config c5 --no-other-configs > $O/bench_c5.json 2> $O/bench_c5.err
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-other-configs > $O/bench_c2_under_rocprof.json 2> $O/stats.err
find $O/stats -name "*kernel_stats.csv" | tail -1 | xargs -I{} cp {} $O/bench_c2_kernel_stats.csv
rm -rf $O/stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats4 -- python3 $R/bench.py --config c4 --steps 20 --no-cpu-baseline --no-extras --no-other-configs > $O/bench_c4_under_rocprof.json 2> $O/stats4.err
find $O/stats4 -name "*kernel_stats.csv" | tail -1 | xargs -I{} cp {} $O/bench_c4_kernel_stats.csv
rm -rf $O/stats4
# the basis stage
cd $R
timeout -k 10 400 python3 tools/gpu_basis_profile.py all > $O/basis_stage_hip_events.txt 2> $O/basis.err
timeout -k 10 300 python3 tools/dev/gpu_energy_map_time.py > $O/energy_map.txt 2>&1
bash tools/dev/pod_timeline.sh fact > $O/pod_timeline_run.txt 2>&1
cp $R/gpurun_out/podtl/pod_rows_timeline.txt $O/pod_rows_kernel_timeline.txt
python3 tools/dev/kernel_timeline.py $R/gpurun_out/podtl/pod_fact_trace.csv k_center_partial > $O/pod_factored_kernel_timeline.txt
rm -f $O/*.err
ls -la $O
