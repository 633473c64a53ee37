# rocprofv3 evidence of round 4 (run on the MI355X box through gpurun; outputs under gpurun_out/r04prof):
#   the default bench line (C2 + compact C4 / C5 legs), the full C4 / C5 lines; kernel-trace --stats of bench.py (C2, C4);
#   PMC passes (FETCH_SIZE, WRITE_SIZE: one counter per pass, kernel-trace only) of the C2 / C4 / C5 sweeps; SQ counters of
#   the C4 kernels (tile Cholesky after the round's changes); HIP events of the basis-stage calls incl. the factored ones
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04prof
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout -k 10 400 python bench.py --config c4 > $O/bench_c4.json 2> $O/bench_c4.err
timeout -k 10 500 python bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/bench_c2_under_rocprof.json 2> $O/stats.err
find $O/stats -name "*kernel_stats.csv" | tail -1 | xargs -I{} cp {} $O/bench_c2_kernel_stats.csv
rm -rf $O/stats
ROMHC_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats4 -- python3 $R/bench.py --config c4 --steps 20 --no-cpu-baseline --no-extras > $O/bench_c4_under_rocprof.json 2> $O/stats4.err
find $O/stats4 -name "*kernel_stats.csv" | tail -1 | xargs -I{} cp {} $O/bench_c4_kernel_stats.csv
rm -rf $O/stats4
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --preroll 0 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_$c.err
  find $O/pmc_$c -name "*counter_collection.csv" | tail -1 | xargs -I{} cp {} $O/bench_c2_$c.csv
  rm -rf $O/pmc_$c
done
python3 $R/tools/pmc_summary.py $O/bench_c2_FETCH_SIZE.csv $O/bench_c2_WRITE_SIZE.csv $O/pmc_traffic.json > /dev/null
for cfg in c4 c5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ROMHC_STREAMS=1 timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p45_$c -- python3 $R/bench.py --config $cfg --steps 1 --warmup 1 --preroll 0 --no-cpu-baseline --no-extras > /dev/null 2> $O/p45.err
    find $O/p45_$c -name "*counter_collection.csv" | tail -1 | xargs -I{} cp {} $O/p45_$c.csv
    rm -rf $O/p45_$c
  done
  python3 $R/tools/pmc_summary.py $O/p45_FETCH_SIZE.csv $O/p45_WRITE_SIZE.csv $O/pmc_traffic_$cfg.json > /dev/null
  rm -f $O/p45_*.csv $O/p45.err
done
# SQ counters (MFMA pipe busy) of the C4 kernels
ROMHC_STREAMS=1 timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/bench.py --config c4 --steps 2 --warmup 1 --preroll 0 --no-cpu-baseline --no-extras > /dev/null 2> $O/sq.err
find $O/sq -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/sq_c4.csv
python3 $R/tools/pmc_sq_summary.py $O/sq_c4.csv > $O/pmc_sq_c4.txt
rm -rf $O/sq $O/sq_c4.csv
# the basis stage
cd $R
timeout -k 10 400 python3 tools/gpu_basis_profile.py all > $O/basis_stage_hip_events.txt 2> $O/basis.err
rm -f $O/*.err
ls -la $O
