# rocprofv3 evidence of round 2 (run on the MI355X box through gpurun; outputs under gpurun_out/r02prof):
#   kernel-trace --stats of bench.py (C2) and of tools/gpu_hbm_kernels.py (C2, C4); PMC passes (FETCH_SIZE, WRITE_SIZE, one
#   counter group per pass, kernel-trace only: gpurun's rule) of the same commands
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02prof
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/bench_c2_under_rocprof.json 2> $O/stats.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_c2_kernel_stats.csv
rm -rf $O/stats
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_$c.err
  find $O/pmc_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/bench_c2_$c.csv
  rm -rf $O/pmc_$c
done
python3 $R/tools/pmc_summary.py $O/bench_c2_FETCH_SIZE.csv $O/bench_c2_WRITE_SIZE.csv $O/pmc_traffic.json > /dev/null
for cfg in c2 c4; do
  CFG=$cfg timeout -k 10 200 python3 $R/tools/gpu_hbm_kernels.py > $O/hbm_kernels_$cfg.txt 2> $O/hbm_$cfg.err
  CFG=$cfg REPS=2 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/hs_$cfg -- python3 $R/tools/gpu_hbm_kernels.py > /dev/null 2> $O/hs_$cfg.err
  find $O/hs_$cfg -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/hbm_kernels_${cfg}_kernel_stats.csv
  rm -rf $O/hs_$cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    CFG=$cfg REPS=2 timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/hp_$c -- python3 $R/tools/gpu_hbm_kernels.py > /dev/null 2> $O/hp_${cfg}_$c.err
    find $O/hp_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/hbm_kernels_${cfg}_$c.csv
    rm -rf $O/hp_$c
  done
  python3 $R/tools/pmc_summary.py $O/hbm_kernels_${cfg}_FETCH_SIZE.csv $O/hbm_kernels_${cfg}_WRITE_SIZE.csv $O/hbm_kernels_${cfg}_pmc_traffic.json > /dev/null
done
# PMC traffic of the C4 / C5 sweeps (read by `bench.py --config c4|c5` as roofline.traffic); one stream, so that a
# launch is the whole 1024- / 4096-system sweep as in the bench's per-kernel pass (by default these sweeps run as two
# concurrent sub-batches)
export ROMHC_STREAMS=1
for cfg in c4 c5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pc_$c -- python3 $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pc_${cfg}_$c.err
    find $O/pc_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/bench_${cfg}_$c.csv
    rm -rf $O/pc_$c
  done
  python3 $R/tools/pmc_summary.py $O/bench_${cfg}_FETCH_SIZE.csv $O/bench_${cfg}_WRITE_SIZE.csv $O/pmc_traffic_$cfg.json > /dev/null
done
unset ROMHC_STREAMS
# rocprofv3 per-kernel statistics of the C4 / C5 sweeps as they run by default (two concurrent sub-batches: a launch
# covers half the systems, and kernels of the two halves overlap)
for cfg in c4 c5; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$cfg -- python3 $R/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_${cfg}_under_rocprof.json 2> $O/st_$cfg.err
  find $O/st_$cfg -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_${cfg}_kernel_stats.csv
  rm -rf $O/st_$cfg
done
rm -f $O/*.err
ls -la $O
