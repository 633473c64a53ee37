# Second evidence run of round 5, after rom_pod lost its Gram stage (the solver sources -- and with them the PMC traffic files of
# tools/profile_r05.sh, which bench.py matches by source hash -- are unchanged): the three bench lines, kernel-trace --stats of
# the default line, HIP events of the basis-stage calls, kernel timelines of rom_pod (rows and factored).  Outputs: gpurun_out/r05prof
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05prof
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout -k 10 400 python bench.py --config c4 --no-other-configs > $O/bench_c4.json 2> $O/bench_c4.err
timeout -k 10 500 python bench.py --config c5 --no-other-configs > $O/bench_c5.json 2> $O/bench_c5.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-other-configs > $O/bench_c2_under_rocprof.json 2> $O/stats.err
find $O/stats -name "*kernel_stats.csv" | tail -1 | xargs -I{} cp {} $O/bench_c2_kernel_stats.csv
rm -rf $O/stats
cd $R
timeout -k 10 400 python3 tools/gpu_basis_profile.py all > $O/basis_stage_hip_events.txt 2> $O/basis.err
bash tools/dev/pod_timeline.sh fact > $O/pod_timeline_run.txt 2>&1
cp $R/gpurun_out/podtl/pod_rows_timeline.txt $O/pod_rows_kernel_timeline.txt
python3 tools/dev/kernel_timeline.py $R/gpurun_out/podtl/pod_fact_trace.csv k_center_partial > $O/pod_factored_kernel_timeline.txt
M=8192 REPS=6 timeout -k 10 200 python3 tools/pod_time.py > $O/pod_time_c3.txt 2>&1
rm -f $O/*.err
ls -la $O
