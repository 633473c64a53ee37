# kernel timeline of rom_pod (rows; `fact` as first argument: rom_pod_factored too) at the C2 geometry under rocprofv3 --kernel-trace (dev tool)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/podtl
rm -rf $O; mkdir -p $O
cd $R
M=1024 REPS=8 timeout -k 10 300 python3 tools/pod_time.py > $O/pod_time.txt 2>&1
cd /tmp && export TMPDIR=/tmp
M=1024 REPS=3 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/pod_time.py > $O/kt.out 2> $O/kt.err
find $O/kt -name "*kernel_trace.csv" | tail -1 | xargs -I{} cp {} $O/pod_rows_trace.csv
rm -rf $O/kt
python3 $R/tools/dev/kernel_timeline.py $O/pod_rows_trace.csv kp_zero_sum_rows > $O/pod_rows_timeline.txt
if [ "$1" = "fact" ]; then
  cd $R
  timeout -k 10 300 python3 tools/pod_factored_prof.py > $O/pod_factored_time.txt 2>&1
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt2 -- python3 $R/tools/pod_factored_prof.py > $O/kt2.out 2> $O/kt2.err
  find $O/kt2 -name "*kernel_trace.csv" | tail -1 | xargs -I{} cp {} $O/pod_fact_trace.csv
  rm -rf $O/kt2
  grep "^rep" $O/pod_factored_time.txt
fi
cat $O/pod_time.txt
tail -1 $O/pod_rows_timeline.txt
