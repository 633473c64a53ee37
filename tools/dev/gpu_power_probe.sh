# Is the sweep power-capped?  Runs `CMD` for a few seconds in the background and samples rocm-smi (power, clocks) beside it.
# usage (GPU box): bash tools/dev/gpu_power_probe.sh "python tools/gpu_ext_time.py"   (REPS / INNER make it long enough)
R=$GRAFT_REPO_ROOT
cd $R
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^=\|^$" | head -20
echo "---- running: $1"
( eval "$1" > gpurun_out/power_probe_cmd.log 2>&1 ) &
PID=$!
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  sleep 1
  rocm-smi --showpower --showclocks 2>&1 | grep -i "Package Power\|sclk" | sed 's/.*: //' | tr '\n' ' '
  echo
  kill -0 $PID 2>/dev/null || break
done
wait $PID
tail -3 gpurun_out/power_probe_cmd.log
