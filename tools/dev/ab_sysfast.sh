set -e
cd $GRAFT_REPO_ROOT
for c in c5 c4; do
  for v in 0 1 2; do
    export ROMHC_X128_SYS_FAST=$v
    timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-extras > gpurun_out/ab_${c}_$v.json 2> gpurun_out/ab_${c}_$v.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab_${c}_$v.json").read().strip().splitlines()[-1])
print("$c sys_fast=$v", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], flush=True)
PY
  done
done
ROMHC_X128_SYS_FAST=2 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tilings or c5 or g1" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1 2; do
export ROMHC_X128_SYS_FAST=$v
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc5_$c -- python3 $R/bench.py --config c5 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/pmc5.err
  find $R/gpurun_out/pmc5_$c -name "*counter_collection.csv" | tail -1 | xargs -I{} cp {} $R/gpurun_out/pmc5_${c}.csv
  rm -rf $R/gpurun_out/pmc5_$c
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc5_FETCH_SIZE.csv $R/gpurun_out/pmc5_WRITE_SIZE.csv $R/gpurun_out/pmc5_traffic_$v.json | grep -A6 extend128 || true
rm -f $R/gpurun_out/pmc5_*.csv
done
