set -e
cd $GRAFT_REPO_ROOT
for c in c5 c4 c2; do
  for v in 0 1; do
    if [ $v = 1 ]; then export ROMHC_X128_SYS_FAST=1; else unset ROMHC_X128_SYS_FAST; fi
    timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-extras > gpurun_out/ab_${c}_$v.json 2> gpurun_out/ab_${c}_$v.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab_${c}_$v.json").read().strip().splitlines()[-1])
print("$c sys_fast=$v", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], flush=True)
PY
  done
done
unset ROMHC_X128_SYS_FAST
ROMHC_X128_SYS_FAST=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tilings or c5 or c4 or g1" 2>&1 | tail -5
