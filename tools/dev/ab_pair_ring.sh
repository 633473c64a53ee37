cd $GRAFT_REPO_ROOT
for v in 2 16; do
  touch romhighcontrast_amd/csrc/rom_fem_dev.h
  t0=$(date +%s.%N)
  make -C romhighcontrast_amd/csrc -j8 EXTRA="-DPAIR_RING_=$v" 2>&1 | tail -2
  t1=$(date +%s.%N)
  echo "=== PAIR_RING $v (build $(echo "$t1 - $t0" | bc) s)"
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print(d['value'], d['ms_per_step'], {n:round(v['avg_ms'],5) for n,v in k.items()})"
done
