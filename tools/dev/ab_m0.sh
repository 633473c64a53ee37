cd $GRAFT_REPO_ROOT
for v in with without with without; do
  cp romhighcontrast_amd/csrc/rom_fem_kernels.hip /tmp/fk_backup.hip
  if [ $v = without ]; then sed -i 's/: "memory", "m0")/: "memory")/' romhighcontrast_amd/csrc/rom_fem_kernels.hip; fi
  touch romhighcontrast_amd/csrc/rom_fem_kernels.hip
  make -C romhighcontrast_amd/csrc -j8 > /dev/null 2>&1 || echo build failed
  echo "=== $v m0 clobber"
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print(d['value'], d['ms_per_step'], {n:round(v['avg_ms'],5) for n,v in k.items()})"
  cp /tmp/fk_backup.hip romhighcontrast_amd/csrc/rom_fem_kernels.hip
done
touch romhighcontrast_amd/csrc/rom_fem_kernels.hip; make -C romhighcontrast_amd/csrc -j8 > /dev/null 2>&1
