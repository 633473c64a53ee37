cd $GRAFT_REPO_ROOT
for v in "" "-DTILE_CHUNK_PROBE"; do
  touch romhighcontrast_amd/csrc/rom_fem_kernels.hip
  make -C romhighcontrast_amd/csrc -j8 EXTRA="$v" > /dev/null 2>&1 || { echo build failed; continue; }
  echo "=== EXTRA='$v'"
  NB=4 N=256 M=4096 DEC=3 REPS=3 INNER=3 timeout -k 10 300 python tools/gpu_solve_time.py 2>&1 | tail -9
  NB=3 N=171 M=1024 DEC=2 REPS=3 INNER=5 timeout -k 10 300 python tools/gpu_solve_time.py 2>&1 | tail -9
done
touch romhighcontrast_amd/csrc/rom_fem_kernels.hip; make -C romhighcontrast_amd/csrc -j8 > /dev/null 2>&1
