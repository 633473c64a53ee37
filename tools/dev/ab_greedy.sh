cd $GRAFT_REPO_ROOT
for v in "1 2" "1 3" "1 5" "1 6" "1 4"; do
  set -- $v
  touch romhighcontrast_amd/csrc/rom_basis.hip
  make -C romhighcontrast_amd/csrc EXTRA="-DGU_SYS_=$1 -DGU_UNROLL_=$2" > /dev/null 2>&1 || { echo build failed $v; continue; }
  echo "=== GU_SYS $1 GU_UNROLL $2"
  timeout -k 10 200 python tools/gpu_basis_profile.py greedy 2>&1 | grep -E "^==|greedy_pass "
done
