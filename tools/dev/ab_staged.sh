# A/B of the staged epilogue of k_extend128 (variant library gpurun_out/libromhc_staged.so, built with -DX128_STAGED=1) (dev tool)
set -e
R=$GRAFT_REPO_ROOT
cd $R
L=$R/tools/dev/libromhc_staged.so
python3 tools/dev/gpu_rows_hash.py > gpurun_out/hash_prod.txt
python3 tools/dev/gpu_rows_hash.py $L > gpurun_out/hash_staged.txt
diff gpurun_out/hash_prod.txt gpurun_out/hash_staged.txt && echo "rows bit-identical"
for cfg in "2 128 1024" "3 171 1024" "4 256 2048"; do
  set -- $cfg
  for rep in 1 2; do
    NB=$1 N=$2 M=$3 REPS=5 python3 tools/gpu_ext_time.py | tail -1
    ROMHC_LIB=$L NB=$1 N=$2 M=$3 REPS=5 python3 tools/gpu_ext_time.py | tail -1 | sed 's/default/staged /'
  done
done
