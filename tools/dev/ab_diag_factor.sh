# k_diag_factor with its pivots / solved entries through LDS (new build) against the build before (libromhc_prev.so):
# bits of the rows, C4 / C5 bench lines alternating (dev tool; logs under gpurun_out/abdf)
# (libromhc_prev.so: the object of the commit before, linked with the current other objects -- `git show HEAD~:.../rom_fem_kernels.hip`, same flags as the Makefile; not kept in the tree)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/abdf
rm -rf $O; mkdir -p $O
cd $R
L=romhighcontrast_amd/csrc
timeout -k 10 200 python3 tools/dev/gpu_rows_hash.py > $O/hash_new.txt 2>&1
timeout -k 10 200 python3 tools/dev/gpu_rows_hash.py $L/libromhc_prev.so > $O/hash_prev.txt 2>&1
if cmp -s $O/hash_new.txt $O/hash_prev.txt; then echo "rows: same bits"; else echo "rows DIFFER"; diff $O/hash_new.txt $O/hash_prev.txt || true; fi
line() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d.get('kernels',{})
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], {n:v for n,v in k.items() if 'diag' in n} if isinstance(k,dict) else '')" $1; }
for rep in 1 2; do
  for c in c4 c5; do
    timeout -k 10 200 python3 bench.py --config $c --steps 20 --no-extras --no-cpu-baseline > $O/${c}_new_$rep.json 2> $O/err.txt
    line $O/${c}_new_$rep.json
    timeout -k 10 200 python3 tools/dev/with_lib.py $L/libromhc_prev.so bench.py --config $c --steps 20 --no-extras --no-cpu-baseline > $O/${c}_prev_$rep.json 2> $O/err.txt
    line $O/${c}_prev_$rep.json
  done
done
