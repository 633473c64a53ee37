# the three fuzzers with seeds / sizes beyond the ones the suite runs (dev tool; logs under gpurun_out/fuzz_more)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fuzz_more
rm -rf $O; mkdir -p $O
cd $R
for s in 11 12 13; do
  SEED=$s CASES=60 timeout -k 10 240 python3 tools/dev/pod_fuzz.py > $O/pod_fuzz_seed$s.txt 2>&1
  tail -1 $O/pod_fuzz_seed$s.txt
done
SEED=21 CASES=25 MMAX=1500 DMAX=20000 timeout -k 10 400 python3 tools/dev/pod_fuzz.py > $O/pod_fuzz_big_seed21.txt 2>&1
tail -1 $O/pod_fuzz_big_seed21.txt
SEED=5 CASES=30 timeout -k 10 300 python3 tests/dev/gpu_builders_fuzz.py > $O/builders_seed5.txt 2>&1
SEED=6 CASES=40 timeout -k 10 300 python3 tests/dev/gpu_api_fuzz.py > $O/api_seed6.txt 2>&1
echo done
