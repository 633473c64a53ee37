R=$GRAFT_REPO_ROOT; cd $R
for cfg in c5 c4; do
  for s in 2 1 2 1; do
    v=$(ROMHC_STREAMS=$s timeout -k 10 200 python3 bench.py --config $cfg --steps 20 --no-cpu-baseline --no-extras --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "$cfg streams=$s: $v"
  done
done
