# sweep rate at C4 / C5 against the number of concurrent sub-batches (ROMHC_STREAMS) (dev tool)
R=$GRAFT_REPO_ROOT; cd $R
for cfg in c4 c5; do
  for s in 1 2 3 4; do
    v=$(ROMHC_STREAMS=$s timeout -k 10 200 python3 bench.py --config $cfg --steps 20 --no-cpu-baseline --no-extras --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "$cfg streams=$s: $v"
  done
done
