# PMC passes over tools/gpu_ext_time.py (dev tool): one counter group per pass, kernel-trace only (gpurun rule)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_ext
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export REPS=1 INNER=2
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -- python3 $R/tools/gpu_ext_time.py > $O/g$i.out 2> $O/g$i.err || echo "group $i failed"
  find $O/g$i -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/g$i.csv || true
  rm -rf $O/g$i
done
python3 - <<PY
import csv, collections, glob
for path in sorted(glob.glob("$O/g*.csv")):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "extend" in k:
            d[(k, r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    for key, v in sorted(d.items()):
        print(path.split("/")[-1], key, "n=%d" % len(v), "avg=%.4g" % (sum(v) / len(v)), "last=%.4g" % v[-1])
PY
