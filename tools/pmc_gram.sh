# SQ counters of the Gram kernel (dev tool; kernel-trace + one --pmc group per pass: gpurun's rule)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_gram
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MS=4096
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -- python3 $R/tools/gpu_gram_time.py > $O/g$i.out 2> $O/g$i.err || echo "group $i failed"
  find $O/g$i -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/g$i.csv || true
  rm -rf $O/g$i
done
python3 - <<PY
import csv, collections, glob
for path in sorted(glob.glob("$O/g*.csv")):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "gram128" in k and "finish" not in k:
            d[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for key, v in sorted(d.items()):
        print(key, "launches=%d" % len(v), "avg=%.5g" % (sum(v) / len(v)))
PY
