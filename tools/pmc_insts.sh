# instruction mix (SQ_INSTS_*) of the extension kernels (dev tool)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_insts
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export REPS=1 INNER=2 VARIANTS="p:ROMHC_EXT_P=1"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
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
            d[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for key, v in sorted(d.items()):
        print(key, "n=%d" % len(v), "avg=%.5g" % (sum(v) / len(v)))
PY
