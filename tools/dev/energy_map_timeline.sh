# where rom_fem_energy_map spends its time at C4 / C5 geometry: HIP-event profile + rocprofv3 kernel trace (dev tool)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/emap
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/dev/gpu_energy_map_time.py > $O/emap_time.txt 2>&1
cat > /tmp/emap_one.py <<'PY'
import sys
sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi
ctx = _ffi.get_context(0)
fem = _ffi.Fem(ctx, 3, 3, 171)
ctx.synchronize()
print(fem.energy_map(3), fem.compact_stride)
ctx.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
cp /tmp/emap_one.py $R/gpurun_out/emap_one.py
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 gpurun_out/emap_one.py > $O/kt.out 2> $O/kt.err
find $O/kt -name "*kernel_trace.csv" | tail -1 | xargs -I{} cp {} $O/emap_trace.csv
rm -rf $O/kt
python3 tools/dev/kernel_timeline.py $O/emap_trace.csv kf_set_identity > $O/emap_timeline.txt
cat $O/emap_time.txt; cat $O/kt.out
awk '$5 > 100.0' $O/emap_timeline.txt | head -60
tail -1 $O/emap_timeline.txt
