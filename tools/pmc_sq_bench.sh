# SQ counters of the kernels of a bench configuration (round 3): MFMA pipe busy share, wave cycles, waits.
# usage (GPU box): bash tools/pmc_sq_bench.sh [c2|c4|c5]    -> gpurun_out/pmc_sq_<cfg>.txt
set -e
R=$GRAFT_REPO_ROOT
CFG=${1:-c2}
O=$R/gpurun_out/pmc_sq_$CFG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/g -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/err.txt
find $O/g -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/sq.csv
python3 $R/tools/pmc_sq_summary.py $O/sq.csv > $R/gpurun_out/pmc_sq_$CFG.txt
rm -rf $O
cat $R/gpurun_out/pmc_sq_$CFG.txt
