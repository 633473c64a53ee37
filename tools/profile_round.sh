set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/v7
mkdir -p $O
cd $R
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pod > $O/under_rocprof.json 2> $O/stats.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pod > /dev/null 2> $O/fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pod > /dev/null 2> $O/write.err
cd $R
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/fetch -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/fetch_counter_collection.csv
find $O/write -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/write_counter_collection.csv
python tools/pmc_summary.py $O/fetch_counter_collection.csv $O/write_counter_collection.csv $O/pmc_traffic.json > /dev/null
rm -rf $O/stats $O/fetch $O/write
head -c 600 $O/bench.json; echo; head -5 $O/kernel_stats.csv
