#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats of the default bench + the two TCC PMC passes (separate runs).
tag=${1:-r01}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_$tag
rm -rf $O; mkdir -p $O/stats $O/fetch $O/write
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_stats.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-roofline > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-roofline > /dev/null 2> $O/write.err
python3 tools/profile_summary.py $O/stats $O/fetch $O/write $tag
cp profiles/${tag}_kernel_stats.csv profiles/pmc_summary.json $O/
head -12 profiles/${tag}_kernel_stats.csv
