#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats of one bench.py workload + the two TCC PMC passes (separate runs, as
# MI355X_MICROARCH.md prescribes; signal forks are forced ON only for the kernel-trace pass, which leaves the queues concurrent,
# and stay off -- the default under any profiler -- for the counter passes, which serialise dispatches), summarised into profiles/<tag>_{kernel_stats.csv,pmc_summary.json,bench.json}.
#   tools/collect_profiles.sh r03                                  the headline (U-Net 256, B=16)
#   tools/collect_profiles.sh r03_c4 --size 512 --steps 20 --warmup 5
#   tools/collect_profiles.sh r03_c3 --model fcn8s --size 512 --classes 21 --batch 8
#   tools/collect_profiles.sh r03_c5 --mode mc --batch 32 --steps 5 --warmup 2
tag=${1:-r03}; shift
args=("$@")
[ ${#args[@]} -eq 0 ] && args=(--steps 50 --warmup 10)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_$tag
rm -rf $O; mkdir -p $O/stats $O/fetch $O/write
cd /tmp && export TMPDIR=/tmp
cd $R
SEG_FORK_SIGNAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py "${args[@]}" --no-cpu-baseline > $O/bench_stats.json 2> $O/stats.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py "${args[@]}" --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-roofline > /dev/null 2> $O/fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py "${args[@]}" --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-roofline > /dev/null 2> $O/write.err || exit 1
python3 tools/profile_summary.py $O/stats $O/fetch $O/write $tag || exit 1
grep '^{' $O/bench_stats.json | tail -1 > profiles/${tag}_bench_under_rocprof.json
cp profiles/${tag}_kernel_stats.csv profiles/${tag}_pmc_summary.json profiles/${tag}_bench_under_rocprof.json $O/
[ -f profiles/pmc_summary.json ] && cp profiles/pmc_summary.json $O/
head -8 profiles/${tag}_kernel_stats.csv
