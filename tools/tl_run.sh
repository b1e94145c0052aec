cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/tl; rm -rf $O; mkdir -p $O
SEG_FORK_SIGNAL=1 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 bench.py --steps 30 --warmup 10 --windows 1 --no-cpu-baseline --no-roofline > $O/bench.json 2> $O/err.txt && python3 tools/timeline.py $O/tr 1 > $O/timeline.txt 2>&1 && python3 tools/conv_micro2.py --size 256 --cfgs 0,1,2,3,4,11,12,13,14,21,22,23,24 > $O/micro256.txt 2>&1
tail -3 $O/bench.json
