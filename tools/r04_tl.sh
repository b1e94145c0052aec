#!/bin/bash
# GPU box: full kernel timeline (start, end, queue, name) of one eager train step that is GPU-bound even under the tracer:
# batch 32 at 256^2 (the host needs ~1.3 ms per step under rocprofv3, the GPU ~1.9) and batch 16 at 512^2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "256 32" "512 16"; do set -- $cfg
  rm -rf gpurun_out/tl_$1
  SEG_FORK_SIGNAL=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$1 -- python3 bench.py --no-graph --size $1 --batch $2 --steps 20 --warmup 10 --windows 1 --no-cpu-baseline --no-roofline > gpurun_out/tl_$1.json 2> gpurun_out/tl_$1.err || exit 1
  python tools/timeline.py gpurun_out/tl_$1 list > gpurun_out/timeline_$1.txt 2>&1
  rm -rf gpurun_out/tl_$1
done
head -12 gpurun_out/timeline_256.txt
