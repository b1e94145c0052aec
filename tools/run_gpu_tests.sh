#!/bin/bash
# GPU box: the -m gpu suite (the form the driver runs), log under gpurun_out/
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 900 "$@" > gpurun_out/gpu_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/gpu_tests.log
tail -5 gpurun_out/gpu_tests.log
