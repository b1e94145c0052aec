#!/bin/bash
# SQ counters of the tiled convolution and the persistent one on the layers VERDICT r03 item 2 names (run on the GPU box):
# conv3_2 / conv4_2 / conv6_1 at 512^2, conv5_2 / conv6_2 at 256^2.  One rocprofv3 --pmc pass of 8 SQ counters per (layer, kernel);
# tools/r04_counters.py folds gpurun_out/pmc_r04_*/ into gpurun_out/r04_sq_counters.json (committed copy under profiles/).
CTR="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES"
CFGS=${CFGS:-"0 104"}
run() { # name hw cin cout
  for cfg in $CFGS; do
    bash tools/pmc.sh r04_$1_cfg$cfg $CTR -- --hw $2 --cin $3 --cout $4 --batch 16 --kind fwd --cfg $cfg --iters 20 || return 1
  done
}
run conv3_2_512 123 128 128 && run conv4_2_512 58 256 256 && run conv6_1_512 48 512 256 && run conv5_2_256 10 512 512 && run conv6_2_256 14 256 256
