cd $GRAFT_REPO_ROOT
for f in "" "-DSEG_RING_NOREAD" "-DSEG_RING_NOMMA"; do echo "=== flags '$f' SEG_RING_ABL=3 (no fills)"; STAMP_FLAGS="$f" SEG_RING_ABL=3 timeout -k 10 200 python tools/stamp_ring.py 58,256,256,16,208 58,256,256,16,204 2>&1 | grep -v "amdgpu.ids\|wg 128\|wg 255\|warning\|note:\|asm volatile\|\^" ; done > gpurun_out/r04_ring_ablate2.txt 2>&1
cat gpurun_out/r04_ring_ablate2.txt
