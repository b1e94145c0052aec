for dw in 1 0; do for it in 1 2 4 8; do
  SEG_FUSE_HEAD_DW=$dw SEG_HEAD_ITERS=$it timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/it_$it.json 2>/dev/null || exit 1
  echo dw $dw iters $it; python tools/bench_line.py gpurun_out/it_$it.json xent
done; done
