#!/bin/bash
# Team tiling (csrc/smoe_team.hip.h) against the regular kernels over the batch sizes below ~16 000 blocks.
# usage: scripts/bench_team.sh  (prints one line per run; profiles/r03/bench_team.txt)
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-22s B=%-6d %-26s %10.1f Mpx-it/s  frac %.3f  ms/launch %.3f  psnr %.2f->%.2f (median %.2f)' % ('x'.join(map(str,d['config']['block_shape']))+' C%d K%d'%(d['config']['channels'],d['config']['kernels']), d['config']['blocks_per_gpu'], d['config']['kernel_variant'], d['value'], r['frac'], r['kernel_ms_per_launch'], d['initial_psnr_db'], d['final_psnr_db'], d['final_median_block_psnr_db']))"; }
for B in 1024 2048 4096 8192 16384; do
  run --blocks $B --tiling 0
  for T in 264 216 416 816; do run --blocks $B --tiling $T; done
done
run --blocks 512 --tiling 0
run --blocks 512 --tiling 264
run --blocks 3072 --tiling 0
run --blocks 3072 --tiling 264
run --blocks 65536 --tiling 216
for B in 1013 4050 8100; do
  run --blocks $B --channels 3 --tiling 0
  for T in 264 216 416 816; do run --blocks $B --channels 3 --tiling $T; done
done
for B in 1020 8160; do
  run --blocks $B --block-shape 16 16 4 --channels 3 --kernels-per-dim 2 2 1 --tiling 0
  for T in 264 416 816; do run --blocks $B --block-shape 16 16 4 --channels 3 --kernels-per-dim 2 2 1 --tiling $T; done
done
