#!/bin/bash
# Throughput of the BASELINE shapes on one GPU (not the headline bench line).
run() { python bench.py --no-cpu-baseline --steps 100 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-22s B=%-6d %-22s %10.1f Mpx-it/s  frac %.3f  ms/launch %.3f  psnr %.2f->%.2f (median %.2f)' % ('x'.join(map(str,d['config']['block_shape']))+' C%d K%d'%(d['config']['channels'],d['config']['kernels']), d['config']['blocks_per_gpu'], d['config']['kernel_variant'], d['value'], r['frac'], r['kernel_ms_per_launch'], d['initial_psnr_db'], d['final_psnr_db'], d['final_median_block_psnr_db']))"; }
run --blocks 65536
run --blocks 1024
run --blocks 1024 --tiling 64
run --blocks 1024 --tiling 16
run --blocks 1024 --tiling 32
run --blocks 2048
run --blocks 4096 --tiling 64
run --blocks 4096 --tiling 32
run --blocks 4096 --tiling 16
run --blocks 8192 --tiling 32
run --blocks 8192 --tiling 16
run --blocks 16384 --tiling 32
run --blocks 16384 --tiling 16
run --blocks 2040 --block-shape 32 32 --channels 3 --kernels-per-dim 2 4
run --blocks 2040 --block-shape 32 32 --channels 3 --kernels-per-dim 2 4 --tiling 16
run --blocks 32400 --channels 3
run --blocks 32400 --channels 3 --tiling 64
run --blocks 4050 --channels 3 --tiling 16
run --blocks 4050 --channels 3 --tiling 32
run --blocks 4050 --channels 3 --tiling 64
run --blocks 65280 --block-shape 16 16 4 --channels 3 --kernels-per-dim 2 2 1
run --blocks 8160 --block-shape 16 16 4 --channels 3 --kernels-per-dim 2 2 1
run --blocks 1020 --block-shape 16 16 4 --channels 3 --kernels-per-dim 2 2 1
run --blocks 1020 --block-shape 16 16 4 --channels 3 --kernels-per-dim 2 2 1 --tiling 64
