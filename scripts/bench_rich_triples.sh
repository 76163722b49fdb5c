#!/bin/bash
# the parameter-rich triples (>= 96 slots) whose 64-lane hoisting kernels are bound to two wavefronts per SIMD: throughput per shape
run() { python bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-10s C%d K%-2d B=%-6d %-22s %10.1f Mpx-it/s  ms/launch %.3f' % ('x'.join(map(str,d['config']['block_shape'])), d['config']['channels'], d['config']['kernels'], d['config']['blocks_per_gpu'], d['config']['kernel_variant'], d['value'], r['kernel_ms_per_launch']))"; }
run --blocks 2040 --block-shape 32 32 --channels 3 --kernels-per-dim 2 4
run --blocks 2040 --block-shape 24 24 --channels 3 --kernels-per-dim 2 4
run --blocks 2040 --block-shape 32 32 --channels 1 --kernels-per-dim 3 4
run --blocks 2040 --block-shape 32 32 --channels 3 --kernels-per-dim 2 3
run --blocks 2040 --block-shape 32 32 --channels 3 --kernels-per-dim 3 3
run --blocks 2040 --block-shape 32 32 --channels 1 --kernels-per-dim 4 4
run --blocks 2040 --block-shape 16 16 4 --channels 3 --kernels-per-dim 2 2 2
run --blocks 2040 --block-shape 16 16 4 --channels 1 --kernels-per-dim 2 2 2
run --blocks 2040 --block-shape 16 16 4 --channels 3 --kernels-per-dim 1 3 2
run --blocks 2048 --block-shape 16 16 --channels 1 --kernels-per-dim 3 4
run --blocks 2048 --block-shape 16 16 --channels 3 --kernels-per-dim 3 3
run --blocks 1024 --block-shape 16 16 --channels 1 --kernels-per-dim 3 4
