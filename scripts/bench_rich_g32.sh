#!/bin/bash
# mid-size batches (32-lane tiling) of the parameter-rich FULL triples
run() { python bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-10s C%d K%-2d B=%-6d %-22s %10.1f Mpx-it/s  ms/launch %.3f' % ('x'.join(map(str,d['config']['block_shape'])), d['config']['channels'], d['config']['kernels'], d['config']['blocks_per_gpu'], d['config']['kernel_variant'], d['value'], r['kernel_ms_per_launch']))"; }
run --blocks 4096 --block-shape 16 16 --channels 3 --kernels-per-dim 2 4
run --blocks 6000 --block-shape 16 16 --channels 3 --kernels-per-dim 2 4
run --blocks 4096 --block-shape 8 8 4 --channels 3 --kernels-per-dim 2 2 1
run --blocks 3100 --block-shape 16 16 --channels 3 --kernels-per-dim 2 4
