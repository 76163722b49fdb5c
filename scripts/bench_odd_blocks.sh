run() { python bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-14s B=%-6d %-22s %10.1f Mpx-it/s  ms/launch %.3f' % ('x'.join(map(str,d['config']['block_shape'])), d['config']['blocks_per_gpu'], d['config']['kernel_variant'], d['value'], r['kernel_ms_per_launch']))"; }
run --blocks 2040 --block-shape 24 24 --channels 3 --kernels-per-dim 2 4
run --blocks 2040 --block-shape 30 20 --channels 3 --kernels-per-dim 2 4
run --blocks 4000 --block-shape 24 24 --channels 3 --kernels-per-dim 2 4
