run() { python bench.py --no-cpu-baseline --steps 100 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-22s B=%-6d %-22s %10.1f Mpx-it/s  frac %.3f  ms/launch %.3f' % ('x'.join(map(str,d['config']['block_shape']))+' C%d K%d'%(d['config']['channels'],d['config']['kernels']), d['config']['blocks_per_gpu'], d['config']['kernel_variant'], d['value'], r['frac'], r['kernel_ms_per_launch']))"; }
run --blocks 2040 --block-shape 32 32 --channels 3 --kernels-per-dim 2 4 --tiling 32
run --blocks 2040 --block-shape 32 32 --channels 3 --kernels-per-dim 2 4 --tiling 64
run --blocks 4080 --block-shape 32 32 --channels 3 --kernels-per-dim 2 4 --tiling 32
run --blocks 4080 --block-shape 32 32 --channels 3 --kernels-per-dim 2 4 --tiling 64
run --blocks 65280 --block-shape 16 16 4 --channels 3 --kernels-per-dim 2 2 1 --tiling 32
run --blocks 8160 --block-shape 16 16 4 --channels 3 --kernels-per-dim 2 2 1 --tiling 32
