run() { python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-12s B=%-6d %-22s %10.1f Mpx-it/s ms/launch %.3f' % ('x'.join(map(str,d['config']['block_shape'])), d['config']['blocks_per_gpu'], d['config']['kernel_variant'], d['value'], r['kernel_ms_per_launch']))"; }
run --blocks 1024 --block-shape 8 8 --tiling 64
run --blocks 1024 --block-shape 16 16 --tiling 64
run --blocks 1024 --block-shape 32 32 --tiling 64
run --blocks 1024 --block-shape 64 64 --tiling 64
run --blocks 2048 --block-shape 16 16 --tiling 64
run --blocks 2048 --block-shape 32 32 --tiling 64
run --blocks 4096 --block-shape 32 32 --tiling 64
