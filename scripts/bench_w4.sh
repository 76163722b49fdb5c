run() { python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('B=%-6d %-22s %10.1f Mpx-it/s  frac %.3f  ms/launch %.3f' % (d['config']['blocks_per_gpu'], d['config']['kernel_variant'], d['value'], r['frac'], r['kernel_ms_per_launch']))"; }
run --blocks 65536
run --blocks 16384
run --blocks 8192
run --blocks 65536 --steps 20 --warmup 5
