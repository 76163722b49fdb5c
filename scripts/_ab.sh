cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
for cfg in "32400 3 16" "32400 3 32" "16200 3 16" "16200 3 32" "8100 3 16" "8100 3 32" "32768 1 16" "32768 1 32"; do set -- $cfg; python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 --blocks $1 --channels $2 --tiling $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['blocks_per_gpu'], 'C', d['config']['channels'], d['config']['kernel_variant'], d['value'], d['roofline']['frac'])"; done
