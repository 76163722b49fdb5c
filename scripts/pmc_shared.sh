#!/bin/bash
# PMC passes for the shared-kernel pass (each pass its own rocprofv3 run).  usage: scripts/pmc_shared.sh <outdir> [bench_shared args]
set -u
OUT=$1; shift
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root: the GPU box exports it; elsewhere derived from this file
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
mkdir -p "$OUT"
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 scripts/bench_shared.py --cpu-iters 0 --steps 20 $ARGS > "$OUT/$name.log" 2>&1; echo "$name rc=$?"; }
export ARGS="${ARGS:---image 2048 2048 --kernels-per-dim 48 48}"
run p1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM
run p3 SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_GDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(out+'/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'shared_pass_kernel' not in k or 'ELb1E' not in k.split('shared_pass_kernel')[1][:40]: continue
        agg['pass'][r['Counter_Name']]+=float(r['Counter_Value'])
        cnt[('pass',r['Counter_Name'])]+=1
for kk in agg:
    print('==',kk)
    for c,v in sorted(agg[kk].items()):
        n=cnt[(kk,c)]
        print(f'  {c:24s} total {v:.4g}  dispatches {n}  per-dispatch {v/n:.4g}')
PY
