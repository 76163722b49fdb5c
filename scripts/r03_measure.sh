#!/bin/bash
# Round-3 measurement set on one MI355X: the bench line (default and the driver's flags), its kernel trace, the PMC passes,
# the shape table, the evaluation pass, the SSIM / quantised / shared-kernel lines.  usage: scripts/r03_measure.sh <outdir>
OUT=${1:-gpurun_out/r03}
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT" || exit 1
mkdir -p "$OUT"
export PYTHONPATH="$ROOT"
timeout -k 10 400 python bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err" || echo "bench default rc=$?"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > "$OUT/bench_n1_steps20.json" 2> "$OUT/bench_n1_steps20.err" || echo "bench steps20 rc=$?"
bash scripts/prof_bench.sh "$OUT" --steps 20 --warmup 5 > "$OUT/prof_bench.log" 2>&1
mv "$OUT/bench_under_rocprof.json" "$OUT/bench_n1_under_rocprof.json" 2>/dev/null
rm -rf "$OUT/trace"
timeout -k 10 900 python3 scripts/pmc_collect.py "$OUT/pmc" > "$OUT/pmc_collect.log" 2>&1 || echo "pmc rc=$?"
cp "$OUT/pmc/pmc_summary.txt" "$OUT/rocprofv3_pmc_summary.txt" 2>/dev/null
rm -rf "$OUT"/pmc/p*_s*/
bash scripts/bench_shapes.sh > "$OUT/bench_shapes.txt" 2>&1
timeout -k 10 300 python scripts/forward_timing.py > "$OUT/forward_timing.txt" 2>&1
timeout -k 10 300 python scripts/bench_ssim.py > "$OUT/bench_ssim_mode.txt" 2>&1
timeout -k 10 300 python scripts/quant_timing.py > "$OUT/quant_mode_timing.txt" 2>&1
timeout -k 10 300 python scripts/bench_shared.py --cpu-iters 1 > "$OUT/bench_shared_mode.json" 2>/dev/null
timeout -k 10 300 python scripts/phase_clocks.py > /dev/null 2>&1   # (needs the diagnostic build; skipped silently otherwise)
echo done
