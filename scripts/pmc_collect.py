#!/usr/bin/env python3
"""PMC passes of the fit kernel under rocprofv3 (each pass its own run: --pmc with --kernel-trace only), reduced to the two
files bench.py reads:

  profiles/pmc.json      per kernel variant: VALU instructions per 64 pixel-iterations, issue utilisation
                         (2 cycles x SQ_INSTS_VALU per SIMD / engine cycles), LDS bank-conflict share
  profiles/traffic.json  per kernel variant: measured HBM bytes per block that do not depend on the iteration count
                         (staging in, parameters / slots in and out) and per block-iteration, from FETCH_SIZE (x2:
                         gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md) + WRITE_SIZE at 100 and at 20
                         iterations per launch

usage (on the GPU box, from the repo root):  python3 scripts/pmc_collect.py <outdir> [bench args ...]
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = {
    "p1": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"],
    "p2": ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_INSTS_VMEM"],
    "p3": ["FETCH_SIZE"],
    "p4": ["WRITE_SIZE"],
}


def run_pass(out, name, counters, bench_args, steps):
    d = os.path.join(out, f"{name}_s{steps}")
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "--",
           "python3", os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "0", "--clock-warm-iters", "0",
           "--no-cpu-baseline", "--no-extras", "--no-reps"] + bench_args
    env = dict(os.environ, TMPDIR="/tmp")
    with open(os.path.join(out, f"{name}_s{steps}.log"), "w") as log:
        rc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, timeout=600).returncode
    line = [l for l in open(os.path.join(out, f"{name}_s{steps}.log")) if l.startswith("{") and '"metric"' in l]
    # a failed pass must not reach the committed profile files: stop here, nothing is written
    if rc != 0:
        sys.exit(f"pmc_collect: pass {name} (steps {steps}) failed with rc {rc}: see {out}/{name}_s{steps}.log")
    if not line:
        sys.exit(f"pmc_collect: pass {name} (steps {steps}) printed no bench line: see {out}/{name}_s{steps}.log")
    cfg = json.loads(line[-1])["config"]
    agg = collections.defaultdict(float)
    cnt = collections.Counter()
    kname = None
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "fit_kernel" not in r["Kernel_Name"]:
                continue
            kname = r["Kernel_Name"]
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
    missing = [c for c in counters if cnt[c] == 0]
    if missing:
        sys.exit(f"pmc_collect: pass {name} (steps {steps}): no fit_kernel dispatch carries {missing}")
    per = {c: agg[c] / cnt[c] for c in agg}
    print(f"{name} steps={steps} rc={rc} dispatches={dict(cnt)}", flush=True)
    return per, cfg, kname


def main():
    out = os.path.abspath(sys.argv[1])
    bench_args = sys.argv[2:]
    os.makedirs(out, exist_ok=True)
    res = {}
    cfg = {}
    kname = None
    for name, counters in PASSES.items():
        per, c, k = run_pass(out, name, counters, bench_args, 100)
        res.update(per)
        cfg = c or cfg
        kname = k or kname
    t20 = {}
    for name in ("p3", "p4"):
        per, _, _ = run_pass(out, name, PASSES[name], bench_args, 20)
        t20.update(per)
    variant = cfg.get("kernel_variant")
    B = cfg.get("blocks_rank0", 0)
    if not variant or not B:
        sys.exit("pmc_collect: the bench line carries no kernel_variant / blocks_rank0: nothing written")
    N = 1
    for s in cfg.get("block_shape", []):
        N *= s
    with open(os.path.join(out, "pmc_summary.txt"), "w") as f:
        f.write(f"kernel {kname}\nvariant {variant}  blocks {B}  pixels/block {N}  iterations/launch 100 (FETCH/WRITE also at 20)\n")
        for c in sorted(res):
            f.write(f"  {c:24s} per dispatch {res[c]:.6g}\n")
        for c in sorted(t20):
            f.write(f"  {c:24s} per dispatch at 20 iterations {t20[c]:.6g}\n")
    simds = 256 * 4
    px_it = B * N * 100
    insts = res.get("SQ_INSTS_VALU", 0.0)
    cyc = res.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    pmc = {"valu_insts_per_wave_px_iter": round(insts / (px_it / 64.0), 3) if px_it else None,
           "issue_util": round(2.0 * (insts / simds) / cyc, 4) if cyc else None,
           "cycles_per_valu_inst": round(cyc / (insts / simds), 3) if insts else None,
           "engine_cycles_per_launch": cyc,
           "lds_bank_conflict_share": round(res.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(res.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0), 4),
           "blocks": B, "iters_per_launch": 100, "source": os.path.relpath(os.path.join(out, "pmc_summary.txt"), ROOT)}
    b100 = (2.0 * res.get("FETCH_SIZE", 0.0) + res.get("WRITE_SIZE", 0.0)) * 1024.0
    b20 = (2.0 * t20.get("FETCH_SIZE", 0.0) + t20.get("WRITE_SIZE", 0.0)) * 1024.0
    per_iter = (b100 - b20) / 80.0
    traffic = {"fixed_bytes_per_block": round((b20 - 20.0 * per_iter) / B, 2) if B else None,
               "bytes_per_block_iter": round(per_iter / B, 4) if B else None,
               "hbm_bytes_per_launch_100": int(b100), "hbm_bytes_per_launch_20": int(b20), "blocks": B,
               "source": os.path.relpath(os.path.join(out, "pmc_summary.txt"), ROOT)}
    for fname, key, ent in (("pmc.json", variant, pmc), ("traffic.json", variant, traffic)):
        path = os.path.join(ROOT, "profiles", fname)
        cur = json.load(open(path)) if os.path.exists(path) else {}
        if fname == "traffic.json":
            cur["_doc"] = ("HBM traffic of smoe_fit from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, at 100 and 20 iterations "
                           "per launch).  gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of coalesced streaming reads "
                           "-> x2; WRITE_SIZE is exact; units KB -> bytes x1024.  bytes per launch = blocks x (fixed + per_iter x iterations).")
        cur[key] = ent
        json.dump(cur, open(path, "w"), indent=1)
    print(json.dumps({"variant": variant, "pmc": pmc, "traffic": traffic}))


if __name__ == "__main__":
    main()
