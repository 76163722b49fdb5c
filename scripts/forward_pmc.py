"""VALU census of the evaluation kernel (loss-only pass, headline size, 16-lane tiling): run under
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace
and reduce the counter CSV with `--reduce <dir>`: instructions per wavefront-pixel and issue utilisation."""
import csv
import glob
import json
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--reduce":
    agg, cnt = {}, {}
    for f in glob.glob(sys.argv[2] + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "forward_kernel" not in r["Kernel_Name"]:
                continue
            agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            cnt[r["Counter_Name"]] = cnt.get(r["Counter_Name"], 0) + 1
    per = {k: agg[k] / cnt[k] for k in agg}
    B, N = 65536, 256
    insts, cyc = per.get("SQ_INSTS_VALU", 0.0), per.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    out = {"per_dispatch": per, "dispatches": cnt.get("SQ_INSTS_VALU", 0),
           "valu_insts_per_wave_px": round(insts / (B * N / 64.0), 2),
           "engine_cycles_per_launch": cyc, "us_at_2p4GHz": round(cyc / 2400.0, 2),
           "issue_util": round(2.0 * (insts / 1024.0) / cyc, 4) if cyc else None,
           "valu_floor_us_at_2_cycles": round(2.0 * (insts / 1024.0) / 2400.0, 2),
           "valu_floor_us_at_3p2_cycles": round(3.2 * (insts / 1024.0) / 2400.0, 2)}
    print(json.dumps(out, indent=1))
    sys.exit(0)

import torch                                                                                # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from steered_mixture_of_experts_amd import blocks as blk                                    # noqa: E402
from steered_mixture_of_experts_amd.engine import BlockEngine, EngineConfig                 # noqa: E402

B, shape, C, kpd, K = 65536, (16, 16), 1, [2, 2], 4
blocks = blk.synthetic_blocks(B, shape, C, 7)
p0 = blk.init_block_params(blocks, kpd)
eng = BlockEngine(EngineConfig(block_shape=shape, channels=C, kernels=K, quantize_pis=True))
eng.set_tiling(16)
T = torch.from_numpy(blk.to_planar(blocks)).cuda()
dp = {k: torch.from_numpy(v).cuda() for k, v in p0.items()}
act = torch.full((B,), 15, dtype=torch.int32, device="cuda")
for _ in range(10):
    eng.forward(T, dp, act, want_recon=False)
torch.cuda.synchronize()
eng.close()
