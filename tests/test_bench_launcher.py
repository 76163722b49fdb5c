"""bench.py --gpus N must produce an N-rank line by itself (no torchrun environment): the parent spawns the ranks
before anything touches the GPU and relays rank 0's JSON line.  Runs here on the gloo backend with the CPU engine
double (tests/fake_engine.py); the line is labelled accordingly and is not a measurement."""
import json
import os
import subprocess
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--clock-warm-iters", "0",
          "--engine-factory", "fake_engine:OracleEngine", "--backend", "gloo"]


def _run(extra, env_extra=None):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), env.get("PYTHONPATH", "")])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + COMMON, env=env, timeout=600,
                         stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0, out.stdout
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_weak():
    d = _run(["--gpus", "2", "--blocks", "12"])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "weak"
    assert d["config"]["total_blocks"] == 24 and d["config"]["blocks_rank0"] == 12
    assert len(d["per_rank_ms"]) == 2 and d["value"] > 0 and d["steps"] == 3 and d["warmup"] == 1
    assert "NOT a measurement" in d["engine"]
    for key in ("metric", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "roofline"):
        assert key in d


def test_strong_scaling_splits_one_image():
    one = _run(["--gpus", "1", "--scaling", "strong", "--image", "48", "80"])
    two = _run(["--gpus", "2", "--scaling", "strong", "--image", "48", "80"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert one["config"]["total_blocks"] == two["config"]["total_blocks"] == 15
    assert one["config"]["blocks_rank0"] == 15 and two["config"]["blocks_rank0"] == 8
    # the same image whatever the number of ranks: the all-reduced PSNR agrees (summation order aside)
    assert abs(one["initial_psnr_db"] - two["initial_psnr_db"]) < 1e-6
    assert abs(one["final_psnr_db"] - two["final_psnr_db"]) < 1e-6


def test_four_ranks_one_of_them_without_blocks():
    """B < ranks: 3 blocks over 4 ranks -- the last rank owns none, runs no kernel and still takes part
    in every collective; the line carries one kernel variant per rank."""
    one = _run(["--gpus", "1", "--scaling", "strong", "--image", "16", "48"])
    four = _run(["--gpus", "4", "--scaling", "strong", "--image", "16", "48"])
    assert four["n_gpus"] == 4 and four["rccl_ranks"] == 4 and len(four["per_rank_ms"]) == 4
    assert four["config"]["total_blocks"] == 3 and four["config"]["blocks_rank0"] == 1
    assert len(four["config"]["kernel_variant_per_rank"]) == 4 and "(0 blocks)" in four["config"]["kernel_variant_per_rank"][3]
    assert abs(one["final_psnr_db"] - four["final_psnr_db"]) < 1e-6
    assert one["state_digest"] == four["state_digest"]           # every block bit-identical whatever the split


def test_short_timed_region_is_repeated_from_cloned_state():
    d = _run(["--gpus", "1", "--blocks", "4"])
    assert d["config"]["reps"] >= 3 and len(d["config"]["timed_s_each_rep"]) == min(12, d["config"]["reps"])
    assert d["config"]["timed_region_s_total"] >= 0.45 or d["config"]["reps"] == 2000       # an externally visible timed region
    once = _run(["--gpus", "1", "--blocks", "4", "--no-reps"])
    assert once["config"]["reps"] == 1
    assert abs(once["final_psnr_db"] - d["final_psnr_db"]) < 1e-9      # every repetition replays the same K steps


def test_torchrun_environment_is_used_as_is(tmp_path):
    """The driver's own launcher (python -m torch.distributed.run ... bench.py --gpus 2) must not spawn again."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), env.get("PYTHONPATH", "")])
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--blocks", "6"] + COMMON, env=env, timeout=600, stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0, out.stdout
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
