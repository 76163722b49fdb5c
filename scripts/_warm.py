"""Bring the engine clocks of an idle MI355X up before a microbenchmark times anything.  From idle the same kernel takes
7.0, 5.3, 5.1, 4.9, 4.8, 4.65, 4.6, 4.6 ... ms per launch (profiles/r01/launch_times.txt): ~25 ms of THIS kind of load
are needed (a loop of small torch matmuls with syncs does not do it).  The warm-up runs the engine's own fit kernel on
scratch copies of the parameters -- with the other tiling where there is one, so that a kernel trace keeps the warm-up
launches in a row of their own."""
import torch


def warm_block_engine(eng, target, params, active, iters=600, other_tiling=True, restore_tiling=0):
    B = active.shape[0]
    p2 = {k: v.clone() for k, v in params.items()}
    st2 = eng.new_adam_state(p2)
    a2 = active.clone()
    if other_tiling:
        name = eng.fit_variant(B)
        eng.set_tiling(64 if "_g16" in name else 16)
        if not eng.fit_variant(B):
            eng.set_tiling(restore_tiling)
    left = iters
    while left > 0:
        eng.fit(target, p2, st2, a2, min(100, left))
        left -= 100
    torch.cuda.synchronize()
    eng.set_tiling(restore_tiling)


def warm_shared_engine(eng, target, params, lists, iters=1200):
    p2 = {k: v.clone() for k, v in params.items()}
    st2 = eng.new_adam_state(p2)
    l2 = lists.clone()
    eng.fit(target, p2, st2, l2, iters)
    torch.cuda.synchronize()
