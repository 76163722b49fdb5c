"""MI355X-native per-block Steered-Mixture-of-Experts fit / reconstruction hot path.

Host code is Python (this package) calling hand-written gfx950 HIP kernels through a C ABI
(``include/smoe_hip.h`` -> ``libsmoe_hip.so``).  The public surface mirrors the reference's
``smoe.Smoe`` / ``smoe_reconstruction`` entry points (see ``smoe.py``).
"""
__all__ = ["engine", "_lib"]
