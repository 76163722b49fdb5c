// Fake-quant helpers shared by the block kernels (smoe_block.hip.h) and the shared-kernel mode (smoe_shared.hip):
// TF fake_quant_with_min_max_{args,vars} (smoe.py:474-538) in fp32, as the TF device kernels compute it.
#pragma once
#include <hip/hip_runtime.h>

#include "smoe_device.h"

namespace smoe {

struct FqRange {
    float nmin, nmax, scale, inv, back;   // nudged range, step, 1/step, offset added back (x - min forms)
    float shift;                          // what is taken off the input first (= back, except radial_as: 0)
    bool zero;                            // min == max == 0: TF outputs zeros and passes the whole gradient
};

__device__ __forceinline__ FqRange fq_fixed(const KernelConsts& kc, int g) {
    FqRange r;
    r.nmin = kc.q_nmin[g]; r.nmax = kc.q_nmax[g]; r.scale = kc.q_scale[g]; r.inv = kc.q_inv[g];
    r.back = 0.0f; r.shift = 0.0f; r.zero = false;
    return r;
}

// TF Nudge() on [rmin, rmax] with levels = 2^bits - 1 (fake_quant_ops_functor.h), fp32 as on the device there.
// offset: fake_quant(x - min, 0, max - min) + min (A_diagonal, nu_e); noshift (radial_as steering, smoe.py:498-504):
// fake_quant(x, 0, max - min) + min -- the reference does not shift the input there; restated as is.
__device__ __forceinline__ FqRange fq_vars(float lo, float hi, float levels, bool offset, bool noshift = false) {
    FqRange r;
    const float rmin = offset ? 0.0f : lo;
    const float rmax = offset ? hi - lo : hi;
    r.back = offset ? lo : 0.0f;
    r.shift = noshift ? 0.0f : r.back;
    r.zero = (rmin == 0.0f) && (rmax == 0.0f);
    r.scale = (rmax - rmin) / levels;
    const float zp = 0.0f - rmin / r.scale;
    const float nzp = (zp < 0.0f) ? 0.0f : ((zp > levels) ? levels : roundf(zp));
    r.nmin = (0.0f - nzp) * r.scale;
    r.nmax = (levels - nzp) * r.scale;
    r.inv = 1.0f / r.scale;
    return r;
}

__device__ __forceinline__ float fq_val(float x, const FqRange& r) {
    const float v = x - r.shift;
    const float cl = fminf(fmaxf(v, r.nmin), r.nmax);
    const float q = floorf((cl - r.nmin) * r.inv + 0.5f) * r.scale + r.nmin;
    return (r.zero ? 0.0f : q) + r.back;
}

}  // namespace smoe
