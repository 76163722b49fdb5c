// smoe_var.hip -- the per-block kernels of ONE (dim, channels, kernels) triple; compiled once per line of
// smoe_variants.def with -DSMOE_D= -DSMOE_C= -DSMOE_K= -DSMOE_FULL= (csrc/Makefile).
#include "smoe_block.hip.h"
#include "smoe_team.hip.h"
#include "smoe_duo.hip.h"

#if !defined(SMOE_D) || !defined(SMOE_C) || !defined(SMOE_K) || !defined(SMOE_FULL)
#error "compile with -DSMOE_D -DSMOE_C -DSMOE_K -DSMOE_FULL (see csrc/Makefile)"
#endif

#define SMOE_CAT_(a, b, c, d, e, f, g) a##b##c##d##e##f##g
#define SMOE_CAT(a, b, c, d, e, f, g) SMOE_CAT_(a, b, c, d, e, f, g)

namespace smoe {
// wavefronts per workgroup of the 16-lane tiling: four while the block images of 16 blocks fit next to three more workgroups
#ifndef SMOE_W16
#if SMOE_D == 2 && SMOE_K * SMOE_C <= 12
#define SMOE_W16 4
#else
#define SMOE_W16 2
#endif
#endif

// host function (a namespace-scope table of host function pointers would also be emitted for the device)
const Variant* SMOE_CAT(variants_d, SMOE_D, c, SMOE_C, k, SMOE_K, )(int* count) {
#if SMOE_FULL
    static const Variant table[] = { SMOE_VARIANT(SMOE_D, SMOE_C, SMOE_K, 16, SMOE_W16), SMOE_VARIANT_BASIC(SMOE_D, SMOE_C, SMOE_K, 32, 2),
                                     SMOE_VARIANT(SMOE_D, SMOE_C, SMOE_K, 64, 2) };
#else
    static const Variant table[] = { SMOE_VARIANT_BASIC(SMOE_D, SMOE_C, SMOE_K, 16, SMOE_W16), SMOE_VARIANT_BASIC(SMOE_D, SMOE_C, SMOE_K, 64, 2) };
#endif
    *count = (int)(sizeof(table) / sizeof(table[0]));
    return table;
}
}  // namespace smoe
