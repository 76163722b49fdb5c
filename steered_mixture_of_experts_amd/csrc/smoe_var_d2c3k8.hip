// Instantiations of the per-block kernels for one (dim, channels, kernels) combination: 16 and 64 lanes per block.
#include "smoe_block.cuh"

namespace smoe {
// host function (a namespace-scope table of host function pointers would also be emitted for the device)
const Variant* variants_d2c3k8() {
    static const Variant table[2] = { SMOE_VARIANT(2, 3, 8, 16, 2), SMOE_VARIANT(2, 3, 8, 64, 2) };
    return table;
}
}  // namespace smoe
