// Host-side helpers of the C ABI (no device code).
//
// rpg_host_f32_to_bf16: fp32 -> bf16 (round to nearest even, NaN -> quiet NaN: the rounding of the device's (__bf16)float and of
// torch's .bfloat16()) of a contiguous host array.  evaluate.evaluate_stream calls it from its staging threads (ctypes releases
// the GIL) to round the node images of the bf16 encoder while they are copied into the pinned H2D buffer: torch's converting
// copy_ did not scale over Python threads (r3: 2.6 k graphs/s with 8 threads, 1.7 k with 32, against 4.2 k for the plain fp32
// staging at 256 x 341).
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../../include/relpose_gnn_hip.h"

extern "C" int rpg_host_f32_to_bf16(const float* src, void* dst_bf16, size_t count) {
    if ((!src || !dst_bf16) && count) return RPG_ERR_BAD_ARG;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    uint16_t* d = reinterpret_cast<uint16_t*>(dst_bf16);
    for (size_t i = 0; i < count; ++i) {
        const uint32_t u = s[i];
        const uint32_t rounded = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
        const uint32_t nan = (u >> 16) | 0x40u;
        d[i] = (uint16_t)(((u & 0x7FFFFFFFu) > 0x7F800000u) ? nan : rounded);
    }
    return RPG_OK;
}
