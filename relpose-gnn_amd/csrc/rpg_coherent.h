// Device-side helpers of the in-kernel combine of split tiles (round 4): agent-scope (sc1) slab accesses and the arrival ticket.
// Their own header (round 5) because they are CODE of the Winograd kernel: the PMC profile of that kernel is stamped
// with the digest of winograd.hip + this file (build.WINOGRAD_SOURCES), so an edit here orphans the committed profile as it should,
// while rpg_common.h (declarations of the other translation units) stays out of the digest.
#pragma once
#include <hip/hip_runtime.h>

namespace rpg {
// Agent-scope ("sc1") 16-byte accesses for data that one workgroup writes and ANOTHER workgroup of the SAME launch reads
// (the partial tiles of the split-K / stream-K launches, combined by the last-arriving workgroup).  The L2 caches of the 8
// XCDs are not coherent with each other for ordinary accesses; the portable way -- __threadfence() = buffer_wbl2 + buffer_inv
// of the whole L2 -- was measured at 2.2x the step time (round 4: 23.0 vs 10.7 ms at configs[1]).  An sc1 store is written
// through to memory and an sc1 load does not hit a stale line, which is how agent-scope atomics are coherent on gfx942 /
// gfx950; with ONLY such accesses to the slabs, ordering needs no cache maintenance: the writers wait for their stores
// (s_waitcnt vmcnt(0)) before the ticket atomic, the reader loads after its own ticket atomic has returned.
// cache-policy immediate of the raw buffer intrinsics on gfx940+: bit 0 = sc0, bit 1 = nt, bit 4 = sc1.
typedef unsigned int rpg_u32x4 __attribute__((ext_vector_type(4)));
constexpr int kAuxAgentScope = 16;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t agent_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void agent_store_f4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned s_off, const float4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rpg_u32x4, v), r, byte_off, s_off, kAuxAgentScope);
}
__device__ __forceinline__ float4 agent_load_f4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned s_off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, s_off, kAuxAgentScope));
}
// Ticket of the arrival counter: true for the workgroup that completes the count.  Every thread of the workgroup calls it
// after its slab stores; `flag` = an int of LDS nobody is using.  The counter goes back to zero for the next launch.
__device__ __forceinline__ bool last_arriver(unsigned* counter, unsigned contributors, int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's sc1 slab stores have been written through
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old + 1u == contributors;
        if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last;
    }
    __syncthreads();
    return *flag != 0;
}

}  // namespace rpg
