// Host-side helpers of the C ABI (no device code).
//
// rpg_host_f32_to_bf16: fp32 -> bf16 (round to nearest even, NaN -> quiet NaN: the rounding of the device's (__bf16)float and of
// torch's .bfloat16()) of a contiguous host array.  evaluate.evaluate_stream calls it from its staging threads (ctypes releases
// the GIL) to round the node images of the bf16 encoder while they are copied into the pinned H2D buffer: torch's converting
// copy_ did not scale over Python threads (r3: 2.6 k graphs/s with 8 threads, 1.7 k with 32, against 4.2 k for the plain fp32
// staging at 256 x 341).
//
// Round 6: the same integer arithmetic on 8 / 16 lanes (AVX2 / AVX-512F, chosen once at run time from the CPU's feature bits: the
// library is built on one machine and runs on another) -- bit-identical by construction, denormals and NaN payloads included (the
// AVX512-BF16 conversion instruction would flush denormal inputs) --, with non-temporal stores (the destination is a pinned staging
// buffer that only the DMA engine reads next).  Measured on the GPU host (tools/probes/c4_legs.py): 19.8 -> 26.8 GB/s of fp32 per
// thread, 143 -> 182 GB/s with 16 threads; the baseline-x86-64 loop was NOT what bounded the host-rounded stream (55 GB/s needed) --
// its host loop was (evaluate.py: post-processing now on a helper thread).
#if !defined(__HIP_DEVICE_COMPILE__)          // (host pass only: hipcc parses a .hip file for the device too, where the CPU builtins do not exist)
#include <immintrin.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../../include/relpose_gnn_hip.h"

namespace {

inline uint16_t round_one(uint32_t u) {
    const uint32_t rounded = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    const uint32_t nan = (u >> 16) | 0x40u;
    return (uint16_t)(((u & 0x7FFFFFFFu) > 0x7F800000u) ? nan : rounded);
}

void round_scalar(const uint32_t* s, uint16_t* d, size_t count) {
    for (size_t i = 0; i < count; ++i) d[i] = round_one(s[i]);
}

__attribute__((target("avx2"))) inline __m256i round8(__m256i u) {
    const __m256i hi = _mm256_srli_epi32(u, 16);
    const __m256i rounded = _mm256_srli_epi32(_mm256_add_epi32(_mm256_add_epi32(u, _mm256_set1_epi32(0x7FFF)), _mm256_and_si256(hi, _mm256_set1_epi32(1))), 16);
    const __m256i nan = _mm256_or_si256(hi, _mm256_set1_epi32(0x40));
    // |x| > inf as a SIGNED compare of the magnitude bits (both operands are < 2^31)
    const __m256i isnan = _mm256_cmpgt_epi32(_mm256_and_si256(u, _mm256_set1_epi32(0x7FFFFFFF)), _mm256_set1_epi32(0x7F800000));
    return _mm256_blendv_epi8(rounded, nan, isnan);
}

__attribute__((target("avx2"))) void round_avx2(const uint32_t* s, uint16_t* d, size_t count) {
    size_t i = 0;
    while (i < count && (reinterpret_cast<uintptr_t>(d + i) & 31)) { d[i] = round_one(s[i]); ++i; }     // aligned 32-byte stores from here on
    for (; i + 16 <= count; i += 16) {
        const __m256i a = round8(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i)));
        const __m256i b = round8(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i + 8)));
        // packus works per 128-bit lane: [a0..3 b0..3 | a4..7 b4..7] -> put the quarters in order (values are < 2^16: no saturation)
        const __m256i p = _mm256_permute4x64_epi64(_mm256_packus_epi32(a, b), 0xD8);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i), p);
    }
    _mm_sfence();
    for (; i < count; ++i) d[i] = round_one(s[i]);
}

__attribute__((target("avx512f"))) void round_avx512(const uint32_t* s, uint16_t* d, size_t count) {
    size_t i = 0;
    while (i < count && (reinterpret_cast<uintptr_t>(d + i) & 31)) { d[i] = round_one(s[i]); ++i; }
    const __m512i c7fff = _mm512_set1_epi32(0x7FFF), one = _mm512_set1_epi32(1), q = _mm512_set1_epi32(0x40);
    const __m512i mag = _mm512_set1_epi32(0x7FFFFFFF), inf = _mm512_set1_epi32(0x7F800000);
    for (; i + 16 <= count; i += 16) {
        const __m512i u = _mm512_loadu_si512(s + i);
        const __m512i hi = _mm512_srli_epi32(u, 16);
        const __m512i rounded = _mm512_srli_epi32(_mm512_add_epi32(_mm512_add_epi32(u, c7fff), _mm512_and_si512(hi, one)), 16);
        const __mmask16 isnan = _mm512_cmpgt_epi32_mask(_mm512_and_si512(u, mag), inf);
        const __m512i r = _mm512_mask_mov_epi32(rounded, isnan, _mm512_or_si512(hi, q));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i), _mm512_cvtepi32_epi16(r));
    }
    _mm_sfence();
    for (; i < count; ++i) d[i] = round_one(s[i]);
}

using RoundFn = void (*)(const uint32_t*, uint16_t*, size_t);

RoundFn pick(int isa) {            // isa: 0 = best the CPU has, 1 = scalar, 2 = AVX2, 3 = AVX-512F; nullptr if the CPU lacks it
    __builtin_cpu_init();
    const bool a2 = __builtin_cpu_supports("avx2"), a5 = __builtin_cpu_supports("avx512f");
    switch (isa) {
        case 0: return a2 ? round_avx2 : (a5 ? round_avx512 : round_scalar);      // (measured on 2 x EPYC 9575F, 16 threads: AVX2 182 GB/s, AVX-512F 161, scalar / SSE2 143)
        case 1: return round_scalar;
        case 2: return a2 ? round_avx2 : nullptr;
        case 3: return a5 ? round_avx512 : nullptr;
        default: return nullptr;
    }
}

}  // namespace

extern "C" int rpg_host_f32_to_bf16(const float* src, void* dst_bf16, size_t count) {
    if ((!src || !dst_bf16) && count) return RPG_ERR_BAD_ARG;
    static const RoundFn best = pick(0);
    best(reinterpret_cast<const uint32_t*>(src), reinterpret_cast<uint16_t*>(dst_bf16), count);
    return RPG_OK;
}

extern "C" int rpg_host_f32_to_bf16_isa(const float* src, void* dst_bf16, size_t count, int isa) {
    if ((!src || !dst_bf16) && count) return RPG_ERR_BAD_ARG;
    const RoundFn f = pick(isa);
    if (!f) return RPG_ERR_BAD_ARG;        // unknown ISA index, or one this CPU does not have
    f(reinterpret_cast<const uint32_t*>(src), reinterpret_cast<uint16_t*>(dst_bf16), count);
    return RPG_OK;
}
#endif  // !__HIP_DEVICE_COMPILE__
