// bf16 encoder path for gfx950 (BASELINE.json configs[2]: "bf16 activations + MFMA conv"): the ResNet (BasicBlock)
// encoder with bf16 NHWC activations and bf16 weights, fp32 accumulation on v_mfma_f32_32x32x16_bf16 (16x the f32
// matrix rate, so these convolutions are HBM/L2-bound), fp32 BatchNorm / residual / ReLU epilogue, bf16 stores.  The
// encoder's last Linear writes fp32 features, so the GNN behind it is the unchanged fp32 path.
//
// The tile engine is the byte-for-byte twin of the f32 one (gemm_f32.hip) at K-step 32 elements = 64 bytes per row:
// 16-byte staging slots, 80-byte LDS pitch (conflict-free ds_read_b128), double-buffered LDS, register staging, one
// barrier per step, LDS-transposed epilogue.  A lane's 16-byte LDS fragment is exactly the 8 consecutive k the bf16
// MFMA wants from it (k = 8*(lane>>5) + 0..7 of a 16-wide chunk), so no k permutation is involved here.
// Looser parity bar than the f32 path: see tests (<= 3e-2 max-norm relative on the forward; stated there).
#include "rpg_common.h"
#ifdef RPG_PATCH_TRACE
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NT = 256;
// BK = bf16 elements per K step (32 = 64 bytes per row, 64 = 128 bytes); LDS pitch BK + 8 elements (80 / 144 bytes: both
// put the 16 rows of a ds_read_b128 lane group on distinct 4-bank slots); BK / 8 sixteen-byte staging slots per row.

struct EpiB {
    const float* scale;        // fp32 per output channel (folded BN) or null
    const float* shift;        // fp32 per output channel or null
    const __bf16* residual;    // bf16 [M][ldc] or null
    void* out;                 // bf16 [M][ldc], or fp32 when out_f32
    int ldc;
    int relu;
    int out_f32;
    // fp32 residual rows for the bf16 Linears of the GNN: row m adds res_f32[(res_idx ? res_idx[m] : m) * ldr + n]
    // and, if given, res2_f32[res2_idx[m] * ldr + n]
    const float* res_f32 = nullptr;
    const int64_t* res_idx = nullptr;
    const float* res2_f32 = nullptr;
    const int64_t* res2_idx = nullptr;
    int ldr = 0;
    // optional SECOND output in bf16 (round 3: the next GEMM's A operand straight from this epilogue instead of a separate
    // f32 -> bf16 pass): out2[m * ld2 + n] = bf16(relu2 ? max(y, 0) : y) of the same y (before the primary's own ReLU);
    // `out` may then be null (bf16-only consumer)
    __bf16* out2 = nullptr;
    int ld2 = 0;
    int relu2 = 0;
    // set by the launcher: the plain convolution case (bf16 output, optional bf16 residual, none of the fp32 / second-output
    // options, N % 4 == 0, a workgroup's rows addressable with 32-bit byte offsets) -> bf16_tile_epilogue_lean_body
    int lean = 0;
};

struct ConvArgsB {
    const __bf16* x;
    int H, W, Cin, KH, KW, stride, pad, Ho, Wo;
    int img_elems = 0;         // elements between consecutive images; 0 = H * W * Cin (a Linear whose A rows have a pitch lda > k)
};

__device__ __forceinline__ uint4 ld16_or_zero(const __bf16* p, bool ok) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (ok) v = *reinterpret_cast<const uint4*>(p);
    return v;
}

// Epilogue shared by both kernels: each wave transposes its accumulators through a private fp32 LDS slab; each lane
// finishes 4 consecutive channels of a row (BatchNorm scale/shift, bf16 residual, ReLU, bf16 or fp32 store).
template <int FM, int FN, int SLAB_BUDGET_BYTES, int NWAVES = 4>
__device__ __forceinline__ void bf16_tile_epilogue(f32x16 (&acc)[FM][FN], unsigned char* lds_raw, const EpiB& ep, int m0,
                                                   int n0, int M, int N, int wm, int wn, int lane, int wave) {
    constexpr int EW = FN * 32, EPITCH = EW + 4, C4 = EW / 4, RPI = 64 / C4, NIT = 32 / RPI;
    static_assert(NWAVES * 32 * EPITCH * 4 <= SLAB_BUDGET_BYTES, "epilogue slab does not fit the staging LDS");
    float* slab = reinterpret_cast<float*>(lds_raw) + wave * (32 * EPITCH);
    const int c4 = lane % C4, r_in = lane / C4;
    const int nb = n0 + wn * EW + 4 * c4;
    const bool n_ok = nb < N;                 // N % 4 == 0
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n_ok && ep.scale) sc = *reinterpret_cast<const float4*>(ep.scale + nb);
    if (n_ok && ep.shift) sh = *reinterpret_cast<const float4*>(ep.shift + nb);
    // The bf16 residual rows of a 32-row fragment are requested as one batch, one fragment ahead (unconditional loads on
    // clamped addresses): read one at a time inside the row loop below, each was a fully exposed round trip (measured at
    // 512 images: a residual cost +122 us on a layer-1 convolution, +58 us on layer 2 -- five times its HBM time).
    bf16x4 rs_cur[NIT], rs_nxt[NIT];
    auto load_res = [&](int i, bf16x4 (&dst)[NIT]) {
        if (!ep.residual) return;
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            const int m = m0 + (wm * FM + i) * 32 + r_in + RPI * t;
            const size_t o = n_ok && m < M ? (size_t)m * ep.ldc + nb : 0;
            dst[t] = *reinterpret_cast<const bf16x4*>(ep.residual + o);
        }
    };
    load_res(0, rs_cur);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        if (i + 1 < FM) load_res(i + 1, rs_nxt);
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                slab[((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * EPITCH + j * 32 + (lane & 31)] = acc[i][j][e];
        __builtin_amdgcn_wave_barrier();
        const int mb = m0 + (wm * FM + i) * 32;
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            const int row = r_in + RPI * t;
            const int m = mb + row;
            const float4 v = *reinterpret_cast<const float4*>(&slab[row * EPITCH + 4 * c4]);
            if (n_ok && m < M) {
                const size_t o = (size_t)m * ep.ldc + nb;
                float4 y;
                y.x = v.x * sc.x + sh.x; y.y = v.y * sc.y + sh.y; y.z = v.z * sc.z + sh.z; y.w = v.w * sc.w + sh.w;
                if (ep.residual) {
                    const bf16x4 rs = rs_cur[t];
                    y.x += (float)rs[0]; y.y += (float)rs[1]; y.z += (float)rs[2]; y.w += (float)rs[3];
                }
                if (ep.res_f32) {
                    const size_t ro = (size_t)(ep.res_idx ? ep.res_idx[m] : (int64_t)m) * ep.ldr + nb;
                    const float4 rs = *reinterpret_cast<const float4*>(ep.res_f32 + ro);
                    y.x += rs.x; y.y += rs.y; y.z += rs.z; y.w += rs.w;
                    if (ep.res2_f32) {
                        const float4 r2 = *reinterpret_cast<const float4*>(ep.res2_f32 + (size_t)ep.res2_idx[m] * ep.ldr + nb);
                        y.x += r2.x; y.y += r2.y; y.z += r2.z; y.w += r2.w;
                    }
                }
                if (ep.out2) {
                    const float f2 = ep.relu2 ? 0.f : -INFINITY;
                    const bf16x4 o2 = {(__bf16)fmaxf(y.x, f2), (__bf16)fmaxf(y.y, f2), (__bf16)fmaxf(y.z, f2), (__bf16)fmaxf(y.w, f2)};
                    *reinterpret_cast<bf16x4*>(ep.out2 + (size_t)m * ep.ld2 + nb) = o2;
                }
                if (ep.relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
                if (!ep.out) {
                } else if (ep.out_f32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(ep.out) + o) = y;
                } else {
                    const bf16x4 ob = {(__bf16)y.x, (__bf16)y.y, (__bf16)y.z, (__bf16)y.w};
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(ep.out) + o) = ob;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < NIT; ++t) rs_cur[t] = rs_nxt[t];
    }
}

// The same epilogue for the plain convolution case (ep.lean), written for instruction count (round 4).  Measured with the
// epilogue compiled out of the patch kernel (tools/probes/patch_ablate.sh): the general version above costs 6.6 us of a 19-us
// 512 x 64 tile, 8.5 us of a 256 x 256 tile -- not memory time (de-phasing the workgroups changes nothing) but ~2,400 instructions
// per wave: a branch per row and option (131 s_cbranch), 64-bit address arithmetic per access (208 v_lshl_add_u64, 96 v_mad_u64),
// SGPR spills (156 v_readlane) and a conservative wait per region.  Here: raw buffer accesses based at the wave's first row
// (invalid rows / columns carry an out-of-range offset: loads return zero, stores are dropped -- no branches), 32-bit offsets
// built incrementally, the residual rows of a fragment requested one fragment ahead, the residual as a template parameter.
typedef unsigned int u32x2e __attribute__((ext_vector_type(2)));
template <int FM, int FN, int NWAVES, bool RES>
__device__ __forceinline__ void bf16_tile_epilogue_lean_body(f32x16 (&acc)[FM][FN], unsigned char* lds_raw, const EpiB& ep, int m0,
                                                             int n0, int M, int N, int wm, int wn, int lane, int wave) {
    constexpr int EW = FN * 32, EPITCH = EW + 4, C4 = EW / 4, RPI = 64 / C4, NIT = 32 / RPI;
    constexpr unsigned OOB = 0x80000000u;
    float* slab = reinterpret_cast<float*>(lds_raw) + wave * (32 * EPITCH);
    const int c4 = lane % C4, r_in = lane / C4;
    const int nb = n0 + wn * EW + 4 * c4;
    const bool n_ok = nb < N;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n_ok && ep.scale) sc = *reinterpret_cast<const float4*>(ep.scale + nb);
    if (n_ok && ep.shift) sh = *reinterpret_cast<const float4*>(ep.shift + nb);
    const int mw = m0 + wm * FM * 32;                        // first row of this wave (wave-uniform)
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<__bf16*>(ep.out) + (size_t)mw * ep.ldc, 0,
                                                                         0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(RES ? ep.residual : reinterpret_cast<const __bf16*>(ep.out)) + (size_t)mw * ep.ldc, 0, 0x7fffffff, 0x00020000);
    const unsigned row_b = 2u * (unsigned)ep.ldc;             // bytes per output row
    const unsigned base = n_ok ? (unsigned)r_in * row_b + 2u * (unsigned)nb : OOB;      // row r_in of fragment 0
    const int rows_left = M - mw;                             // rows of this wave's range that exist (may be <= 0)
    const bool relu = ep.relu != 0;          // (a select, not a max with -inf: a NaN accumulator of a non-ReLU convolution stays NaN, as in the general epilogue)
    auto off = [&](int i, int t) -> unsigned {
        const int r = i * 32 + r_in + RPI * t;
        return (n_ok && r < rows_left) ? base + (unsigned)(i * 32 + RPI * t) * row_b : OOB;
    };
    u32x2e rs_cur[NIT], rs_nxt[NIT];
    auto load_res = [&](int i, u32x2e (&dst)[NIT]) {
        if (!RES) return;
#pragma unroll
        for (int t = 0; t < NIT; ++t) dst[t] = __builtin_bit_cast(u32x2e, __builtin_amdgcn_raw_buffer_load_b64(rr, off(i, t), 0, 0));
    };
    load_res(0, rs_cur);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        if (i + 1 < FM) load_res(i + 1, rs_nxt);
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                slab[((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * EPITCH + j * 32 + (lane & 31)] = acc[i][j][e];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            const float4 v = *reinterpret_cast<const float4*>(&slab[(r_in + RPI * t) * EPITCH + 4 * c4]);
            float4 y;
            y.x = v.x * sc.x + sh.x; y.y = v.y * sc.y + sh.y; y.z = v.z * sc.z + sh.z; y.w = v.w * sc.w + sh.w;
            if (RES) {
                const bf16x4 rs = __builtin_bit_cast(bf16x4, rs_cur[t]);
                y.x += (float)rs[0]; y.y += (float)rs[1]; y.z += (float)rs[2]; y.w += (float)rs[3];
            }
            const bf16x4 ob = {(__bf16)(relu ? fmaxf(y.x, 0.f) : y.x), (__bf16)(relu ? fmaxf(y.y, 0.f) : y.y), (__bf16)(relu ? fmaxf(y.z, 0.f) : y.z),
                               (__bf16)(relu ? fmaxf(y.w, 0.f) : y.w)};
#if defined(RPG_PATCH_ABL) && (RPG_PATCH_ABL & 4)      // diagnostic: all stores of the wave land in one 64-KB window (L2-resident): the
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2e, ob), ro, off(i, t) & 0xffffu, 0, 0);      // epilogue without its HBM writes
#else
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2e, ob), ro, off(i, t), 0, 0);
#endif
        }
        __builtin_amdgcn_wave_barrier();
        if (RES) {
#pragma unroll
            for (int t = 0; t < NIT; ++t) rs_cur[t] = rs_nxt[t];
        }
    }
}

// dispatch: the lean form for plain convolutions, the general one otherwise (both forms live in every kernel; `lean` is uniform)
template <int FM, int FN, int SLAB_BUDGET_BYTES, int NWAVES = 4>
__device__ __forceinline__ void bf16_tile_epilogue_any(f32x16 (&acc)[FM][FN], unsigned char* lds_raw, const EpiB& ep, int m0,
                                                       int n0, int M, int N, int wm, int wn, int lane, int wave) {
    if (ep.lean) {
        if (ep.residual) bf16_tile_epilogue_lean_body<FM, FN, NWAVES, true>(acc, lds_raw, ep, m0, n0, M, N, wm, wn, lane, wave);
        else bf16_tile_epilogue_lean_body<FM, FN, NWAVES, false>(acc, lds_raw, ep, m0, n0, M, N, wm, wn, lane, wave);
    } else {
        bf16_tile_epilogue<FM, FN, SLAB_BUDGET_BYTES, NWAVES>(acc, lds_raw, ep, m0, n0, M, N, wm, wn, lane, wave);
    }
}

template <int BM, int BN, int WM, int WN, int BK>
__global__ __launch_bounds__(NT) void conv_bf16_kernel(ConvArgsB a, const __bf16* __restrict__ Wt, int M, int N, int K,
                                                       EpiB ep, int tiles_n) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int LD = BK + 8, SLOTS = BK / 8, ROWS_PER_PASS = NT / SLOTS;
    constexpr int FM = BM / WM / 32, FN = BN / WN / 32;
    constexpr int RA = BM / ROWS_PER_PASS, RW = BN / ROWS_PER_PASS;
    static_assert(RA >= 1 && RW >= 1, "tile too small for this K step");
    constexpr int STAGE = (BM + BN) * LD;                 // elements per LDS buffer
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __bf16* lds = reinterpret_cast<__bf16*>(lds_raw);

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3, q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int slot = tid % SLOTS, srow = tid / SLOTS;

    // ---- A loader state: implicit-GEMM over NHWC bf16, k = (kh, kw, c)
    const __bf16* img[RA];
    int hi0[RA], wi0[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
        const int m = m0 + srow + ROWS_PER_PASS * j;
        if (m < M) {
            const int wo = m % a.Wo;
            const int t = m / a.Wo;
            img[j] = a.x + (size_t)(t / a.Ho) * (a.img_elems ? a.img_elems : a.H * a.W * a.Cin);
            hi0[j] = (t % a.Ho) * a.stride - a.pad;
            wi0[j] = wo * a.stride - a.pad;
        } else {
            img[j] = a.x;
            hi0[j] = -(1 << 24);
            wi0[j] = 0;
        }
    }
    int kc, kw_, kh_;
    {
        const int k0 = 8 * slot;
        kc = k0 % a.Cin;
        const int t = k0 / a.Cin;
        kw_ = t % a.KW;
        kh_ = t / a.KW;
    }
    const __bf16* wrow[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int n = n0 + srow + ROWS_PER_PASS * j;
        wrow[j] = (n < N) ? Wt + (size_t)n * K : nullptr;
    }
    int kk = 8 * slot;

    uint4 ra[RA], rw[RW];
    auto fetch = [&]() {
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int hi = hi0[j] + kh_, wi = wi0[j] + kw_;
            const bool ok = (kh_ < a.KH) && ((unsigned)hi < (unsigned)a.H) && ((unsigned)wi < (unsigned)a.W);
            ra[j] = ld16_or_zero(img[j] + ((hi * a.W + wi) * a.Cin + kc), ok);
        }
#pragma unroll
        for (int j = 0; j < RW; ++j) rw[j] = ld16_or_zero(wrow[j] + kk, (wrow[j] != nullptr) && (kk < K));
    };
    auto advance = [&]() {
        kc += BK;
        while (kc >= a.Cin) {
            kc -= a.Cin;
            if (++kw_ == a.KW) { kw_ = 0; ++kh_; }
        }
        kk += BK;
    };
    auto stage = [&](int buf) {
        __bf16* As = lds + buf * STAGE;
        __bf16* Ws = As + BM * LD;
#pragma unroll
        for (int j = 0; j < RA; ++j)
            *reinterpret_cast<uint4*>(&As[(srow + ROWS_PER_PASS * j) * LD + 8 * slot]) = ra[j];
#pragma unroll
        for (int j = 0; j < RW; ++j)
            *reinterpret_cast<uint4*>(&Ws[(srow + ROWS_PER_PASS * j) * LD + 8 * slot]) = rw[j];
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    fetch();
    stage(0);
    __syncthreads();
    const int nk = (K + BK - 1) / BK;
    const int a_off = (wm * FM * 32 + (lane & 31)) * LD + 8 * (lane >> 5);
    const int b_off = (BM + wn * FN * 32 + (lane & 31)) * LD + 8 * (lane >> 5);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            advance();
            fetch();
        }
        const __bf16* L = lds + cur * STAGE;
#pragma unroll
        for (int kb = 0; kb < BK; kb += 16) {
            bf16x8 av[FM], bv[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i)
                av[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&L[a_off + i * 32 * LD + kb]));
#pragma unroll
            for (int j = 0; j < FN; ++j)
                bv[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&L[b_off + j * 32 * LD + kb]));
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (more) stage(cur ^ 1);
        __syncthreads();
    }

    bf16_tile_epilogue<FM, FN, 2 * STAGE * 2>(acc, lds_raw, ep, m0, n0, M, N, wm, wn, lane, wave);
}

// ---------------------------------------------------------------------------------------------------------------
// The kernel for Cin % 64 == 0 (every convolution but the stem), built by the rules measured for the f32 engine
// (gemm_f32.hip, tile_mainloop_b): nothing but MFMAs issues in clumps.  K step 64 elements = 128-byte LDS rows of
// eight 16-byte chunks, unpadded, chunk c of row r stored at c ^ ((r >> 1) & 7) (conflict-free ds_read_b128 lane
// groups, contiguous ds_write_b128 rows); two LDS images; raw buffer loads whose lane-dependent offsets are
// K-invariant (rebuilt in a rarely taken branch when the kernel tap changes; out-of-image taps and ragged rows carry
// an out-of-range offset = hardware zero fill) with the K position in the scalar offset; the stage writes of step
// t+1, the loads of step t+2 and the operand reads of the next MFMA group are spread one or two at a time behind the
// 4 x FM*FN MFMAs of step t (v_mfma_f32_32x32x16_bf16: 32 cycles each, LDS / VMEM instructions hide beside it); one
// barrier per step, placed before the last MFMA group, which already reads the next image.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(NT) void conv_bf16_fast_kernel(ConvArgsB a, const __bf16* __restrict__ Wt, int M, int N, int K,
                                                            EpiB ep, int tiles_n) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int BK = 64, LD = BK, SLOTS = BK / 8, ROWS_PER_PASS = NT / SLOTS;      // 32 rows per staging pass
    constexpr int FM = BM / WM / 32, FN = BN / WN / 32;
    constexpr int RA = BM / ROWS_PER_PASS, RW = BN / ROWS_PER_PASS, NJ = RA + RW;
    constexpr int KB = BK / 16, G = FM * FN, NR = FM + FN;
    constexpr int STAGE = (BM + BN) * LD;                 // elements per LDS image
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __bf16* lds = reinterpret_cast<__bf16*>(lds_raw);

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3, q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int slot = tid % SLOTS, srow = tid / SLOTS;

    // ---- A rows: element offset of the row's image relative to the tile's first image, top-left input pixel
    const int n_first = m0 / (a.Ho * a.Wo);
    const int img = a.img_elems ? a.img_elems : a.H * a.W * a.Cin;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x) + (size_t)n_first * img, 0,
                                                                          0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Wt), 0, 0x7fffffff, 0x00020000);
    int base[RA], hi0[RA], wi0[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
        const int m = m0 + srow + ROWS_PER_PASS * j;
        base[j] = -1; hi0[j] = 0; wi0[j] = 0;
        if (m < M) {
            const int wo = m % a.Wo;
            const int t = m / a.Wo;
            base[j] = (t / a.Ho - n_first) * img;
            hi0[j] = (t % a.Ho) * a.stride - a.pad;
            wi0[j] = wo * a.stride - a.pad;
        }
    }
    unsigned voff[RA], woff[RW], weff[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int n = n0 + srow + ROWS_PER_PASS * j;
        woff[j] = n < N ? 2u * (unsigned)(n * K + 8 * slot) : OOB;
    }
    int kh = 0, kw = 0, c0 = 0, k0 = 0;                   // wave-uniform position of the next fetch (K % 64 == 0)
    auto refresh = [&]() {                                // offsets of tap (kh, kw); everything invalid past K
        asm volatile("");                                 // keeps the callers' branch
        const bool live = k0 < K;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int hi = hi0[j] + kh, wi = wi0[j] + kw;
            const bool ok = live && base[j] >= 0 && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            voff[j] = ok ? 2u * (unsigned)(base[j] + (hi * a.W + wi) * a.Cin + 8 * slot) : OOB;
        }
#pragma unroll
        for (int j = 0; j < RW; ++j) weff[j] = live ? woff[j] : OOB;
    };
    refresh();
    // Staging registers (A rows, then W rows), TWO sets: K step t stages step t+1 from set (t+1) & 1 and then loads step t+3
    // into the same set, so a load has two and a half steps to land.  (With one set -- loaded half a step before it is
    // staged -- a step of 16 MFMAs x 32 cycles hid nothing of a 2-5 k-cycle round trip: the 128 x 128 kernel ran at 3.1 k
    // cycles per step and pair of workgroups, 1 k of them MFMA.)
    uint4 rr[2][NJ];
    auto load_job = [&](int qj, int set) {
        if (qj < RA)
            rr[set][qj] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsa, voff[qj], 2u * (unsigned)c0, 0));
        else
            rr[set][qj] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsw, weff[qj - RA], 2u * (unsigned)k0, 0));
    };
    auto next_k = [&]() {
        k0 += BK;
        c0 += BK;
        if (c0 >= a.Cin || k0 >= K) {
            if (c0 >= a.Cin) { c0 = 0; if (++kw == a.KW) { kw = 0; ++kh; } }
            refresh();
        }
    };
    const int st_off = srow * LD + 8 * (slot ^ ((srow >> 1) & 7));      // ROWS_PER_PASS is a multiple of 16: same swizzle per pass
    auto write_job = [&](int qj, int im, int set) {
        const int rrow = qj < RA ? ROWS_PER_PASS * qj : BM + ROWS_PER_PASS * (qj - RA);
        *reinterpret_cast<uint4*>(&lds[im + st_off + rrow * LD]) = rr[set][qj];
    };
    // operand fragment of lane (row = lane & 31, k half = lane >> 5) for the 16-wide k group g: logical chunk 2g + half
    const int lrow = lane & 31, sw = (lrow >> 1) & 7, half = lane >> 5;
    const int a_row = (wm * FM * 32 + lrow) * LD, b_row = (BM + wn * FN * 32 + lrow) * LD;
    bf16x8 fr[2][NR];
    auto read_job = [&](int set, int rj, int g, int im) {
        const int off = (rj < FM ? a_row + rj * 32 * LD : b_row + (rj - FM) * 32 * LD) + 8 * ((2 * g + half) ^ sw);
        fr[set][rj] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&lds[im + off]));
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // side jobs of MFMA group g: the NR operand reads of the next group, then (g < KB-2 .. ) stage writes / loads
    auto kstep = [&](int cur, int nxt, int set_n) {       // set_n: the register set of the step staged into nxt
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < KB; ++g) {
            const int set = g & 1;
            // jobs of this group: NR reads, then writes (first half of the step) or loads (second half)
            constexpr int WPG = (NJ + KB / 2 - 1) / (KB / 2);          // writes (loads) per group
            constexpr int JOBS = NR + WPG, PER = (JOBS + G - 1) / G;
#pragma unroll
            for (int ms = 0; ms < G; ++ms) {
                const int i = ms / FN, j = ms % FN;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[set][i], fr[set][FM + j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                    const int job = ms * PER + u;
                    if (job < NR) {
                        if (g + 1 < KB) read_job(set ^ 1, job, g + 1, cur);
                        else read_job(0, job, 0, nxt);
                    } else if (job < JOBS) {
                        const int w = (g % (KB / 2)) * WPG + (job - NR);
                        if (w < NJ) {
                            if (g < KB / 2) write_job(w, nxt, set_n);
                            else load_job(w, set_n);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g == KB - 2) {
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        next_k();
    };

#pragma unroll
    for (int qj = 0; qj < NJ; ++qj) load_job(qj, 0);      // step 0
    next_k();
#pragma unroll
    for (int qj = 0; qj < NJ; ++qj) load_job(qj, 1);      // step 1
    next_k();
#pragma unroll
    for (int qj = 0; qj < NJ; ++qj) write_job(qj, 0, 0);
#pragma unroll
    for (int qj = 0; qj < NJ; ++qj) load_job(qj, 0);      // step 2
    next_k();
    __syncthreads();
#pragma unroll
    for (int rj = 0; rj < NR; ++rj) read_job(0, rj, 0, 0);
    const int nk = K / BK;
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        kstep(0, STAGE, 1);
        kstep(STAGE, 0, 0);
    }
    if (kt < nk) kstep(0, STAGE, 1);
    __syncthreads();
    bf16_tile_epilogue<FM, FN, 2 * STAGE * 2>(acc, lds_raw, ep, m0, n0, M, N, wm, wn, lane, wave);
}

// ---------------------------------------------------------------------------------------------------------------
// The LDS-DMA kernel (round 3).  PMC of the interleaved kernel above on layer 3 showed 37 % MFMA-busy next to 40 % LDS-busy
// and 43 % TA-busy with waves waiting on LDS a quarter of their cycles: every operand byte crossed the register file twice
// (buffer_load -> VGPR -> ds_write_b128) and a 64 x 64 wave tile reads one LDS fragment per MFMA.  Here
//   * global -> LDS goes by `buffer_load_dwordx4 ... lds` (LDS-DMA): no staging registers, no ds_write instructions; a wave
//     instruction moves one 1-KB piece = 64 / CH rows of CH 16-byte chunks to M0 + 16 * lane (lane-linear destination), so
//     the XOR swizzle that keeps ds_read_b128 conflict-free is applied to the SOURCE chunk each lane fetches;
//   * out-of-image taps, ragged rows and the steps past K carry an out-of-range offset: the DMA writes zeros (probed:
//     tools/probes/glds_probe.hip), so borders cost nothing and every step issues the same number of loads -- which is
//     what makes a COUNTED s_waitcnt vmcnt((ST - 2) * pieces) possible: ST LDS images, loads run ST - 1 steps ahead of the
//     MFMAs and are never drained inside the loop (raw s_barrier, not __syncthreads(), which would wait vmcnt(0));
//   * 8 waves (two per SIMD) on tiles up to 256 x 256: a 128 x 64 wave tile reads 6 fragments per 8 MFMAs;
//   * one barrier per K step: it publishes step t's pieces and retires image (t - 1) % ST, which the loads issued right
//     behind it (step t + ST - 1) overwrite.
// Requirements: Cin % BK == 0 (a K step never straddles a kernel tap), 32-bit offsets (checked by the launcher).
typedef __attribute__((address_space(3))) void* lds_void_ptr;
// (a plain function, not spelled inside the kernel template: the builtin does not exist for the host target, and an
// instantiation that names it is silently dropped from the host side -- no launch stub, an undefined symbol at load time)
__device__ __forceinline__ void dma_piece16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_dst, unsigned voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)lds_dst, 16, voffset, soffset, 0, 0);
}

template <int BM, int BN, int WM, int WN, int BK, int ST>
__device__ __forceinline__ void conv_bf16_dma_body(const ConvArgsB& a, const __bf16* __restrict__ Wt, int M, int N, int K, const EpiB& ep,
                                                   int tiles_n, int bid, int nwg) {
    constexpr int NW = WM * WN;
    constexpr int CH = BK / 8, RPG = 64 / CH;                  // 16-byte chunks per row; rows per 1-KB piece
    constexpr int NA = BM / RPG, NB = BN / RPG;                // pieces per image
    static_assert(NA % NW == 0 && NB % NW == 0, "pieces must divide evenly over the waves");
    static_assert(BK == 64 || BK == 32, "K step");
    constexpr int JA = NA / NW, JB = NB / NW, NJ = JA + JB;    // pieces per wave and step
    constexpr int FM = BM / WM / 32, FN = BN / WN / 32, KB = BK / 16, NR = FM + FN;
    constexpr int IMG_B = (BM + BN) * BK * 2;                  // bytes per LDS image
    constexpr unsigned OOB = 0x80000000u;
    static_assert((ST - 2) * NJ < 64, "vmcnt is 6 bits");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

    const int xcd = bid & 7, loc = bid >> 3, q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- DMA side.  Wave w moves A pieces w, w + NW, ... and B pieces w, w + NW, ...; lane l of a piece: row l / CH,
    // physical chunk l % CH <- logical chunk (l % CH) ^ swz(row).  swz(row) = (row >> 1) & 7 (128-byte rows) or
    // (row >> 2) & 3 (64-byte rows): the 16 rows of every ds_read_b128 lane group then sit on 16 distinct 16-byte slots of the
    // 256-byte bank row.  Pieces start on multiples of RPG rows and NW is even, so the swizzle is the same for all of a lane's pieces.
    const int r_in = lane / CH, pc = lane % CH;
    const int swz_w = BK == 64 ? ((r_in >> 1) + 4 * (wave & 1)) & 7 : (r_in >> 2) & 3;
    const int lc = pc ^ swz_w;
    const int n_first = m0 / (a.Ho * a.Wo);
    const int img = a.img_elems ? a.img_elems : a.H * a.W * a.Cin;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x) + (size_t)n_first * img, 0,
                                                                          0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Wt), 0, 0x7fffffff, 0x00020000);
    int base[JA], hi0[JA], wi0[JA];
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        const int m = m0 + (wave + NW * j) * RPG + r_in;
        base[j] = -1; hi0[j] = 0; wi0[j] = 0;
        if (m < M) {
            const int wo = m % a.Wo;
            const int t = m / a.Wo;
            base[j] = (t / a.Ho - n_first) * img;
            hi0[j] = (t % a.Ho) * a.stride - a.pad;
            wi0[j] = wo * a.stride - a.pad;
        }
    }
    unsigned voff[JA], woff[JB], weff[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int n = n0 + (wave + NW * j) * RPG + r_in;
        woff[j] = n < N ? 2u * (unsigned)(n * K + 8 * lc) : OOB;
    }
    int kh = 0, kw = 0, c0 = 0, k0 = 0;                   // wave-uniform position of the next fetch (K % BK == 0)
    auto refresh = [&]() {                                // offsets of tap (kh, kw); everything out of range past K
        asm volatile("");                                 // keeps the callers' branch
        const bool live = k0 < K;
#pragma unroll
        for (int j = 0; j < JA; ++j) {
            const int hi = hi0[j] + kh, wi = wi0[j] + kw;
            const bool ok = live && base[j] >= 0 && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            voff[j] = ok ? 2u * (unsigned)(base[j] + (hi * a.W + wi) * a.Cin + 8 * lc) : OOB;
        }
#pragma unroll
        for (int j = 0; j < JB; ++j) weff[j] = live ? woff[j] : OOB;
    };
    refresh();
    // piece jj (0 .. NJ-1: A pieces, then B pieces) of the next K step -> image im (compile-time indices)
    auto issue_piece = [&](int jj, int im) {
        if (jj < JA) dma_piece16(rsa, lds_raw + im * IMG_B + (wave + NW * jj) * 1024, voff[jj], 2 * c0);
        else dma_piece16(rsw, lds_raw + im * IMG_B + BM * BK * 2 + (wave + NW * (jj - JA)) * 1024, weff[jj - JA], 2 * k0);
    };
    auto advance_k = [&]() {
        k0 += BK;
        c0 += BK;
        if (c0 >= a.Cin || k0 >= K) {
            if (c0 >= a.Cin) { c0 = 0; if (++kw == a.KW) { kw = 0; ++kh; } }
            refresh();
        }
    };

    // ---- MFMA side: fragment of lane (row = lane & 31, k half = lane >> 5) for the 16-wide k group g = logical chunk 2g + half
    const int lrow = lane & 31, half = lane >> 5;
    const int sw = BK == 64 ? (lrow >> 1) & 7 : (lrow >> 2) & 3;
    unsigned cbyte[KB];                                    // byte offset of (row lrow, chunk of group g) inside a 32-row block
#pragma unroll
    for (int g = 0; g < KB; ++g) cbyte[g] = (unsigned)(lrow * BK * 2 + 16 * ((2 * g + half) ^ sw));
    const int a_blk = wm * FM * 32 * BK * 2, b_blk = (BM + wn * FN * 32) * BK * 2;       // wave-uniform byte offsets
    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // One K step on image im; the NJ pieces of step t + ST - 1 go to image in.  The schedule is pinned (sched_barrier after
    // every MFMA slot, as in the f32 engines): left to itself hipcc keeps ONE fragment set and serialises read - wait - MFMA,
    // which leaves the matrix pipe idle for an LDS round trip per k group (measured r3: 53 % MFMA rate inside a tile whether
    // the fill traffic was 64 KB or 37 KB per step).  Here group g + 1's fragments are read into the other register set one
    // at a time behind group g's MFMAs, and the DMA pieces are spread over the step's MFMA slots.
    bf16x8 fr[2][NR];
    auto read_frag = [&](int set, int g, int rj, int im) {
        const unsigned char* L = lds_raw + im * IMG_B;
        const unsigned char* p = rj < FM ? L + a_blk + rj * 32 * BK * 2 + cbyte[g] : L + b_blk + (rj - FM) * 32 * BK * 2 + cbyte[g];
        fr[set][rj] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p));
    };
    auto compute = [&](int im, int in) {
        constexpr int G = FM * FN, SLOTS = KB * G;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rj = 0; rj < NR; ++rj) read_frag(0, 0, rj, im);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < KB; ++g) {
#pragma unroll
            for (int ms = 0; ms < G; ++ms) {
                const int i = ms / FN, j = ms % FN, slot = g * G + ms;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[g & 1][i], fr[g & 1][FM + j], acc[i][j], 0, 0, 0);
                // side jobs of this MFMA slot: fragment reads of the next group (NR of them over G slots), DMA pieces (NJ over SLOTS)
                if (g + 1 < KB) {
#pragma unroll
                    for (int rk = ms * NR / G; rk < (ms + 1) * NR / G; ++rk)       // B fragments first: the group's first MFMAs need them
                        read_frag((g + 1) & 1, g + 1, rk < FN ? FM + rk : rk - FN, im);
                }
#pragma unroll
                for (int pj = slot * NJ / SLOTS; pj < (slot + 1) * NJ / SLOTS; ++pj) issue_piece(pj, in);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        advance_k();
    };

    const int nk = K / BK;
    // prologue: steps 0 .. ST-2 in flight (steps past the end are all out of range: zero pieces, same instruction count)
#pragma unroll
    for (int s = 0; s < ST - 1; ++s) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) issue_piece(jj, s);
        advance_k();
    }
    // step t: its pieces have landed when at most (ST - 2) * NJ younger loads of this wave are outstanding; the barrier makes
    // that true for every wave's pieces and retires image (t - 1) % ST = (t + ST - 1) % ST, the target of the loads issued next.
    // lgkmcnt(0): this wave's operand reads of step t - 1 must be COMPLETE, not merely issued, before any wave may overwrite
    // that image -- hipcc moves the last MFMAs of a step (register-only instructions, which a "memory" clobber does not pin)
    // and the wait for their operands below the barrier; without the explicit wait a DMA piece of another wave could land
    // under a read still queued in a busy LDS (seen once: the 128 x 64 tile with three workgroups per CU on layer 1, wrong
    // in a few elements out of 8 M)
#define RPG_DMA_STEP(IMX)                                                              \
    do {                                                                               \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((ST - 2) * NJ) : "memory"); \
        __builtin_amdgcn_s_barrier();                                                  \
        asm volatile("" ::: "memory");   /* no LDS read of this step above the barrier */ \
        compute(IMX, (IMX + ST - 1) % ST);                                             \
    } while (0)
    int t = 0;
    for (; t + ST <= nk; t += ST) {
        RPG_DMA_STEP(0);
        RPG_DMA_STEP(1);
        if (ST > 2) RPG_DMA_STEP(2 % ST);
        if (ST > 3) RPG_DMA_STEP(3 % ST);
    }
    if (t < nk) { RPG_DMA_STEP(0); ++t; }
    if (t < nk) { RPG_DMA_STEP(1); ++t; }
    if (ST > 3 && t < nk) { RPG_DMA_STEP(2 % ST); ++t; }
#undef RPG_DMA_STEP
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the zero pieces of the steps past K: LDS becomes the epilogue slabs
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bf16_tile_epilogue_any<FM, FN, ST * IMG_B, NW>(acc, lds_raw, ep, m0, n0, M, N, wm, wn, lane, wave);
}

template <int BM, int BN, int WM, int WN, int BK, int ST>
__global__ __launch_bounds__(64 * WM * WN) void conv_bf16_dma_kernel(ConvArgsB a, const __bf16* __restrict__ Wt, int M, int N, int K,
                                                                       EpiB ep, int tiles_n) {
    conv_bf16_dma_body<BM, BN, WM, WN, BK, ST>(a, Wt, M, N, K, ep, tiles_n, (int)blockIdx.x, (int)gridDim.x);
}

// Round 6: TWO convolutions in one launch -- the 3x3 / stride-2 convolution and the 1x1 / stride-2 shortcut of
// a down-sampling BasicBlock (torchvision BasicBlock.downsample, reached from modules/posenet.py:1037), which read the same input.
// Workgroups [0, t1) are the tiles of the first, the rest those of the second (arguments selected by the block index: scalar
// selects); same tile template, same arithmetic per output as the two launches.  What it buys: one launch gap less per
// down-sampling block and the shortcut's tiles filling the partly empty last round of the 3x3's (392 / 196 tiles on 256 CUs).
struct DmaOne {
    ConvArgsB a;
    const __bf16* Wt;
    int M, N, K, tiles_n;
    EpiB ep;
};
template <int BM, int BN, int WM, int WN, int BK, int ST>
__global__ __launch_bounds__(64 * WM * WN) void conv_bf16_dma_pair_kernel(DmaOne c1, DmaOne c2, int t1) {
    const bool first = (int)blockIdx.x < t1;
    const DmaOne& c = first ? c1 : c2;
    conv_bf16_dma_body<BM, BN, WM, WN, BK, ST>(c.a, c.Wt, c.M, c.N, c.K, c.ep, c.tiles_n, first ? (int)blockIdx.x : (int)blockIdx.x - t1,
                                               first ? t1 : (int)gridDim.x - t1);
}

// ---------------------------------------------------------------------------------------------------------------
// The patch kernel (round 3): 3x3 / stride 1 / pad 1 convolutions with the INPUT PATCH resident in LDS.
//
// Measured on the LDS-DMA kernel above (r3, 512 images): a 256 x 256 tile of layer 4 runs at 53 % of a CU's bf16 matrix rate
// and what bounds it is the operand fill from L2 -- 64 KB per 64-deep K step, of which the A half is the SAME input pixels
// nine times over (im2col: every pixel is fetched once per kernel tap).  Here K is walked channel-chunk major, tap minor:
//   * per 32-channel chunk the input pixels under the tile are fetched ONCE into LDS as a patch [rows][P slots][32 ch]
//     (64-byte slots, chunk-swizzled by slot) with a zero halo, and the nine taps read their A fragments from it at shifted
//     slots: address(lane, tap) = slot(lane) + kh * P + kw -- one v_add per ds_read_b128;
//   * the patch rows are VIRTUAL image rows: image n occupies rows n (H + 2) + 1 .. + H, rows n (H + 2) and n (H + 2) + H + 1
//     are zero (top / bottom padding), slots 0 and W + 1 .. P - 1 of every row are zero; all zeros come from out-of-range
//     DMA lanes, so a tile of BM consecutive output pixels may span rows and images freely;
//   * P is a multiple of 16 slots: the swizzle term ((slot >> 2) & 3) of a lane's address then depends on kw only, and the
//     per-lane address table is 3 (kw) x 2 (k group) x FM registers for the whole kernel;
//   * the weights of one (tap, chunk) -- BN rows of 64 bytes -- stream through three LDS stages, two steps ahead; the patch
//     of the next chunk goes into the second patch buffer one 1-KB piece per wave and step during the current chunk's taps
//     0..7; every step issues exactly 1 + JB loads per wave (dummy pieces land in a scratch KB), which keeps the counted
//     s_waitcnt vmcnt(1 + JB) exact.
// L2 -> LDS traffic per 32-deep step of a 256 x 256 tile: 16 KB of weights + ~3-6 KB of patch, against 32 KB before.
// Requirements: Cin % 64 == 0 (an even number of chunks), patch <= 64 KB per buffer, 32-bit offsets (launcher).
struct PatchArgs {
    const __bf16* x;
    int H, W, Cin, Nimg, M, N, tiles_n, n_tiles;
    int m_base;                            // first output row of tile 0 (a launch may cover the rows [m_base, m_end) only: split launches)
    int P, PR, patch_bytes, n_pieces;      // slots per patch row, patch rows, bytes per patch buffer (multiple of 1 KB), 1-KB pieces
    // floor(2^32 / d) + 1 for d = H + 2, H * W, W, P / 16: the divisions of the tile set-up as one mul_hi each (exact for
    // numerator * d < 2^32; every numerator here is a piece index, a patch row or a pixel offset inside the tile: < 2^15).  Round 4: the
    // set-up was ~700 instructions before the first load of a tile went out, three quarters of them hipcc's expansion of 24
    // integer divisions per lane.
    unsigned mg_hv, mg_hw, mg_w, mg_p16;
#ifdef RPG_PATCH_TRACE
    unsigned long long* trace;             // probe builds (tools/probes/patch_trace.sh): 8 s_memtime stamps per workgroup
#endif
};
#ifdef RPG_PATCH_TRACE
#define RPG_PATCH_STAMP(K) do { if (a.trace && threadIdx.x == 0) { a.trace[(size_t)blockIdx.x * 16 + (K)] = __builtin_amdgcn_s_memtime(); if ((K) == 0 || (K) == 7) a.trace[(size_t)blockIdx.x * 16 + 8 + (K)] = wall_clock64(); } } while (0)
#else
#define RPG_PATCH_STAMP(K) do { } while (0)
#endif
// magic 0 = divisor 1 (2^32 + 1 does not fit): mul_hi gives 0 and the numerator is added back -- a select, not a branch
__device__ __forceinline__ int div_magic(int n, unsigned magic) { return (int)(__umulhi((unsigned)n, magic) + (magic ? 0u : (unsigned)n)); }

__device__ __forceinline__ void dma_piece16_raw(__amdgpu_buffer_rsrc_t rsrc, unsigned lds_byte_off, unsigned char* lds_base,
                                                unsigned voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(lds_base + lds_byte_off), 16, voffset, soffset, 0, 0);
}

// PS = persistent (round 4): one workgroup per CU walks tiles bid, bid + grid, ...; during a tile's LAST chunk the patch slot of the
// steps loads the NEXT tile's first chunk into patch buffer 0 and the weight slot wraps to the next tile's first steps, and the epilogue
// slabs live beside patch buffer 1 (LDS layout: buffer 0 | weight stages | dummy | buffer 1 / slabs) -- so a tile's ~10 k cycles of
// set-up + first-load latency (phase trace: 36 % of a layer-1 tile's life, plus ~2 us of workgroup turnaround) run under the previous
// tile's last chunk and epilogue.  Requires tiles_n == 1 (the weight offsets are those of the previous tile).
template <int BM, int BN, int WM, int WN, int NS, bool PS = false>
__global__ __launch_bounds__(64 * WM * WN) void conv3x3_bf16_patch_kernel(PatchArgs a, const __bf16* __restrict__ Wt, EpiB ep) {
    // NS = weight stages.  3: the weights of step s + 2 are issued at step s and may be read from step s + 2's barrier on.
    // 4: issued THREE steps ahead, waited for one step early -- at step s's barrier the weights of step s + 1 are already
    // visible, so step s + 1's group-0 weight fragments are read behind step s's last MFMAs (like the A fragments, which come
    // from the resident patch): nothing is left exposed behind a barrier but the barrier itself.  (Measured r3 with 3 stages:
    // ~350 of a step's ~1400 cycles were the two weight reads + the wait in front of the first MFMA.)
    static_assert(NS == 3 || NS == 4, "weight stages");
    constexpr int NW = WM * WN;
    static_assert(NW == 8, "8 waves");
    constexpr int FM = BM / WM / 32, FN = BN / WN / 32, NR = FM + FN;
    // weight pieces (16 rows x 64 B) per wave and step; a 64-channel tile has only 4: the other waves issue a dummy piece (every
    // wave must issue the same number of loads per step for the counted waits)
    constexpr int NBP = BN / 16;
    constexpr int JB = NBP >= NW ? NBP / NW : 1;
    static_assert(NBP % NW == 0 || NBP < NW, "weight pieces must divide over the waves");
    constexpr int BSTAGE = BN * 64;                           // bytes per weight stage
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NL = 1 + JB;                                // loads per wave and step
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3, q8 = nwg >> 3, r8 = nwg & 7;
    int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    int m0 = a.m_base + (PS ? tile * BM : (tile / a.tiles_n) * BM);
    const int n0 = PS ? 0 : (tile % a.tiles_n) * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int H = a.H, W = a.W, HW = a.H * a.W, P = a.P, HV = a.H + 2;
    const int NC = a.Cin / 32;
    RPG_PATCH_STAMP(0);                                   // workgroup entry

    // ---- LDS layout (byte offsets): patch buffer 0 at 0; one-shot form: buffer 1 | weight stages | dummy KB; persistent form: weight
    // stages | dummy KB | buffer 1, which doubles as the epilogue slabs while buffer 0 and the stages take the next tile's first loads
    const unsigned off_b1 = PS ? (unsigned)a.patch_bytes + (unsigned)NS * BSTAGE + 1024u : (unsigned)a.patch_bytes;
    const unsigned off_st = PS ? (unsigned)a.patch_bytes : 2u * (unsigned)a.patch_bytes;
    const unsigned dummy_off = off_st + (unsigned)NS * BSTAGE;       // 1 KB nobody reads
    const int img = HW * a.Cin;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Wt), 0, 0x7fffffff, 0x00020000);

    // ---- geometry of a tile: first image, first virtual row of the patch (wave-uniform: one real division), and the patch pieces of
    // this wave: piece q = wave + 8 t (t = 0 .. 7) covers 16-byte chunks 64 q .. 64 q + 63 of the buffer;
    // chunk ci = (slot ci >> 2, physical chunk ci & 3) <- logical chunk (ci & 3) ^ ((slot >> 2) & 3) of that pixel.
    // A piece is 16 consecutive slots and P % 16 == 0, so a piece lies inside ONE patch row: its row, image and validity are
    // wave-uniform (scalar unit: two s_mul_hi per piece), and a lane adds only its column -- five vector instructions per piece.
    // (Phase trace of round 4: the per-lane form cost ~6 k cycles of a 27-k-cycle layer-1 tile before the first load went out:
    // 220 vector instructions, 48 of them quarter-rate integer multiplies, 16 exec-mask branches.)
    int n_first, rem_first, v0;
    unsigned pvoff[8];
    auto geometry = [&](int m0_, bool valid, int& nf, int& rf, int& v0_) {
        nf = m0_ / HW;
        rf = m0_ - nf * HW;                                          // pixel of m0 inside its image
        v0_ = nf * HV + div_magic(rf, a.mg_w);                       // = v(m0) - 1: the row above the tile's first pixel
        const int s_l = lane >> 2;                                   // the lane's slot inside its piece
        const int lc0 = (lane & 3) ^ ((s_l >> 2) & 3);               // (slot >> 2) & 3 == (s_l >> 2) & 3: pieces start on multiples of 16 slots
        const int p16 = P >> 4;                                      // pieces per patch row
        const int vbase = v0_ - nf * HV;                             // patch row 0 relative to the first image's virtual rows
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int q = wave + NW * t;                             // uniform
            const int prow = div_magic(q, a.mg_p16);
            const int col0 = (q - prow * p16) * 16;
            // Row-parity term of the swizzle (round 5): physical chunk = logical ^ ((slot >> 2) & 3) ^ 2 (patch row & 1).  With P a
            // multiple of 16 the same column of two ADJACENT patch rows lies a multiple of 16 slots apart -- the same banks under the
            // slot term alone -- and on 14-wide maps (P = 16) a fragment's 32 consecutive pixels cover 2.3 rows: lanes l and l + 14
            // of one ds_read_b128 lane group hit the same 16 bytes' banks (r4 PMC: SQ_LDS_BANK_CONFLICT = 45 % of the LDS cycles of the
            // 256 x 256 kernel).  Parity is XOR-decomposable over the kernel row: tap kh of a pixel reads row + kh, i.e. the table
            // entry of the OTHER k group when kh is odd (read_a) -- no extra registers, no extra instructions.
            const int lc8 = 8 * (lc0 ^ (2 * (prow & 1)));
            const int vrel = vbase + prow;                           // < PR + HV
            const int nrel = div_magic(vrel, a.mg_hv);
            const int rr = vrel - nrel * HV - 1;
            const bool rowok = valid && q < a.n_pieces && prow < a.PR && (unsigned)rr < (unsigned)H && nf + nrel < a.Nimg;
            const int rowbase = ((nrel * H + rr) * W - 1) * a.Cin;   // element offset of column "pcol = 0" (the left halo slot) of that image row
            const int pcol = col0 + s_l;
            const unsigned off = 2u * (unsigned)(rowbase + __mul24(pcol, a.Cin) + lc8);
            pvoff[t] = (rowok && (unsigned)(pcol - 1) < (unsigned)W) ? off : OOB;
        }
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x) + (size_t)nf * img, 0, 0x7fffffff, 0x00020000);
    };
    __amdgpu_buffer_rsrc_t rs_p = geometry(m0, true, n_first, rem_first, v0);       // the patch loads' resource: based at the tile's first image
    // persistent form, while a tile's last chunk runs: the patch slot loads the NEXT tile's chunk 0 (pvoff / rs_p already the next
    // tile's; c_off cancels the chunk index in the scalar offset) and the weight slot wraps to its first steps
    int c_off = 0;
    bool wrap_live = false;
    // weight pieces: piece j covers rows 16 (wave + 8 j) .. + 15 of the stage; lane: row l >> 2, physical chunk l & 3
    unsigned woff[JB];
    auto set_woff = [&](int l) {
        const int r_in = l >> 2;
        const int lc = (l & 3) ^ ((r_in >> 2) & 3);
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const int n = n0 + (wave + NW * j) * 16 + r_in;
            woff[j] = n < a.N ? 2u * (unsigned)(n * 9 * a.Cin + 8 * lc) : OOB;
        }
    };
    set_woff(lane);
    // patch piece t of chunk `chunk` -> patch buffer `buf` (dummy target when this wave has no t-th piece)
    auto issue_patch = [&](int t, int buf, int chunk) {
        const bool real = (wave + NW * t) < a.n_pieces;
        const unsigned dst = real ? (buf ? off_b1 : 0u) + (unsigned)((wave + NW * t) * 1024) : dummy_off;
        dma_piece16_raw(rs_p, dst, lds_raw, (chunk < NC || (PS && wrap_live)) ? pvoff[t] : OOB, chunk * 64 + (PS ? c_off : 0));
    };
    auto issue_dummy = [&]() { dma_piece16_raw(rsw, dummy_off, lds_raw, OOB, 0); };
    // the weights of (chunk, tap) -> stage `stage`
    auto issue_w = [&](int stage, int chunk, int tap) {
        const bool live = chunk < NC || (PS && wrap_live);
        const int cw = (PS && chunk >= NC) ? 0 : chunk;                // past the tile's end: the next tile's first chunk
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            if (NBP < NW && wave >= NBP) issue_dummy();
            else dma_piece16_raw(rsw, off_st + (unsigned)(stage * BSTAGE + (wave + NW * j) * 1024), lds_raw,
                                 live ? woff[j] : OOB, (tap * a.Cin + cw * 32) * 2);
        }
    };

    unsigned a3[FM][3][2];
    unsigned cb[2];
    f32x16 acc[FM][FN];

    // One step = one tap of one chunk on patch buffer `buf` and weight stage tap % 3: two k groups of FM * FN MFMAs.  Pinned
    // schedule (see the LDS-DMA kernel): group 1's fragments are read behind group 0's MFMAs; and because the patch of a chunk
    // is complete from its first tap on, the NEXT tap's group-0 A fragments are read behind group 1's MFMAs, before the
    // barrier -- after it only the two weight fragments of the new stage (and at a chunk's first tap the A fragments) are
    // exposed.  Loads of the step: one patch piece of the next chunk (taps 0..7; tap 8: a dummy) in the first MFMA slot, the
    // weights of step s + 2 spread over the following slots.
    bf16x8 fr[2][NR];
    // a3 is CARRIED through the chunk loop (it moves to the other patch buffer at every chunk end, see RPG_PATCH_CHUNK): written as
    // a3 + buf * patch_bytes + kh * P * 64 with a loop-invariant a3, hipcc hoisted all 2 x 9 x FM x 2 sums out of the unrolled step
    // loop and spilled them (100 VGPRs in the four-stage form)
    auto read_a = [&](int set, int g, int i, int tap) {
        const unsigned aoff = (unsigned)((tap / 3) * P * 64);                             // wave-uniform
        fr[set][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds_raw + (a3[i][tap % 3][g ^ ((tap / 3) & 1)] + aoff)));
    };
    auto flip_patch = [&](int buf) {                           // after a chunk on buffer `buf`: the next chunk reads the other one
        const unsigned d = buf == 0 ? off_b1 : 0u - off_b1;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) { a3[i][kw][0] += d; a3[i][kw][1] += d; }
    };
    auto read_b = [&](int set, int g, int j, int stage) {
        fr[set][FM + j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds_raw + cb[g] + stage * BSTAGE + j * 32 * 64));
    };
    // `sidx` = the step's index modulo NS (compile time: a chunk has 9 steps; with 3 stages 9 % 3 == 0 -> tap % 3, with 4 stages
    // the caller unrolls 4 chunks = 36 steps = 0 mod 4)
    auto step = [&](int buf, int tap, int chunk, int sbase, bool first, bool last) {
        // stage of this step / of the steps it loads or prefetches for: compile-time with 3 stages (9 % 3 == 0), a wave-uniform
        // runtime value with 4 (sbase = 9 * chunk mod 4; unrolling 4 chunks to make it a constant spilled 131 VGPRs)
        auto stage_of = [&](int d) { return NS == 3 ? (tap + d) % 3 : (sbase + tap + d) & 3; };
        const int sidx = 0;
        (void)sidx;
        constexpr int G = FM * FN;
        constexpr int AHEAD = NS - 1;                          // weights are issued this many steps ahead
        __builtin_amdgcn_sched_barrier(0);
#ifdef RPG_PATCH_TRACE
        if (first && tap == 0) RPG_PATCH_STAMP(3);        // the first step's barrier has passed: patch chunk 0 + first weights landed
        if (chunk == 1 && tap == 0) RPG_PATCH_STAMP(4);   // chunk 0 done (nine steps)
#endif
        if (tap == 0) {
#pragma unroll
            for (int i = 0; i < FM; ++i) read_a(0, 0, i, tap);
        }
        if (NS == 3 || first) {
#pragma unroll
            for (int j = 0; j < FN; ++j) read_b(0, 0, j, stage_of(0));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ms = 0; ms < G; ++ms) {                      // k group 0
            const int i = ms / FN, j = ms % FN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[0][i], fr[0][FM + j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int rk = ms * NR / G; rk < (ms + 1) * NR / G; ++rk) {
                if (rk < FN) read_b(1, 1, rk, stage_of(0));
                else read_a(1, 1, rk - FN, tap);
            }
            if (ms == 0) {
                if (tap < 8) issue_patch(tap, buf ^ 1, chunk + 1);
                else issue_dummy();
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ms = 0; ms < G; ++ms) {                      // k group 1
            const int i = ms / FN, j = ms % FN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[1][i], fr[1][FM + j], acc[i][j], 0, 0, 0);
            if (ms == 0) {
                if (tap + AHEAD < 9) issue_w(stage_of(AHEAD), chunk, tap + AHEAD);
                else issue_w(stage_of(AHEAD), chunk + 1, tap + AHEAD - 9);
            }
            // next tap's group-0 A fragments (set 0 is free once group 0 has issued), early in the group: the wait in front of
            // the barrier must not find them still in flight
            if (tap < 8 && ms < FM) read_a(0, 0, ms, tap + 1);
            // four stages: the next step's group-0 weight fragments too (its weights were waited for at THIS step's barrier)
            if (NS == 4 && !last && ms >= FM && ms < FM + FN) read_b(0, 0, ms - FM, stage_of(1));
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue: the first chunk's patch, then the weights of the first NS - 1 steps; a dummy piece in front of each of the
    // last NS - 2 weight sets makes the loads that a step's wait leaves outstanding exactly whole steps' worth (NL each)
#pragma unroll
    for (int t = 0; t < 8; ++t) issue_patch(t, 0, 0);
    issue_w(0, 0, 0);
    issue_dummy();
    issue_w(1, 0, 1);
    if (NS == 4) {
        issue_dummy();
        issue_w(2, 0, 2);
    }
    RPG_PATCH_STAMP(1);                                   // set-up of the load addresses done, prologue loads issued

    // step s: everything but the loads of the last NS - 3 + 1 steps has landed: with 3 stages the weights of step s (issued at
    // s - 2; step s - 1's loads may be in flight), with 4 stages the weights of step s + 1 (issued at s - 2) as well
#ifndef RPG_PATCH_ABL
#define RPG_PATCH_ABL 0                    // diagnostic builds (tools/probes/patch_ablate.sh): 1 no epilogue | 2 no per-step wait + barrier
#endif                                     // (wrong results, timing only)
#define RPG_PATCH_STEP(BUF, TAP, CHUNK, SIDX, FIRST, LAST)                                \
    do {                                                                                 \
        if (!(RPG_PATCH_ABL & 2)) {                                                      \
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NL) : "memory");          \
            __builtin_amdgcn_s_barrier();                                                \
        }                                                                                \
        asm volatile("" ::: "memory");                                                   \
        step(BUF, TAP, CHUNK, SIDX, FIRST, LAST);                                        \
    } while (0)
#define RPG_PATCH_CHUNK(BUF, CHUNK, S0)                                                                                  \
    do {                                                                                                                 \
        const bool f0 = (CHUNK) == 0, l8 = (CHUNK) + 1 == NC;                                                            \
        const int sb = (S0);                                                                                             \
        RPG_PATCH_STEP(BUF, 0, CHUNK, sb, f0, false); RPG_PATCH_STEP(BUF, 1, CHUNK, sb, false, false);                   \
        RPG_PATCH_STEP(BUF, 2, CHUNK, sb, false, false); RPG_PATCH_STEP(BUF, 3, CHUNK, sb, false, false);                \
        RPG_PATCH_STEP(BUF, 4, CHUNK, sb, false, false); RPG_PATCH_STEP(BUF, 5, CHUNK, sb, false, false);                \
        RPG_PATCH_STEP(BUF, 6, CHUNK, sb, false, false); RPG_PATCH_STEP(BUF, 7, CHUNK, sb, false, false);                \
        RPG_PATCH_STEP(BUF, 8, CHUNK, sb, false, l8);                                                                    \
        flip_patch(BUF);                                                                                                 \
    } while (0)

    int sb0 = 0;                                          // steps done before this tile, modulo 4 (stage rotation of the four-stage form)
    bool first_tile = true;
    for (;;) {
        // persistent form: every lane-derived address is rebuilt per tile from an opaque copy of the lane index -- kept live across the
        // epilogue (which needs the whole register file beside the accumulators) they were spilled to scratch (130 VGPRs on the 512 x 128 tile)
        int lt = lane;
        if (PS) {
            asm volatile("" : "+v"(lt));
            if (!first_tile) set_woff(lt);
        }
        const int half = lt >> 5;
        // B fragment addresses (inside stage 0): row lrow of the wave's j-th 32-channel block, chunk 2 g + half
        {
            const int lrow = lt & 31, sw = (lrow >> 2) & 3;
#pragma unroll
            for (int g = 0; g < 2; ++g)
                cb[g] = off_st + (unsigned)(wn * FN * 32 * 64 + lrow * 64 + 16 * ((2 * g + half) ^ sw));
        }
        // ---- A fragment addresses: pixel m of lane (i, lane & 31) -> slot s0 = (v(m) - 1 - v0) * P + c; tap (kh, kw) reads slot
        // s0 + kh * P + kw; byte address = slot * 64 + 16 * (chunk ^ ((slot >> 2) & 3)), chunk = 2 g + half.  (Set up AFTER the tile's
        // first loads have been issued: these instructions run under the first patch's round trip instead of in front of it.)
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            int m = m0 + (wm * FM + i) * 32 + (lt & 31);
            m = m < a.M ? m : a.M - 1;
            const int rel = rem_first + (m - m0);                    // pixel offset from the first image's origin: < HW + BM
            const int nr = div_magic(rel, a.mg_hw), rem = rel - nr * HW, r = div_magic(rem, a.mg_w), c = rem - r * W;
            const int prow0 = (n_first + nr) * HV + r - v0;          // patch row of the pixel's kh = 0 taps
            const int s0 = prow0 * P + c;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int t = s0 + kw;
                const int e = half ^ ((t >> 2) & 3) ^ (2 * (prow0 & 1));
                a3[i][kw][0] = (unsigned)(t * 64 + 16 * e);
                a3[i][kw][1] = (unsigned)(t * 64 + 16 * (e ^ 2));
            }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        RPG_PATCH_STAMP(2);                               // fragment addresses / accumulators set up: the K loop starts waiting

        // persistent form: the tile after this one (its geometry is computed in front of the last chunk, below)
        const int tile_n = tile + nwg, m0_n = a.m_base + tile_n * BM;
        const bool has_next = PS && tile_n < a.n_tiles;
        int nf_n = 0, rf_n = 0, v0_n = 0;
        if (PS && !first_tile) {
            // the previous tile's stores are acknowledged and this tile's first chunk + first weights (issued during its last chunk)
            // have landed: the counted waits of the steps start from zero again
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        for (int cc = 0; cc < NC; cc += 2) {
            RPG_PATCH_CHUNK(0, cc, (sb0 + 9 * cc) & 3);
            if (PS && cc + 2 >= NC) {
                // in front of the tile's LAST chunk (on buffer 1; NC is even): from here on the loads are the next tile's.  Without a
                // next tile every offset is out of range (zero pieces, same instruction count).
                rs_p = geometry(m0_n, has_next, nf_n, rf_n, v0_n);
                c_off = -NC * 64;
                wrap_live = has_next;
            }
            RPG_PATCH_CHUNK(1, cc + 1, (sb0 + 9 * cc + 9) & 3);
        }
        if (PS) {
            // every wave has finished reading buffer 1 and the stages of this tile; buffer 0 / the stages are being filled for the next
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        RPG_PATCH_STAMP(5);                               // K loop done
        if (RPG_PATCH_ABL & 1) {           // keep the accumulators alive, store (almost) nothing
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) sacc += acc[i][j][0] + acc[i][j][9];
            if (sacc == 123.456f) reinterpret_cast<__bf16*>(ep.out)[tid] = (__bf16)sacc;
        } else {
            if constexpr (PS) {       // plain convolutions only (launcher): the lean epilogue, slabs beside patch buffer 1
                if (ep.residual) bf16_tile_epilogue_lean_body<FM, FN, NW, true>(acc, lds_raw + off_b1, ep, m0, n0, a.M, a.N, wm, wn, lane, wave);
                else bf16_tile_epilogue_lean_body<FM, FN, NW, false>(acc, lds_raw + off_b1, ep, m0, n0, a.M, a.N, wm, wn, lane, wave);
            } else {
                bf16_tile_epilogue_any<FM, FN, 160 * 1024, NW>(acc, lds_raw, ep, m0, n0, a.M, a.N, wm, wn, lane, wave);
            }
        }
        if (!PS || !has_next) break;
        tile = tile_n; m0 = m0_n; n_first = nf_n; rem_first = rf_n; v0 = v0_n;
        c_off = 0;
        wrap_live = false;
        sb0 = (sb0 + 9 * NC) & 3;
        first_tile = false;
    }
#undef RPG_PATCH_CHUNK
#undef RPG_PATCH_STEP
#ifdef RPG_PATCH_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    RPG_PATCH_STAMP(6);                                   // epilogue instructions issued (the stores may still be in flight)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RPG_PATCH_STAMP(7);                                   // ... and acknowledged
#endif
}

int g_bf16_lean_epi = 1;     // RPG_TUNE_BF16_LEAN_EPI: the branch-free epilogue for plain convolutions (0: the general one, rounds 1-3)
int g_bf16_fused_stem = 1;   // RPG_TUNE_FUSED_STEM also selects the bf16 encoder's fused stem (stem_bf16.hip)
int g_bf16_chunk = 0;        // RPG_TUNE_BF16_CHUNK: images per depth-first group of the bf16 encoder's identity-block runs (0 = off)
int g_bf16_chunk_mb = 64;    // ... for activation tensors of at least this many MB
int g_bf16_linear_dma = 0;   // RPG_TUNE_BF16_LINEAR_DMA: 0 = the interleaved buffer-load kernel | 10 + i: LDS-DMA configuration i
int g_bf16_dma = 1;      // RPG_TUNE_BF16_DMA: 0 off | 1 by shape | 10 + i: configuration i of launch_dma_config wherever it is eligible
int g_bf16_tile = -1;    // RPG_TUNE_BF16_TILE: -1 auto | 0: 64x64 | 1: 128x128 | 2: 256x64 | 3: 128x64 (interleaved kernel only)
int g_bf16_fast = 1;     // RPG_TUNE_BF16_FAST: the interleaved buffer-load kernel where eligible
int g_bf16_bk = 32;      // measured on MI355X at 64 graphs: K step 32 -> 9.97 ms/step, 64 -> 12.5 (the 72-KB LDS image halves occupancy)

template <int BM, int BN, int WM, int WN, int BK>
void launch_tile(const ConvArgsB& a, const __bf16* w, int M, int N, int K, const EpiB& ep, hipStream_t s) {
    constexpr int lds = 2 * (BM + BN) * (BK + 8) * 2;
    auto kern = conv_bf16_kernel<BM, BN, WM, WN, BK>;
    if (lds > 64 * 1024) {
        static bool once = false;
        if (!once) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            once = true;
        }
    }
    const int tn = (N + BN - 1) / BN, tm = (M + BM - 1) / BM;
    hipLaunchKernelGGL(kern, dim3(tm * tn), dim3(NT), lds, s, a, w, M, N, K, ep, tn);
}

template <int BM, int BN, int WM, int WN>
void launch_fast(const ConvArgsB& a, const __bf16* w, int M, int N, int K, const EpiB& ep, hipStream_t s) {
    constexpr int lds = 2 * (BM + BN) * 64 * 2;
    auto kern = conv_bf16_fast_kernel<BM, BN, WM, WN>;
    if (lds > 64 * 1024) {
        static bool once = false;
        if (!once) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            once = true;
        }
    }
    const int tn = (N + BN - 1) / BN, tm = (M + BM - 1) / BM;
    hipLaunchKernelGGL(kern, dim3(tm * tn), dim3(NT), lds, s, a, w, M, N, K, ep, tn);
}

template <int BM, int BN, int WM, int WN, int BK, int ST>
void launch_dma(const ConvArgsB& a, const __bf16* w, int M, int N, int K, const EpiB& ep, hipStream_t s) {
    constexpr int lds = ST * (BM + BN) * BK * 2;              // (256 x 64 / K 64 / 2 images = 80 KB exactly: two workgroups per CU)
    static_assert(lds <= 160 * 1024, "LDS images do not fit a CU");
    auto kern = conv_bf16_dma_kernel<BM, BN, WM, WN, BK, ST>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    static bool once[64] = {};
    if (!once[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        once[dev] = true;
    }
    const int tn = (N + BN - 1) / BN, tm = (M + BM - 1) / BM;
    hipLaunchKernelGGL(kern, dim3(tm * tn), dim3(64 * WM * WN), lds, s, a, w, M, N, K, ep, tn);
}

template <int BM, int BN, int WM, int WN, int BK, int ST>
void launch_dma_pair(const DmaOne& c1in, const DmaOne& c2in, hipStream_t s) {
    constexpr int lds = ST * (BM + BN) * BK * 2;
    auto kern = conv_bf16_dma_pair_kernel<BM, BN, WM, WN, BK, ST>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    static bool once[64] = {};
    if (!once[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        once[dev] = true;
    }
    DmaOne c1 = c1in, c2 = c2in;
    c1.tiles_n = (c1.N + BN - 1) / BN; c2.tiles_n = (c2.N + BN - 1) / BN;
    const int t1 = ((c1.M + BM - 1) / BM) * c1.tiles_n, t2 = ((c2.M + BM - 1) / BM) * c2.tiles_n;
    hipLaunchKernelGGL(kern, dim3(t1 + t2), dim3(64 * WM * WN), lds, s, c1, c2, t1);
}
int g_bf16_pair = 1;     // RPG_TUNE_BF16_PAIR: the strided 3x3 + 1x1 shortcut pair of a down-sampling block in one launch

int g_bf16_stages = 4;   // weight stages of the patch kernel (experiments: 3 = never the prefetching form); RPG_TUNE_BF16_PATCH + 10
// RPG_TUNE_BF16_PERSIST: the persistent form of the patch kernel (cross-tile prefetch) on the 64-channel layer.  OFF by default: the kernel
// itself is 10 % (in the model) to 15 % (stand-alone) faster, but 256 resident workgroups hold every CU for the whole launch, and the
// default two-stream schedule lives on the other stream's workgroups slipping into this launch's gaps: configs[2] 13.53-13.60 k graphs/s
// without it, 13.26-13.39 k with it (same box, alternating runs; one-stream kernel sums equal).
int g_bf16_persist = 0;
int g_bf16_tail = 1;     // RPG_TUNE_BF16_TAIL: bit 0 (default): 512 x 128 launches hand the rows beyond the last full round to 256 x 128 tiles (layer 2 at 512
                         // images: -4..5 %); bit 1: 256 x 256 launches likewise to 160 x 256 tiles (layer 3: measured no gain -- the partly empty
                         // second round of the single launch already runs 1.46x faster per tile: the kernels are bound chip-wide, not per CU)
int g_bf16_patch = 1;    // RPG_TUNE_BF16_PATCH: the patch kernel for 3x3 / stride-1 convolutions: 0 off | 1 by shape | 2 wherever eligible

// The patch kernel, if the shape is eligible (3x3, stride 1, pad 1, Cin % 64 == 0, patch <= 64 pieces, 32-bit offsets, LDS fits)
// m_begin / m_end: the launch covers the output rows [m_begin, m_end) only (m_end = 0: to M); the tiles start at m_begin
template <int BM, int BN, int WM, int WN, int NS, bool PS = false>
bool launch_patch(const ConvArgsB& c, const __bf16* w, int nimg, int M, int N, const EpiB& ep, hipStream_t s, int m_begin = 0, int m_end = 0) {
    if (c.KH != 3 || c.KW != 3 || c.stride != 1 || c.pad != 1 || (c.Cin & 63) || c.img_elems) return false;
    const int H = c.H, W = c.W, HW = H * W;
    PatchArgs a{};
    if (m_end <= 0 || m_end > M) m_end = M;
    if (m_begin < 0 || m_begin >= m_end) return false;
    // a.M bounds the rows a tile may write / read as pixels: the end of this launch's range
    a.x = c.x; a.H = H; a.W = W; a.Cin = c.Cin; a.Nimg = nimg; a.M = m_end; a.N = N; a.m_base = m_begin;
    a.tiles_n = (N + BN - 1) / BN;
    a.P = (W + 2 + 15) / 16 * 16;
    const int rows = (BM - 1) / W + 2, imgs = (BM - 1) / HW + 2;
    a.PR = rows + 2 * (imgs - 1) + 2;
    a.patch_bytes = (a.PR * a.P * 64 + 1023) / 1024 * 1024;
    a.n_pieces = a.patch_bytes / 1024;
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((1ULL << 32) / (unsigned)d) + 1u; };      // 0: divisor 1 (2^32 + 1 does not fit)
    a.mg_hv = magic(H + 2); a.mg_hw = magic(HW); a.mg_w = magic(W); a.mg_p16 = magic(a.P / 16);
    // exactness of v_mul_hi(n, magic): n * d < 2^32 with n < patch slots (4096), patch rows + H + 2, H W + BM, H W respectively
    if ((long)(HW + BM) * HW >= (1L << 32) || (long)(a.PR + H + 2 + 2) * (H + 2) >= (1L << 32)) return false;
    constexpr int slab = 8 * 32 * ((BN / WN / 32) * 32 + 4) * 4;
    const int tm = (m_end - m_begin + BM - 1) / BM;
    a.n_tiles = tm * a.tiles_n;
    int lds = 2 * a.patch_bytes + NS * BN * 64 + 1024;
    int grid = a.n_tiles;
    if (PS) {
        // persistent form: one workgroup per CU, more than one round of tiles, one channel tile (the weight offsets carry over), the
        // epilogue slabs beside patch buffer 1 (kernel comment)
        grid = rpg::num_cus();
        if (!ep.lean || a.tiles_n != 1 || a.n_tiles <= grid || (long)m_begin + ((long)a.n_tiles + grid) * BM >= (1L << 31)) return false;      // (the tile after the last is still an int)
        lds = a.patch_bytes + NS * BN * 64 + 1024 + (a.patch_bytes > slab ? a.patch_bytes : slab);
    }
    if (a.n_pieces > 64 || lds > 160 * 1024 || lds < slab) return false;
    if ((long)(imgs + 1) * HW * c.Cin * 2 >= (1L << 31) || (long)N * 9 * c.Cin * 2 >= (1L << 31)) return false;
    auto kern = conv3x3_bf16_patch_kernel<BM, BN, WM, WN, NS, PS>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    static bool once[64] = {};
    if (!once[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once[dev] = true;
    }
#ifdef RPG_PATCH_TRACE
    // probe build: with RPG_PATCH_TRACE=1 in the environment every launch is synchronous and prints the median phase lengths of its
    // workgroups (s_memtime ticks) to stderr
    static const bool tracing = getenv("RPG_PATCH_TRACE") != nullptr;
    const int nwg_trace = grid;
    static unsigned long long* tbuf = nullptr;
    static int tcap = 0;
    if (tracing) {
        if (nwg_trace > tcap) {
            if (tbuf) (void)hipFree(tbuf);
            (void)hipMalloc(reinterpret_cast<void**>(&tbuf), (size_t)nwg_trace * 128);
            tcap = nwg_trace;
        }
        (void)hipMemsetAsync(tbuf, 0, (size_t)nwg_trace * 128, s);
        a.trace = tbuf;
    }
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a, w, ep);
#ifdef RPG_PATCH_TRACE
    if (tracing) {
        std::vector<unsigned long long> h((size_t)nwg_trace * 16);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), tbuf, h.size() * 8, hipMemcpyDeviceToHost);
        auto median = [&](int k0, int k1) {
            std::vector<long long> d;
            for (int g = 0; g < nwg_trace; ++g) d.push_back((long long)(h[(size_t)g * 16 + k1] - h[(size_t)g * 16 + k0]));
            std::sort(d.begin(), d.end());
            return d[d.size() / 2];
        };
        // wall_clock64 (100 MHz) at entry / exit: the launch's span, the sum of the workgroup lives, and the s_memtime rate
        unsigned long long lo = ~0ULL, hi = 0;
        double life_wall = 0, life_mem = 0;
        for (int g = 0; g < nwg_trace; ++g) {
            lo = std::min(lo, h[(size_t)g * 16 + 8]); hi = std::max(hi, h[(size_t)g * 16 + 15]);
            life_wall += (double)(h[(size_t)g * 16 + 15] - h[(size_t)g * 16 + 8]);
            life_mem += (double)(h[(size_t)g * 16 + 7] - h[(size_t)g * 16]);
        }
        fprintf(stderr, "patch_trace <%d,%d,NS%d%s> wgs %d M %d N %d Cin %d | set-up+issue %lld  frag set-up %lld  wait first data %lld  chunk0 %lld  "
                "rest of K loop %lld  epilogue issue %lld  store ack %lld | life %lld ticks = %.2f us (s_memtime %.0f MHz) | launch span %.1f us, "
                "sum of lives / 256 CUs %.1f us\n", BM, BN, NS, PS ? ",persistent" : "", nwg_trace, M, N, c.Cin,
                median(0, 1), median(1, 2), median(2, 3), median(3, 4), median(4, 5), median(5, 6), median(6, 7), median(0, 7),
                median(8, 15) * 0.01, life_mem / life_wall * 100.0, (double)(hi - lo) * 0.01, life_wall * 0.01 / 256.0);
    }
#endif
    return true;
}

#include "block_bf16.inc"

#ifdef RPG_PROBE_WS64
// the weights-stationary layer-1 experiment of round 3 (correct, not faster): lives in tools/probes/, compiled in on request only
#include "../../tools/probes/conv3x3_bf16_ws64.inc"
#endif

// configuration table of the LDS-DMA kernel (index = RPG_TUNE_BF16_DMA - 10); returns false if the index is unknown or the
// shape is not eligible for it (Cin % BK)
bool launch_dma_config(int cfg, const ConvArgsB& a, const __bf16* w, int M, int N, int K, const EpiB& ep, hipStream_t s) {
    const bool k64 = a.Cin % 64 == 0, k32 = a.Cin % 32 == 0;
    switch (cfg) {
        case 0: if (!k64) return false; launch_dma<256, 256, 2, 4, 64, 2>(a, w, M, N, K, ep, s); return true;   // 128 KB, wave 128 x 64
        case 1: if (!k64) return false; launch_dma<256, 128, 4, 2, 64, 3>(a, w, M, N, K, ep, s); return true;   // 144 KB, wave 64 x 64
        case 2: if (!k64) return false; launch_dma<256, 128, 2, 2, 64, 3>(a, w, M, N, K, ep, s); return true;   // 4 waves, wave 128 x 64
        case 3: if (!k64) return false; launch_dma<128, 128, 2, 2, 64, 2>(a, w, M, N, K, ep, s); return true;   // 64 KB: 2 workgroups / CU
        case 4: if (!k64) return false; launch_dma<256, 64, 4, 2, 64, 3>(a, w, M, N, K, ep, s); return true;    // 120 KB, wave 64 x 32
        case 5: if (!k64) return false; launch_dma<256, 64, 4, 2, 64, 2>(a, w, M, N, K, ep, s); return true;    // 80 KB: 2 workgroups / CU
        case 6: if (!k32) return false; launch_dma<256, 256, 2, 4, 32, 4>(a, w, M, N, K, ep, s); return true;   // 128 KB, 3 steps ahead
        case 7: if (!k32) return false; launch_dma<256, 128, 4, 2, 32, 3>(a, w, M, N, K, ep, s); return true;   // 72 KB: 2 workgroups / CU
        case 8: if (!k64) return false; launch_dma<128, 128, 2, 2, 64, 3>(a, w, M, N, K, ep, s); return true;   // 96 KB
        case 9: if (!k32) return false; launch_dma<128, 64, 2, 2, 32, 4>(a, w, M, N, K, ep, s); return true;    // 48 KB: 3 workgroups / CU
        // (measured and removed, r3: 512 x 128 / 1024 x 64 / 512 x 64 tiles with 128 x 64 wave tiles for the narrow layers 1-2 --
        // 280-290 us on layer 1 against 205 us for configuration 5, 190 us on layer 2 against 160 us for configuration 7: fewer,
        // fatter workgroups lose more to the one-workgroup-per-CU prologue / epilogue than the wave tile wins in LDS reads)
        default: return false;
    }
}

// ---- streaming kernels on bf16 NHWC tensors (8 elements = 16 bytes per lane) ----
inline int capped_grid(long items) {
    long g = (items + NT - 1) / NT;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// fp32 (or already-bf16) [n][3][h][w] -> bf16 [n][h][w][8] (channels 3..7 zero)
template <typename TIn>
__global__ __launch_bounds__(NT) void nchw3_to_nhwc8_bf16_kernel(const TIn* __restrict__ x, uint4* __restrict__ y,
                                                                 long npix, int hw) {
    for (long p = (long)blockIdx.x * NT + threadIdx.x; p < npix; p += (long)gridDim.x * NT) {
        const long n = p / hw;
        const int qd = (int)(p - n * hw);
        const TIn* b = x + n * 3 * (long)hw + qd;
        const bf16x8 v = {(__bf16)b[0], (__bf16)b[hw], (__bf16)b[2 * (long)hw], (__bf16)0.f,
                          (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        y[p] = __builtin_bit_cast(uint4, v);
    }
}

__global__ __launch_bounds__(NT) void maxpool3x3s2_bf16_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int h,
                                                               int w, int c8, int ho, int wo, long total) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i % c8);
        long t = i / c8;
        const int ox = (int)(t % wo);
        t /= wo;
        const int oy = (int)(t % ho);
        const long n = t / ho;
        const uint4* img = x + n * (long)h * w * c8;
        float m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)w) continue;
                const bf16x8 v = __builtin_bit_cast(bf16x8, img[((long)iy * w + ix) * c8 + c]);
#pragma unroll
                for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], (float)v[k]);
            }
        }
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (__bf16)m[k];
        y[i] = __builtin_bit_cast(uint4, o);
    }
}

// [n][hw][c] bf16 -> [n][c] bf16 (fp32 sum, one rounding at the end)
__global__ __launch_bounds__(NT) void global_avgpool_bf16_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int hw,
                                                                 int c8, long total) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i % c8);
        const long n = i / c8;
        const uint4* p = x + n * (long)hw * c8 + c;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int qd = 0; qd < hw; ++qd) {
            const bf16x8 v = __builtin_bit_cast(bf16x8, p[(long)qd * c8]);
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += (float)v[k];
        }
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (__bf16)(s[k] / (float)hw);
        y[i] = __builtin_bit_cast(uint4, o);
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int conv_out(int x, int k, int s, int p) { return (x + 2 * p - k) / s + 1; }

}  // namespace

namespace rpg {

void bf16_set_bk(int bk) { g_bf16_bk = bk; }
void bf16_set_fast(int on) { g_bf16_fast = on; }
void bf16_set_tile(int t) { g_bf16_tile = t; }
void bf16_set_dma(int v) { g_bf16_dma = v; }
void bf16_set_patch(int v) { g_bf16_stages = v >= 10 ? 3 : 4; g_bf16_patch = v % 10; }
void bf16_set_fused_stem(int on) { g_bf16_fused_stem = on; }
void bf16_set_lean_epi(int on) { g_bf16_lean_epi = on; }
void bf16_set_persist(int v) { g_bf16_persist = v; }
void bf16_set_fuse_block(int v) { g_bf16_fuse_block = v; }
void bf16_set_tail(int v) { g_bf16_tail = v & 7; }
void bf16_set_linear_dma(int v) { g_bf16_linear_dma = v; }
void bf16_set_chunk(int images, int min_mb) { g_bf16_chunk = images; g_bf16_chunk_mb = min_mb; }
#ifdef RPG_PROBE_WS64
int bf16_set_ws64(int v) { g_bf16_ws64 = v; return RPG_OK; }
#else
int bf16_set_ws64(int v) { return v == 0 ? RPG_OK : RPG_ERR_BAD_ARG; }      // the probe kernel is not in this build
#endif

// fp32 [rows][ld_src] (first `cols` columns) -> bf16 dst[rows][ld_dst] at column offset col_off (cols % 8 == 0)
__global__ __launch_bounds__(NT) void f32_to_bf16_kernel(const float* __restrict__ src, int ld_src, __bf16* __restrict__ dst,
                                                         int ld_dst, int col_off, long total8, int cols8) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total8; i += (long)gridDim.x * NT) {
        const long r = i / cols8;
        const int c = (int)(i - r * cols8) * 8;
        const float4 a = *reinterpret_cast<const float4*>(src + r * ld_src + c);
        const float4 b = *reinterpret_cast<const float4*>(src + r * ld_src + c + 4);
        const bf16x8 v = {(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w, (__bf16)b.x, (__bf16)b.y, (__bf16)b.z, (__bf16)b.w};
        *reinterpret_cast<uint4*>(dst + r * ld_dst + col_off + c) = __builtin_bit_cast(uint4, v);
    }
}

int launch_f32_to_bf16(const float* src, int ld_src, void* dst, int ld_dst, int col_off, long rows, int cols, hipStream_t s) {
    if (!src || !dst || rows <= 0 || cols <= 0 || (cols & 7) || (ld_src & 3) || (ld_dst & 7) || (col_off & 7) ||
        !aligned16(src) || !aligned16(dst))
        return RPG_ERR_BAD_ARG;
    const long total8 = rows * (cols / 8);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(capped_grid(total8)), dim3(NT), 0, s, src, ld_src,
                       reinterpret_cast<__bf16*>(dst), ld_dst, col_off, total8, cols / 8);
    RPG_CHECK_LAUNCH("f32_to_bf16");
    return RPG_OK;
}

// out[m][n_out] (fp32) = act(a[m][k] (bf16) * w[n_out][k]^T (bf16) + bias + residual rows): the Linear of the bf16 GNN
// as a 1x1 convolution over an m-pixel "image"; residual: fp32 rows, optionally gathered (see EpiB).
int launch_linear_bf16(const void* a, const void* w, const float* bias, const float* res, const int64_t* res_idx,
                       const float* res2, const int64_t* res2_idx, int ldr, float* out, int m, int k, int n_out, int relu,
                       hipStream_t s) {
    LinearBf16Out o{};
    o.out = out; o.out_f32 = 1;
    return launch_linear_bf16_ex(a, k, w, bias, res, res_idx, res2, res2_idx, ldr, o, m, k, n_out, relu, s);
}

// The general form: A rows with pitch lda >= k (a column block of a wider bf16 matrix), primary output fp32 or bf16 or
// none, optional second bf16 output with its own pitch and ReLU (EpiB::out2): the Linears of the bf16 GNN hand their result
// to the next GEMM in bf16 from the epilogue.
int launch_linear_bf16_ex(const void* a, int lda, const void* w, const float* bias, const float* res, const int64_t* res_idx,
                          const float* res2, const int64_t* res2_idx, int ldr, const LinearBf16Out& o, int m, int k, int n_out,
                          int relu, hipStream_t s) {
    if (!a || !w || (!o.out && !o.out2) || m <= 0 || k <= 0 || n_out <= 0 || (k & 7) || (n_out & 3) || lda < k || (lda & 7) ||
        !aligned16(a) || !aligned16(w) || (o.out && !aligned16(o.out)) || (bias && !aligned16(bias)) ||
        (res && (!aligned16(res) || (ldr & 3) || ldr < n_out)) || (res2 && (!res || !res2_idx || !aligned16(res2))) ||
        (o.out2 && ((reinterpret_cast<uintptr_t>(o.out2) & 7) || (o.ld2 & 3) || o.ld2 < n_out)))
        return RPG_ERR_BAD_ARG;
    ConvArgsB ca{reinterpret_cast<const __bf16*>(a), 1, 1, k, 1, 1, 1, 0, 1, 1};
    ca.img_elems = lda == k ? 0 : lda;
    EpiB ep{nullptr, bias, nullptr, o.out, n_out, relu, o.out_f32};
    ep.res_f32 = res; ep.res_idx = res_idx; ep.res2_f32 = res2; ep.res2_idx = res2_idx; ep.ldr = ldr;
    ep.out2 = reinterpret_cast<__bf16*>(o.out2); ep.ld2 = o.ld2; ep.relu2 = o.relu2;
    const __bf16* wp = reinterpret_cast<const __bf16*>(w);
    const int slot = timing_begin(RPG_TIMER_LINEAR, s);
    const bool fast = g_bf16_fast && k % 64 == 0 && 258L * lda * 2 < (1L << 31) && (long)n_out * k * 2 < (1L << 31);
    const long t128 = (long)((m + 127) / 128) * ((n_out + 127) / 128);
    // RPG_TUNE_BF16_LINEAR_DMA = 10 + i: configuration i of the LDS-DMA kernel for the edge-row Linears (a 1 x 1 convolution over an
    // m-pixel image; the general epilogue carries the gathers and the second output)
    const bool dma_ok = 1026L * lda * 2 < (1L << 31) && (long)n_out * k * 2 < (1L << 31);
    if (g_bf16_linear_dma >= 10 && dma_ok && t128 >= 192 && launch_dma_config(g_bf16_linear_dma - 10, ca, wp, m, n_out, k, ep, s)) {
    } else if (fast) {
        if (t128 >= 192) launch_fast<128, 128, 2, 2>(ca, wp, m, n_out, k, ep, s);
        else launch_fast<64, 64, 2, 2>(ca, wp, m, n_out, k, ep, s);
    } else {
        if (t128 >= 192) launch_tile<128, 128, 2, 2, 32>(ca, wp, m, n_out, k, ep, s);
        else launch_tile<64, 64, 2, 2, 32>(ca, wp, m, n_out, k, ep, s);
    }
    timing_end(slot, 2.0 * (double)m * n_out * (double)k, s);
    RPG_CHECK_LAUNCH("linear_bf16");
    return RPG_OK;
}

void bf16_set_pair(int v) { g_bf16_pair = v; }

// conv A (kh x kw / stride / pad) and conv B (1 x 1 / same stride, pad 0) of the same input x [n][h][w][cin] -> ya, yb (both
// [n][ho][wo][cout], bf16, BN folded, ReLU on A only per `relu_a`); false: not eligible (the caller launches them one by one)
bool launch_conv_pair_bf16(const void* x, const void* wa, const float* sa, const float* ha, void* ya, const void* wb_, const float* sb,
                           const float* hb, void* yb, int n, int h, int wd, int cin, int cout, int k, int stride, int pad, hipStream_t s) {
    if (!g_bf16_pair || g_bf16_dma != 1 || !x || !wa || !wb_ || !ya || !yb || n <= 0 || (cin & 31) || (cout & 3)) return false;
    const int ho = conv_out(h, k, stride, pad), wo = conv_out(wd, k, stride, pad);
    if (ho <= 0 || wo <= 0 || ho != conv_out(h, 1, stride, 0) || wo != conv_out(wd, 1, stride, 0)) return false;
    const long M = (long)n * ho * wo, Ka = (long)k * k * cin, Kb = cin;
    if (M < 8192 || M >= (1L << 31) || Ka >= (1 << 24)) return false;
    const long span1k = 1024 / ((long)ho * wo) + 2;
    if (span1k * h * wd * cin * 2 >= (1L << 31) || (long)cout * Ka * 2 >= (1L << 31)) return false;
    if (!aligned16(x) || !aligned16(wa) || !aligned16(wb_) || !aligned16(ya) || !aligned16(yb)) return false;
    DmaOne c1{}, c2{};
    c1.a = ConvArgsB{reinterpret_cast<const __bf16*>(x), h, wd, cin, k, k, stride, pad, ho, wo};
    c2.a = ConvArgsB{reinterpret_cast<const __bf16*>(x), h, wd, cin, 1, 1, stride, 0, ho, wo};
    c1.Wt = reinterpret_cast<const __bf16*>(wa); c2.Wt = reinterpret_cast<const __bf16*>(wb_);
    c1.M = c2.M = (int)M; c1.N = c2.N = cout; c1.K = (int)Ka; c2.K = (int)Kb;
    c1.ep = EpiB{sa, ha, nullptr, ya, cout, 1, 0};
    c2.ep = EpiB{sb, hb, nullptr, yb, cout, 0, 0};
    const bool lean = g_bf16_lean_epi && 1024L * cout * 2 < (1L << 31);
    c1.ep.lean = lean; c2.ep.lean = lean;
    const int slot = timing_begin(RPG_TIMER_CONV, s);
    // the tile template of launch_conv_bf16's choice for the 3x3 (configuration 7 up to 128 channels, 0 above where 256 x 256 tiles
    // fill three quarters of the CUs); the shortcut takes the same (its own choice by shape is the same in ResNet34)
    const long t256 = ((M + 255) / 256) * ((cout + 255) / 256);
    if (cout <= 128 || 4 * t256 < 3L * num_cus() || cin % 64) launch_dma_pair<256, 128, 4, 2, 32, 3>(c1, c2, s);
    else launch_dma_pair<256, 256, 2, 4, 64, 2>(c1, c2, s);
    timing_end(slot, 2.0 * (double)M * cout * (double)(Ka + Kb), s);
    return hipGetLastError() == hipSuccess;
}

int launch_conv_bf16(const void* x, const void* w, const float* scale, const float* shift, const void* residual, void* y,
                     int n, int h, int wd, int cin, int cout, int kh, int kw, int stride, int pad, int relu, int out_f32,
                     hipStream_t s) {
    if (!x || !w || !y || n <= 0 || h <= 0 || wd <= 0 || cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 ||
        pad < 0 || (cin & 7) || (cout & 3) || !aligned16(x) || !aligned16(w) || !aligned16(y) ||
        (scale && !aligned16(scale)) || (shift && !aligned16(shift)) || (residual && (reinterpret_cast<uintptr_t>(residual) & 7)))
        return RPG_ERR_BAD_ARG;
    const int ho = conv_out(h, kh, stride, pad), wo = conv_out(wd, kw, stride, pad);
    if (ho <= 0 || wo <= 0) return RPG_ERR_BAD_ARG;
    const long M = (long)n * ho * wo, K = (long)kh * kw * cin;
    if (M >= (1L << 31) || K >= (1 << 24) || (long)h * wd * cin >= (1L << 31)) return RPG_ERR_BAD_ARG;
    ConvArgsB a{reinterpret_cast<const __bf16*>(x), h, wd, cin, kh, kw, stride, pad, ho, wo};
    EpiB ep{scale, shift, reinterpret_cast<const __bf16*>(residual), y, cout, relu, out_f32};
    // plain convolution: the lean epilogue (RPG_TUNE_BF16_LEAN_EPI); a workgroup's <= 1024 rows within 32-bit byte offsets
    ep.lean = g_bf16_lean_epi && !out_f32 && (cout & 3) == 0 && 1024L * cout * 2 < (1L << 31);
    const __bf16* wp = reinterpret_cast<const __bf16*>(w);
    const int slot = timing_begin(RPG_TIMER_CONV, s);
    const bool big_k = g_bf16_bk == 64 && K >= 128;
    // the interleaved buffer-load kernel: whole 64-element K steps per kernel tap, 32-bit offsets relative to the
    // first image of a tile (whose <= 256 rows span at most 256 / (ho*wo) + 2 images)
    const long span = 256 / ((long)ho * wo) + 2;
    const bool fast = g_bf16_fast && cin % 64 == 0 && span * h * wd * cin * 2 < (1L << 31) && (long)cout * K * 2 < (1L << 31);
    const long span1k = 1024 / ((long)ho * wo) + 2;          // the widest DMA tile has 1024 rows
    const bool dma_ok = span1k * h * wd * cin * 2 < (1L << 31) && (long)cout * K * 2 < (1L << 31);
    // RPG_TUNE_BF16_DMA = 1: the LDS-DMA kernel by shape (measured at 256 / 512 images, tools/conv_bench.py --dma-sweep,
    // profiles/r3_bf16_dma_sweep_*.txt): 256 x 64 tiles with two workgroups per CU for 64 output channels (layer 1: the operand
    // fill from L2 bounds it, N is too small to amortise the A tile), 256 x 128 / K step 32 / two workgroups per CU up to 128
    // channels and wherever 256 x 256 tiles would leave a quarter of the CUs idle, 256 x 256 (wave tile 128 x 64) otherwise
    int dma_cfg = -1;
    if (dma_ok && g_bf16_dma == 1 && M >= 8192 && cin % 32 == 0) {
        const long t256 = ((M + 255) / 256) * ((cout + 255) / 256);
        if (cout <= 64) dma_cfg = cin % 64 == 0 ? 5 : -1;
        else if (cout <= 128 || 4 * t256 < 3L * num_cus() || cin % 64) dma_cfg = 7;
        else dma_cfg = 0;
    } else if (dma_ok && g_bf16_dma >= 10) {
        dma_cfg = g_bf16_dma - 10;                    // forced configuration (experiments)
    }
    // 3x3 / stride 1: the patch kernel (input pixels fetched once per 32-channel chunk instead of once per tap)
    bool done = false;
#ifdef RPG_PROBE_WS64
    if (g_bf16_ws64 && cin == 64 && cout == 64 && M >= 8192)
        done = g_bf16_ws64 == 2 ? launch_ws64<2>(a, wp, n, (int)M, cout, ep, s) : launch_ws64<3>(a, wp, n, (int)M, cout, ep, s);
#endif
    if (!done && g_bf16_patch && kh == 3 && kw == 3 && stride == 1 && pad == 1 && (g_bf16_patch >= 2 || M >= 8192)) {
        // measured (tools/conv_bench.py --dma-sweep, r3): from 256 output channels up the patch kernel beats the im2col DMA kernel
        // by 5-10 %; 256 x 256 tiles where they occupy at least three quarters of the CUs, else 256 x 128 (layer 4 at 256 images:
        // 98 vs 196 tiles, 98 vs 68 us); at 128 channels the DMA kernel's 256 x 128 / K-step-32 configuration is faster
        const long t256 = ((M + 255) / 256) * ((cout + 255) / 256);
        // Tile by output width (every variant has 128 x 64 or 64 x 64 wave tiles on 8 waves): 256 x 256 from 256 channels up where
        // that fills three quarters of the CUs, else 256 x 128; 512 x 128 for 128 channels, 512 x 64 for 64 -- the narrow layers
        // need the tall tile to put 8-16 MFMAs per wave between two barriers.  Four weight stages (next step's weight fragments
        // read before the barrier) where the LDS allows and it measured faster, else three.
        const bool big = M >= 65536 || g_bf16_patch >= 2;
        // Tail re-tiling (round 5, RPG_TUNE_BF16_TAIL): M = 49 * 2^k pixels puts 49 * 2^j tiles on 256 CUs -- 3.06 rounds on layer 2,
        // 1.53 on layer 3 at 512 images -- and the last, mostly empty round costs a whole tile time.  Splitting the tail tiles along
        // K (what the fp32 Winograd kernel does) would move fp32 partial slabs of 256 KB per part through HBM: more than the round it
        // saves.  Instead the rows beyond the last FULL round go to a second launch of the same kernel family with SMALLER tiles
        // (160 x 256 on 8 x 1 waves / 256 x 128): one short round instead of a full one, no partial sums, same arithmetic per output.
        const int cus = num_cus();
        auto tail_split = [&](int bm, int bm_tail) -> int {        // rows of the main launch, or 0: do not split
            if (!g_bf16_tail || cout > 256 || M >= (1L << 30)) return 0;
            const long tiles = (M + bm - 1) / bm;
            const long main_tiles = tiles / cus * cus, rem = tiles - main_tiles;
            if (main_tiles == 0 || rem == 0) return 0;
            const long tail_rows = M - main_tiles * bm;
            return (tail_rows + bm_tail - 1) / bm_tail <= cus ? (int)(main_tiles * bm) : 0;
        };
        if (g_bf16_patch == 3) {
            done = launch_patch<256, 128, 4, 2, 3>(a, wp, n, (int)M, cout, ep, s);
        } else if (g_bf16_patch == 4 && cout > 128) {
            // round-6 experiment (RPG_TUNE_BF16_PATCH = 4): 512 x 128 tiles on the 256- / 512-channel layers too -- 15 instead of 21 bytes of
            // L2 -> LDS fill per output.  Measured at 512 images: layer 3 124-127 us against 127-130, layer 4 126-137 against 119-123: the
            // fill volume is not what bounds these layers either
            done = launch_patch<512, 128, 4, 2, 4>(a, wp, n, (int)M, cout, ep, s) || launch_patch<512, 128, 4, 2, 3>(a, wp, n, (int)M, cout, ep, s);
        } else if (cout > 128 && 4 * t256 >= 3L * num_cus()) {
            const int m_main = (g_bf16_tail & 2) && cout <= 256 ? tail_split(256, 160) : 0;
            done = (g_bf16_stages != 3 && launch_patch<256, 256, 2, 4, 4>(a, wp, n, (int)M, cout, ep, s, 0, m_main)) ||
                   launch_patch<256, 256, 2, 4, 3>(a, wp, n, (int)M, cout, ep, s, 0, m_main);
            if (done && m_main && !launch_patch<160, 256, 1, 8, 3>(a, wp, n, (int)M, cout, ep, s, m_main, 0))
                done = launch_patch<256, 256, 2, 4, 3>(a, wp, n, (int)M, cout, ep, s, m_main, 0);
        } else if (cout > 128) {
            done = launch_patch<256, 128, 4, 2, 3>(a, wp, n, (int)M, cout, ep, s);
        } else if (cout > 64 && big) {
            // (the persistent form of this tile was built and measured slower -- 141 -> 179 us: at 256 VGPRs the next tile's piece
            // addresses, computed in front of the last chunk with the accumulators live, spill 65-113 registers)
            // (bit 2, round-6 experiment: the tail on 128 x 128 tiles -- twice the workgroups, half the rows each)
            const int m_main = (g_bf16_tail & 1) ? tail_split(512, (g_bf16_tail & 4) ? 128 : 256) : 0;
            done = (g_bf16_stages != 3 && launch_patch<512, 128, 4, 2, 4>(a, wp, n, (int)M, cout, ep, s, 0, m_main)) ||
                   launch_patch<512, 128, 4, 2, 3>(a, wp, n, (int)M, cout, ep, s, 0, m_main);
            if (done && m_main && (g_bf16_tail & 4) && launch_patch<128, 128, 4, 2, 3>(a, wp, n, (int)M, cout, ep, s, m_main, 0)) {
            } else if (done && m_main && !launch_patch<256, 128, 4, 2, 3>(a, wp, n, (int)M, cout, ep, s, m_main, 0))
                done = launch_patch<512, 128, 4, 2, 3>(a, wp, n, (int)M, cout, ep, s, m_main, 0);
        } else if (g_bf16_patch >= 2 || big) {
            // 64 output channels (layer 1).  Round 3 measured 232-245 us at 512 images against 205-222 us for the im2col DMA kernel
            // with two 80-KB workgroups per CU (configuration 5: they hide each other's epilogue) and kept this tile for tests only.
            // With the lean epilogue of round 4 the order flipped where it matters: 216-223 / 229-239 us (without / with residual)
            // against 214-215 / 246-248, and the whole configs[2] step gains 2.4 % (12.79 -> 13.10 k graphs/s, two streams:
            // profiles/r4_bf16_epilogue_experiments.txt) -- a quarter of the L2 -> LDS traffic leaves more of the chip to the other stream.
            done = (g_bf16_persist && g_bf16_stages != 3 && launch_patch<512, 64, 8, 1, 4, true>(a, wp, n, (int)M, cout, ep, s)) ||
                   (g_bf16_stages != 3 && launch_patch<512, 64, 8, 1, 4>(a, wp, n, (int)M, cout, ep, s)) ||
                   launch_patch<512, 64, 8, 1, 3>(a, wp, n, (int)M, cout, ep, s);
        }
    }
    if (done) {
    } else if (dma_cfg >= 0 && launch_dma_config(dma_cfg, a, wp, (int)M, cout, (int)K, ep, s)) {
    } else if (fast && g_bf16_tile >= 0) {       // RPG_TUNE_BF16_TILE: forced tile of the interleaved kernel (experiments)
        if (g_bf16_tile == 0) launch_fast<64, 64, 2, 2>(a, wp, (int)M, cout, (int)K, ep, s);
        else if (g_bf16_tile == 1) launch_fast<128, 128, 2, 2>(a, wp, (int)M, cout, (int)K, ep, s);
        else if (g_bf16_tile == 3) launch_fast<128, 64, 4, 1>(a, wp, (int)M, cout, (int)K, ep, s);
        else launch_fast<256, 64, 4, 1>(a, wp, (int)M, cout, (int)K, ep, s);
    } else if (fast) {
        // Cout = 64 on many rows (layer 1): 128 x 64 (three workgroups per CU; 238 vs 251 us with 256 x 64 at 512 images)
        if (cout <= 64 && M >= 65536) launch_fast<128, 64, 4, 1>(a, wp, (int)M, cout, (int)K, ep, s);
        else if (cout <= 64 || (long)((M + 127) / 128) * ((cout + 127) / 128) < 256) launch_fast<64, 64, 2, 2>(a, wp, (int)M, cout, (int)K, ep, s);
        else launch_fast<128, 128, 2, 2>(a, wp, (int)M, cout, (int)K, ep, s);
    } else if (cout <= 64 && M >= 65536) {
        if (big_k) launch_tile<256, 64, 4, 1, 64>(a, wp, (int)M, cout, (int)K, ep, s);
        else launch_tile<256, 64, 4, 1, 32>(a, wp, (int)M, cout, (int)K, ep, s);
    } else if (cout <= 64 || (long)((M + 127) / 128) * ((cout + 127) / 128) < 256) {
        launch_tile<64, 64, 2, 2, 32>(a, wp, (int)M, cout, (int)K, ep, s);
    } else {
        if (big_k) launch_tile<128, 128, 2, 2, 64>(a, wp, (int)M, cout, (int)K, ep, s);
        else launch_tile<128, 128, 2, 2, 32>(a, wp, (int)M, cout, (int)K, ep, s);
    }
    timing_end(slot, 2.0 * (double)M * cout * (double)K, s);
    RPG_CHECK_LAUNCH("conv2d_bn_act_bf16");
    return RPG_OK;
}

}  // namespace rpg

extern "C" int rpg_conv2d_bn_act_nhwc_bf16(const void* x, const void* w_ohwi, const float* scale, const float* shift,
                                           const void* residual, void* y, int n, int h, int w, int cin, int cout, int kh,
                                           int kw, int stride, int pad, int relu, int out_f32, void* stream) {
    return rpg::launch_conv_bf16(x, w_ohwi, scale, shift, residual, y, n, h, w, cin, cout, kh, kw, stride, pad, relu, out_f32,
                                 rpg::as_stream(stream));
}

extern "C" size_t rpg_resnet_bf16_workspace_bytes(int n, int h, int w, const int* planes) {
    if (n <= 0 || h <= 0 || w <= 0 || !planes) return 0;
    const int h1 = conv_out(h, 7, 2, 3), w1 = conv_out(w, 7, 2, 3), h2 = conv_out(h1, 3, 2, 1), w2 = conv_out(w1, 3, 2, 1);
    size_t blk = 0;
    int hh = h2, ww = w2;
    for (int l = 0; l < 4; ++l) {
        if (l > 0) { hh = conv_out(hh, 3, 2, 1); ww = conv_out(ww, 3, 2, 1); }
        const size_t sz = (size_t)n * hh * ww * planes[l];
        if (sz > blk) blk = sz;
    }
    return align_up((size_t)n * h * w * 8 * 2, 256) + align_up((size_t)n * h1 * w1 * planes[0] * 2, 256) +
           4 * align_up(blk * 2, 256) + align_up((size_t)n * planes[3] * 2, 256) + 7 * rpg::kWorkspaceSkew;
}

// tensors: per conv {w_ohwi bf16, scale f32, shift f32} (stem Cin padded to 8), then fc weight bf16 [feat][512], bias f32.
static int resnet_forward_bf16_impl(const void* const* tensors, int n_tensors, const int* blocks, const int* planes, int feat_dim,
                                   const void* x_nchw_any, int x_is_bf16, int n, int h, int w, float* feat, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    const float* x_nchw = reinterpret_cast<const float*>(x_nchw_any);       // (only read as fp32 when !x_is_bf16)
    if (!tensors || !blocks || !planes || !x_nchw || !feat || !workspace || n <= 0 || h <= 0 || w <= 0 || feat_dim <= 0 ||
        (feat_dim & 3))
        return RPG_ERR_BAD_ARG;
    int expect = 3 + 2, cin = planes[0];
    for (int l = 0; l < 4; ++l) {
        if (planes[l] & 7) return RPG_ERR_BAD_ARG;
        for (int b = 0; b < blocks[l]; ++b) {
            const int stride = (l > 0 && b == 0) ? 2 : 1;
            expect += 6 + ((stride != 1 || cin != planes[l]) ? 3 : 0);
            cin = planes[l];
        }
    }
    // optional last tensor: the operands of the fused stem kernel (stem_bf16.hip; 64-channel stems)
    if (n_tensors != expect && n_tensors != expect + 1) return RPG_ERR_BAD_ARG;
    const void* stem_pack = n_tensors == expect + 1 ? tensors[expect] : nullptr;
    for (int i = 0; i < n_tensors; ++i)
        if (!tensors[i]) return RPG_ERR_BAD_ARG;
    if (workspace_bytes < rpg_resnet_bf16_workspace_bytes(n, h, w, planes)) return RPG_ERR_WORKSPACE;
    hipStream_t s = rpg::as_stream(stream);
    const int h1 = conv_out(h, 7, 2, 3), w1 = conv_out(w, 7, 2, 3), h2 = conv_out(h1, 3, 2, 1), w2 = conv_out(w1, 3, 2, 1);
    size_t blk = 0;
    {
        int hh = h2, ww = w2;
        for (int l = 0; l < 4; ++l) {
            if (l > 0) { hh = conv_out(hh, 3, 2, 1); ww = conv_out(ww, 3, 2, 1); }
            const size_t sz = (size_t)n * hh * ww * planes[l];
            if (sz > blk) blk = sz;
        }
    }
    char* base = reinterpret_cast<char*>(workspace);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base + off;
        off += align_up(bytes, 256) + rpg::kWorkspaceSkew;      // see rpg_common.h: de-aliases the HBM channels
        return reinterpret_cast<void*>(p);
    };
    void* in8 = take((size_t)n * h * w * 8 * 2);
    void* stem = take((size_t)n * h1 * w1 * planes[0] * 2);
    void* buf[4];
    for (int i = 0; i < 4; ++i) buf[i] = take(blk * 2);
    void* pool = take((size_t)n * planes[3] * 2);

    int rc, ti = 0;
    if (stem_pack && g_bf16_fused_stem && rpg::stem_pool_bf16_supported(n, h, w, planes[0])) {
        // fp32 NCHW -> conv7x7/2 + BN + ReLU + maxpool3x3/2 -> pooled bf16 NHWC, one kernel
        if ((rc = rpg::launch_stem_pool_bf16(x_nchw_any, x_is_bf16, stem_pack, (const float*)tensors[1], (const float*)tensors[2], buf[0],
                                             n, h, w, s)) != RPG_OK)
            return rc;
    } else {
        // three-kernel stem (RPG_TUNE_FUSED_STEM = 0, non-64-channel stems, no stem pack): re-layout (from fp32, or from images
        // the host already rounded to bf16 -- the same bits either way), generic convolution on the 8-channel image, max-pool
        const long npix = (long)n * h * w;
        if (x_is_bf16)
            hipLaunchKernelGGL(nchw3_to_nhwc8_bf16_kernel<__bf16>, dim3(capped_grid(npix)), dim3(NT), 0, s,
                               reinterpret_cast<const __bf16*>(x_nchw_any), reinterpret_cast<uint4*>(in8), npix, h * w);
        else
            hipLaunchKernelGGL(nchw3_to_nhwc8_bf16_kernel<float>, dim3(capped_grid(npix)), dim3(NT), 0, s, x_nchw,
                               reinterpret_cast<uint4*>(in8), npix, h * w);
        if ((rc = rpg::launch_conv_bf16(in8, tensors[0], (const float*)tensors[1], (const float*)tensors[2], nullptr, stem, n, h,
                                        w, 8, planes[0], 7, 7, 2, 3, 1, 0, s)) != RPG_OK)
            return rc;
        const int c8 = planes[0] / 8;
        const long total = (long)n * h2 * w2 * c8;
        hipLaunchKernelGGL(maxpool3x3s2_bf16_kernel, dim3(capped_grid(total)), dim3(NT), 0, s,
                           reinterpret_cast<const uint4*>(stem), reinterpret_cast<uint4*>(buf[0]), h1, w1, c8, h2, w2, total);
    }
    ti += 3;
    int cur = 0, hh = h2, ww = w2;
    cin = planes[0];
    for (int l = 0; l < 4; ++l) {
        for (int b = 0; b < blocks[l]; ++b) {
            const int stride = (l > 0 && b == 0) ? 2 : 1;
            const int c = planes[l];
            const bool ds = (stride != 1 || cin != c);
            const int ho = conv_out(hh, 3, stride, 1), wo = conv_out(ww, 3, stride, 1);
            // Depth-first over image groups (round 4, RPG_TUNE_BF16_CHUNK): a run of identity blocks (stride 1, no downsample)
            // on LARGE activation tensors is walked group of images by group of images -- conv1, conv2 of every block of the
            // run on `g_bf16_chunk` images, then the next group -- so that a group's intermediate tensors (and the residual) are
            // re-read from the 256-MB Infinity Cache instead of HBM.  Images are independent: same kernels, same arithmetic,
            // only the pointers and the image count of a launch change.
            const size_t act_bytes = (size_t)n * hh * ww * c * 2;
            if (!ds && g_bf16_chunk > 0 && n > g_bf16_chunk && act_bytes >= ((size_t)g_bf16_chunk_mb << 20)) {
                int run = 1;                                   // identity blocks b .. b + run - 1 (all that follow in this layer)
                while (b + run < blocks[l]) ++run;
                const size_t img = (size_t)hh * ww * c * 2;    // bytes per image of every tensor in the run
                for (int i0 = 0; i0 < n; i0 += g_bf16_chunk) {
                    const int ni = n - i0 < g_bf16_chunk ? n - i0 : g_bf16_chunk;
                    int cc = cur, tj = ti;
                    for (int r = 0; r < run; ++r) {
                        char* X = static_cast<char*>(buf[cc]) + i0 * img;
                        char* T = static_cast<char*>(buf[(cc + 1) & 3]) + i0 * img;
                        char* Y = static_cast<char*>(buf[(cc + 2) & 3]) + i0 * img;
                        if ((rc = rpg::launch_conv_bf16(X, tensors[tj], (const float*)tensors[tj + 1], (const float*)tensors[tj + 2],
                                                        nullptr, T, ni, hh, ww, c, c, 3, 3, 1, 1, 1, 0, s)) != RPG_OK)
                            return rc;
                        if ((rc = rpg::launch_conv_bf16(T, tensors[tj + 3], (const float*)tensors[tj + 4], (const float*)tensors[tj + 5],
                                                        X, Y, ni, hh, ww, c, c, 3, 3, 1, 1, 1, 0, s)) != RPG_OK)
                            return rc;
                        tj += 6;
                        cc = (cc + 2) & 3;
                    }
                }
                ti += 6 * run;
                cur = (cur + 2 * run) & 3;
                b += run - 1;
                cin = c;
                continue;
            }
            void* X = buf[cur];
            void* T = buf[(cur + 1) & 3];
            void* Y = buf[(cur + 2) & 3];
            void* D = buf[(cur + 3) & 3];
            // 64-channel identity block (layer 1): conv1 + BN + ReLU + conv2 + BN + identity + ReLU in one kernel, the intermediate
            // on chip (block_bf16.inc; bit-identical to the two launches below)
            if (!ds && c == 64 && g_bf16_fuse_block && (long)n * hh * ww >= 8192) {
                const int slot = rpg::timing_begin(RPG_TIMER_CONV, s);
                if (launch_block64_fused(X, tensors[ti], (const float*)tensors[ti + 1], (const float*)tensors[ti + 2], tensors[ti + 3],
                                         (const float*)tensors[ti + 4], (const float*)tensors[ti + 5], Y, n, hh, ww, s)) {
                    rpg::timing_end(slot, 2.0 * 2.0 * (double)n * hh * ww * 64.0 * 9.0 * 64.0, s);
                    RPG_CHECK_LAUNCH("basicblock64_bf16");
                    ti += 6;
                    cur = (cur + 2) & 3;
                    continue;
                }
                rpg::timing_end(slot, 0.0, s);
            }
            const void* identity = X;
            bool paired = false;
            if (ds && stride == 2)
                paired = rpg::launch_conv_pair_bf16(X, tensors[ti], (const float*)tensors[ti + 1], (const float*)tensors[ti + 2], T, tensors[ti + 6],
                                                    (const float*)tensors[ti + 7], (const float*)tensors[ti + 8], D, n, hh, ww, cin, c, 3, stride, 1, s);
            if (paired) {
                identity = D;
            } else if ((rc = rpg::launch_conv_bf16(X, tensors[ti], (const float*)tensors[ti + 1], (const float*)tensors[ti + 2],
                                                   nullptr, T, n, hh, ww, cin, c, 3, 3, stride, 1, 1, 0, s)) != RPG_OK) {
                return rc;
            }
            if (ds && !paired) {
                if ((rc = rpg::launch_conv_bf16(X, tensors[ti + 6], (const float*)tensors[ti + 7], (const float*)tensors[ti + 8],
                                                nullptr, D, n, hh, ww, cin, c, 1, 1, stride, 0, 0, 0, s)) != RPG_OK)
                    return rc;
                identity = D;
            }
            if ((rc = rpg::launch_conv_bf16(T, tensors[ti + 3], (const float*)tensors[ti + 4], (const float*)tensors[ti + 5],
                                            identity, Y, n, ho, wo, c, c, 3, 3, 1, 1, 1, 0, s)) != RPG_OK)
                return rc;
            ti += ds ? 9 : 6;
            cur = (cur + 2) & 3;
            hh = ho; ww = wo; cin = c;
        }
    }
    {
        const int c8 = cin / 8;
        const long total = (long)n * c8;
        hipLaunchKernelGGL(global_avgpool_bf16_kernel, dim3(capped_grid(total)), dim3(NT), 0, s,
                           reinterpret_cast<const uint4*>(buf[cur]), reinterpret_cast<uint4*>(pool), hh * ww, c8, total);
    }
    // fc as a 1x1 convolution on a 1x1 image: [n][1][1][cin] x [feat][1][1][cin], bias in `shift`, fp32 output
    return rpg::launch_conv_bf16(pool, tensors[ti], nullptr, (const float*)tensors[ti + 1], nullptr, feat, n, 1, 1, cin,
                                 feat_dim, 1, 1, 1, 0, 0, 1, s);
}

extern "C" int rpg_resnet_forward_bf16(const void* const* tensors, int n_tensors, const int* blocks, const int* planes,
                                       int feat_dim, const float* x_nchw, int n, int h, int w, float* feat, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    return resnet_forward_bf16_impl(tensors, n_tensors, blocks, planes, feat_dim, x_nchw, 0, n, h, w, feat, workspace, workspace_bytes,
                                    stream);
}

// the same forward on node images that are already bf16 (rounded from fp32 on the host, so that the H2D copy is half the size);
// with or without the fused stem's operands (the three-kernel stem's re-layout pass takes bf16 pixels too, round 4)
extern "C" int rpg_resnet_forward_bf16_xbf16(const void* const* tensors, int n_tensors, const int* blocks, const int* planes,
                                             int feat_dim, const void* x_nchw_bf16, int n, int h, int w, float* feat, void* workspace,
                                             size_t workspace_bytes, void* stream) {
    return resnet_forward_bf16_impl(tensors, n_tensors, blocks, planes, feat_dim, x_nchw_bf16, 1, n, h, w, feat, workspace,
                                    workspace_bytes, stream);
}

extern "C" int rpg_basicblock64_bf16(const void* x, const void* w1_ohwi, const float* scale1, const float* shift1, const void* w2_ohwi,
                                      const float* scale2, const float* shift2, void* y, int n, int h, int w, void* stream) {
    if (!x || !y || x == y || !rpg::aligned16(x) || !rpg::aligned16(y) || !rpg::aligned16(w1_ohwi) || !rpg::aligned16(w2_ohwi)) return RPG_ERR_BAD_ARG;
    if (!launch_block64_fused(x, w1_ohwi, scale1, shift1, w2_ohwi, scale2, shift2, y, n, h, w, rpg::as_stream(stream))) return RPG_ERR_BAD_ARG;
    RPG_CHECK_LAUNCH("basicblock64_bf16");
    return RPG_OK;
}

extern "C" int rpg_f32_to_bf16(const float* src, int ld_src, void* dst, int ld_dst, int col_off, long rows, int cols,
                               void* stream) {
    return rpg::launch_f32_to_bf16(src, ld_src, dst, ld_dst, col_off, rows, cols, rpg::as_stream(stream));
}

extern "C" int rpg_linear_bf16(const void* a, const void* weight, const float* bias, const float* residual,
                               const int64_t* res_idx, const float* residual2, const int64_t* res2_idx, int ldr, float* out,
                               int m, int k, int n_out, int relu, void* stream) {
    return rpg::launch_linear_bf16(a, weight, bias, residual, res_idx, residual2, res2_idx, ldr, out, m, k, n_out, relu,
                                   rpg::as_stream(stream));
}

// ------------------------------------------------------------------------------------------------------------------------------
// Measurement aid (no reference counterpart): what the bf16 matrix pipe of THIS device sustains chip-wide with
// nothing else drawing power -- `iters` x 16 v_mfma_f32_32x32x16_bf16 per wave on 4 independent accumulators, operands from
// `operands` (65,536 x 16 bytes of bf16: the caller chooses zeros / random / ReLU-like data; on MI355X the clock under this load is
// 2.38 / 1.70 / 2.0 GHz respectively, tools/probes/mfma_power_probe_bf16.hip).  bench.py prices the bf16 convolutions against it next
// to the data-sheet peak.  The caller times the launch (HIP events on `stream`); FLOP = workgroups * 8 * iters * 16 * 32768.
__global__ __launch_bounds__(512) void mfma_pipe_probe_bf16_kernel(const uint4* __restrict__ src, float* sink, long iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, src[(t * 8 + i) & 65535]);
        b[i] = __builtin_bit_cast(bf16x8, src[(t * 8 + 4 + i) & 65535]);
    }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (long it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[k], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[(k + 1) & 3], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(k + 1) & 3], b[k], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(k + 2) & 3], b[(k + 3) & 3], c3, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 123.456f) sink[0] = s;
}

extern "C" int rpg_probe_mfma_bf16(const void* operands, long iters, int workgroups, float* sink, void* stream) {
    if (!operands || !sink || iters <= 0 || iters > (1L << 24) || workgroups <= 0 || workgroups > 65536 || !rpg::aligned16(operands))
        return RPG_ERR_BAD_ARG;
    hipLaunchKernelGGL(mfma_pipe_probe_bf16_kernel, dim3(workgroups), dim3(512), 0, rpg::as_stream(stream),
                       reinterpret_cast<const uint4*>(operands), sink, iters);
    RPG_CHECK_LAUNCH("probe_mfma_bf16");
    return RPG_OK;
}
