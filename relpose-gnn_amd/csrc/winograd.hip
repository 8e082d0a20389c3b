// 3x3 / stride 1 / pad 1 convolution as a 1-D Winograd F(4,3) along the image width, on f32 MFMA (gfx950).
//
// Why: the f32 matrix pipe (v_mfma_f32_32x32x2_f32) is the roofline of the ResNet34 encoder; F(4,3) produces 4 output
// pixels of a row from 6 input pixels with 6 multiplies per (kernel row, channel) instead of 12, i.e. HALF the MFMA
// work, and stays pure fp32 (the transforms are small fp32 linear combinations; measured error ~2e-6 per layer).
// 29 of the 36 convolutions of ResNet34 (all 3x3 stride-1 ones) take this path.
//
//   y[4t+i] = sum_xi AT[i][xi] * M[xi],   M[xi][tile][cout] = sum_{kh,c} V[xi][tile][kh,c] * U[xi][cout][kh,c]
//   V[xi] = sum_j BT[xi][j] * d[j]   (d = the 6 input pixels 4t-1 .. 4t+4 of row ho+kh-1, zero outside the image)
//   U[xi] = sum_j G[xi][j] * w[cout][kh][j][c]   (precomputed once per weight load by wino43_weights_kernel)
//
// So one convolution = 6 independent GEMMs  [tiles x 3*Cin] * [3*Cin x Cout]  whose accumulators a lane combines
// in registers at the end (the output transform is lane-local: the MFMA C layout puts the same (tile, cout) element
// of all 6 products in the same lane); a wave owns 32 tiles x 32 channels x 6 positions = 6 accumulators of 32x32.
// The kernels share the addressing, the LDS layout (64-byte rows of four 16-byte chunks, XOR-swizzled by row) and
// the epilogue (output transform, LDS transpose, BatchNorm / residual / ReLU through raw buffer accesses):
//   * wino43_conv8_kernel: 8 waves on 128 tiles x 64 channels, two LDS images of a 16-wide K step, one barrier per step,
//     every load / LDS access placed singly behind an MFMA; its first blocks are the split-K variant for the tiles beyond
//     the last full round of CUs, finished by wino43_fixup_kernel;
//   * wino43_conv8p_kernel (the one that matters: every launch with more tiles than CUs): the same K step in a persistent
//     workgroup that walks its tiles with the load pipeline running across them;
//   * wino43_conv_kernel: 4 waves on 64 tiles x 64 channels, one LDS image, two workgroups per CU: small grids;
// (A short-K variant with two 4-wave workgroups per CU, K step 8, was built, measured and removed in round 3: its K loop was
// 6 % slower and ate what the overlap of one workgroup's prologue / epilogue with the other's MFMAs won)
// The design rules come from tools/probes/mfma_shadow_probe.hip: VALU time does not hide behind f32 MFMAs on gfx950.
//
// Reference op replaced: nn.Conv2d(3x3, stride 1, pad 1) + nn.BatchNorm2d (eval) (+ identity) + ReLU of a torchvision
// BasicBlock, reached from /root/reference/python/niantic/modules/posenet.py:1037.
#include "rpg_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int NT = 256;
constexpr int BK = 16, LD = BK;          // K step = LDS pitch (floats): no padding, 16-byte chunks XOR-swizzled by row
constexpr int P = 6;                     // Winograd positions
constexpr int BMT = 64, BN = 64;         // tiles x output channels per workgroup
constexpr int A_FLOATS = P * BMT * LD, B_FLOATS = P * BN * LD;
constexpr int LDS_BYTES = (A_FLOATS + B_FLOATS) * (int)sizeof(float);     // 49,152

struct Epi {
    const float* scale;
    const float* shift;
    const float* residual;
    float* out;
    int relu;
};

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 lin(float a, const float4& x, float b, const float4& y) {
    return make_float4(a * x.x + b * y.x, a * x.y + b * y.y, a * x.z + b * y.z, a * x.w + b * y.w);
}
__device__ __forceinline__ float4 add(const float4& x, const float4& y) {
    return make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}
__device__ __forceinline__ float4 sub(const float4& x, const float4& y) {
    return make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
}

// Epilogue of one wave: output transform of its 6 accumulators (lane-local: the MFMA C layout puts the same (tile,
// cout) element of all 6 products in the same lane), then an LDS transpose to 16-byte row segments and the fused
// BatchNorm / residual / ReLU / store, two of the four pixel columns at a time.  mw0 = first tile of the wave's 32,
// nw0 = first output channel of its 32 (both WAVE-UNIFORM and passed as scalars: the caller derives them from
// readfirstlane(wave id), which keeps the two buffer resources in SGPRs -- with a lane-derived wave id the compiler
// wrapped every one of the 32 buffer accesses in a waterfall loop); slab = 64 x 36 floats of LDS private to the wave.
// Nothing overlaps an epilogue when a CU holds one workgroup, so it is written for latency and instruction count (it is
// issue-bound: ~1500 instructions per wave before this version):
//   * the 16 row addresses of a lane come from ONE integer division (the rest is incremental);
//   * residual reads and output stores are raw buffer accesses based at the wave's first image row (invalid rows /
//     columns / channels carry an out-of-range offset: reads return zero, stores are dropped), hence branch-free;
//   * all 16 residual reads (both halves) are issued before the output transform: one exposed round trip, hidden
//     behind the transform's arithmetic, instead of one per half;
//   * the output transform works on register PAIRS (elements e, e+1 of an accumulator = tiles t, t+1 of the same
//     channel) with packed f32 instructions and shares the sums / differences m1+-m2, m3+-m4 between the four outputs:
//     11 packed instructions per pair instead of ~30 scalar ones.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wino43_epilogue(const f32x16 (&acc)[P], float* slab, int lane, int mw0, int nw0, int M,
                                                int Tw, int W, int Cout, const Epi& ep) {
    constexpr int EP = 32 + 4;                              // slab pitch: 32 channels + pad
    constexpr unsigned OOB = 0x80000000u;
    const int c4 = lane & 7, pr = lane >> 3;                // 8 lanes cover a row's 32 channels; 8 rows per pass
    const int nb = nw0 + 4 * c4;
    const bool n_ok = nb < Cout;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = f4zero();
    if (n_ok && ep.scale) sc = *reinterpret_cast<const float4*>(ep.scale + nb);
    if (n_ok && ep.shift) sh = *reinterpret_cast<const float4*>(ep.shift + nb);

    // this lane stores pixel column 4*tw + (pr & 1) [+ 2 in the second half] of tiles mw0 + (pr >> 1) + 4*it, it = 0..7
    const int t_first = mw0 / Tw;                           // wave-uniform: first image row (n*H + ho) of the wave
    const size_t row0 = (size_t)t_first * W * Cout;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(ep.out + row0, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(ep.residual ? ep.residual + row0 : ep.out + row0), 0, 0x7fffffff, 0x00020000);
    const int step_t = 4 / Tw, step_tw = 4 % Tw;            // wave-uniform
    int mt = mw0 + (pr >> 1);
    int trel = mt / Tw - t_first, tw = mt % Tw;
    unsigned voff[2][8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int wo = 4 * tw + (pr & 1);
        const bool ok = n_ok && mt < M;
        const unsigned o = 4u * (unsigned)((trel * W + wo) * Cout + nb);
        voff[0][it] = ok && wo < W ? o : OOB;
        voff[1][it] = ok && wo + 2 < W ? o + 8u * (unsigned)Cout : OOB;
        mt += 4;
        trel += step_t;
        tw += step_tw;
        if (tw >= Tw) { tw -= Tw; ++trel; }
    }
    float4 rs[2][8];
    if (ep.residual) {
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int it = 0; it < 8; ++it)
                rs[half][it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, voff[half][it], 0, 0));
    } else {
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int it = 0; it < 8; ++it) rs[half][it] = f4zero();
    }
    const float floor_v = ep.relu ? 0.f : -INFINITY;        // ReLU without a branch per store
    // output transform y = AT m (AT of F(4,3), points 0, +-1, +-2, inf) on pairs of elements
    f32x2 y[4][8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = 2 * q;
        const f32x2 m0 = {acc[0][e], acc[0][e + 1]}, m1 = {acc[1][e], acc[1][e + 1]}, m2 = {acc[2][e], acc[2][e + 1]};
        const f32x2 m3 = {acc[3][e], acc[3][e + 1]}, m4 = {acc[4][e], acc[4][e + 1]}, m5 = {acc[5][e], acc[5][e + 1]};
        const f32x2 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
        const f32x2 u = d34 + d34;                          // 2 (m3 - m4), exact
        y[0][q] = (m0 + s12) + s34;                         // y0 = m0 + m1 + m2 + m3 + m4
        y[1][q] = d12 + u;                                  // y1 = (m1 - m2) + 2 (m3 - m4)
        y[2][q] = s12 + 4.f * s34;                          // y2 = (m1 + m2) + 4 (m3 + m4)
        y[3][q] = (d12 + 4.f * u) + m5;                     // y3 = (m1 - m2) + 8 (m3 - m4) + m5
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = 2 * q;
            const int trow = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);  // tile within the wave's 32 (element e; e+1: +1)
            float* sp = slab + (2 * trow) * EP + (lane & 31);
            sp[0] = y[2 * half][q].x;
            sp[EP] = y[2 * half + 1][q].x;
            sp[2 * EP] = y[2 * half][q].y;
            sp[3 * EP] = y[2 * half + 1][q].y;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int prow = pr + 8 * it;                               // 0..63 = (tile, pixel-in-half)
            const float4 v = *reinterpret_cast<const float4*>(&slab[prow * EP + 4 * c4]);
            const f32x2 r0 = {rs[half][it].x, rs[half][it].y}, r1 = {rs[half][it].z, rs[half][it].w};
            f32x2 o0 = f32x2{v.x, v.y} * f32x2{sc.x, sc.y} + f32x2{sh.x, sh.y} + r0;
            f32x2 o1 = f32x2{v.z, v.w} * f32x2{sc.z, sc.w} + f32x2{sh.z, sh.w} + r1;
            const float4 yv = make_float4(fmaxf(o0.x, floor_v), fmaxf(o0.y, floor_v), fmaxf(o1.x, floor_v), fmaxf(o1.y, floor_v));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, yv), ro, voff[half][it], 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// The same epilogue with a smaller register footprint, for the persistent kernel (which keeps the NEXT tile's fetch state
// and in-flight loads alive across it): one half (two of the four pixel columns) at a time -- residual reads of a half
// are issued before that half's output transform, those of the second half as soon as the first half's transform
// registers are free -- and the second half's offsets are rebuilt from the first half's + a validity mask.  mid() runs
// after the second output transform, when the accumulators are dead (the persistent kernel re-issues the next tile's
// K-step-1 loads there).
template <bool RES, class Mid>
__device__ __forceinline__ void wino43_epilogue_lean_body(const f32x16 (&acc)[P], float* slab, int lane, int mw0, int nw0, int M,
                                                          int Tw, int W, int Cout, const Epi& ep, Mid mid) {
    constexpr int EP = 32 + 4;
    constexpr unsigned OOB = 0x80000000u;
    const int c4 = lane & 7, pr = lane >> 3;
    const int nb = nw0 + 4 * c4;
    const bool n_ok = nb < Cout;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = f4zero();
    if (n_ok && ep.scale) sc = *reinterpret_cast<const float4*>(ep.scale + nb);
    if (n_ok && ep.shift) sh = *reinterpret_cast<const float4*>(ep.shift + nb);
    const int t_first = mw0 / Tw;
    const size_t row0 = (size_t)t_first * W * Cout;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(ep.out + row0, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(ep.residual ? ep.residual + row0 : ep.out + row0), 0, 0x7fffffff, 0x00020000);
    const int step_t = 4 / Tw, step_tw = 4 % Tw;
    int mt = mw0 + (pr >> 1);
    int trel = mt / Tw - t_first, tw = mt % Tw;
    unsigned voff[8], ok2 = 0;             // first-half offsets; bit it of ok2: the second half's pixel column exists
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int wo = 4 * tw + (pr & 1);
        const bool ok = n_ok && mt < M;
        voff[it] = ok && wo < W ? 4u * (unsigned)((trel * W + wo) * Cout + nb) : OOB;
        if (ok && wo + 2 < W) ok2 |= 1u << it;
        mt += 4;
        trel += step_t;
        tw += step_tw;
        if (tw >= Tw) { tw -= Tw; ++trel; }
    }
    const unsigned half_b = 8u * (unsigned)Cout;
    auto off = [&](int half, int it) -> unsigned {
        return half == 0 ? voff[it] : ((ok2 >> it) & 1u) ? voff[it] + half_b : OOB;
    };
    int relu_i = ep.relu;                               // opaque: hoisted out of the caller's tile loop, the select below
    asm volatile("" : "+s"(relu_i));                     // became a VGPR that was spilled around the K loop
    const float floor_v = relu_i ? 0.f : -INFINITY;
    float4 rs[2][8];
    auto load_res = [&](int half) {
        if (RES) {
#pragma unroll
            for (int it = 0; it < 8; ++it)
                rs[half][it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, off(half, it), 0, 0));
        }
    };
    auto transform_to_slab = [&](int half) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = 2 * q;
            const f32x2 m1 = {acc[1][e], acc[1][e + 1]}, m2 = {acc[2][e], acc[2][e + 1]};
            const f32x2 m3 = {acc[3][e], acc[3][e + 1]}, m4 = {acc[4][e], acc[4][e + 1]};
            f32x2 ya, yb;
            if (half == 0) {
                const f32x2 m0 = {acc[0][e], acc[0][e + 1]};
                const f32x2 d34 = m3 - m4;
                ya = (m0 + (m1 + m2)) + (m3 + m4);
                yb = (m1 - m2) + (d34 + d34);
            } else {
                const f32x2 m5 = {acc[5][e], acc[5][e + 1]};
                const f32x2 d34 = m3 - m4, u = d34 + d34;
                ya = (m1 + m2) + 4.f * (m3 + m4);
                yb = ((m1 - m2) + 4.f * u) + m5;
            }
            const int trow = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            float* sp = slab + (2 * trow) * EP + (lane & 31);
            sp[0] = ya.x;
            sp[EP] = yb.x;
            sp[2 * EP] = ya.y;
            sp[3 * EP] = yb.y;
        }
    };
    auto finish = [&](int half) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int prow = pr + 8 * it;
            const float4 v = *reinterpret_cast<const float4*>(&slab[prow * EP + 4 * c4]);
            f32x2 o0 = f32x2{v.x, v.y} * f32x2{sc.x, sc.y} + f32x2{sh.x, sh.y};
            f32x2 o1 = f32x2{v.z, v.w} * f32x2{sc.z, sc.w} + f32x2{sh.z, sh.w};
            if (RES) {                     // (without a residual nothing waits on memory here: zero-filling rs[] instead made
                o0 += f32x2{rs[half][it].x, rs[half][it].y};       // the compiler wait for ALL outstanding stores and loads
                o1 += f32x2{rs[half][it].z, rs[half][it].w};       // before overwriting registers they might still target)
            }
            const float4 yv = make_float4(fmaxf(o0.x, floor_v), fmaxf(o0.y, floor_v), fmaxf(o1.x, floor_v), fmaxf(o1.y, floor_v));
#ifdef RPG_ABL_NOSTORE                     // ablation (tools/probes/wino_ablate.sh): only a value that never occurs is stored
            if (yv.x == 1234.5678f)
#endif
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, yv), ro, off(half, it), 0, 0);
        }
    };
    load_res(0);
    transform_to_slab(0);
    __builtin_amdgcn_wave_barrier();
    finish(0);
    load_res(1);
    __builtin_amdgcn_wave_barrier();
    transform_to_slab(1);
    mid();                                 // the caller's loads for its next tile: the accumulators are dead from here on
    __builtin_amdgcn_wave_barrier();
    finish(1);
    __builtin_amdgcn_wave_barrier();
}

template <class Mid>
__device__ __forceinline__ void wino43_epilogue_lean(const f32x16 (&acc)[P], float* slab, int lane, int mw0, int nw0, int M,
                                                     int Tw, int W, int Cout, const Epi& ep, Mid mid) {
    if (ep.residual) wino43_epilogue_lean_body<true>(acc, slab, lane, mw0, nw0, M, Tw, W, Cout, ep, mid);
    else wino43_epilogue_lean_body<false>(acc, slab, lane, mw0, nw0, M, Tw, W, Cout, ep, mid);
}

// Split-K variant: the wave's raw output-transformed tile goes to dst[512 px][64 ch] of its workgroup's slab
// (px = 4 * tile-in-workgroup + pixel column); no masking, the fix-up kernel knows what is valid.
struct NoMid { __device__ __forceinline__ void operator()() const {} };
template <class Mid = NoMid>
__device__ __forceinline__ void wino43_epilogue_partial(const f32x16 (&acc)[P], float* slab, int lane, int mw, int nw,
                                                        float* __restrict__ dst, Mid mid = Mid()) {
    constexpr int EP = 32 + 4;
    const int c4 = lane & 7, pr = lane >> 3;
    const __amdgpu_buffer_rsrc_t rd = rpg::agent_rsrc(dst);           // dst is workgroup-uniform
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float m1 = acc[1][e], m2 = acc[2][e], m3 = acc[3][e], m4 = acc[4][e];
            float ya, yb;
            if (half == 0) {
                ya = acc[0][e] + m1 + m2 + m3 + m4;
                yb = (m1 - m2) + 2.f * (m3 - m4);
            } else {
                ya = (m1 + m2) + 4.f * (m3 + m4);
                yb = (m1 - m2) + 8.f * (m3 - m4) + acc[5][e];
            }
            const int trow = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            slab[(2 * trow) * EP + (lane & 31)] = ya;
            slab[(2 * trow + 1) * EP + (lane & 31)] = yb;
        }
        if (half == 1) mid();                                           // the accumulators are dead from here on
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int prow = pr + 8 * it;                               // (tile, pixel-in-half)
            const float4 v = *reinterpret_cast<const float4*>(&slab[prow * EP + 4 * c4]);
            const int px = 4 * (mw + (prow >> 1)) + 2 * half + (prow & 1);
            // agent-scope store (written through the XCD's L2): another workgroup of this launch may add the slabs up
            rpg::agent_store_f4(rd, 4u * (unsigned)(px * BN + nw + 4 * c4), 0, v);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// x [n][H][W][Cin] -> y [n][H][W][Cout];  U [6][Cout][3][Cin];  M = n*H*Tw tiles, Tw = ceil(W/4)
__global__ __launch_bounds__(NT, 2) void wino43_conv_kernel(const float* __restrict__ x, const float* __restrict__ U,
                                                         int H, int W, int Cin, int Cout, int Tw, int M, Epi ep,
                                                         int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;                     // [P][BMT][LD]
    float* Bs = lds + A_FLOATS;          // [P][BN][LD]
    const int K = 3 * Cin;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3, q = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
    const int m0 = (tile / tiles_n) * BMT;
    const int n0 = (tile % tiles_n) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int row = tid >> 2, slot = tid & 3;          // this thread stages tile-row `row`, k-slot `slot`

    // ---- staging addresses.  All global reads are raw buffer loads: an invalid lane carries the offset 0x80000000
    // >= num_records and the hardware returns zeros, so image borders and ragged tiles cost neither a branch nor a
    // select on the data.  Everything that depends on the lane is K-invariant and lives in 7 VGPRs (va[6], vb); what
    // changes with the K step is wave-uniform and goes into the instruction's scalar offset:
    //   A: pixel (ho-1+kh, wi0+j), channels c0+4*slot..   = base' + va[j] + 4*(kh*W*Cin + c0)   (base' = image - one row)
    //   B: U[xi][n0+row][kh][c0+4*slot..]                  = U + vb + 4*(xi*Cout*K + kh*Cin + c0)
    // Only the validity of the image row (top / bottom border) depends on both lane and kh: one bit test + 6 selects.
    const int n_first = (m0 / Tw) / H;
    const size_t img_floats = (size_t)H * W * Cin;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(x + n_first * img_floats) - (size_t)W * Cin, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, 0x7fffffff, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int m = m0 + row;
    unsigned va[P];
    unsigned rowbits = 0;                 // bit kh set <=> image row ho-1+kh exists (and the tile itself does)
    {
        int img_off = 0, ho = 0, wi0 = -(1 << 24);
        if (m < M) {
            const int tw = m % Tw;
            const int t = m / Tw;
            ho = t % H;
            img_off = (t / H - n_first) * (int)img_floats;
            wi0 = 4 * tw - 1;
            rowbits = (ho > 0 ? 1u : 0u) | 2u | (ho < H - 1 ? 4u : 0u);
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int wi = wi0 + j;
            va[j] = (unsigned)wi < (unsigned)W ? 4u * (unsigned)(img_off + (ho * W + wi) * Cin + 4 * slot) : OOB;
        }
    }
    const int nrow = n0 + row;
    const unsigned vb = nrow < Cout ? 4u * (unsigned)(nrow * K + 4 * slot) : OOB;
    const unsigned ustride_b = (unsigned)Cout * K * 4u;
    const int nk = 3 * ((Cin + BK - 1) / BK);      // K steps: each kernel row is walked in steps of 16 channels
    int f_kt = 0, f_kh = 0, f_c0 = 0;     // the K step the next fetch() loads (wave-uniform)

    float4 d[P], ub[P];
    // fetch_one(i), i = 0..11: load i of the 12 of K step f_kt (A pixels 0..5, then U positions 0..5); fetch_next()
    // moves on to the following step.  Split like this so that the main loop can drop the loads one at a time between
    // MFMAs.
    auto fetch_one = [&](int i) {
        const bool cv = f_c0 + 4 * slot < Cin;            // channel tail of a kernel row when Cin % 16 != 0
        if (i < P) {
            const bool rv = ((rowbits >> f_kh) & 1u) && cv;
            const unsigned sa = 4u * (unsigned)(f_kh * W * Cin + f_c0);
            d[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra, rv ? va[i] : OOB, sa, 0));
        } else {
            const unsigned vbe = (f_kt < nk && cv) ? vb : OOB;
            const unsigned sb = 4u * (unsigned)(f_kh * Cin + f_c0) + (unsigned)(i - P) * ustride_b;
            ub[i - P] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rb, vbe, sb, 0));
        }
    };
    auto fetch_next = [&]() {
        ++f_kt;
        f_c0 += BK;
        if (f_c0 >= Cin) { f_c0 = 0; ++f_kh; }
    };
    auto stage = [&]() {
        // input transform V = BT d  (BT of F(4,3), interpolation points 0, +-1, +-2, inf)
        const float4 t0 = lin(4.f, d[0], -5.f, d[2]);                  // 4 d0 - 5 d2
        const float4 v0 = add(t0, d[4]);
        const float4 s12 = lin(-4.f, d[2], 1.f, d[4]);                 // -4 d2 + d4
        const float4 q12 = lin(4.f, d[1], -1.f, d[3]);                 //  4 d1 - d3
        const float4 v1 = sub(s12, q12);
        const float4 v2 = add(s12, q12);
        const float4 s34 = sub(d[4], d[2]);                            // d4 - d2
        const float4 q34 = lin(2.f, d[3], -2.f, d[1]);                 // 2 d3 - 2 d1
        const float4 v3 = add(s34, q34);
        const float4 v4 = sub(s34, q34);
        const float4 v5 = add(lin(4.f, d[1], -5.f, d[3]), d[5]);
        // chunk (row, slot) lives at slot ^ ((row >> 2) & 3): with a 64-byte pitch this makes both the 8-lane groups
        // of these ds_write_b128 (2 rows x 4 chunks = 32 distinct banks) and the 16-lane groups of the ds_read_b128
        // below (rows {0-3,12-15,20-27} / {4-11,16-19,28-31} of one logical chunk) conflict-free
        const int sw = 4 * (slot ^ ((row >> 2) & 3));
        float* ap = As + row * LD + sw;
        *reinterpret_cast<float4*>(ap + 0 * BMT * LD) = v0;
        *reinterpret_cast<float4*>(ap + 1 * BMT * LD) = v1;
        *reinterpret_cast<float4*>(ap + 2 * BMT * LD) = v2;
        *reinterpret_cast<float4*>(ap + 3 * BMT * LD) = v3;
        *reinterpret_cast<float4*>(ap + 4 * BMT * LD) = v4;
        *reinterpret_cast<float4*>(ap + 5 * BMT * LD) = v5;
        float* bp = Bs + row * LD + sw;
#pragma unroll
        for (int xi = 0; xi < P; ++xi) *reinterpret_cast<float4*>(bp + xi * BN * LD) = ub[xi];
    };

    f32x16 acc[P];
#pragma unroll
    for (int xi = 0; xi < P; ++xi)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[xi][e] = 0.f;

    const int rsw = 4 * ((lane >> 5) ^ ((lane >> 2) & 3));            // logical chunk (lane>>5) [+2 for kb = 8: ^ 8 floats]
    const int a_off = (wm * 32 + (lane & 31)) * LD + rsw;
    const int b_off = (wn * 32 + (lane & 31)) * LD + rsw;
#pragma unroll
    for (int i = 0; i < 2 * P; ++i) fetch_one(i);
    fetch_next();
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                 // every wave is done reading the previous step's LDS image
        stage();
        __syncthreads();
        // 48 MFMAs in 6 groups of 8 = (k half, position pair), consecutive MFMAs alternating between two accumulators.
        // A wave issues in order, so whatever it issues in a clump is time it cannot issue MFMAs: the 16 ds_read_b128
        // of the next group's operands and the 12 buffer loads of K step kt+1 (all-invalid after the last step) are
        // therefore placed one at a time behind individual MFMAs, and scheduling barriers pin that placement.
        float4 fa[2][2], fb[2][2];
        auto frag_one = [&](int g, int set, int i) {          // i = 0..3: a[0], b[0], a[1], b[1]
            const int kb = (g / 3) * 8, xi = 2 * (g % 3) + (i >> 1);
            if (i & 1) fb[set][i >> 1] = *reinterpret_cast<const float4*>(&Bs[xi * BN * LD + (b_off ^ kb)]);
            else       fa[set][i >> 1] = *reinterpret_cast<const float4*>(&As[xi * BMT * LD + (a_off ^ kb)]);
        };
#pragma unroll
        for (int i = 0; i < 4; ++i) frag_one(0, 0, i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int set = g & 1, x0 = 2 * (g % 3);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = i & 1, e = i >> 1;
                const float av = e == 0 ? fa[set][j].x : e == 1 ? fa[set][j].y : e == 2 ? fa[set][j].z : fa[set][j].w;
                const float bv = e == 0 ? fb[set][j].x : e == 1 ? fb[set][j].y : e == 2 ? fb[set][j].z : fb[set][j].w;
                acc[x0 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[x0 + j], 0, 0, 0);
                if (i < 4 && g + 1 < 6) frag_one(g + 1, set ^ 1, i);
                if (i >= 4 && i < 6) fetch_one(2 * g + (i - 4));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        fetch_next();
    }
    __syncthreads();                     // LDS becomes the epilogue slabs

    const int ws = __builtin_amdgcn_readfirstlane(wave);       // wave-uniform by construction: scalar addressing below
    wino43_epilogue(acc, lds + ws * (64 * 36), lane, m0 + (ws >> 1) * 32, n0 + (ws & 1) * 32, M, Tw, W, Cout, ep);
}

// ---------------------------------------------------------------------------------------------------------------
// The large-problem kernel: 8 waves on 128 tiles (= 512 output pixels) x 64 output channels, TWO LDS images of a K
// step (2 x 72 KB, one workgroup = two waves per SIMD per CU) and ONE barrier per K step.
//
// Measured on gfx950 (tools/probes/mfma_shadow_probe.hip): v_mfma_f32_32x32x2_f32 and ordinary VALU instructions of a
// SIMD do NOT overlap (each VALU op costs ~4.4 cycles of matrix-pipe time with two waves per SIMD, ~8 with one), while
// ds_read / ds_write / buffer_load issue mostly hides behind MFMAs, and a wave issues in order.  Hence:
//   * nothing runs in a separate phase: the 9 stage pieces of K step kt+1 (-> the other image), the 9 buffer loads
//     of K step kt+2 and the 24 operand reads of step kt are placed one at a time behind individual MFMAs
//     (scheduling barriers pin the placement), so a wave never stops issuing MFMAs except at the one barrier;
//   * VALU work per K step is cut to the input transform itself (24 packed FMAs / adds): all addressing that depends
//     on the lane is K-invariant (effective offsets are refreshed only when the kernel row or the channel tail
//     changes, a few times per kernel), the K step goes into the scalar offset of the buffer instruction, and the
//     K loop is unrolled by two so the image toggles at compile time.
// Staging: thread t stages A row t>>2 (128 tiles), k-slot t&3: 6 pixels -> 6 transformed chunks; and U row (t>>2)&63
// of positions 3*(t>>8) .. +2: 3 chunks.  Wave w owns tiles 32*(w>>1).., channels 32*(w&1)..
constexpr int NT8 = 512, BMT8 = 128;
constexpr int A8_FLOATS = P * BMT8 * LD, IMG8_FLOATS = A8_FLOATS + B_FLOATS;      // 12288 + 6144 floats = 72 KB
constexpr int LDS8_BYTES = 2 * IMG8_FLOATS * (int)sizeof(float);                  // 147,456

typedef float f32x2 __attribute__((ext_vector_type(2)));
struct F4 { f32x2 lo, hi; };             // a 16-byte chunk as two packed pairs
__device__ __forceinline__ F4 to_f4(const float4& v) { return F4{f32x2{v.x, v.y}, f32x2{v.z, v.w}}; }
__device__ __forceinline__ float4 to_float4(const F4& v) { return make_float4(v.lo.x, v.lo.y, v.hi.x, v.hi.y); }
// Packed f32 VALU written as instructions: beside MFMAs hipcc splits packed f32 arithmetic into scalar halves (twice
// the instructions), which is the wrong trade here because VALU time is not hidden behind the f32 MFMAs.  The scale
// is an inline constant (+-2, +-4) broadcast to both halves.
#define RPG_PK_FMA(NAME, CONST)                                                                               \
    __device__ __forceinline__ f32x2 NAME(f32x2 x, f32x2 y) {                                                 \
        f32x2 r;                                                                                              \
        asm("v_pk_fma_f32 %0, %1, " CONST ", %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(y));               \
        return r;                                                                                             \
    }
RPG_PK_FMA(pk_fma_p4, "4.0")
RPG_PK_FMA(pk_fma_m4, "-4.0")
RPG_PK_FMA(pk_fma_p2, "2.0")
RPG_PK_FMA(pk_fma_m2, "-2.0")
#undef RPG_PK_FMA
__device__ __forceinline__ f32x2 pk_add(f32x2 x, f32x2 y) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 x, f32x2 y) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
#define RPG_F4_OP(NAME, OP) \
    __device__ __forceinline__ F4 NAME(const F4& x, const F4& y) { return F4{OP(x.lo, y.lo), OP(x.hi, y.hi)}; }
RPG_F4_OP(fma4_p4, pk_fma_p4)   // 4 x + y
RPG_F4_OP(fma4_m4, pk_fma_m4)   // -4 x + y
RPG_F4_OP(fma4_p2, pk_fma_p2)
RPG_F4_OP(fma4_m2, pk_fma_m2)
RPG_F4_OP(add4, pk_add)
RPG_F4_OP(sub4, pk_sub)
#undef RPG_F4_OP

// Split-K tail: the tiles [tile_base, ..) that do not fill a whole round of CUs are cut along K into `parts` workgroups each,
// which store their raw partial output tile (after the output transform: it is linear) as [512 px][64 ch] in slab
// (tile - tile_base) * parts + part of `partial`; wino43_fixup_kernel adds the parts in k order and applies the epilogue.
// The n_split = (tiles - tile_base) * parts split workgroups are the FIRST blocks of the same launch as the whole tiles
// (round 2): as a launch of their own (round 1) they ran after the main kernel on a third of the CUs for ~25 us per
// convolution; now they are over before the first round of whole tiles ends and only the fix-up kernel follows.
// arrive (round 4): per tail tile arrival counters, zero between launches.  Non-null: the workgroup that stores the LAST of a
// tile's `parts` slabs adds them in k order and applies the epilogue itself (wino43_combine_last); null: wino43_fixup_kernel
// does it in a launch of its own.
struct Split { int tile_base, parts, n_split; float* partial; unsigned* arrive; };

#ifdef RPG_WINO_TRACE
// Timeline instrumentation (tools/probes/wino_trace.sh): per workgroup {HW_ID, s_memtime at entry, after the prologue, after
// the K loop, at exit} -> g_wino_trace[5 * blockIdx.x ..]
__device__ unsigned long long* g_wino_trace = nullptr;
#define RPG_TRACE(i)                                                                         \
    do {                                                                                     \
        if (g_wino_trace && threadIdx.x == 0) g_wino_trace[5 * blockIdx.x + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define RPG_TRACE(i) do {} while (0)
#endif

// BatchNorm / residual / ReLU / store of one 16-byte group of a summed tail tile: pixel p (0..511) x channel quad c4 (0..15)
__device__ __forceinline__ void wino43_finish_quad(const float4& s, const Epi& ep, int tile, int p, int c4, int M, int Tw, int W,
                                                   int Cout, int tiles_n) {
    const int m = (tile / tiles_n) * BMT8 + (p >> 2), nb = (tile % tiles_n) * BN + 4 * c4;
    if (m >= M || nb >= Cout) return;
    const int t = m / Tw, wo = 4 * (m - t * Tw) + (p & 3);
    if (wo >= W) return;
    const size_t o = ((size_t)t * W + wo) * Cout + nb;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = f4zero(), rs = f4zero();
    if (ep.scale) sc = *reinterpret_cast<const float4*>(ep.scale + nb);
    if (ep.shift) sh = *reinterpret_cast<const float4*>(ep.shift + nb);
    if (ep.residual) rs = *reinterpret_cast<const float4*>(ep.residual + o);
    float4 y;
    y.x = s.x * sc.x + sh.x + rs.x; y.y = s.y * sc.y + sh.y + rs.y;
    y.z = s.z * sc.z + sh.z + rs.z; y.w = s.w * sc.w + sh.w + rs.w;
    if (ep.relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
    *reinterpret_cast<float4*>(ep.out + o) = y;
}

// In-kernel combine of the split-K parts (round 4), called by every thread of an 8-wave workgroup right after it has stored
// partial slab `part` of tail tile tt (agent-scope stores, see rpg_common.h).  Thread 0 takes a ticket from the tile's arrival
// counter, and the workgroup that holds the last ticket adds all `parts` slabs in ascending k order (its own included, read
// back like the others: the bits do not depend on who came last), applies the epilogue and puts the counter back to zero --
// the work of wino43_fixup_kernel without the launch.  Nobody waits (no spinning: the workgroups of two streams share the
// CUs).  `flag`: one int of LDS that no wave is using.  8 of a thread's 16 groups at a time: 8 loads in flight per thread.
__device__ __forceinline__ void wino43_combine_last(const Split& sp, int tt, const Epi& ep, int M, int Tw, int W, int Cout,
                                                    int tiles_n, int* flag) {
    if (!rpg::last_arriver(sp.arrive + tt, (unsigned)sp.parts, flag)) return;
    const __amdgpu_buffer_rsrc_t rs = rpg::agent_rsrc(sp.partial + (size_t)tt * sp.parts * (BMT8 * 4 * BN));
    constexpr unsigned SLAB_B = BMT8 * 4 * BN * 4;
    const int tile = sp.tile_base + tt;
    const int tid_c = threadIdx.x;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        const unsigned vo = 16u * (unsigned)(tid_c + h * 8 * NT8);          // group idx = tid + NT8 * (8 h + c): px = idx >> 4, c4 = idx & 15
        float4 sum[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) sum[c] = rpg::agent_load_f4(rs, vo + 16u * c * NT8, 0);
        for (int i = 1; i < sp.parts; ++i) {
            float4 v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = rpg::agent_load_f4(rs, vo + 16u * c * NT8, (unsigned)i * SLAB_B);
#pragma unroll
            for (int c = 0; c < 8; ++c) { sum[c].x += v[c].x; sum[c].y += v[c].y; sum[c].z += v[c].z; sum[c].w += v[c].w; }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int idx = tid_c + NT8 * (8 * h + c);
            wino43_finish_quad(sum[c], ep, tile, idx >> 4, idx & 15, M, Tw, W, Cout, tiles_n);
        }
    }
}

__global__ __launch_bounds__(NT8) void wino43_conv8_kernel(const float* __restrict__ x, const float* __restrict__ U, int H,
                                                          int W, int Cin, int Cout, int Tw, int M, Epi ep, int tiles_n,
                                                          Split sp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef RPG_WINO_TRACE
    if (g_wino_trace && threadIdx.x == 0) g_wino_trace[5 * blockIdx.x] = __builtin_amdgcn_s_getreg(0xF804);   // HW_ID
#endif
    RPG_TRACE(1);
    const int K = 3 * Cin;
    // K is walked CHANNEL-BLOCK major, kernel row minor: step t covers channels 16 (t / 3) .. + 15 of kernel row t % 3.
    // The three kernel rows of a channel block read the SAME input rows (tile row r reads image rows r-1, r, r+1), so
    // with this order an input row's 64-byte segment is fetched from HBM once and re-read from L2 / L1 within three
    // consecutive K steps.  (Kernel-row major, the order of round 1, re-read it after Cin/16 steps, by when the resident
    // workgroups of an XCD (32 x 160 KB at layer 1) had pushed it out of the 4-MB L2: PMC showed 2.0x the algorithmic
    // read traffic, FETCH_SIZE calibrated with tools/probes/fetch_calib_probe.hip.)
    const int kpr = (Cin + BK - 1) / BK;               // channel blocks = K steps per kernel row
    int tile, kb = 0, nk = 3 * kpr;                    // this workgroup's tile and K-step range [kb, nk), multiples of 3
    const bool is_split = (int)blockIdx.x < sp.n_split;       // workgroup-uniform
    if (is_split) {
        const int tt = blockIdx.x / sp.parts, part = blockIdx.x - tt * sp.parts;
        tile = sp.tile_base + tt;
        kb = 3 * (part * kpr / sp.parts);
        nk = 3 * ((part + 1) * kpr / sp.parts);
    } else {
        const int nwg = gridDim.x - sp.n_split, bid = blockIdx.x - sp.n_split;
        const int xcd = bid & 7, loc = bid >> 3, q = nwg >> 3, r8 = nwg & 7;
        tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
    }
    const int m0 = (tile / tiles_n) * BMT8;
    const int n0 = (tile % tiles_n) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int row = tid >> 2, slot = tid & 3;          // A: tile-row `row` (0..127), k-slot `slot`
    const int brow = row & 63, bhalf = tid >> 8;       // B: channel-row brow, positions 3*bhalf .. 3*bhalf+2

    // ---- K-invariant staging addresses (see wino43_conv_kernel for the buffer-load conventions)
    const int n_first = (m0 / Tw) / H;
    const size_t img_floats = (size_t)H * W * Cin;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(x + n_first * img_floats) - (size_t)W * Cin, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, 0x7fffffff, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned va[P], va3[3][P];            // va3[kh]: offsets of the six pixels for kernel row kh, OOB where the row is outside
    unsigned rowbits = 0;                 // bit kh set <=> image row ho-1+kh exists (and the tile itself does)
    {
        const int m = m0 + row;
        int img_off = 0, ho = 0, wi0 = -(1 << 24);
        if (m < M) {
            const int tw = m % Tw;
            const int t = m / Tw;
            ho = t % H;
            img_off = (t / H - n_first) * (int)img_floats;
            wi0 = 4 * tw - 1;
            rowbits = (ho > 0 ? 1u : 0u) | 2u | (ho < H - 1 ? 4u : 0u);
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int wi = wi0 + j;
            va[j] = (unsigned)wi < (unsigned)W ? 4u * (unsigned)(img_off + (ho * W + wi) * Cin + 4 * slot) : OOB;
        }
    }
    const unsigned ustride_b = (unsigned)Cout * K * 4u;
    const unsigned vb = n0 + brow < Cout ? 4u * (unsigned)((n0 + brow) * K + 4 * slot) + 3u * bhalf * ustride_b : OOB;
    unsigned vb_eff;
    int f_kt = kb, f_c0 = (kb / 3) * BK;  // the K step / channel block the next fetches load (wave-uniform)
    const unsigned row_b = 4u * (unsigned)(W * Cin), krow_b = 4u * (unsigned)Cin;       // bytes per image row / per kernel row of U
    auto refresh = [&]() {                // effective offsets of channel block f_c0: row borders, channel tail, past the end
        asm volatile("" ::: "memory");    // keeps this a (rarely taken) branch
        const bool cv = f_kt < nk && f_c0 + 4 * slot < Cin;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const bool rv = cv && ((rowbits >> kh) & 1u);
#pragma unroll
            for (int j = 0; j < P; ++j) va3[kh][j] = rv ? va[j] : OOB;
        }
        vb_eff = cv ? vb : OOB;
    };
    refresh();

    float4 d[P], ub[3];
    // fetch_one(i, kh), i = 0..8: pixels 0..5 of kernel row kh (compile-time), then this thread's 3 U positions
    auto fetch_one = [&](int i, int kh) {
        if (i < P) {
            const unsigned sa = (unsigned)kh * row_b + 4u * (unsigned)f_c0;
            d[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra, va3[kh][i], sa, 0));
        } else {
            const unsigned sb = (unsigned)kh * krow_b + 4u * (unsigned)f_c0 + (unsigned)(i - P) * ustride_b;
            ub[i - P] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rb, vb_eff, sb, 0));
        }
    };
    // after the fetches of kernel row kh: next step; a new channel block starts after kernel row 2
    auto fetch_next = [&](int kh) {
        ++f_kt;
        if (kh == 2) {
            f_c0 += BK;
            if (f_c0 + BK > Cin || f_kt >= nk) refresh();       // rare: channel tail (Cin % 16 != 0) and past the end
        }
    };

    // stage_one(p, img), p = 0..8: piece p of the K step held in d / ub -> LDS image img (float offset).  p < 6: the
    // input transform V[p] = BT d (BT of F(4,3), points 0, +-1, +-2, inf) in 12 packed FMA / add per 4 channels;
    // p >= 6: U chunk p-6.  Chunk (row, slot) lives at slot ^ ((row >> 2) & 3): conflict-free ds_write_b128 and
    // ds_read_b128 with an unpadded 64-byte pitch.
    const int st_a = row * LD + 4 * (slot ^ ((row >> 2) & 3));
    const int st_b = A8_FLOATS + (3 * bhalf) * BN * LD + brow * LD + 4 * (slot ^ ((brow >> 2) & 3));
    auto stage_one = [&](int p, int img) {
        if (p >= P) {
            *reinterpret_cast<float4*>(lds + img + st_b + (p - P) * BN * LD) = ub[p - P];
            return;
        }
        // r = d4 - d2 and t = d3 - d1 are shared by four of the six outputs
        const F4 d0 = to_f4(d[0]), d1 = to_f4(d[1]), d2 = to_f4(d[2]), d3 = to_f4(d[3]), d4 = to_f4(d[4]), d5 = to_f4(d[5]);
        F4 v;
        if (p == 0) v = fma4_p4(sub4(d0, d2), sub4(d4, d2));                   // 4 d0 - 5 d2 + d4 = 4 (d0 - d2) + r
        else if (p == 5) v = fma4_m4(sub4(d3, d1), sub4(d5, d3));              // 4 d1 - 5 d3 + d5 = -4 t + (d5 - d3)
        else if (p <= 2) {
            const F4 sx = fma4_m4(d2, d4);                                     // d4 - 4 d2
            const F4 tx = fma4_m4(d1, d3);                                     // d3 - 4 d1
            v = p == 1 ? add4(sx, tx) : sub4(sx, tx);
        } else {
            const F4 r = sub4(d4, d2), t = sub4(d3, d1);
            v = p == 3 ? fma4_p2(t, r) : fma4_m2(t, r);                        // r +- 2 t
        }
        *reinterpret_cast<float4*>(lds + img + st_a + p * BMT8 * LD) = to_float4(v);
    };

    f32x16 acc[P];
#pragma unroll
    for (int xi = 0; xi < P; ++xi)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[xi][e] = 0.f;

    const int rsw = 4 * ((lane >> 5) ^ ((lane >> 2) & 3));            // logical chunk (lane>>5) [+2 for kb = 8: ^ 8 floats]
    const int a_off = (wm * 32 + (lane & 31)) * LD + rsw;
    const int b_off = A8_FLOATS + (wn * 32 + (lane & 31)) * LD + rsw;
    float4 fa[2][2], fb[2][2];
    auto frag_one = [&](int g, int set, int i, int img) {             // i = 0..3: a[0], b[0], a[1], b[1]
        const int kb = (g / 3) * 8, xi = 2 * (g % 3) + (i >> 1);
        if (i & 1) fb[set][i >> 1] = *reinterpret_cast<const float4*>(&lds[img + xi * BN * LD + (b_off ^ kb)]);
        else       fa[set][i >> 1] = *reinterpret_cast<const float4*>(&lds[img + xi * BMT8 * LD + (a_off ^ kb)]);
    };
    // one K step on image `cur` (compile-time float offset), staging step kt+1 into `nxt`, fetching step kt+2 (kernel
    // row khf, compile time):
    //   groups 0-5, MFMAs 0-3: the 4 operand reads of the next group (group 5: of the next K step's group 0, from nxt)
    //   groups 0-2, MFMAs 4-7: stage pieces 0..8 -> nxt        groups 3-5, MFMAs 4-7: buffer loads 0..8
    //   barrier after group 4: every wave has written nxt and issued its last reads of cur
    auto kstep = [&](int cur, int nxt, int khf) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int set = g & 1, x0 = 2 * (g % 3);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = i & 1, e = i >> 1;
                const float av = e == 0 ? fa[set][j].x : e == 1 ? fa[set][j].y : e == 2 ? fa[set][j].z : fa[set][j].w;
                const float bv = e == 0 ? fb[set][j].x : e == 1 ? fb[set][j].y : e == 2 ? fb[set][j].z : fb[set][j].w;
                acc[x0 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[x0 + j], 0, 0, 0);
                if (i < 4) {
                    if (g < 5) frag_one(g + 1, set ^ 1, i, cur);
                    else frag_one(0, 0, i, nxt);
                } else {
                    const int s = 4 * (g % 3) + (i - 4);              // 0..11, pieces 0..8 used
                    if (s < 9) {
                        if (g < 3) stage_one(s, nxt);
                        else fetch_one(s, khf);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g == 4) {
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        fetch_next(khf);
    };

    // local step s = kt - kb (kb is a multiple of 3) runs kernel row s % 3 on image s % 2 and fetches kernel row (s+2) % 3
#pragma unroll
    for (int i = 0; i < 9; ++i) fetch_one(i, 0);
    fetch_next(0);
#pragma unroll
    for (int p = 0; p < 9; ++p) stage_one(p, 0);
#pragma unroll
    for (int i = 0; i < 9; ++i) fetch_one(i, 1);
    fetch_next(1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) frag_one(0, 0, i, 0);
    RPG_TRACE(2);
    int kt = kb;
    for (; kt + 5 < nk; kt += 6) {
        kstep(0, IMG8_FLOATS, 2);
        kstep(IMG8_FLOATS, 0, 0);
        kstep(0, IMG8_FLOATS, 1);
        kstep(IMG8_FLOATS, 0, 2);
        kstep(0, IMG8_FLOATS, 0);
        kstep(IMG8_FLOATS, 0, 1);
    }
    if (kt < nk) {                       // (nk - kb) % 6 == 3: three more steps
        kstep(0, IMG8_FLOATS, 2);
        kstep(IMG8_FLOATS, 0, 0);
        kstep(0, IMG8_FLOATS, 1);
    }
    __syncthreads();                     // LDS becomes the epilogue slabs
    RPG_TRACE(3);
    const int ws = __builtin_amdgcn_readfirstlane(wave);       // wave-uniform by construction: scalar addressing below
    if (is_split) {
        wino43_epilogue_partial(acc, lds + ws * (64 * 36), lane, (ws >> 1) * 32, (ws & 1) * 32,
                                sp.partial + (size_t)blockIdx.x * (BMT8 * 4 * BN));
        if (sp.arrive) wino43_combine_last(sp, tile - sp.tile_base, ep, M, Tw, W, Cout, tiles_n, reinterpret_cast<int*>(lds));
    } else
        wino43_epilogue(acc, lds + ws * (64 * 36), lane, m0 + (ws >> 1) * 32, n0 + (ws & 1) * 32, M, Tw, W, Cout, ep);
#ifdef RPG_WINO_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    RPG_TRACE(4);
}

// ---------------------------------------------------------------------------------------------------------------
// The persistent form of the 8-wave kernel (round 2): one workgroup per CU walks a list of work items -- item =
// blockIdx.x + i * gridDim.x over [split-K parts | whole tiles], same item -> tile mapping as above -- and the load /
// stage pipeline runs ACROSS items: during the last two K steps of an item the buffer loads already fetch steps 0 and 1 of
// the next item, the last step stages the next item's step 0 into the other LDS image, and the epilogue uses the image the
// last step read for its slabs.  What that removes per tile: the cold prologue (address set-up, one exposed HBM round
// trip, first stage + barrier: 6.5-10.7 k cycles, profiles/r1_wino43_phase_cycles.txt) and the gap between a workgroup's
// exit and its successor's first instruction on the CU (one 144-KB-LDS workgroup per CU: the dispatcher cannot overlap
// them; SQ_BUSY_CU_CYCLES showed CUs without any wave 10-11 % of a layer-1/2 launch against 5 % on the one-round layers
// 3-4).  What stays: the address set-up of the next item (inside the rarely taken branch of fetch_next, ~1.5 k cycles of
// VALU) and the epilogue (its registers are the reason for wino43_epilogue_lean: the next item's 9 loads in flight and its
// offsets stay live across it).  Requirements (else the launcher takes the kernel above): Cin % 16 == 0 -- every item is
// then a whole number of channel blocks (3 K steps), so the kernel-row phase is a compile-time constant across items --
// and H >= 2 (see fix_row1).
__global__ __launch_bounds__(NT8) void wino43_conv8p_kernel(const float* __restrict__ x, const float* __restrict__ U, int H,
                                                           int W, int Cin, int Cout, int Tw, int M, Epi ep, int tiles_n,
                                                           Split sp, int n_items) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef RPG_ABL_STAGGER                     // diagnostic (tools/probes/wino_ablate.sh): start the workgroups out of phase
    {
        const unsigned long long t0 = __builtin_readcyclecounter();
        const unsigned long long wait = (unsigned long long)((blockIdx.x >> 3) & 7) * RPG_ABL_STAGGER;
        while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
#endif
    const int K = 3 * Cin;
    const int kpr = Cin / BK;                          // channel blocks per kernel row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int row = tid >> 2, slot = tid & 3;
    const int brow = row & 63, bhalf = tid >> 8;
    const size_t img_floats = (size_t)H * W * Cin;
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, 0x7fffffff, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const unsigned ustride_b = (unsigned)Cout * K * 4u;
    const unsigned row_b = 4u * (unsigned)(W * Cin), krow_b = 4u * (unsigned)Cin;

    // ---- fetch-side state: the item whose K steps the buffer loads are walking (up to two steps ahead of the MFMAs)
    int f_item = blockIdx.x, f_m0 = 0, f_n0 = 0, f_kb = 0, f_nk = 0, f_kt = 0, f_c0 = 0;
    const float* f_base = x;
    unsigned va3[3][P], vb_eff;
    auto decode = [&]() {                 // fetch state of item f_item (everything out of range when there is none)
        asm volatile("" ::: "memory");
        int tid_d = tid;                  // opaque: keeps the lane-dependent part of this out of registers between calls
        asm volatile("" : "+v"(tid_d));
        const int row = tid_d >> 2, slot = tid_d & 3, brow = row & 63, bhalf = tid_d >> 8;
        const bool valid = f_item < n_items;
        int tile, kb = 0, nk = 3 * kpr;
        if (f_item < sp.n_split) {
            const int tt = f_item / sp.parts, part = f_item - tt * sp.parts;
            tile = sp.tile_base + tt;
            kb = 3 * (part * kpr / sp.parts);
            nk = 3 * ((part + 1) * kpr / sp.parts);
        } else {
            const int nwg = n_items - sp.n_split, bid = f_item - sp.n_split;
            const int xcd = bid & 7, loc = bid >> 3, q = nwg >> 3, r8 = nwg & 7;
            tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
        }
        f_m0 = (tile / tiles_n) * BMT8;
        f_n0 = (tile - (tile / tiles_n) * tiles_n) * BN;
        const int n_first = (f_m0 / Tw) / H;
        f_base = x + n_first * img_floats - (size_t)W * Cin;
        const int m = f_m0 + row;
        int img_off = 0, ho = 0, wi0 = -(1 << 24);
        unsigned rowbits = 0;
        if (valid && m < M) {
            const int t = m / Tw;
            const int tw = m - t * Tw;
            const int n = t / H;
            ho = t - n * H;
            img_off = (n - n_first) * (int)img_floats;
            wi0 = 4 * tw - 1;
            rowbits = (ho > 0 ? 1u : 0u) | 2u | (ho < H - 1 ? 4u : 0u);
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int wi = wi0 + j;
            const unsigned o = (unsigned)wi < (unsigned)W ? 4u * (unsigned)(img_off + (ho * W + wi) * Cin + 4 * slot) : OOB;
            // Kernel row 1 stays switched off until fix_row1(): the loads of the new item's K step 1 that the last K step
            // of the current item would issue are skipped that way (an out-of-range offset costs no memory access) -- their
            // registers are needed by the epilogue, which issues them itself (refetch) once the accumulators are dead.
            va3[0][j] = (rowbits & 1u) ? o : OOB;
            va3[1][j] = OOB;
            va3[2][j] = (rowbits & 4u) ? o : OOB;
        }
        vb_eff = valid && f_n0 + brow < Cout ? 4u * (unsigned)((f_n0 + brow) * K + 4 * slot) + 3u * bhalf * ustride_b : OOB;
        f_kb = kb;
        f_kt = kb;
        f_nk = valid ? nk : 0x3fffffff;   // no item: never advance again
        f_c0 = (kb / 3) * BK;
    };
    // row ho exists whenever the tile does, and H >= 2 (launcher) makes one of its neighbours exist too: its offsets are
    // the smaller (= valid, OOB is the largest unsigned value in use) of the two neighbours' offsets
    auto fix_row1 = [&]() {
#pragma unroll
        for (int j = 0; j < P; ++j) va3[1][j] = va3[0][j] < va3[2][j] ? va3[0][j] : va3[2][j];
    };
    decode();
    fix_row1();

    float4 d[P], ub[3];
    auto fetch_one = [&](int i, int kh) {
        if (i < P) {
            const __amdgpu_buffer_rsrc_t ra =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(f_base), 0, 0x7fffffff, 0x00020000);
            const unsigned sa = (unsigned)kh * row_b + 4u * (unsigned)f_c0;
            d[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra, va3[kh][i], sa, 0));
        } else {
            const unsigned sb = (unsigned)kh * krow_b + 4u * (unsigned)f_c0 + (unsigned)(i - P) * ustride_b;
            ub[i - P] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rb, vb_eff, sb, 0));
        }
    };
    auto fetch_next = [&](int kh) {
        ++f_kt;
        if (kh == 2) {
            f_c0 += BK;
            if (f_kt >= f_nk) {           // the item's last K step has been fetched: on to the next item of this workgroup
                f_item += gridDim.x;
                decode();
            }
        }
    };

    const int st_a = row * LD + 4 * (slot ^ ((row >> 2) & 3));
    const int st_b = A8_FLOATS + (3 * bhalf) * BN * LD + brow * LD + 4 * (slot ^ ((brow >> 2) & 3));
    // The two LDS images are addressed through registers (roles X / Y, not compile-time offsets): an item of an odd number
    // of channel blocks (3, 9, ... K steps) ends on image X, and the roles are swapped so that the next item again starts on X.
    // (offsets in 16-byte units into lds4: keeps the accesses ds_read_b128 / ds_write_b128 now that they are not constants)
    struct Img { int a[2], b[2], sa, sb; };         // operand reads of A / B at k half 0 / 1, stage writes of A / B
    float4* lds4 = reinterpret_cast<float4*>(lds);
    auto stage_one = [&](int p, const Img& I) {
        if (p >= P) {
            lds4[I.sb + (p - P) * (BN * LD / 4)] = ub[p - P];
            return;
        }
#ifdef RPG_ABL_NOXF                        // ablation: no input transform (wrong results, timing only)
        lds4[I.sa + p * (BMT8 * LD / 4)] = d[p];
        return;
#endif
        const F4 d0 = to_f4(d[0]), d1 = to_f4(d[1]), d2 = to_f4(d[2]), d3 = to_f4(d[3]), d4 = to_f4(d[4]), d5 = to_f4(d[5]);
        F4 v;
        if (p == 0) v = fma4_p4(sub4(d0, d2), sub4(d4, d2));
        else if (p == 5) v = fma4_m4(sub4(d3, d1), sub4(d5, d3));
        else if (p <= 2) {
            const F4 sx = fma4_m4(d2, d4);
            const F4 tx = fma4_m4(d1, d3);
            v = p == 1 ? add4(sx, tx) : sub4(sx, tx);
        } else {
            const F4 r = sub4(d4, d2), t = sub4(d3, d1);
            v = p == 3 ? fma4_p2(t, r) : fma4_m2(t, r);
        }
        lds4[I.sa + p * (BMT8 * LD / 4)] = to_float4(v);
    };

    f32x16 acc[P];
    const int rsw = 4 * ((lane >> 5) ^ ((lane >> 2) & 3));
    const int a_off = (wm * 32 + (lane & 31)) * LD + rsw;
    const int b_off = A8_FLOATS + (wn * 32 + (lane & 31)) * LD + rsw;
    float4 fa[2][2], fb[2][2];
    constexpr int IMG4 = IMG8_FLOATS / 4;
    Img X{{a_off >> 2, (a_off ^ 8) >> 2}, {b_off >> 2, (b_off ^ 8) >> 2}, st_a >> 2, st_b >> 2};
    Img Y{{X.a[0] + IMG4, X.a[1] + IMG4}, {X.b[0] + IMG4, X.b[1] + IMG4}, X.sa + IMG4, X.sb + IMG4};
    int ybase = IMG8_FLOATS;               // float offset of image Y (wave-uniform)
    auto frag_one = [&](int g, int set, int i, const Img& I) {
        const int xi = 2 * (g % 3) + (i >> 1);
        if (i & 1) fb[set][i >> 1] = lds4[I.b[g / 3] + xi * (BN * LD / 4)];
        else       fa[set][i >> 1] = lds4[I.a[g / 3] + xi * (BMT8 * LD / 4)];
    };
    auto kstep = [&](const Img& cur, const Img& nxt, int khf) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int set = g & 1, x0 = 2 * (g % 3);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = i & 1, e = i >> 1;
                const float av = e == 0 ? fa[set][j].x : e == 1 ? fa[set][j].y : e == 2 ? fa[set][j].z : fa[set][j].w;
                const float bv = e == 0 ? fb[set][j].x : e == 1 ? fb[set][j].y : e == 2 ? fb[set][j].z : fb[set][j].w;
                acc[x0 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[x0 + j], 0, 0, 0);
                if (i < 4) {
                    if (g < 5) frag_one(g + 1, set ^ 1, i, cur);
                    else frag_one(0, 0, i, nxt);
                } else {
                    const int s = 4 * (g % 3) + (i - 4);
                    if (s < 9) {
                        if (g < 3) stage_one(s, nxt);
                        else fetch_one(s, khf);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#ifndef RPG_ABL_NOBAR                      // ablation: no barrier (wrong results, timing only)
            if (g == 4) {
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        }
        fetch_next(khf);
    };

    // ---- the compute side starts on the first item: steps 0 and 1 the slow way
    int c_item = f_item, c_m0 = f_m0, c_n0 = f_n0, c_ns = f_nk - f_kb;
#pragma unroll
    for (int i = 0; i < 9; ++i) fetch_one(i, 0);
    fetch_next(0);
#pragma unroll
    for (int p = 0; p < 9; ++p) stage_one(p, X);
#pragma unroll
    for (int i = 0; i < 9; ++i) fetch_one(i, 1);
    fetch_next(1);
    const int ws = __builtin_amdgcn_readfirstlane(wave);
    for (;;) {
#ifdef RPG_WINO_TRACE
        if (g_wino_trace && threadIdx.x == 0) {
            g_wino_trace[5 * c_item] = __builtin_amdgcn_s_getreg(0xF804);
            g_wino_trace[5 * c_item + 1] = __builtin_readcyclecounter();
        }
#endif
#pragma unroll
        for (int xi = 0; xi < P; ++xi)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[xi][e] = 0.f;
        __syncthreads();                 // step 0 of this item is in image X (and the previous item's slabs are done with)
#pragma unroll
        for (int i = 0; i < 4; ++i) frag_one(0, 0, i, X);
#ifdef RPG_WINO_TRACE
        if (g_wino_trace && threadIdx.x == 0) g_wino_trace[5 * c_item + 2] = __builtin_readcyclecounter();
#endif
        int s = 0;
        for (; s + 6 <= c_ns; s += 6) {
            kstep(X, Y, 2);
            kstep(Y, X, 0);
            kstep(X, Y, 1);
            kstep(Y, X, 2);
            kstep(X, Y, 0);
            kstep(Y, X, 1);
        }
        if (s < c_ns) {                  // an odd number of channel blocks: three more steps, ending on X -> swap the roles
            kstep(X, Y, 2);
            kstep(Y, X, 0);
            kstep(X, Y, 1);
            const Img t = X;
            X = Y;
            Y = t;
            ybase = IMG8_FLOATS - ybase;
        }
        __syncthreads();                 // image Y (read by the last step) becomes the epilogue slabs; X holds the next item's step 0
        float* slab = lds + ybase + ws * (64 * 36);
#ifdef RPG_WINO_TRACE
        if (g_wino_trace && threadIdx.x == 0) g_wino_trace[5 * c_item + 3] = __builtin_readcyclecounter();
#endif
        // The loads of the next item's K step 1 were skipped by the last K step (kernel row 1 switched off, see decode) so
        // that the epilogue has their registers; they are issued after its second output transform, when the accumulator
        // registers are free, and land under the second half's stores, the barrier and the first MFMAs of the next item.
        auto refetch = [&]() {
            fix_row1();
#pragma unroll
            for (int i = 0; i < 9; ++i) fetch_one(i, 1);
        };
        // the epilogue's lane-dependent addressing is item-invariant; left to itself the compiler hoists all of it out of
        // the item loop and spills it around the K loop (82 VGPRs, reloaded from scratch on the critical path)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        // (No in-kernel combine of the split parts here, unlike wino43_conv8_kernel: this kernel sits at 256 VGPRs with no
        // scratch, and with wino43_combine_last inlined after the partial epilogue hipcc spilled five LDS-address registers
        // whose reloads -- each an s_waitcnt vmcnt(0) -- landed inside the K loop (round 4, -Rpass-analysis=kernel-resource-usage:
        // 20-44 bytes of scratch per lane).  Its launches are the many-tile ones, where the 6-us fix-up launch is 2 % of the kernel.)
        if (c_item < sp.n_split)
            wino43_epilogue_partial(acc, slab, lane_e, (ws >> 1) * 32, (ws & 1) * 32,
                                    sp.partial + (size_t)c_item * (BMT8 * 4 * BN), refetch);
        else
            wino43_epilogue_lean(acc, slab, lane_e, c_m0 + (ws >> 1) * 32, c_n0 + (ws & 1) * 32, M, Tw, W, Cout, ep, refetch);
#ifdef RPG_WINO_TRACE
        if (g_wino_trace && threadIdx.x == 0) g_wino_trace[5 * c_item + 4] = __builtin_readcyclecounter();
#endif
        if (f_item >= n_items) break;    // the fetch side is on the item after this one
        c_item = f_item;
        c_m0 = f_m0;
        c_n0 = f_n0;
        c_ns = f_nk - f_kb;
    }
}

// Sums the `parts` partial slabs of tail tile blockIdx.x / 32 in k order and applies BatchNorm / residual / ReLU (the separate
// fix-up launch: RPG_TUNE_INKERNEL_FIXUP = 0, and launches whose part count is above what one workgroup should add up).
__global__ __launch_bounds__(256) void wino43_fixup_kernel(const float* __restrict__ partial, Epi ep, int M, int Tw, int W,
                                                           int Cout, int tiles_n, Split sp) {
    const int tt = blockIdx.x >> 5, idx = (blockIdx.x & 31) * 256 + threadIdx.x;      // 512 px x 16 channel quads
    const float* src = partial + (size_t)tt * sp.parts * (BMT8 * 4 * BN) + 4 * idx;
    float4 s = *reinterpret_cast<const float4*>(src);
    for (int i = 1; i < sp.parts; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)i * (BMT8 * 4 * BN));
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    wino43_finish_quad(s, ep, sp.tile_base + tt, idx >> 4, idx & 15, M, Tw, W, Cout, tiles_n);
}

// U[xi][co][kh][c] = sum_j G[xi][j] * w[co][kh][j][c], evaluated in double and rounded once.
__global__ __launch_bounds__(NT) void wino43_weights_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout,
                                                            int Cin, long total) {
    const long i = (long)blockIdx.x * NT + threadIdx.x;       // over [co][kh][c]
    if (i >= total) return;
    const int c = (int)(i % Cin);
    const long t = i / Cin;
    const int kh = (int)(t % 3);
    const long co = t / 3;
    const float* wp = w + ((co * 3 + kh) * 3) * (long)Cin + c;
    const double g0 = wp[0], g1 = wp[Cin], g2 = wp[2 * (long)Cin];
    const double u[6] = {g0 / 4.0, -(g0 + g1 + g2) / 6.0, -(g0 - g1 + g2) / 6.0,
                         g0 / 24.0 + g1 / 12.0 + g2 / 6.0, g0 / 24.0 - g1 / 12.0 + g2 / 6.0, g2};
    const long plane = (long)Cout * 3 * Cin;
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) U[xi * plane + i] = (float)u[xi];
}

int g_wino_split = 1;                    // RPG_TUNE_WINO_SPLIT: split-K tail of the 8-wave kernel
int g_wino_split_steps = 3;              // ... and the least number of K steps a part of a tile gets in the one-workgroup-per-tile form (values >= 2 of the same key;
                                         // one 8-node graph, r3: 4 -> 1.63 ms per forward, 3 or 2 -> 1.50; 6 -> 1.61; the benched 32-graph launches do not get here:
                                         // their part count is bound by CUs / tail tiles)
int g_wino = 1;                          // RPG_TUNE_WINOGRAD: 0 off | 1 auto | 2 / 3: always the 4-wave / 8-wave kernel
int g_wino_min_blocks = 16;              // ... | n >= 16: auto, Winograd from n blocks of 64 tiles x 64 channels up (default 16; round 6: the per-layer rule of small batches)
int g_wino_persist = 1;                  // RPG_TUNE_WINO_PERSIST: the persistent 8-wave kernel when a launch has more tiles than CUs
int g_wino_combine_max = 8;              // most parts of a tail tile that its last-arriving workgroup adds up itself (one CU reads parts x 128 KB:
                                         // beyond this the 32-blocks-per-tile fix-up launch is the faster way); RPG_TUNE_INKERNEL_FIXUP >= 2 sets it

}  // namespace

#ifdef RPG_PROBE_WINO2D
#include "../../tools/probes/winograd2d.hip"      // the nested F(4x2, 3x3) experiment of round 4 (correct, slower): opt-in
#endif

namespace rpg {

#ifndef RPG_PROBE_WINO2D
void wino2d_set(int) {}                           // RPG_TUNE_WINO2D is accepted (0) and ignored in a build without the probe kernel
#endif
bool wino_enabled() { return g_wino != 0; }
void wino_set(int on) { if (on >= 16) { g_wino = 1; g_wino_min_blocks = on; } else { g_wino = on; g_wino_min_blocks = 16; } }
void wino_split_set(int v) { g_wino_split = v != 0; g_wino_split_steps = v >= 2 ? v : 3; }
void wino_short_set(int) {}              // RPG_TUNE_WINO_SHORT: retired with the short-K kernel (accepted, ignored)
void wino_persist_set(int on) { g_wino_persist = on; }      // 2: also for launches of at most one tile per CU
void wino_combine_max_set(int v) { g_wino_combine_max = v; }

// Winograd needs (a) 32-bit buffer offsets (checked again by the launcher) and (b) enough work to occupy the chip.  Since
// the 8-wave kernel cuts a grid that does not fill the CUs along K (split-K tail), that is little: from 16 units of 64
// tiles x 64 channels on it beats the direct kernel + stream-K (measured on the whole forward: 2 graphs 1068 -> 1198
// graphs/s, 4: 1400 -> 1525, 8: 1891 -> 2149; 1 graph: equal).
bool wino_pays(int n, int h, int w, int cin, int cout) {
    if (!g_wino || (cin & 3) || (cout & 3)) return false;
    const long tiles = (long)n * h * ((w + 3) / 4);
    const long blocks = ((tiles + BMT - 1) / BMT) * ((cout + BN - 1) / BN);
    const int tw = (w + 3) / 4;
    return blocks >= g_wino_min_blocks && (long)h * w * cin * 4 * 67 < (1L << 31) && 6L * cout * 3 * cin * 4 < (1L << 31) &&
           (32L / tw + 3) * w * cout * 4 < (1L << 31);
}

int launch_conv_wino(const float* x, const float* u, const float* scale, const float* shift, const float* residual,
                     float* y, int n, int h, int w, int cin, int cout, int relu, hipStream_t s) {
    if (!x || !u || !y || n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || (cin & 3) || (cout & 3) ||
        !aligned16(x) || !aligned16(u) || !aligned16(y) || (scale && !aligned16(scale)) || (shift && !aligned16(shift)) ||
        (residual && !aligned16(residual)))
        return RPG_ERR_BAD_ARG;
#ifdef RPG_PROBE_WINO2D
    // probe builds: the nested 2-D form F(4x2, 3x3) (tools/probes/winograd2d.hip) where RPG_TUNE_WINO2D selects it; its weights
    // follow the 1-D image in the same buffer
    if (wino2d_takes(n, h, w, cin, cout))
        return launch_conv_wino2d(x, u + (size_t)18 * cout * cin, scale, shift, residual, y, n, h, w, cin, cout, relu, s);
#endif
    const int tw = (w + 3) / 4;
    const long M = (long)n * h * tw;
    // 32-bit buffer offsets: a workgroup's 64 tiles span at most 65 images (+ 3 rows of scalar offset); U is
    // addressed from its base
    // the epilogue addresses the output rows of a wave's 32 tiles (<= 32/tw + 2 image rows) from its first row
    if (M >= (1L << 31) || (long)h * w * cin * 4 * 67 >= (1L << 31) || 6L * cout * 3 * cin * 4 >= (1L << 31) ||
        (32L / tw + 3) * w * cout * 4 >= (1L << 31))
        return RPG_ERR_BAD_ARG;
    int dev = 0;                                   // function attributes are per device
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    static bool attr[64] = {};
    if (!attr[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino43_conv_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr[dev] = true;
    }
    const int tn = (cout + BN - 1) / BN;
    Epi ep{scale, shift, residual, y, relu};
    const int slot = timing_begin(RPG_TIMER_CONV_WINO, s);
    double executed = 0.0;                 // matrix-pipe FLOP: workgroups x K steps x 48 MFMAs x waves x 4096
    const long tm8 = (M + BMT8 - 1) / BMT8;
    const bool fits8 = (long)h * w * cin * 4 * 131 < (1L << 31);     // a workgroup's 128 tiles span at most 129 images
    if (fits8 && (g_wino == 3 || (g_wino == 1 && tm8 * tn >= 8))) {
        // large problems (or RPG_TUNE_WINOGRAD = 3): 8 waves on 128 tiles, double-buffered, one workgroup per CU.
        // The tiles beyond the last full round of CUs would cost a whole extra round (784 tiles on 256 CUs: a 4th
        // round for 2 % of the work): they are cut along K into floor(CUs / tail) parts (>= g_wino_split_steps K steps each) whose partial
        // tiles a small fix-up kernel adds in k order.
        static bool attr8[64] = {};
        if (!attr8[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino43_conv8_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS8_BYTES);
            attr8[dev] = true;
        }
        const long T = tm8 * tn;
        const int S = num_cus(), kpr = (cin + BK - 1) / BK, nk = 3 * kpr;
        executed = (double)T * nk * 48.0 * 8.0 * 4096.0;
        long t_main = T;
        Split sp{0, 1, 0, nullptr, nullptr};
        const long tail = T % S;
        if (g_wino_persist && cin % BK == 0 && (T > S || g_wino_persist == 2) && h >= 2) {
            // more tiles than CUs: the persistent kernel, S workgroups walking items b, b + S, ...; the tail tiles are cut
            // into parts of whole channel blocks (3 K steps) handed to the first workgroups
            static bool attr8p[64] = {};
            if (!attr8p[dev]) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino43_conv8p_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, LDS8_BYTES);
                attr8p[dev] = true;
            }
            if (g_wino_split && tail > 0) {
                int parts = (int)(S / tail);
                if (parts > kpr) parts = kpr;
                if (parts >= 2) {
                    sp.partial = stream_scratch(s, (size_t)tail * parts * (BMT8 * 4 * BN) * sizeof(float));
                    if (sp.partial) { sp.parts = parts; t_main = T - tail; sp.tile_base = (int)t_main; sp.n_split = (int)(tail * parts); }
                }
            }
            const long items = sp.n_split + t_main;
            hipLaunchKernelGGL(wino43_conv8p_kernel, dim3((unsigned)(items < S ? items : S)), dim3(NT8), LDS8_BYTES, s, x, u, h, w,
                               cin, cout, tw, (int)M, ep, tn, sp, (int)items);
            if (sp.n_split) {
                hipStream_t fs = fixup_hop_begin(s);                                     // RPG_TUNE_FIXUP_PRIO: high-priority companion stream
                hipLaunchKernelGGL(wino43_fixup_kernel, dim3((unsigned)(T - t_main) * 32), dim3(256), 0, fs, sp.partial, ep, (int)M,
                                   tw, w, cout, tn, sp);
                fixup_hop_end(s, fs);
            }
            timing_end(slot, 2.0 * (double)n * h * w * cout * 9.0 * cin, s, executed);
            RPG_CHECK_LAUNCH("conv3x3_wino43");
            return RPG_OK;
        }
        if (g_wino_split && tail > 0) {
            int parts = (int)(S / tail);
            if (parts > kpr) parts = kpr;               // a part is at least one channel block (3 K steps)
            if (parts > nk / g_wino_split_steps && nk / g_wino_split_steps >= 2) parts = nk / g_wino_split_steps;
            if (parts >= 2) {
                unsigned* counters = nullptr;
                sp.partial = stream_scratch(s, (size_t)tail * parts * (BMT8 * 4 * BN) * sizeof(float), &counters);
                if (sp.partial) { sp.parts = parts; t_main = T - tail; sp.tile_base = (int)t_main; sp.n_split = (int)(tail * parts); }
                if (sp.partial && inkernel_fixup_enabled() && parts <= g_wino_combine_max) sp.arrive = counters;
            }
        }
        // one launch: the split workgroups first, then the whole tiles
        hipLaunchKernelGGL(wino43_conv8_kernel, dim3((unsigned)(sp.n_split + t_main)), dim3(NT8), LDS8_BYTES, s, x, u, h, w, cin,
                           cout, tw, (int)M, ep, tn, sp);
        if (sp.n_split && !sp.arrive) {
            hipStream_t fs = fixup_hop_begin(s);
            hipLaunchKernelGGL(wino43_fixup_kernel, dim3((unsigned)(T - t_main) * 32), dim3(256), 0, fs, sp.partial, ep, (int)M, tw, w,
                               cout, tn, sp);
            fixup_hop_end(s, fs);
        }
    } else {
        const int tm = (int)((M + BMT - 1) / BMT);
        executed = (double)tm * tn * (3 * ((cin + BK - 1) / BK)) * 48.0 * 4.0 * 4096.0;
        hipLaunchKernelGGL(wino43_conv_kernel, dim3(tm * tn), dim3(NT), LDS_BYTES, s, x, u, h, w, cin, cout, tw, (int)M, ep, tn);
    }
    timing_end(slot, 2.0 * (double)n * h * w * cout * 9.0 * cin, s, executed);      // algorithmic = direct-convolution FLOP
    RPG_CHECK_LAUNCH("conv3x3_wino43");
    return RPG_OK;
}

}  // namespace rpg

#ifdef RPG_WINO_TRACE
extern "C" int rpg_wino_trace_set(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wino_trace), &buf, sizeof(buf)) == hipSuccess ? RPG_OK : RPG_ERR_LAUNCH;
}
#endif

// floats of the transformed-weight buffer of one convolution: U [6][cout][3][cin] (+ the nested image [24][cout][cin] behind it
// in a probe build with the 2-D kernel, tools/probes/winograd2d.hip)
extern "C" size_t rpg_wino43_weights_floats(int cout, int cin) {
    if (cout <= 0 || cin <= 0) return 0;
#ifdef RPG_PROBE_WINO2D
    return rpg::wino_weight_floats(cout, cin);
#else
    return (size_t)18 * cout * cin;
#endif
}

extern "C" int rpg_wino43_transform_weights_f32(const float* w_ohwi, float* u, int cout, int cin, void* stream) {
    if (!w_ohwi || !u || cout <= 0 || cin <= 0) return RPG_ERR_BAD_ARG;
    const long total = (long)cout * 3 * cin;
    hipLaunchKernelGGL(wino43_weights_kernel, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, rpg::as_stream(stream),
                       w_ohwi, u, cout, cin, total);
    RPG_CHECK_LAUNCH("wino43_transform_weights");
#ifdef RPG_PROBE_WINO2D
    return rpg::launch_wino2d_weights(w_ohwi, u + (size_t)18 * cout * cin, cout, cin, rpg::as_stream(stream));
#else
    return RPG_OK;
#endif
}

extern "C" int rpg_conv3x3_wino43_bn_act_nhwc_f32(const float* x, const float* u, const float* scale, const float* shift,
                                                  const float* residual, float* y, int n, int h, int w, int cin, int cout,
                                                  int relu, void* stream) {
    return rpg::launch_conv_wino(x, u, scale, shift, residual, y, n, h, w, cin, cout, relu, rpg::as_stream(stream));
}
