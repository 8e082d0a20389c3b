// f32 MFMA tile engine for gfx950 (MI355X) and its two front-ends:
//   * implicit-GEMM convolution over NHWC activations with a fused BatchNorm/residual/ReLU epilogue
//     (replaces aten conv2d + batch_norm + relu reached from torchvision resnet34,
//      /root/reference/python/niantic/modules/posenet.py:1037);
//   * nn.Linear over a never-materialised row-wise concatenation of up to three gathered sources
//     (replaces torch.cat + index_select + addmm, my_gnn_layer.py:236-239,304-311; posenet.py:1053-1055).
//
// Design (CDNA4): C[M][N] = A[M][K] * W[N][K]^T with both operands K-contiguous.  A workgroup of 4 waves
// (one per SIMD) owns a BM x BN tile; each wave owns FM x FN fragments of 32x32 and accumulates with
// v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD = the 157 TFLOP/s f32 matrix peak).  K is walked in
// steps of BK=16 through a double-buffered LDS image [rows][BK+4] (80-byte pitch: the 16-lane groups of a
// ds_read_b128 land on 16 distinct 4-bank slots).  A lane (i, h) reads 4 consecutive k of row i at offset
// 4h and feeds them to 4 MFMAs, so the two k-slices of one MFMA are k and k+4: a permutation of the
// summation order that A and W share.  Global->LDS staging goes through registers (the im2col / gather
// addressing is per-row, and the padded pitch rules out LDS-DMA), one barrier per step.  Two main loops:
//   * tile_mainloop_b + ConvLoaderB / ConvLoaderTap / GatherLoaderB (whenever every K segment is a multiple of BK: all
//     the shapes of the hot path): raw buffer loads with K-invariant lane offsets and the K position in the scalar
//     offset, and every load / LDS access placed singly behind an MFMA, because VALU time does not hide behind f32
//     MFMAs on gfx950 and a wave issues in order (tools/probes/mfma_shadow_probe.hip);
//   * tile_mainloop + ConvLoader / GatherLoader: any shape (ragged K, unaligned segments), loads before / stage
//     writes after the MFMAs of a step.
// Tiles: 128x128, 256x64 (Cout = 64), 128x64 and 64x64 (small problems); a stream-K pass + fix-up kernel balances the
// tiles that do not fill a round of resident workgroups.  Workgroup ids are remapped so that the tiles sharing an A
// row-panel run on one XCD (shared L2).
#include <atomic>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "rpg_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int NTHREADS = 256;
// K-step BK in {16, 32}: LDS pitch BK+4 floats; a staging pass covers NTHREADS / (BK/4) rows of BK/4 float4 slots.

struct Epilogue {
    const float* scale;      // per output column, or null
    const float* shift;      // per output column (bias / folded BN shift), or null
    const float* residual;   // [M][ldc] or null; row m reads row res_idx[m] (pitch ldr) when res_idx is given
    float* out;              // [M][ldc]
    int ldc;
    int relu;
    const int64_t* res_idx = nullptr;     // gather index of `residual` rows, or null (= row m, pitch ldc)
    const float* residual2 = nullptr;     // second gathered residual (pitch ldr), or null
    const int64_t* res2_idx = nullptr;
    int ldr = 0;                          // pitch of gathered residual rows
};

// byte-free helpers: element offset of row m's residual(s)
__device__ __forceinline__ size_t res_off(const Epilogue& ep, int m) {
    return ep.res_idx ? (size_t)ep.res_idx[m] * ep.ldr : (size_t)m * ep.ldc;
}

__device__ __forceinline__ float4 ld4_or_zero(const float* p, bool ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) v = *reinterpret_cast<const float4*>(p);
    return v;
}

// ------------------------------------------------------------------------------------------------
// A-operand loaders.  Each thread owns one 4-float k-slot (tid % SLOTS) of R rows (tid / SLOTS) + ROWS_PER_PASS j.
// ------------------------------------------------------------------------------------------------
struct ConvArgs {
    const float* x;
    int H, W, Cin, KH, KW, stride, pad, Ho, Wo;
};

template <int R, int BK>
struct ConvLoader {
    static constexpr bool kBuffer = false;
    static constexpr int SLOTS = BK / 4, ROWS_PER_PASS = NTHREADS / SLOTS;
    const float* img[R];
    int hi0[R], wi0[R];
    int H, W, Cin, KH, KW;
    int kh, kw, c;

    __device__ __forceinline__ void init(const ConvArgs& a, int m0, int M, int tid, int kbase) {
        H = a.H; W = a.W; Cin = a.Cin; KH = a.KH; KW = a.KW;
        const int r0 = tid / SLOTS;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int m = m0 + r0 + ROWS_PER_PASS * j;
            if (m < M) {
                const int wo = m % a.Wo;
                const int t = m / a.Wo;
                const int ho = t % a.Ho;
                const int n = t / a.Ho;
                img[j] = a.x + (size_t)n * a.H * a.W * a.Cin;
                hi0[j] = ho * a.stride - a.pad;
                wi0[j] = wo * a.stride - a.pad;
            } else {
                img[j] = a.x;
                hi0[j] = -(1 << 24);   // fails every bounds check
                wi0[j] = 0;
            }
        }
        const int k0 = kbase + 4 * (tid % SLOTS);
        c = k0 % Cin;
        const int t = k0 / Cin;
        kw = t % KW;
        kh = t / KW;
    }
    __device__ __forceinline__ void fetch(float4 (&v)[R]) const {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int hi = hi0[j] + kh, wi = wi0[j] + kw;
            const bool ok = (kh < KH) && ((unsigned)hi < (unsigned)H) && ((unsigned)wi < (unsigned)W);
            v[j] = ld4_or_zero(img[j] + ((hi * W + wi) * Cin + c), ok);
        }
    }
    __device__ __forceinline__ void advance() {
        c += BK;
        while (c >= Cin) {
            c -= Cin;
            if (++kw == KW) { kw = 0; ++kh; }
        }
    }
};

struct GatherArgs {
    const float* a[3];
    const int64_t* idx[3];
    int ld[3];
    int w0, w01, K;    // segment boundaries along k: [0,w0) [w0,w01) [w01,K)
};

template <int R, int BK>
struct GatherLoader {
    static constexpr bool kBuffer = false;
    static constexpr int SLOTS = BK / 4, ROWS_PER_PASS = NTHREADS / SLOTS;
    const float* r0p[R];
    const float* r1p[R];
    const float* r2p[R];
    int w0, w01, K, k;

    __device__ __forceinline__ void init(const GatherArgs& a, int m0, int M, int tid, int kbase) {
        w0 = a.w0; w01 = a.w01; K = a.K;
        const int r0 = tid / SLOTS;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int m = m0 + r0 + ROWS_PER_PASS * j;
            r0p[j] = r1p[j] = r2p[j] = nullptr;
            if (m < M) {
                const int64_t i0 = a.idx[0] ? a.idx[0][m] : (int64_t)m;
                r0p[j] = a.a[0] + (size_t)i0 * a.ld[0];
                if (a.a[1]) {
                    const int64_t i1 = a.idx[1] ? a.idx[1][m] : (int64_t)m;
                    r1p[j] = a.a[1] + (size_t)i1 * a.ld[1];
                }
                if (a.a[2]) {
                    const int64_t i2 = a.idx[2] ? a.idx[2][m] : (int64_t)m;
                    r2p[j] = a.a[2] + (size_t)i2 * a.ld[2];
                }
            }
        }
        k = kbase + 4 * (tid % SLOTS);
    }
    __device__ __forceinline__ void fetch(float4 (&v)[R]) const {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const float* base;
            int kk;
            if (k < w0) { base = r0p[j]; kk = k; }
            else if (k < w01) { base = r1p[j]; kk = k - w0; }
            else { base = r2p[j]; kk = k - w01; }
            v[j] = ld4_or_zero(base + kk, (base != nullptr) && (k < K));
        }
    }
    __device__ __forceinline__ void advance() { k += BK; }
};

// ------------------------------------------------------------------------------------------------
// Buffer-load loaders (tile_mainloop_b).  Measured on gfx950 (tools/probes/mfma_shadow_probe.hip): VALU instructions
// do not overlap with f32 MFMAs on a SIMD and a wave issues in order, so per-step address arithmetic and branchy
// conditional loads come straight out of matrix-pipe time.  These loaders keep everything lane-dependent K-invariant:
// a row is a 32-bit byte offset into a raw buffer resource (0x80000000 = out of range = the hardware returns zeros,
// which serves both ragged tiles and convolution padding), the K position goes into the instruction's scalar offset,
// and the offsets are rebuilt only when the source segment / kernel tap changes (a rarely taken branch).
// Preconditions (checked on the host, else the general loaders above are used): every segment length (gather widths,
// Cin) is a multiple of BK, and the offsets fit in 32 bits.
// ------------------------------------------------------------------------------------------------
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7fffffff, 0x00020000);
}

template <int R, int BK>
struct ConvLoaderB {
    static constexpr bool kBuffer = true;
    static constexpr int SLOTS = BK / 4, ROWS_PER_PASS = NTHREADS / SLOTS;
    __amdgpu_buffer_rsrc_t rs;
    int base[R], hi0[R], wi0[R];      // element offset of the row's image (relative to the tile's first image); -1 = no row
    int H, W, Cin, KW, slot4;
    unsigned voff[R];
    int kh, kw, c0, k0, kend;         // wave-uniform position of the next fetch

    __device__ __forceinline__ void init(const ConvArgs& a, int m0, int M, int tid, int kbase, int kend_) {
        H = a.H; W = a.W; Cin = a.Cin; KW = a.KW; slot4 = 4 * (tid % SLOTS); kend = kend_;
        const int r0 = tid / SLOTS;
        const int n_first = m0 / (a.Ho * a.Wo);
        const int img = a.H * a.W * a.Cin;
        rs = make_rsrc(a.x + (size_t)n_first * img);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int m = m0 + r0 + ROWS_PER_PASS * j;
            base[j] = -1; hi0[j] = 0; wi0[j] = 0;
            if (m < M) {
                const int wo = m % a.Wo;
                const int t = m / a.Wo;
                base[j] = (t / a.Ho - n_first) * img;
                hi0[j] = (t % a.Ho) * a.stride - a.pad;
                wi0[j] = wo * a.stride - a.pad;
            }
        }
        k0 = kbase;
        c0 = kbase % Cin;
        const int t = kbase / Cin;
        kw = t % KW;
        kh = t / KW;
        refresh();
    }
    __device__ __forceinline__ void refresh() {           // offsets of tap (kh, kw); everything invalid past kend
        asm volatile("");                                  // keeps the callers' branch: if-converted this is ~8 VALU / row
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int hi = hi0[j] + kh, wi = wi0[j] + kw;
            const bool ok = base[j] >= 0 && k0 < kend && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
            voff[j] = ok ? 4u * (unsigned)(base[j] + (hi * W + wi) * Cin + slot4) : OOB;
        }
    }
    __device__ __forceinline__ float4 fetch_one(int j) const { return buf_ld4(rs, voff[j], 4u * (unsigned)c0); }
    __device__ __forceinline__ void advance() {
        k0 += BK;
        c0 += BK;
        if (c0 >= Cin || k0 >= kend) {
            if (c0 >= Cin) { c0 = 0; if (++kw == KW) { kw = 0; ++kh; } }
            refresh();
        }
    }
};

// The 4-channel stem (Cin == 4: one 16-byte k-slot = all channels of ONE kernel tap): thread slot s walks taps
// s, s + SLOTS, ...  Validity of (row, tap) is a K-invariant 64-bit mask per row (bit tap = pixel inside the image),
// the tap's offset relative to the row's top-left pixel is per-thread and advances incrementally, so a load costs a
// 64-bit shift, a test, an add and a select instead of the general loader's bounds arithmetic and conditional load.
template <int R, int BK>
struct ConvLoaderTap {
    static constexpr bool kBuffer = true;
    static constexpr int SLOTS = BK / 4, ROWS_PER_PASS = NTHREADS / SLOTS;
    __amdgpu_buffer_rsrc_t rs;
    unsigned long long mask[R];       // bit t: tap t of this row reads a pixel inside the image
    int rowoff[R];                    // byte offset of the row's (hi0, wi0) pixel relative to the tile's first image (may be < 0)
    int W, KW, taps;
    int tap, kh, kw, k0, kend;        // this thread's current tap; k0 wave-uniform

    __device__ __forceinline__ void init(const ConvArgs& a, int m0, int M, int tid, int kbase, int kend_) {
        W = a.W; KW = a.KW; taps = a.KH * a.KW; kend = kend_;
        const int r0 = tid / SLOTS;
        const int n_first = m0 / (a.Ho * a.Wo);
        const int img = a.H * a.W * 4;
        rs = make_rsrc(a.x + (size_t)n_first * img);
        // (n, ho, wo) of the first row by division, of the others incrementally (rows are ROWS_PER_PASS pixels apart);
        // the tap mask is the outer product of the valid kernel rows and the valid kernel columns
        int m = m0 + r0;
        int wo = m % a.Wo;
        int t = m / a.Wo;
        int ho = t % a.Ho;
        int n = t / a.Ho - n_first;
        const int dwo = ROWS_PER_PASS % a.Wo, dho = ROWS_PER_PASS / a.Wo;     // wave-uniform
        const unsigned long long ones = (1ull << a.KW) - 1ull;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            mask[j] = 0ull; rowoff[j] = 0;
            if (m < M) {
                const int hi0 = ho * a.stride - a.pad, wi0 = wo * a.stride - a.pad;
                rowoff[j] = 4 * (n * img + (hi0 * a.W + wi0) * 4);
                const int xlo = max(0, -wi0), xhi = min(a.KW, a.W - wi0);        // valid kernel columns [xlo, xhi)
                const unsigned long long cm = xhi > xlo ? (ones >> (a.KW - (xhi - xlo))) << xlo : 0ull;
                const int ylo = max(0, -hi0), yhi = min(a.KH, a.H - hi0);        // valid kernel rows [ylo, yhi)
                unsigned long long mk = 0ull;
                for (int y = ylo; y < yhi; ++y) mk |= cm << (y * a.KW);
                mask[j] = mk;
            }
            m += ROWS_PER_PASS;
            wo += dwo; ho += dho;
            if (wo >= a.Wo) { wo -= a.Wo; ++ho; }
            while (ho >= a.Ho) { ho -= a.Ho; ++n; }
        }
        k0 = kbase;
        tap = kbase / 4 + tid % SLOTS;
        kh = tap / KW;
        kw = tap - kh * KW;
    }
    __device__ __forceinline__ float4 fetch_one(int j) const {
        const bool ok = tap < taps && k0 < kend && ((mask[j] >> tap) & 1ull);
        return buf_ld4(rs, ok ? (unsigned)(rowoff[j] + 16 * (kh * W + kw)) : OOB, 0u);
    }
    __device__ __forceinline__ void advance() {
        k0 += BK;
        tap += SLOTS;
        kw += SLOTS;                  // KW >= 4 (host-checked): one wrap step for 4 slots, two for 8
        if (kw >= KW) { kw -= KW; ++kh; }
        if constexpr (SLOTS > 4) { if (kw >= KW) { kw -= KW; ++kh; } }
    }
};

template <int R, int BK>
struct GatherLoaderB {
    static constexpr bool kBuffer = true;
    static constexpr int SLOTS = BK / 4, ROWS_PER_PASS = NTHREADS / SLOTS;
    const float* p0;
    const float* p1;
    const float* p2;
    __amdgpu_buffer_rsrc_t rs;             // resource of the current segment (rebuilt from the pointer in refresh)
    unsigned off0[R], off1[R], off2[R], voff[R];
    int w0, w01, k0, kend, seg_begin, seg_end;

    __device__ __forceinline__ void init(const GatherArgs& a, int m0, int M, int tid, int kbase, int kend_) {
        w0 = a.w0; w01 = a.w01; kend = kend_;
        const int slot4 = 4 * (tid % SLOTS), r0 = tid / SLOTS;
        p0 = a.a[0];
        p1 = a.a[1] ? a.a[1] : a.a[0];
        p2 = a.a[2] ? a.a[2] : a.a[0];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int m = m0 + r0 + ROWS_PER_PASS * j;
            off0[j] = off1[j] = off2[j] = OOB;
            if (m < M) {
                const int64_t i0 = a.idx[0] ? a.idx[0][m] : (int64_t)m;
                off0[j] = 4u * (unsigned)(i0 * a.ld[0] + slot4);
                if (a.a[1]) {
                    const int64_t i1 = a.idx[1] ? a.idx[1][m] : (int64_t)m;
                    off1[j] = 4u * (unsigned)(i1 * a.ld[1] + slot4);
                }
                if (a.a[2]) {
                    const int64_t i2 = a.idx[2] ? a.idx[2][m] : (int64_t)m;
                    off2[j] = 4u * (unsigned)(i2 * a.ld[2] + slot4);
                }
            }
        }
        k0 = kbase;
        refresh();
    }
    __device__ __forceinline__ void refresh() {           // the segment holding k0; everything invalid past kend
        asm volatile("");
        const float* p;
        if (k0 < w0) {
            p = p0; seg_begin = 0; seg_end = w0;
#pragma unroll
            for (int j = 0; j < R; ++j) voff[j] = off0[j];
        } else if (k0 < w01) {
            p = p1; seg_begin = w0; seg_end = w01;
#pragma unroll
            for (int j = 0; j < R; ++j) voff[j] = off1[j];
        } else {
            p = p2; seg_begin = w01; seg_end = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < R; ++j) voff[j] = off2[j];
        }
        rs = make_rsrc(p);
        if (k0 >= kend) {
#pragma unroll
            for (int j = 0; j < R; ++j) voff[j] = OOB;
        }
    }
    __device__ __forceinline__ float4 fetch_one(int j) const { return buf_ld4(rs, voff[j], 4u * (unsigned)(k0 - seg_begin)); }
    __device__ __forceinline__ void advance() {
        k0 += BK;
        if (k0 >= seg_end || k0 >= kend) refresh();
    }
};

// ------------------------------------------------------------------------------------------------
// The tile engine
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int BK>
struct Tile {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static constexpr int LDS_LD = BK + 4;
    static constexpr int SLOTS = BK / 4, ROWS_PER_PASS = NTHREADS / SLOTS;
    static constexpr int FM = BM / WM / 32, FN = BN / WN / 32;
    static constexpr int RA = BM / ROWS_PER_PASS, RW = BN / ROWS_PER_PASS;
    static constexpr int STAGE = (BM + BN) * LDS_LD;                    // floats per LDS buffer
    static constexpr int LDS_BYTES = 2 * STAGE * (int)sizeof(float);
    static_assert(FM >= 1 && FN >= 1 && RA >= 1 && RW >= 1, "tile too small");
};

// acc += A[m0.., ks*BK .. ke*BK) * W[n0.., same k)^T.  Ends with a workgroup barrier: LDS is free afterwards.
template <int BM, int BN, int WM, int WN, int BK, template <int, int> class Loader, class Args>
__device__ __forceinline__ void tile_mainloop(const Args& args, const float* __restrict__ Wt, int ldw, int M, int N,
                                              int K, int m0, int n0, int ks, int ke, float* lds,
                                              f32x16 (&acc)[Tile<BM, BN, WM, WN, BK>::FM][Tile<BM, BN, WM, WN, BK>::FN]) {
    using T = Tile<BM, BN, WM, WN, BK>;
    constexpr int LDS_LD = T::LDS_LD, ROWS_PER_PASS = T::ROWS_PER_PASS, FM = T::FM, FN = T::FN, RA = T::RA, RW = T::RW;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int slot = tid % T::SLOTS, srow = tid / T::SLOTS;

    Loader<RA, BK> la;
    la.init(args, m0, M, tid, ks * BK);
    const float* wrow[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int n = n0 + srow + ROWS_PER_PASS * j;
        wrow[j] = (n < N) ? Wt + (size_t)n * ldw : nullptr;
    }
    int kw_ = ks * BK + 4 * slot;   // this thread's k position in W

    float4 ra[RA], rw[RW];
    auto fetch_w = [&]() {
#pragma unroll
        for (int j = 0; j < RW; ++j) rw[j] = ld4_or_zero(wrow[j] + kw_, (wrow[j] != nullptr) && (kw_ < K));
    };
    auto stage = [&](int buf) {
        float* As = lds + buf * T::STAGE;
        float* Ws = As + BM * LDS_LD;
#pragma unroll
        for (int j = 0; j < RA; ++j)
            *reinterpret_cast<float4*>(&As[(srow + ROWS_PER_PASS * j) * LDS_LD + 4 * slot]) = ra[j];
#pragma unroll
        for (int j = 0; j < RW; ++j)
            *reinterpret_cast<float4*>(&Ws[(srow + ROWS_PER_PASS * j) * LDS_LD + 4 * slot]) = rw[j];
    };

    la.fetch(ra);
    fetch_w();
    stage(0);
    __syncthreads();

    const int a_off = (wm * FM * 32 + (lane & 31)) * LDS_LD + 4 * (lane >> 5);
    const int b_off = (BM + wn * FN * 32 + (lane & 31)) * LDS_LD + 4 * (lane >> 5);
    const int nsteps = ke - ks;
    for (int kt = 0; kt < nsteps; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1 < nsteps);
        if (more) {
            la.advance();
            kw_ += BK;
            la.fetch(ra);
            fetch_w();
        }
        const float* L = lds + cur * T::STAGE;
#pragma unroll
        for (int kb = 0; kb < BK; kb += 8) {
            float av[FM][4], bv[FN][4];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const float4 t = *reinterpret_cast<const float4*>(&L[a_off + i * 32 * LDS_LD + kb]);
                av[i][0] = t.x; av[i][1] = t.y; av[i][2] = t.z; av[i][3] = t.w;
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const float4 t = *reinterpret_cast<const float4*>(&L[b_off + j * 32 * LDS_LD + kb]);
                bv[j][0] = t.x; bv[j][1] = t.y; bv[j][2] = t.z; bv[j][3] = t.w;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c], bv[j][c], acc[i][j], 0, 0, 0);
        }
        if (more) stage(cur ^ 1);
        __syncthreads();
    }
}

// The same contract as tile_mainloop for the buffer loaders, with every non-MFMA instruction placed behind an
// individual MFMA (scheduling barriers pin the placement) so that a wave never stops issuing MFMAs except at the one
// barrier per K step.  A step is BK/8 groups of FM*FN*4 MFMAs; behind the first FM+FN MFMAs of a group go the operand
// reads of the next group (last group: of the next step's first group, from the other LDS image), behind the others
// first the RA+RW stage writes of step kt+1 (data loaded during step kt-1), then the RA+RW buffer loads of step kt+2
// into the same registers.  The barrier sits before the last group: by then every wave has written image kt+1 and
// issued its last reads of image kt.
template <int BM, int BN, int WM, int WN, int BK, template <int, int> class Loader, class Args>
__device__ __forceinline__ void tile_mainloop_b(const Args& args, const float* __restrict__ Wt, int ldw, int M, int N,
                                                int K, int m0, int n0, int ks, int ke, float* lds,
                                                f32x16 (&acc)[Tile<BM, BN, WM, WN, BK>::FM][Tile<BM, BN, WM, WN, BK>::FN]) {
    using T = Tile<BM, BN, WM, WN, BK>;
    constexpr int LDS_LD = T::LDS_LD, ROWS_PER_PASS = T::ROWS_PER_PASS, FM = T::FM, FN = T::FN, RA = T::RA, RW = T::RW;
    constexpr int KB = BK / 8, G = FM * FN * 4, NR = FM + FN, NJ = RA + RW, SPARE = G - NR;
    static_assert(KB >= 2 && (KB & 1) == 0, "operand register sets alternate per group");
    static_assert((KB - 1) * SPARE >= NJ && KB * SPARE >= 2 * NJ, "not enough MFMAs to hide the staging behind");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int slot = tid % T::SLOTS, srow = tid / T::SLOTS;
    const int kend = ke * BK;

    Loader<RA, BK> la;
    la.init(args, m0, M, tid, ks * BK, kend);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(Wt);
    unsigned woff[RW], weff[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int n = n0 + srow + ROWS_PER_PASS * j;
        woff[j] = n < N ? 4u * (unsigned)(n * ldw + 4 * slot) : OOB;
    }
    int kw_ = ks * BK;                 // wave-uniform K position of the next W fetch
    auto refresh_w = [&]() {           // K tail (K % BK != 0) and the steps past the end read zeros
        asm volatile("");
        const bool live = kw_ < kend && kw_ + 4 * slot < K;
#pragma unroll
        for (int j = 0; j < RW; ++j) weff[j] = live ? woff[j] : OOB;
    };
    refresh_w();
    float4 rr[NJ];                     // staging registers: A rows 0..RA-1, then W rows
    auto load_job = [&](int q) {
        if (q < RA) rr[q] = la.fetch_one(q);
        else rr[q] = buf_ld4(rsw, weff[q - RA], 4u * (unsigned)kw_);
    };
    auto next_k = [&]() {
        la.advance();
        kw_ += BK;
        if (kw_ + BK > K || kw_ >= kend) refresh_w();
    };
    const int st_off = srow * LDS_LD + 4 * slot;
    auto write_job = [&](int q, int img) {
        const int r = q < RA ? ROWS_PER_PASS * q : BM + ROWS_PER_PASS * (q - RA);
        *reinterpret_cast<float4*>(&lds[img + st_off + r * LDS_LD]) = rr[q];
    };
    const int a_off = (wm * FM * 32 + (lane & 31)) * LDS_LD + 4 * (lane >> 5);
    const int b_off = (BM + wn * FN * 32 + (lane & 31)) * LDS_LD + 4 * (lane >> 5);
    float4 fr[2][NR];                  // operand fragments: [set][a_0..a_FM-1, b_0..b_FN-1]
    auto read_job = [&](int set, int r, int g, int img) {
        const int off = (r < FM ? a_off + r * 32 * LDS_LD : b_off + (r - FM) * 32 * LDS_LD) + 8 * g;
        fr[set][r] = *reinterpret_cast<const float4*>(&lds[img + off]);
    };
    auto comp = [](const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; };
    auto kstep = [&](int cur, int nxt) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < KB; ++g) {
            const int set = g & 1;
#pragma unroll
            for (int ms = 0; ms < G; ++ms) {
                const int c = ms / (FM * FN), i = (ms / FN) % FM, j = ms % FN;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(fr[set][i], c), comp(fr[set][FM + j], c), acc[i][j], 0, 0, 0);
                if (ms < NR) {
                    if (g + 1 < KB) read_job(set ^ 1, ms, g + 1, cur);
                    else read_job(0, ms, 0, nxt);
                } else {
                    const int q = g * SPARE + (ms - NR);
                    if (q < NJ) write_job(q, nxt);
                    else if (q < 2 * NJ) load_job(q - NJ);
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
    for (int q = 0; q < NJ; ++q) load_job(q);
    next_k();
#pragma unroll
    for (int q = 0; q < NJ; ++q) write_job(q, 0);
#pragma unroll
    for (int q = 0; q < NJ; ++q) load_job(q);
    next_k();
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NR; ++r) read_job(0, r, 0, 0);
    const int nsteps = ke - ks;
    int kt = 0;
    for (; kt + 1 < nsteps; kt += 2) {
        kstep(0, T::STAGE);
        kstep(T::STAGE, 0);
    }
    if (kt < nsteps) kstep(0, T::STAGE);
    __syncthreads();
}

// Epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5).
// EPI_LDS: each wave transposes its accumulators through a private LDS slab (the staging buffers are free after the
// main loop's last barrier) so that global traffic is 16 bytes per lane: a row of the wave tile is FN*32 contiguous
// floats = FN*8 lanes; residual loads / output stores are whole 128..256-byte row segments.  Slab: 32 rows x
// (FN*32 + 4) floats per wave, processed once per FM fragment row.  Needs N % 4 == 0 and 16-byte aligned rows.
template <int BM, int BN, int WM, int WN, int BK, bool EPI_LDS>
__device__ __forceinline__ void tile_epilogue(float* lds, const Epilogue& ep, int m0, int n0, int M, int N,
                                              f32x16 (&acc)[Tile<BM, BN, WM, WN, BK>::FM][Tile<BM, BN, WM, WN, BK>::FN]) {
    using T = Tile<BM, BN, WM, WN, BK>;
    constexpr int FM = T::FM, FN = T::FN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    if constexpr (EPI_LDS) {
        constexpr int EW = FN * 32, EP = EW + 4, C4 = EW / 4, RPI = 64 / C4, NIT = 32 / RPI;
        static_assert(4 * 32 * EP <= 2 * T::STAGE, "epilogue slab does not fit the staging LDS");
        float* slab = lds + wave * (32 * EP);
        const int c4 = lane % C4, r_in = lane / C4;
        const int nb = n0 + wn * EW + 4 * c4;                 // first of this lane's 4 output columns
        const bool n_ok = nb < N;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n_ok && ep.scale) sc = *reinterpret_cast<const float4*>(ep.scale + nb);
        if (n_ok && ep.shift) sh = *reinterpret_cast<const float4*>(ep.shift + nb);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    slab[((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * EP + j * 32 + (lane & 31)] = acc[i][j][e];
            __builtin_amdgcn_wave_barrier();
            const int mb = m0 + (wm * FM + i) * 32;
            float4 v[NIT], rs[NIT];
#pragma unroll
            for (int t = 0; t < NIT; ++t) {
                const int row = r_in + RPI * t;
                v[t] = *reinterpret_cast<const float4*>(&slab[row * EP + 4 * c4]);
                rs[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ep.residual && n_ok && (mb + row) < M) {
                    rs[t] = *reinterpret_cast<const float4*>(ep.residual + res_off(ep, mb + row) + nb);
                    if (ep.residual2) {
                        const float4 r2 = *reinterpret_cast<const float4*>(ep.residual2 + (size_t)ep.res2_idx[mb + row] * ep.ldr + nb);
                        rs[t].x += r2.x; rs[t].y += r2.y; rs[t].z += r2.z; rs[t].w += r2.w;
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < NIT; ++t) {
                const int m = mb + r_in + RPI * t;
                float4 o;
                o.x = v[t].x * sc.x + sh.x + rs[t].x;
                o.y = v[t].y * sc.y + sh.y + rs[t].y;
                o.z = v[t].z * sc.z + sh.z + rs[t].z;
                o.w = v[t].w * sc.w + sh.w + rs[t].w;
                if (ep.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                if (n_ok && m < M) *reinterpret_cast<float4*>(ep.out + (size_t)m * ep.ldc + nb) = o;
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        const int col_l = lane & 31, row_l = 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + (wn * FN + j) * 32 + col_l;
            if (n >= N) continue;
            const float sc = ep.scale ? ep.scale[n] : 1.f;
            const float sh = ep.shift ? ep.shift[n] : 0.f;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int mb = m0 + (wm * FM + i) * 32 + row_l;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mb + (e & 3) + 8 * (e >> 2);
                    if (m < M) {
                        float v = acc[i][j][e] * sc + sh;
                        const size_t o = (size_t)m * ep.ldc + n;
                        if (ep.residual) {
                            v += ep.residual[res_off(ep, m) + n];
                            if (ep.residual2) v += ep.residual2[(size_t)ep.res2_idx[m] * ep.ldr + n];
                        }
                        if (ep.relu) v = fmaxf(v, 0.f);
                        ep.out[o] = v;
                    }
                }
            }
        }
    }
}

template <int FM, int FN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[FM][FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
}

// Data-parallel kernel: one workgroup per output tile, whole K range.
template <int BM, int BN, int WM, int WN, int BK, bool EPI_LDS, template <int, int> class Loader, class Args>
// the 256x64 / BK=16 tiles (Cout = 64: the stem) have short K loops; three workgroups per CU overlap their prologues
// and epilogues
__global__ __launch_bounds__(NTHREADS, (BM == 256 && BK == 16) ? 3 : 1) void gemm_tile_kernel(Args args, const float* __restrict__ Wt, int ldw,
                                                             int M, int N, int K, Epilogue ep, int tiles_n) {
    using T = Tile<BM, BN, WM, WN, BK>;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 * STAGE floats
    // XCD-aware, bijective remap of the linear workgroup id: workgroup b runs on XCD b % 8, so give every
    // XCD a contiguous run of tiles (tile_n fastest => neighbours share the A row panel in that XCD's L2).
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3, q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;
    f32x16 acc[T::FM][T::FN];
    zero_acc(acc);
    if constexpr (Loader<1, BK>::kBuffer)
        tile_mainloop_b<BM, BN, WM, WN, BK, Loader, Args>(args, Wt, ldw, M, N, K, m0, n0, 0, (K + BK - 1) / BK, lds, acc);
    else
        tile_mainloop<BM, BN, WM, WN, BK, Loader, Args>(args, Wt, ldw, M, N, K, m0, n0, 0, (K + BK - 1) / BK, lds, acc);
    tile_epilogue<BM, BN, WM, WN, BK, EPI_LDS>(lds, ep, m0, n0, M, N, acc);
}

// Stream-K kernel for the tiles [tile_base, tile_base + R) that do not fill a whole round of resident workgroups:
// their R*nk k-steps are dealt out evenly, `its_per` consecutive steps per workgroup.  A workgroup that covers a
// tile's whole K range finishes it with the normal epilogue; otherwise it stores the raw partial tile [BM][BN] in
// slab (g + t) of `partial` (g = workgroup, t = tile - tile_base: unique and increasing along the stream), and
// streamk_fixup_kernel sums a tile's slabs in k order and applies the epilogue.
template <int BM, int BN, int WM, int WN, int BK, bool EPI_LDS, template <int, int> class Loader, class Args>
__global__ __launch_bounds__(NTHREADS) void gemm_streamk_kernel(Args args, const float* __restrict__ Wt, int ldw,
                                                                int M, int N, int K, Epilogue ep, int tiles_n,
                                                                int tile_base, int total_its, int nk, int its_per,
                                                                float* __restrict__ partial) {
    using T = Tile<BM, BN, WM, WN, BK>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int g = blockIdx.x;
    int it = g * its_per;
    const int it_end = min(it + its_per, total_its);
    while (it < it_end) {
        const int t = it / nk;
        const int kb = it - t * nk;
        const int ke = min(nk, kb + (it_end - it));
        const int tile = tile_base + t;
        const int m0 = (tile / tiles_n) * BM;
        const int n0 = (tile % tiles_n) * BN;
        f32x16 acc[T::FM][T::FN];
        zero_acc(acc);
        if constexpr (Loader<1, BK>::kBuffer)
            tile_mainloop_b<BM, BN, WM, WN, BK, Loader, Args>(args, Wt, ldw, M, N, K, m0, n0, kb, ke, lds, acc);
        else
            tile_mainloop<BM, BN, WM, WN, BK, Loader, Args>(args, Wt, ldw, M, N, K, m0, n0, kb, ke, lds, acc);
        if (kb == 0 && ke == nk) {
            tile_epilogue<BM, BN, WM, WN, BK, EPI_LDS>(lds, ep, m0, n0, M, N, acc);
        } else {
            const Epilogue raw{nullptr, nullptr, nullptr, partial + (size_t)(g + t) * (BM * BN), BN, 0};
            tile_epilogue<BM, BN, WM, WN, BK, true>(lds, raw, 0, 0, BM, BN, acc);
        }
        __syncthreads();          // the epilogue slabs alias the staging buffers of the next segment
        it += ke - kb;
    }
}

template <int BM, int BN>
__global__ __launch_bounds__(NTHREADS) void streamk_fixup_kernel(const float* __restrict__ partial, Epilogue ep, int M,
                                                                 int N, int tiles_n, int tile_base, int nk, int its_per,
                                                                 int vec) {
    constexpr int C4 = BN / 4, CH = BM * C4 / NTHREADS;
    const int t = blockIdx.x / CH, chunk = blockIdx.x % CH;
    const int g_first = (t * nk) / its_per, g_last = ((t + 1) * nk - 1) / its_per;
    if (g_first == g_last) return;                      // finished by a single workgroup with the normal epilogue
    const int idx4 = chunk * NTHREADS + threadIdx.x;
    const int row = idx4 / C4, c4 = idx4 % C4;
    const float* p = partial + (size_t)(g_first + t) * (BM * BN) + row * BN + 4 * c4;
    float4 s = *reinterpret_cast<const float4*>(p);
    for (int g = g_first + 1; g <= g_last; ++g) {       // ascending k order
        p += BM * BN;
        const float4 v = *reinterpret_cast<const float4*>(p);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int tile = tile_base + t;
    const int m = (tile / tiles_n) * BM + row;
    const int n = (tile % tiles_n) * BN + 4 * c4;
    if (m >= M || n >= N) return;
    const size_t o = (size_t)m * ep.ldc + n;
    if (vec) {
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f), rs = sh;
        if (ep.scale) sc = *reinterpret_cast<const float4*>(ep.scale + n);
        if (ep.shift) sh = *reinterpret_cast<const float4*>(ep.shift + n);
        if (ep.residual) {
            rs = *reinterpret_cast<const float4*>(ep.residual + res_off(ep, m) + n);
            if (ep.residual2) {
                const float4 r2 = *reinterpret_cast<const float4*>(ep.residual2 + (size_t)ep.res2_idx[m] * ep.ldr + n);
                rs.x += r2.x; rs.y += r2.y; rs.z += r2.z; rs.w += r2.w;
            }
        }
        float4 v;
        v.x = s.x * sc.x + sh.x + rs.x; v.y = s.y * sc.y + sh.y + rs.y;
        v.z = s.z * sc.z + sh.z + rs.z; v.w = s.w * sc.w + sh.w + rs.w;
        if (ep.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(ep.out + o) = v;
    } else {
        const float sv[4] = {s.x, s.y, s.z, s.w};
        for (int c = 0; c < 4 && n + c < N; ++c) {
            float v = sv[c] * (ep.scale ? ep.scale[n + c] : 1.f) + (ep.shift ? ep.shift[n + c] : 0.f);
            if (ep.residual) {
                v += ep.residual[res_off(ep, m) + n + c];
                if (ep.residual2) v += ep.residual2[(size_t)ep.res2_idx[m] * ep.ldr + n + c];
            }
            if (ep.relu) v = fmaxf(v, 0.f);
            ep.out[o + c] = v;
        }
    }
}

enum TileShape { TILE_128x128 = 0, TILE_256x64 = 1, TILE_64x64 = 2, TILE_128x64 = 3 };

// Tuning knobs (rpg_set_tuning): -1 = automatic.
int g_force_tile = -1;
int g_tile_128x64 = 1;                // prefer 128x64 over 128x128 tiles for problems smaller than 1.5 big tiles per CU
int g_bk = 0;                        // 0 = automatic: 32 for the 128x128 tile (2 workgroups/CU), else 16
int g_epi_lds = 1;
int g_streamk = 1;
int g_gnn_split = 1;
constexpr int SK_MIN_ITS = 8;        // at least this many k-steps per stream-K workgroup
constexpr int SK_MIN_NK = 32;        // tiles with fewer k-steps are not worth splitting (fix-up traffic dominates)

// Library-owned scratch pool of the fine-grained entry points (see rpg_common.h): one block per (device, stream) that
// only grows; a grown block's predecessor is retired, not freed (a captured HIP graph or work still in flight may
// hold its address), and nothing is allocated while the stream is being captured.
struct Scratch { float* p = nullptr; size_t bytes = 0; };
std::mutex g_scratch_mu;
std::map<std::pair<int, hipStream_t>, Scratch> g_scratch;
std::vector<float*> g_retired;
struct ScratchTls { float* p = nullptr; size_t bytes = 0; bool on = false; };
thread_local ScratchTls t_scratch;

float* get_scratch(hipStream_t s, size_t bytes) {
    if (t_scratch.on) return bytes <= t_scratch.bytes ? t_scratch.p : nullptr;     // caller workspace (composite forwards)
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    Scratch& sc = g_scratch[{dev, s}];
    if (sc.bytes < bytes) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (cap != hipStreamCaptureStatusNone) return nullptr;                      // no allocation inside a capture
        float* np = nullptr;
        const size_t want = bytes + bytes / 4;
        if (hipMalloc(reinterpret_cast<void**>(&np), want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (sc.p) g_retired.push_back(sc.p);
        sc.p = np;
        sc.bytes = want;
    }
    return sc.p;
}

constexpr int MAX_DEV = 64;
int cu_count() {
    static std::atomic<int> n[MAX_DEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) dev = 0;
    int v = n[dev].load(std::memory_order_relaxed);
    if (!v) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

inline TileShape pick_tile(int M, int N, int K) {
    if (g_force_tile >= 0 && g_force_tile <= 3) return (TileShape)g_force_tile;
    if (N <= 64) return (M >= 256 * 256) ? TILE_256x64 : TILE_64x64;
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (t128 >= 384) return TILE_128x128;                 // >= 1.5 big tiles per CU
    // fewer big tiles than CUs can balance: fine with stream-K when K is long enough to split, else go small
    // (128x64 halves the tile so that twice as many workgroups share the work before stream-K has to split K: measured
    // on the GNN Linears, M = 1792: N = 768 85 -> 68 us, N = 2048 142 -> 136 us; M = 256, N = 2048 63 -> 57 us)
    if (g_streamk && K >= 32 * SK_MIN_NK && (long)M * N >= 128L * 128 * 8) return g_tile_128x64 ? TILE_128x64 : TILE_128x128;
    return TILE_64x64;
}

thread_local double t_executed = 0.0;      // FLOP the matrix pipe issues for the last launch_one (whole padded tiles)

template <int BM, int BN, int WM, int WN, int BK, bool EPI, template <int, int> class Loader, class Args>
int launch_one(const Args& args, const float* Wt, int ldw, int M, int N, int K, const Epilogue& ep, bool vec_ok,
               hipStream_t s) {
    using T = Tile<BM, BN, WM, WN, BK>;
    auto kern = gemm_tile_kernel<BM, BN, WM, WN, BK, EPI, Loader, Args>;
    auto kern_sk = gemm_streamk_kernel<BM, BN, WM, WN, BK, EPI, Loader, Args>;
    constexpr int lds = T::LDS_BYTES;
    static std::atomic<int> occ_dev[MAX_DEV];          // resident workgroups per CU of this instantiation, per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) dev = 0;
    int occ = occ_dev[dev].load(std::memory_order_relaxed);
    if (!occ) {
        if (lds > 64 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern_sk), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        }
        int o1 = 0, o2 = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o1, kern, NTHREADS, lds) != hipSuccess) o1 = 1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o2, kern_sk, NTHREADS, lds) != hipSuccess) o2 = 1;
        occ = o1 < o2 ? o1 : o2;
        if (occ < 1) occ = 1;
        (void)hipGetLastError();
        occ_dev[dev].store(occ, std::memory_order_relaxed);
    }
    const int tn = (N + BN - 1) / BN, tm = (M + BM - 1) / BM;
    const int tiles = tm * tn, nk = (K + BK - 1) / BK;
    const int slots = cu_count() * occ;
    t_executed = 2.0 * (double)tiles * BM * BN * (double)nk * BK;
    int t_dp = tiles, its_per = 0, g_sk = 0;
    float* partial = nullptr;
    if (g_streamk && nk >= SK_MIN_NK && tiles % slots != 0) {
        const int rem = tiles % slots;
        const long total = (long)rem * nk;
        its_per = (int)((total + slots - 1) / slots);
        if (its_per < SK_MIN_ITS) its_per = SK_MIN_ITS;
        g_sk = (int)((total + its_per - 1) / its_per);
        if (g_sk > rem) {             // stream-K spreads the remainder over more workgroups than tiles: worth it
            partial = get_scratch(s, (size_t)(g_sk + rem) * BM * BN * sizeof(float));
            if (partial) t_dp = tiles - rem;
        }
    }
    if (t_dp > 0)
        hipLaunchKernelGGL(kern, dim3(t_dp), dim3(NTHREADS), lds, s, args, Wt, ldw, M, N, K, ep, tn);
    if (t_dp < tiles) {
        const int rem = tiles - t_dp;
        hipLaunchKernelGGL(kern_sk, dim3(g_sk), dim3(NTHREADS), lds, s, args, Wt, ldw, M, N, K, ep, tn, t_dp, rem * nk, nk,
                           its_per, partial);
        constexpr int CH = BM * (BN / 4) / NTHREADS;
        hipLaunchKernelGGL((streamk_fixup_kernel<BM, BN>), dim3(rem * CH), dim3(NTHREADS), 0, s, partial, ep, M, N, tn,
                           t_dp, nk, its_per, vec_ok ? 1 : 0);
    }
    return 0;
}

template <int BK, bool EPI, template <int, int> class Loader, class Args>
void launch_shape(TileShape t, const Args& args, const float* Wt, int ldw, int M, int N, int K, const Epilogue& ep,
                  bool vec_ok, hipStream_t s) {
    switch (t) {
        case TILE_128x128: launch_one<128, 128, 2, 2, BK, EPI, Loader, Args>(args, Wt, ldw, M, N, K, ep, vec_ok, s); break;
        case TILE_256x64: launch_one<256, 64, 4, 1, BK, EPI, Loader, Args>(args, Wt, ldw, M, N, K, ep, vec_ok, s); break;
        case TILE_128x64: launch_one<128, 64, 2, 2, BK, EPI, Loader, Args>(args, Wt, ldw, M, N, K, ep, vec_ok, s); break;
        default: launch_one<64, 64, 2, 2, BK, EPI, Loader, Args>(args, Wt, ldw, M, N, K, ep, vec_ok, s); break;
    }
}

int g_fast = 1;                      // RPG_TUNE_FAST_LOADER: buffer-load loaders + interleaved main loop where eligible

// seg_align: 32 / 16 if every K segment of the A operand (gather widths, Cin) is a multiple of it and all offsets of
// the buffer loaders fit in 32 bits, else 0 (general loaders).
// seg_align == -1: the A operand is a 4-channel convolution input (ConvLoaderTap via LoaderT; any K).
template <template <int, int> class Loader, template <int, int> class LoaderB, template <int, int> class LoaderT, class Args>
int launch_tiles(const Args& args, const float* Wt, int ldw, int M, int N, int K, const Epilogue& ep, hipStream_t s,
                 int seg_align) {
    const TileShape t = pick_tile(M, N, K);
    const int bk = g_bk ? g_bk : (((t == TILE_128x128 || t == TILE_128x64) && K >= 256) ? 32 : 16);
    // 16-byte epilogue accesses need 4-column groups: N % 4 == 0 and 16-byte aligned rows
    const bool vec_ok = (N % 4 == 0) && (ep.ldc % 4 == 0) && rpg::aligned16(ep.out) &&
                        (!ep.residual || rpg::aligned16(ep.residual)) && (!ep.residual2 || rpg::aligned16(ep.residual2)) &&
                        (ep.ldr % 4 == 0) && (!ep.scale || rpg::aligned16(ep.scale)) &&
                        (!ep.shift || rpg::aligned16(ep.shift));
    const bool epi = g_epi_lds && vec_ok;
    const bool fast = g_fast && epi && seg_align > 0 && seg_align % bk == 0 && K % bk == 0 &&
                      (long)N * ldw * 4 < (1L << 31);
    if (g_fast && epi && seg_align == -1 && (long)N * ldw * 4 < (1L << 31)) {
        if (bk == 32) launch_shape<32, true, LoaderT, Args>(t, args, Wt, ldw, M, N, K, ep, vec_ok, s);
        else launch_shape<16, true, LoaderT, Args>(t, args, Wt, ldw, M, N, K, ep, vec_ok, s);
        return 0;
    }
    if (bk == 32) {
        if (fast) launch_shape<32, true, LoaderB, Args>(t, args, Wt, ldw, M, N, K, ep, vec_ok, s);
        else if (epi) launch_shape<32, true, Loader, Args>(t, args, Wt, ldw, M, N, K, ep, vec_ok, s);
        else launch_shape<32, false, Loader, Args>(t, args, Wt, ldw, M, N, K, ep, vec_ok, s);
    } else {
        if (fast) launch_shape<16, true, LoaderB, Args>(t, args, Wt, ldw, M, N, K, ep, vec_ok, s);
        else if (epi) launch_shape<16, true, Loader, Args>(t, args, Wt, ldw, M, N, K, ep, vec_ok, s);
        else launch_shape<16, false, Loader, Args>(t, args, Wt, ldw, M, N, K, ep, vec_ok, s);
    }
    return 0;
}

}  // namespace

namespace rpg {

bool gnn_split_enabled() { return g_gnn_split != 0; }
float* stream_scratch(hipStream_t s, size_t bytes) { return get_scratch(s, bytes); }
ScratchScope::ScratchScope(void* p, size_t bytes) {
    t_scratch.p = static_cast<float*>(p);
    t_scratch.bytes = bytes;
    t_scratch.on = true;
}
ScratchScope::~ScratchScope() { t_scratch = ScratchTls{}; }
size_t split_scratch_bytes() { return (size_t)2 * 3 * cu_count() * 65536; }
int num_cus() { return cu_count(); }

int launch_conv(const float* x, const float* w, const float* scale, const float* shift, const float* residual,
                float* y, int n, int h, int wd, int cin, int cout, int kh, int kw, int stride, int pad, int relu,
                hipStream_t s, int alg_cin) {
    if (!x || !w || !y || n <= 0 || h <= 0 || wd <= 0 || cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0 ||
        stride <= 0 || pad < 0 || (cin & 3) || !aligned16(x) || !aligned16(w) || !aligned16(y))
        return RPG_ERR_BAD_ARG;
    const int ho = (h + 2 * pad - kh) / stride + 1, wo = (wd + 2 * pad - kw) / stride + 1;
    if (ho <= 0 || wo <= 0) return RPG_ERR_BAD_ARG;
    const long M = (long)n * ho * wo;
    const long K = (long)kh * kw * cin;
    if (M >= (1L << 31) || K >= (1 << 24) || (long)h * wd * cin >= (1L << 31)) return RPG_ERR_BAD_ARG;
    ConvArgs a{x, h, wd, cin, kh, kw, stride, pad, ho, wo};
    Epilogue ep{scale, shift, residual, y, cout, relu};
    const int slot = timing_begin(RPG_TIMER_CONV, s);
    // buffer loaders: offsets are relative to the first image of a tile, whose <= 256 rows span at most
    // 256 / (ho*wo) + 2 images
    const long span = 256 / ((long)ho * wo) + 2;
    int seg = span * h * wd * cin * 4 >= (1L << 31) ? 0 : (cin % 32 == 0 ? 32 : (cin % 16 == 0 ? 16 : 0));
    if (cin == 4 && kh * kw <= 64 && kw >= 4 && span * h * wd * cin * 4 < (1L << 31)) seg = -1;      // the stem: one tap per k-slot
    launch_tiles<ConvLoader, ConvLoaderB, ConvLoaderTap, ConvArgs>(a, w, (int)K, (int)M, cout, (int)K, ep, s, seg);
    timing_end(slot, 2.0 * (double)M * cout * (double)kh * kw * (alg_cin > 0 ? alg_cin : cin), s, t_executed);
    RPG_CHECK_LAUNCH("conv2d_bn_act");
    return RPG_OK;
}

int launch_linear(const GatherSrc& src, const float* weight, const float* bias, const float* residual, float* out,
                  int m, int n_out, int relu, hipStream_t s, const GatherRes* gres) {
    if (src.n < 1 || src.n > 3 || !weight || !out || m <= 0 || n_out <= 0 || !aligned16(weight)) return RPG_ERR_BAD_ARG;
    GatherArgs a{};
    int K = 0;
    for (int i = 0; i < 3; ++i) {
        if (i < src.n) {
            if (!src.a[i] || src.width[i] <= 0 || (src.width[i] & 3) || (src.ld[i] & 3) || src.ld[i] < src.width[i] ||
                !aligned16(src.a[i]))
                return RPG_ERR_BAD_ARG;
            a.a[i] = src.a[i]; a.idx[i] = src.idx[i]; a.ld[i] = src.ld[i];
            K += src.width[i];
        } else {
            a.a[i] = nullptr; a.idx[i] = nullptr; a.ld[i] = 0;
        }
    }
    a.w0 = src.width[0];
    a.w01 = src.n >= 2 ? src.width[0] + src.width[1] : K;
    a.K = K;
    Epilogue ep{nullptr, bias, residual, out, n_out, relu};
    if (gres) {
        if (!gres->res1 || !gres->idx1 || gres->ld < n_out || (gres->res2 && !gres->idx2)) return RPG_ERR_BAD_ARG;
        ep.residual = gres->res1; ep.res_idx = gres->idx1;
        ep.residual2 = gres->res2; ep.res2_idx = gres->idx2;
        ep.ldr = gres->ld;
    }
    const int slot = timing_begin(RPG_TIMER_LINEAR, s);
    int seg = 32;
    for (int i = 0; i < src.n; ++i) {
        if (src.width[i] % 32) seg = src.width[i] % 16 ? 0 : (seg ? 16 : 0);
        const long rows = src.idx[i] ? src.rows[i] : (long)m;          // a gathered source needs its row count
        if (rows <= 0 || rows * src.ld[i] * 4 >= (1L << 31)) seg = 0;
    }
    launch_tiles<GatherLoader, GatherLoaderB, GatherLoaderB, GatherArgs>(a, weight, K, m, n_out, K, ep, s, seg);
    timing_end(slot, 2.0 * (double)m * n_out * (double)K, s, t_executed);
    RPG_CHECK_LAUNCH("linear_gather");
    return RPG_OK;
}

}  // namespace rpg

extern "C" int rpg_conv2d_bn_act_nhwc_f32(const float* x, const float* w_ohwi, const float* scale, const float* shift,
                                          const float* residual, float* y, int n, int h, int w, int cin, int cout,
                                          int kh, int kw, int stride, int pad, int relu, void* stream) {
    return rpg::launch_conv(x, w_ohwi, scale, shift, residual, y, n, h, w, cin, cout, kh, kw, stride, pad, relu,
                            rpg::as_stream(stream));
}

extern "C" int rpg_linear_gather_f32(int n_src, const float* const* a, const int64_t* const* idx, const int* ld,
                                     const int* width, const float* weight, const float* bias, const float* residual,
                                     float* out, int m, int n_out, int relu, void* stream) {
    if (n_src < 1 || n_src > 3 || !a || !ld || !width) return RPG_ERR_BAD_ARG;
    rpg::GatherSrc src{};
    src.n = n_src;
    for (int i = 0; i < n_src; ++i) {
        src.a[i] = a[i];
        src.idx[i] = idx ? idx[i] : nullptr;
        src.ld[i] = ld[i];
        src.width[i] = width[i];
    }
    return rpg::launch_linear(src, weight, bias, residual, out, m, n_out, relu, rpg::as_stream(stream));
}

extern "C" int rpg_release_scratch(void) {
    if (hipDeviceSynchronize() != hipSuccess) {
        rpg::set_last_error("release_scratch", hipGetLastError());
        return RPG_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    for (auto& kv : g_scratch)
        if (kv.second.p) (void)hipFree(kv.second.p);
    g_scratch.clear();
    for (float* p : g_retired) (void)hipFree(p);
    g_retired.clear();
    return RPG_OK;
}

extern "C" int rpg_set_tuning(int key, int value) {
    switch (key) {
        case RPG_TUNE_TILE: g_force_tile = value; return RPG_OK;
        case RPG_TUNE_BK: if (value != 0 && value != 16 && value != 32) return RPG_ERR_BAD_ARG; g_bk = value; return RPG_OK;
        case RPG_TUNE_EPILOGUE: g_epi_lds = value != 0; return RPG_OK;
        case RPG_TUNE_STREAMK: g_streamk = value != 0; return RPG_OK;
        case RPG_TUNE_BF16_BK: if (value != 32 && value != 64) return RPG_ERR_BAD_ARG; rpg::bf16_set_bk(value); return RPG_OK;
        case RPG_TUNE_GNN_SPLIT: g_gnn_split = value != 0; return RPG_OK;
        case RPG_TUNE_FAST_LOADER: g_fast = value != 0; return RPG_OK;
        case RPG_TUNE_WINO_SPLIT: rpg::wino_split_set(value != 0); return RPG_OK;
        case RPG_TUNE_BF16_FAST: rpg::bf16_set_fast(value != 0); return RPG_OK;
        case RPG_TUNE_FUSED_STEM: rpg::stem_pool_set(value != 0); return RPG_OK;
        case RPG_TUNE_WINOGRAD: if (value < 0 || value > 3) return RPG_ERR_BAD_ARG; rpg::wino_set(value); return RPG_OK;
        default: return RPG_ERR_BAD_ARG;
    }
}
