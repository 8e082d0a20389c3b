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
// addressing is per-row, and the padded pitch rules out LDS-DMA); the loads for step t+1 are issued before
// the MFMAs of step t and written to the other buffer after them, one barrier per step.
// Workgroup ids are remapped so that the tiles sharing an A row-panel run on one XCD (shared L2).
#include "rpg_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 16;
constexpr int LDS_LD = BK + 4;
constexpr int NTHREADS = 256;
constexpr int ROWS_PER_PASS = NTHREADS / (BK / 4);   // 64 rows of 4 float4 slots

struct Epilogue {
    const float* scale;      // per output column, or null
    const float* shift;      // per output column (bias / folded BN shift), or null
    const float* residual;   // [M][ldc] or null
    float* out;              // [M][ldc]
    int ldc;
    int relu;
};

__device__ __forceinline__ float4 ld4_or_zero(const float* p, bool ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) v = *reinterpret_cast<const float4*>(p);
    return v;
}

// ------------------------------------------------------------------------------------------------
// A-operand loaders.  Each thread owns one 4-float k-slot (tid & 3) of R rows (tid >> 2) + 64 j.
// ------------------------------------------------------------------------------------------------
struct ConvArgs {
    const float* x;
    int H, W, Cin, KH, KW, stride, pad, Ho, Wo;
};

template <int R>
struct ConvLoader {
    const float* img[R];
    int hi0[R], wi0[R];
    int H, W, Cin, KH, KW;
    int kh, kw, c;

    __device__ __forceinline__ void init(const ConvArgs& a, int m0, int M, int tid) {
        H = a.H; W = a.W; Cin = a.Cin; KH = a.KH; KW = a.KW;
        const int r0 = tid >> 2;
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
        const int k0 = 4 * (tid & 3);
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

template <int R>
struct GatherLoader {
    const float* r0p[R];
    const float* r1p[R];
    const float* r2p[R];
    int w0, w01, K, k;

    __device__ __forceinline__ void init(const GatherArgs& a, int m0, int M, int tid) {
        w0 = a.w0; w01 = a.w01; K = a.K;
        const int r0 = tid >> 2;
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
        k = 4 * (tid & 3);
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
// The tile kernel
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, template <int> class Loader, class Args>
__global__ __launch_bounds__(NTHREADS) void gemm_tile_kernel(Args args, const float* __restrict__ Wt, int ldw,
                                                             int M, int N, int K, Epilogue ep, int tiles_n) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int FM = BM / WM / 32, FN = BN / WN / 32;
    constexpr int RA = BM / ROWS_PER_PASS, RW = BN / ROWS_PER_PASS;
    static_assert(FM >= 1 && FN >= 1 && RA >= 1 && RW >= 1, "tile too small");
    __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * LDS_LD];

    // XCD-aware, bijective remap of the linear workgroup id: workgroup b runs on XCD b % 8, so give every
    // XCD a contiguous run of tiles (tile_n fastest => neighbours share the A row panel in that XCD's L2).
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3, q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int slot = tid & 3, srow = tid >> 2;

    Loader<RA> la;
    la.init(args, m0, M, tid);
    const float* wrow[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int n = n0 + srow + ROWS_PER_PASS * j;
        wrow[j] = (n < N) ? Wt + (size_t)n * ldw : nullptr;
    }
    int kw_ = 4 * slot;   // this thread's k position in W

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[RA], rw[RW];
    auto fetch_w = [&]() {
#pragma unroll
        for (int j = 0; j < RW; ++j) rw[j] = ld4_or_zero(wrow[j] + kw_, (wrow[j] != nullptr) && (kw_ < K));
    };
    auto stage = [&](int buf) {
        float* As = lds[buf];
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

    const int nk = (K + BK - 1) / BK;
    const int a_off = (wm * FM * 32 + (lane & 31)) * LDS_LD + 4 * (lane >> 5);
    const int b_off = (BM + wn * FN * 32 + (lane & 31)) * LDS_LD + 4 * (lane >> 5);

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1 < nk);
        if (more) {
            la.advance();
            kw_ += BK;
            la.fetch(ra);
            fetch_w();
        }
        const float* L = lds[cur];
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

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
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
                    if (ep.residual) v += ep.residual[o];
                    if (ep.relu) v = fmaxf(v, 0.f);
                    ep.out[o] = v;
                }
            }
        }
    }
}

enum TileShape { TILE_128x128, TILE_256x64, TILE_64x64 };

inline TileShape pick_tile(int M, int N) {
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (N <= 64) return (M >= 256 * 256) ? TILE_256x64 : TILE_64x64;
    if (t128 >= 384) return TILE_128x128;     // >= 1.5 workgroups per CU of the big tile
    return TILE_64x64;
}

template <template <int> class Loader, class Args>
int launch_tiles(const Args& args, const float* Wt, int ldw, int M, int N, int K, const Epilogue& ep, hipStream_t s) {
    switch (pick_tile(M, N)) {
        case TILE_128x128: {
            const int tn = (N + 127) / 128, tm = (M + 127) / 128;
            hipLaunchKernelGGL((gemm_tile_kernel<128, 128, 2, 2, Loader, Args>), dim3(tm * tn), dim3(NTHREADS), 0, s,
                               args, Wt, ldw, M, N, K, ep, tn);
            break;
        }
        case TILE_256x64: {
            const int tn = (N + 63) / 64, tm = (M + 255) / 256;
            hipLaunchKernelGGL((gemm_tile_kernel<256, 64, 4, 1, Loader, Args>), dim3(tm * tn), dim3(NTHREADS), 0, s,
                               args, Wt, ldw, M, N, K, ep, tn);
            break;
        }
        default: {
            const int tn = (N + 63) / 64, tm = (M + 63) / 64;
            hipLaunchKernelGGL((gemm_tile_kernel<64, 64, 2, 2, Loader, Args>), dim3(tm * tn), dim3(NTHREADS), 0, s,
                               args, Wt, ldw, M, N, K, ep, tn);
            break;
        }
    }
    return 0;
}

}  // namespace

namespace rpg {

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
    launch_tiles<ConvLoader, ConvArgs>(a, w, (int)K, (int)M, cout, (int)K, ep, s);
    timing_end(slot, 2.0 * (double)M * cout * (double)kh * kw * (alg_cin > 0 ? alg_cin : cin), s);
    RPG_CHECK_LAUNCH("conv2d_bn_act");
    return RPG_OK;
}

int launch_linear(const GatherSrc& src, const float* weight, const float* bias, const float* residual, float* out,
                  int m, int n_out, int relu, hipStream_t s) {
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
    const int slot = timing_begin(RPG_TIMER_LINEAR, s);
    launch_tiles<GatherLoader, GatherArgs>(a, weight, K, m, n_out, K, ep, s);
    timing_end(slot, 2.0 * (double)m * n_out * (double)K, s);
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
