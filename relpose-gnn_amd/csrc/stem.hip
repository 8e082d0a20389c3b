// The ResNet stem as ONE kernel for gfx950: conv 7x7 / stride 2 / pad 3 (3 -> 64 channels) + BatchNorm (eval) + ReLU +
// max-pool 3x3 / stride 2 / pad 1, reading the reference's NCHW input directly and writing the pooled NHWC tensor.
//
// Replaces (reference: torchvision resnet34 conv1 / bn1 / relu / maxpool, reached from
// /root/reference/python/niantic/modules/posenet.py:1037 after the view of :1035) what round 1 ran as three kernels:
// NCHW -> NHWC4 re-layout (56 us at 256 images), the generic implicit-GEMM kernel on a 4-channel image (747 us: K = 7*7*4
// = 196 padded to 208, a quarter of the MACs on zero channels) and a max-pool pass (227 us, 822 MB written and re-read).
//
// Design.  A workgroup (8 waves) owns a tile of 4 pooled rows x TWp <= 56 pooled columns of one image, all 64 channels:
//   * the input patch it needs (23 rows x (4 TWp + 7) columns x 3 planes, zero outside the image) is staged in LDS once;
//   * the 9 x (2 TWp + 1) convolution outputs under those pooled pixels are computed 32 pixels x 64 channels at a time
//     with v_mfma_f32_32x32x2_f32 (exact fp32): K = 3*7*7 = 147 taps = 74 pairs, a wave keeps ALL 74 x 2 weight operands
//     in registers for its whole life (the kernel is persistent: one workgroup per CU walks the tiles), so the K loop is
//     one ds_read_b32 (the A operand, straight from the patch: the two k-slices of an MFMA are two taps a constant
//     distance apart, so half-wave 1 reads at base + delta) and two MFMAs, with no barrier and no staging;
//   * the BatchNorm scale is folded into the weight operands at pack time, its shift and the ReLU are applied to the
//     accumulators (a lane holds one channel), and every convolution output is folded into the <= 4 pooled cells it
//     belongs to with an LDS atomic max (values are >= 0 after the ReLU, so the unsigned-integer max of the bit patterns
//     is the float max and 0 is the identity); which cells a pixel feeds is a small per-tile table in LDS;
//   * the pooled tile (a contiguous NHWC block per pooled row) is flushed with 16-byte stores.
// The convolution rows / columns on a tile border are computed by both neighbours (9/8 x 113/112 of the MACs at 224x224);
// nothing but the input (154 MB at 256 images) and the pooled output (205 MB) touches HBM.
#include "rpg_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int ST_NT = 512;                 // 8 waves, one workgroup per CU
constexpr int PH = 4;                      // pooled rows per tile
constexpr int CR = 2 * PH + 1;             // convolution rows per tile (9)
constexpr int IR = 2 * CR + 5;             // input rows per tile (23)
constexpr int TWP_MAX = 56;                // pooled columns per tile
constexpr int PITCH = 4 * TWP_MAX + 8;     // 232 floats per patch row (4 TWp + 7 used)
constexpr int KP = 74;                     // tap pairs (147 taps + 1 zero)
constexpr int NF = 2;                      // 32-channel fragments (64 output channels)
constexpr int NFRAG_MAX = (CR * (2 * TWP_MAX + 1) + 31) / 32;     // 32 fragments of 32 pixels
constexpr int PATCH_FLOATS = 3 * IR * PITCH;                      // 16008
constexpr int POOL_FLOATS = (PH * TWP_MAX + 1) * 64;              // 224 cells + 1 trash cell
constexpr int TAB_DWORDS = NFRAG_MAX * 32 * 4;                    // per convolution pixel: byte offsets of its <= 4 pooled cells
constexpr int ST_LDS_BYTES = (PATCH_FLOATS + POOL_FLOATS + TAB_DWORDS) * 4;   // 137,764

struct StemArgs {
    const float* x;        // [N][3][H][W]
    const float* wpack;    // [KP][NF][64]: lane l of fragment nf holds scale[ch] * W[ch = 32 nf + (l & 31)][tap A (l < 32) or B of the pair]
    const float* shift;    // [64] folded BatchNorm shift (the scale is folded into wpack)
    float* out;            // [N][Hp][Wp][64]
    int N, H, W, Hc, Wc, Hp, Wp;
    int TWp, tiles_x, tiles_y, RW, nfrag, total_tiles;
};

// float offset of tap (c, kh, kw) inside the patch, relative to a pixel's top-left tap
__host__ __device__ constexpr int tap_off(int c, int kh, int kw) { return (c * IR + kh) * PITCH + kw; }
// pair kp = (tap A, tap B = tap A + delta of its class): class 0: next column, 1: next row, 2: next plane
__host__ __device__ constexpr int pair_class(int kp) { return kp < 63 ? 0 : (kp < 72 ? 1 : (kp == 72 ? 2 : 0)); }
__host__ __device__ constexpr int pair_off(int kp) {
    return kp < 63 ? tap_off(kp / 21, (kp % 21) / 3, 2 * (kp % 3))
                   : (kp < 72 ? tap_off((kp - 63) / 3, 2 * ((kp - 63) % 3), 6) : (kp == 72 ? tap_off(0, 6, 6) : tap_off(2, 6, 6)));
}

__global__ __launch_bounds__(ST_NT) void stem_pool_kernel(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* patch = lds;                                                   // [3][IR][PITCH]
    unsigned* pooled = reinterpret_cast<unsigned*>(lds + PATCH_FLOATS);   // [PH * TWp + trash][64], bit patterns of floats >= 0
    uint4* tab = reinterpret_cast<uint4*>(lds + PATCH_FLOATS + POOL_FLOATS);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // the wave's weight operands: 148 registers, loaded once (the kernel is persistent)
    float breg[KP][NF];
#pragma unroll
    for (int kp = 0; kp < KP; ++kp)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) breg[kp][nf] = a.wpack[(kp * NF + nf) * 64 + lane];
    float sh[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) sh[nf] = a.shift[32 * nf + (lane & 31)];
    // The operands must not look like pending VMEM results inside the tile loop: with the next tile's patch loads in
    // flight there, the compiler's wait-count pass (in-order vmcnt, states merged at the loop header) made the first MFMAs
    // of every fragment wait for ALL of them.  Wait once here and re-define the registers through an empty asm.
    __builtin_amdgcn_s_waitcnt(0x0F70);                                    // vmcnt(0)
#pragma unroll
    for (int kp = 0; kp < KP; ++kp)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) asm volatile("" : "+v"(breg[kp][nf]));
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) asm volatile("" : "+v"(sh[nf]));
    const unsigned trash = (unsigned)(PH * a.TWp) * 256u;                 // byte offset of the trash cell inside `pooled`
    const unsigned lane_b = 4u * (unsigned)(lane & 31);
    // patch staging: thread t copies column t & 255 of patch rows (t >> 8) + 2 u, u = 0 .. 34 (69 rows = 3 planes x 23)
    const int pcol = tid & 255, prow0 = __builtin_amdgcn_readfirstlane(tid >> 8);
    constexpr int NROW = 3 * IR, NPV = (NROW + 1) / 2;
    float pv[NPV];
    // patch of a tile -> registers, zero outside the image.  Branch-free per lane (a row outside the image is a uniform
    // skip, a column outside it a clamped address + select), so all 35 loads of a thread are in flight together.
    auto fetch_patch = [&](int t) {
        const int n = t / (a.tiles_y * a.tiles_x);
        const int tr = t - n * (a.tiles_y * a.tiles_x);
        const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
        const int iy0 = 2 * (2 * ty * PH - 1) - 3, ix0 = 2 * (2 * tx * a.TWp - 1) - 3;
        const float* img = a.x + (size_t)n * 3 * a.H * a.W;
        const int ix = ix0 + pcol;
        const int ixc = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
        // every load is unconditional on a clamped address (rows and columns outside the image are zeroed where pv is
        // consumed): one load instruction with a scalar row base per element, no branches, nothing waits here
#pragma unroll
        for (int u = 0; u < NPV; ++u) {
            int rr = prow0 + 2 * u;                              // wave-uniform: c * IR + r
            rr = rr < NROW ? rr : NROW - 1;
            const int c = rr / IR, r = rr - c * IR;
            int iy = iy0 + r;
            iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
#ifndef ST_NO_PATCH
            pv[u] = img[((size_t)c * a.H + iy) * a.W + ixc];
#else
            pv[u] = 1.f;
#endif
        }
    };
    if ((int)blockIdx.x < a.total_tiles) fetch_patch(blockIdx.x);

    for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        const int n = tile / (a.tiles_y * a.tiles_x);
        const int tr = tile - n * (a.tiles_y * a.tiles_x);
        const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
        const int P0 = ty * PH, Q0 = tx * a.TWp;
        const int cy0 = 2 * P0 - 1, cx0 = 2 * Q0 - 1;            // first convolution row / column of the tile
        const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;          // first input row / column of the patch
        const int npix = CR * a.RW;
        // ---- phase 1: clear the pooled tile, build the pixel -> cell table, write the prefetched patch to LDS
        {
            for (int i = tid; i < (PH * a.TWp + 1) * 16; i += ST_NT) reinterpret_cast<uint4*>(pooled)[i] = make_uint4(0u, 0u, 0u, 0u);
            for (int p = tid; p < a.nfrag * 32; p += ST_NT) {
                const int ey = p / a.RW, ex = p - ey * a.RW;
                const int cy = cy0 + ey, cx = cx0 + ex;
                unsigned c4[4] = {trash, trash, trash, trash};
                if (p < npix && (unsigned)cy < (unsigned)a.Hc && (unsigned)cx < (unsigned)a.Wc) {
                    // pooled rows r with 2r-1 <= cy <= 2r+1 (one for even cy, two for odd), tile-local; same for columns
                    const int r0 = (cy >> 1) - P0, r1 = ((cy + 1) >> 1) - P0, q0 = (cx >> 1) - Q0, q1 = ((cx + 1) >> 1) - Q0;
                    const bool r0v = (unsigned)r0 < (unsigned)PH && P0 + r0 < a.Hp, r1v = r1 != r0 && (unsigned)r1 < (unsigned)PH && P0 + r1 < a.Hp;
                    const bool q0v = (unsigned)q0 < (unsigned)a.TWp && Q0 + q0 < a.Wp, q1v = q1 != q0 && (unsigned)q1 < (unsigned)a.TWp && Q0 + q1 < a.Wp;
                    if (r0v && q0v) c4[0] = (unsigned)(r0 * a.TWp + q0) * 256u;
                    if (r0v && q1v) c4[1] = (unsigned)(r0 * a.TWp + q1) * 256u;
                    if (r1v && q0v) c4[2] = (unsigned)(r1 * a.TWp + q0) * 256u;
                    if (r1v && q1v) c4[3] = (unsigned)(r1 * a.TWp + q1) * 256u;
                }
                tab[p] = make_uint4(c4[0], c4[1], c4[2], c4[3]);
            }
            if (pcol < PITCH) {
                const bool col_ok = pcol < 4 * a.TWp + 7 && (unsigned)(ix0 + pcol) < (unsigned)a.W;
#pragma unroll
                for (int u = 0; u < NPV; ++u) {
                    const int rr = prow0 + 2 * u;
                    const int r = rr - (rr / IR) * IR;
                    const bool ok = col_ok && (unsigned)(iy0 + r) < (unsigned)a.H;
                    if (rr < NROW) patch[rr * PITCH + pcol] = ok ? pv[u] : 0.f;
                }
            }
        }
        __syncthreads();
        // the next tile's patch travels from HBM to registers while this tile's MFMAs run (all workgroups stage at about the
        // same time: done in the open, the 228 MB of patches were a bandwidth-bound burst of ~80 us per launch)
        if (tile + (int)gridDim.x < a.total_tiles) fetch_patch(tile + gridDim.x);
        // ---- phase 2: 32 convolution pixels x 64 channels per fragment, fragments dealt round-robin to the waves
        for (int f = wave; f < a.nfrag; f += ST_NT / 64) {
            int p = f * 32 + (lane & 31);
            p = p < npix ? p : npix - 1;
            const int oy = p / a.RW, ox = p - oy * a.RW;
            const float* q = patch + (2 * oy * PITCH + 2 * ox);
            const float* q0 = q + (h ? 1 : 0);                   // half-wave 1 holds the second tap of a pair: next column,
            const float* q1 = q + (h ? PITCH : 0);               // next row,
            const float* q2 = q + (h ? IR * PITCH : 0);          // or next plane
            f32x16 acc[NF];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nf][e] = 0.f;
#pragma unroll
            for (int kp = 0; kp < KP; ++kp) {
                const int cls = pair_class(kp);
                const float av = (cls == 0 ? q0 : (cls == 1 ? q1 : q2))[pair_off(kp)];
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
#ifndef ST_NO_MFMA
                    acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, breg[kp][nf], acc[nf], 0, 0, 0);
#else
                    acc[nf][kp & 15] += av * breg[kp][nf];
#endif
                }
            }
            // + shift, ReLU, then fold every convolution output into the pooled cells it belongs to (LDS atomic max on the
            // bit patterns: exact for non-negative floats)
            const int p0 = f * 32 + 4 * h;
#ifdef ST_NO_EPI
            if (a.N == -5)
#endif
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint4 cells = tab[p0 + (e & 3) + 8 * (e >> 2)];          // uniform over a half-wave: broadcast read
                const unsigned v0 = __float_as_uint(fmaxf(acc[0][e] + sh[0], 0.f));
                const unsigned v1 = __float_as_uint(fmaxf(acc[1][e] + sh[1], 0.f));
                const unsigned co[4] = {cells.x, cells.y, cells.z, cells.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // a pixel feeds 1, 2 or 4 cells by the parity of its row / column, the same for both half-waves except
                    // across a row end: skip the LDS atomics (~8 LDS cycles each, the LDS is shared by the CU) when no lane
                    // has a target
                    if (k > 0 && __builtin_amdgcn_ballot_w64(co[k] != trash) == 0) continue;
                    unsigned* cell = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(pooled) + (co[k] + lane_b));
#ifndef ST_NO_ATOMICS
                    atomicMax(cell, v0);
                    atomicMax(cell + 32, v1);
#else
                    asm volatile("" ::"v"(cell), "v"(v0), "v"(v1));
#endif
                }
            }
        }
        __syncthreads();
        // ---- phase 3: flush the pooled tile (NHWC: a pooled row of the tile is one contiguous block)
        for (int i = tid; i < PH * a.TWp * 16; i += ST_NT) {
            const int c4 = i & 15, cell = i >> 4;
            const int r = cell / a.TWp, qq = cell - r * a.TWp;
            const int py = P0 + r, px = Q0 + qq;
            if (py < a.Hp && px < a.Wp)
                reinterpret_cast<uint4*>(a.out)[(((size_t)n * a.Hp + py) * a.Wp + px) * 16 + c4] = reinterpret_cast<const uint4*>(pooled)[i];
        }
        __syncthreads();
    }
}

}  // namespace

namespace rpg {

static int g_fused_stem = 1;                     // RPG_TUNE_FUSED_STEM
void stem_pool_set(int on) { g_fused_stem = on; }
bool stem_pool_supported(int h, int w, int cout) { return g_fused_stem && cout == 64 && h >= 1 && w >= 1; }

int launch_stem_pool(const float* x_nchw, const float* wpack, const float* shift, float* out, int n, int h, int w,
                     hipStream_t s) {
    if (!x_nchw || !wpack || !shift || !out || n <= 0 || h <= 0 || w <= 0 || !aligned16(out)) return RPG_ERR_BAD_ARG;
    StemArgs a{};
    a.x = x_nchw; a.wpack = wpack; a.shift = shift; a.out = out;
    a.N = n; a.H = h; a.W = w;
    a.Hc = (h + 6 - 7) / 2 + 1; a.Wc = (w + 6 - 7) / 2 + 1;
    a.Hp = (a.Hc + 2 - 3) / 2 + 1; a.Wp = (a.Wc + 2 - 3) / 2 + 1;
    a.tiles_x = (a.Wp + TWP_MAX - 1) / TWP_MAX;
    a.TWp = (a.Wp + a.tiles_x - 1) / a.tiles_x;
    a.tiles_y = (a.Hp + PH - 1) / PH;
    a.RW = 2 * a.TWp + 1;
    a.nfrag = (CR * a.RW + 31) / 32;
    const long total = (long)n * a.tiles_y * a.tiles_x;
    if (total >= (1L << 31) || (long)n * 3 * h * w >= (1L << 40)) return RPG_ERR_BAD_ARG;
    a.total_tiles = (int)total;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    static bool attr[64] = {};
    if (!attr[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_pool_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  ST_LDS_BYTES);
        attr[dev] = true;
    }
    const int slot = timing_begin(RPG_TIMER_CONV, s);
    const int grid = (int)(total < num_cus() ? total : num_cus());
    hipLaunchKernelGGL(stem_pool_kernel, dim3(grid), dim3(ST_NT), ST_LDS_BYTES, s, a);
    // algorithmic: the 7x7x3 convolution on every output pixel; executed: fragments x 74 pairs x 2 MFMAs x 4096
    timing_end(slot, 2.0 * (double)n * a.Hc * a.Wc * 64.0 * 147.0, s, (double)total * a.nfrag * KP * NF * 4096.0);
    RPG_CHECK_LAUNCH("stem_conv_bn_relu_maxpool");
    return RPG_OK;
}

}  // namespace rpg

extern "C" int rpg_stem_conv7x7s2_bn_relu_maxpool_f32(const float* x_nchw, const float* wpack, const float* shift,
                                                      float* y_nhwc, int n, int h, int w, void* stream) {
    return rpg::launch_stem_pool(x_nchw, wpack, shift, y_nhwc, n, h, w, rpg::as_stream(stream));
}

// Host-side description of wpack for the packers (relpose-gnn_amd/params.py mirrors it): HOST out arrays of 74 entries.
extern "C" int rpg_stem_pair_table(int* tap_a /* [74][3] = c, kh, kw */, int* tap_b /* [74][3], c = -1: zero weight */) {
    if (!tap_a || !tap_b) return RPG_ERR_BAD_ARG;
    for (int kp = 0; kp < KP; ++kp) {
        const int off = pair_off(kp);
        const int c = off / (IR * PITCH), kh = (off / PITCH) % IR, kw = off % PITCH;
        tap_a[3 * kp] = c; tap_a[3 * kp + 1] = kh; tap_a[3 * kp + 2] = kw;
        const int cls = pair_class(kp);
        int cb = c + (cls == 2), khb = kh + (cls == 1), kwb = kw + (cls == 0);
        if (cb > 2 || khb > 6 || kwb > 6) cb = -1;
        tap_b[3 * kp] = cb; tap_b[3 * kp + 1] = khb; tap_b[3 * kp + 2] = kwb;
    }
    return RPG_OK;
}
