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
//     distance apart, so half-wave 1 reads at base + delta) and two MFMAs, with no barrier and no staging; a fragment is
//     4 x 8 pixels, so that a lane ends up with a 4 x 4 pixel block of its channel;
//   * the BatchNorm scale is folded into the weight operands at pack time, its shift and the ReLU are applied to the
//     accumulators (a lane holds one channel), the lane's 4 x 4 block is max-reduced in registers into the <= 9 pooled
//     cells it touches, and those partial maxima are combined across lanes / fragments with LDS atomic max (values are >= 0
//     after the ReLU, so the unsigned-integer max of the bit patterns is the float max and 0 is the identity);
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
constexpr int NFRAG_MAX = (CR * (2 * TWP_MAX + 1) + 31) / 32;     // 32 fragments of 32 pixels (28 blocks + 4 rows of 32 at TWp = 56)
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
    int nbx, nblk, ncl;    // 4 x 8-pixel block fragments: nbx per block row, nblk = 2 nbx in all; ncl = RW - 8 nbx leftover columns
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
    for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        const int n = tile / (a.tiles_y * a.tiles_x);
        const int tr = tile - n * (a.tiles_y * a.tiles_x);
        const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
        const int P0 = ty * PH, Q0 = tx * a.TWp;
        const int cy0 = 2 * P0 - 1, cx0 = 2 * Q0 - 1;            // first convolution row / column of the tile
        const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;          // first input row / column of the patch
        const int npix = CR * a.RW;
        // ---- phase 1: stage the input patch (zero outside the image), clear the pooled tile, build the leftover-pixel table.
        // The loads come first and are unconditional on clamped addresses (rows / columns outside the image are zeroed when
        // the value is written to LDS), so all 35 of a thread are in flight together behind the table arithmetic.  (Keeping
        // them in flight across the previous tile's MFMA phase instead was measured: no gain, and 35 more live registers.)
        {
            float pv[NPV];
            {
                const float* img = a.x + (size_t)n * 3 * a.H * a.W;
                const int ix = ix0 + pcol;
                const int ixc = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
#pragma unroll
                for (int u = 0; u < NPV; ++u) {
                    int rr = prow0 + 2 * u;                      // wave-uniform: c * IR + r
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
            }
            for (int i = tid; i < (PH * a.TWp + 1) * 16; i += ST_NT) reinterpret_cast<uint4*>(pooled)[i] = make_uint4(0u, 0u, 0u, 0u);
            // leftover pixels (the 9th convolution row, then the columns right of the last block, rows 0..7): pixel -> cells
            for (int l = tid; l < (a.nfrag - a.nblk) * 32; l += ST_NT) {
                int ey = CR - 1, ex = l;
                bool in = l < a.RW;
                if (!in && a.ncl > 0) {
                    const int l2 = l - a.RW;
                    ey = l2 / a.ncl;
                    ex = 8 * a.nbx + (l2 - ey * a.ncl);
                    in = ey < CR - 1;
                }
                const int cy = cy0 + ey, cx = cx0 + ex;
                unsigned c4[4] = {trash, trash, trash, trash};
                if (in && (unsigned)cy < (unsigned)a.Hc && (unsigned)cx < (unsigned)a.Wc) {
                    // pooled rows r with 2r-1 <= cy <= 2r+1 (one for even cy, two for odd), tile-local; same for columns
                    const int r0 = (cy >> 1) - P0, r1 = ((cy + 1) >> 1) - P0, q0 = (cx >> 1) - Q0, q1 = ((cx + 1) >> 1) - Q0;
                    const bool r0v = (unsigned)r0 < (unsigned)PH && P0 + r0 < a.Hp, r1v = r1 != r0 && (unsigned)r1 < (unsigned)PH && P0 + r1 < a.Hp;
                    const bool q0v = (unsigned)q0 < (unsigned)a.TWp && Q0 + q0 < a.Wp, q1v = q1 != q0 && (unsigned)q1 < (unsigned)a.TWp && Q0 + q1 < a.Wp;
                    if (r0v && q0v) c4[0] = (unsigned)(r0 * a.TWp + q0) * 256u;
                    if (r0v && q1v) c4[1] = (unsigned)(r0 * a.TWp + q1) * 256u;
                    if (r1v && q0v) c4[2] = (unsigned)(r1 * a.TWp + q0) * 256u;
                    if (r1v && q1v) c4[3] = (unsigned)(r1 * a.TWp + q1) * 256u;
                }
                tab[l] = make_uint4(c4[0], c4[1], c4[2], c4[3]);
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
        // ---- phase 2: 32 convolution pixels x 64 channels per fragment, fragments dealt round-robin to the waves.
        // Fragments 0 .. nblk-1 are 4 rows x 8 columns of convolution pixels: in the MFMA result a lane then holds a 4 x 4
        // pixel block of its channel (rows = register quads, columns 4 h .. 4 h + 3), and since a tile starts on odd
        // convolution coordinates (2 P0 - 1, 2 Q0 - 1) that block touches three pooled rows x three pooled columns: it is reduced
        // in registers and costs <= 9 LDS atomics per channel half instead of 36.  The remaining pixels (9th row, right-hand
        // columns) go 32 in a row through the per-pixel table.
        for (int f = wave; f < a.nfrag; f += ST_NT / 64) {
            const bool blk = f < a.nblk;
            int oy, ox, by = 0, bx = 0;
            if (blk) {
                by = f / a.nbx; bx = f - by * a.nbx;
                oy = 4 * by + ((lane & 31) >> 3);
                ox = 8 * bx + (lane & 7);
            } else {
                const int l = (f - a.nblk) * 32 + (lane & 31);
                oy = CR - 1; ox = l;
                if (l >= a.RW) {
                    const int l2 = l - a.RW;
                    oy = a.ncl > 0 ? l2 / a.ncl : CR;
                    ox = 8 * a.nbx + (l2 - oy * a.ncl);
                    if (oy >= CR - 1) { oy = 0; ox = 0; }        // padding lanes of the last fragment: any valid address
                }
            }
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
            // + shift, ReLU, then fold the convolution outputs into the pooled cells they belong to (LDS atomic max on the
            // bit patterns: exact for non-negative floats)
#ifdef ST_NO_EPI
            if (a.N == -5)
#endif
            if (blk) {
                // element e of an accumulator = pixel (row e >> 2, column 4 h + (e & 3)) of the block
                const int cyb = cy0 + 4 * by, cxs = cx0 + 8 * bx + 4 * h;
                bool rok[4], cok[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    rok[t] = (unsigned)(cyb + t) < (unsigned)a.Hc;
                    cok[t] = (unsigned)(cxs + t) < (unsigned)a.Wc;
                }
                // The block starts on odd convolution coordinates (row 2 r - 1 with r = P0 + 2 by, column 2 q - 1 with q = Q0 + 4 bx
                // + 2 h), so its rows {0}, {0,1,2}, {2,3} belong to pooled rows r-1, r, r+1 and its columns likewise to q-1, q,
                // q+1: nine cells, reduced in registers (the missing rows / columns of a cell come from the neighbouring blocks
                // through the same atomics).
                const int ra = 2 * by, qa = 4 * bx + 2 * h;
                unsigned cell[3][3];
#pragma unroll
                for (int ri = 0; ri < 3; ++ri)
#pragma unroll
                    for (int qi = 0; qi < 3; ++qi) {
                        const int r = ra - 1 + ri, qq = qa - 1 + qi;
                        const bool ok = (unsigned)r < (unsigned)PH && P0 + r < a.Hp && (unsigned)qq < (unsigned)a.TWp && Q0 + qq < a.Wp;
                        cell[ri][qi] = (ok ? (unsigned)(r * a.TWp + qq) * 256u : trash) + lane_b;
                    }
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    float c[4][3];                               // per block row: column 0, max over columns 0-2, max over 2-3
#pragma unroll
                    for (int y = 0; y < 4; ++y) {
                        float v[4];
#pragma unroll
                        for (int xx = 0; xx < 4; ++xx) {
                            const float t = fmaxf(acc[nf][4 * y + xx] + sh[nf], 0.f);
                            v[xx] = rok[y] && cok[xx] ? t : 0.f;  // outside the image: 0, the identity of max over values >= 0
                        }
                        c[y][0] = v[0];
                        c[y][1] = fmaxf(fmaxf(v[0], v[1]), v[2]);
                        c[y][2] = fmaxf(v[2], v[3]);
                    }
#pragma unroll
                    for (int qi = 0; qi < 3; ++qi) {
                        const float m0 = c[0][qi];                                           // pooled row r - 1: block row 0
                        const float m1 = fmaxf(fmaxf(c[0][qi], c[1][qi]), c[2][qi]);         // pooled row r: rows 0-2
                        const float m2 = fmaxf(c[2][qi], c[3][qi]);                          // pooled row r + 1: rows 2-3
                        const float m[3] = {m0, m1, m2};
#pragma unroll
                        for (int ri = 0; ri < 3; ++ri) {
                            // the cells of row r - 1 / column q - 1 exist only away from the tile's top / left edge: skip the
                            // atomic when no lane has a target
                            if ((ri == 0 || qi == 0) && __builtin_amdgcn_ballot_w64(cell[ri][qi] != trash + lane_b) == 0) continue;
#ifndef ST_NO_ATOMICS
                            atomicMax(reinterpret_cast<unsigned*>(reinterpret_cast<char*>(pooled) + cell[ri][qi]) + 32 * nf,
                                      __float_as_uint(m[ri]));
#else
                            asm volatile("" ::"v"(cell[ri][qi]), "v"(m[ri]));
#endif
                        }
                    }
                }
            } else {
                const int l0 = (f - a.nblk) * 32 + 4 * h;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint4 cells = tab[l0 + (e & 3) + 8 * (e >> 2)];      // uniform over a half-wave: broadcast read
                    const unsigned v0 = __float_as_uint(fmaxf(acc[0][e] + sh[0], 0.f));
                    const unsigned v1 = __float_as_uint(fmaxf(acc[1][e] + sh[1], 0.f));
                    const unsigned co[4] = {cells.x, cells.y, cells.z, cells.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // a pixel feeds 1, 2 or 4 cells by the parity of its row / column: skip the LDS atomics (~8 LDS cycles
                        // each, the LDS is shared by the CU) when no lane has a target
                        if (k > 0 && __builtin_amdgcn_ballot_w64(co[k] != trash) == 0) continue;
                        unsigned* cellp = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(pooled) + (co[k] + lane_b));
#ifndef ST_NO_ATOMICS
                        atomicMax(cellp, v0);
                        atomicMax(cellp + 32, v1);
#else
                        asm volatile("" ::"v"(cellp), "v"(v0), "v"(v1));
#endif
                    }
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


// ================================================================================================================================
// Round 6: the fp32 stem as STRIPS MARCHING DOWN THE IMAGE (the design of stem_bf16.hip's stem_strip_bf16_kernel on the f32 matrix
// pipe).  The tile kernel above stages a patch in LDS, runs 74 tap-pair MFMAs per fragment out of it and pools through LDS, three
// barriers per tile: 0.69 MFMA-busy at 645 us per 256 images, unchanged since round 2.  Here a WAVE owns 32
// convolution columns (15 pooled ones) x 64 channels of a band of pooled rows and walks down the convolution rows:
//   * v_mfma_f32_32x32x2_f32 takes two taps per instruction (k = lane >> 5).  Per channel: the 7 kernel rows x 3 horizontal pairs
//     (kw 0|1, 2|3, 4|5) and the seventh column as VERTICAL pairs (kh 0|1, 2|3, 4|5) + (6, 6) alone: 25 MFMAs, 75 per 32-channel half
//     and convolution row for 73.5 tap pairs (the tile kernel: 74);
//   * a lane's operands for input row r are 4 floats -- columns 2 ox - 3 + h, + 2, + 4 (the horizontal pairs; h = lane >> 5) and
//     2 ox + 3 (the seventh column) --, loaded as 4 dwords straight from global memory into the 8-slot register window
//     [channel][row & 7]: convolution row oy uses rows 2 oy - 3 .. 2 oy + 3, two rows leave and two arrive per step (the row loop is
//     unrolled four deep: static register names); the arriving rows are requested at the start of a step, one for this step's LAST
//     MFMAs (kernel row 6: ~9,000 cycles later), one for the next step.  Columns outside the image carry an out-of-range offset
//     per dword (zeros), rows outside are zeroed in a wave-uniform branch: no masks, no conversions;
//   * weights (BatchNorm scale folded in, fp32) are the A operand, read from a 38-KB LDS copy two MFMAs ahead; accumulators come
//     out pixel-major.  max-pool in registers as in the bf16 kernel (running v_max_f32 over the rows of a pooled row; lanes ordered
//     even columns | odd columns: right neighbour by DPP row_shl:1, the odd column by ds_swizzle); the shift is added once per
//     POOLED value (max(a + s) = max(a) + s), then ReLU, 16-byte stores.
// Work item = (image, band of pooled rows, strip, 32-channel half); four items per workgroup, two waves per SIMD.
// Measured (profiles/r6_stem_f32_strips.txt): 599 us per 256 images against 640-684 for the tile kernel (0.76 against 0.69 of the
// f32 matrix peak executed), the configs[1] step 10.46-10.52 against 10.53-10.56 ms.  OPT-IN (RPG_TUNE_FUSED_STEM bit 7), not the
// default: every fp32 bar holds with it (<= 1e-4 everywhere, stem <= 1e-5), but over 8 seeds x 2 shapes its abs-pose distance from
// the float64 oracle is 0.3-2.8x the CPU fp32 reference's own (mean 1.1x) where the tile kernel's is 0.2-1.5x (mean 0.8x) -- a
// different summation order re-rolling a x50-amplified rounding noise --, and the noise-floor test's 1.5x ratio
// (tests/test_hip_bench_geometry.py) would have to move for a 0.4 % gain.  It did not.
// ================================================================================================================================
constexpr int SF_NT = 256, SF_TP = 15, SF_NW = 75;
struct StemSFArgs {
    const float* x;
    const float* wpack;    // [2 nf][75][64 lanes] (params.pack_stem_pairs, second part)
    const float* shift;
    float* out;
    int N, H, W, Hc, Wc, Hp, Wp;
    int TP, tiles_x, BH, bands;
    unsigned items;        // N * bands * tiles_x * 2
};

__global__ __launch_bounds__(SF_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void stem_strip_f32_kernel(StemSFArgs a) {
    constexpr int NFH = 1;                                // 32-channel halves per wave (two overflow the register file: 96 window + 2 x 32 registers)
    constexpr unsigned SENT = 0x80000000u;
    __shared__ float wl[2 * SF_NW * 64];
    __shared__ __attribute__((aligned(16))) float shl[64];
    const int tid = threadIdx.x, lane = tid & 63, l = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * SF_NW * 64; i += SF_NT) wl[i] = a.wpack[i];
    if (tid < 64) shl[tid] = a.shift[tid];
    __syncthreads();
    const unsigned b = blockIdx.x, nb = gridDim.x;
    unsigned wg = b;
    if ((nb & 15u) == 0) { const unsigned xcd = b & 7u, k = b >> 3; wg = ((k >> 1) * 8u + xcd) * 2u + (k & 1u); }
    const unsigned item = wg * 4u + (unsigned)wave;
    if (item >= a.items) return;
    const int nf = NFH == 1 ? (int)(item & 1u) : 0;       // first 32-channel half of this wave
    unsigned rest = NFH == 1 ? item >> 1 : item;
    const int strip = (int)(rest % (unsigned)a.tiles_x); rest /= (unsigned)a.tiles_x;
    const int band = (int)(rest % (unsigned)a.bands);
    const int n = (int)(rest / (unsigned)a.bands);
    const int Q0 = strip * a.TP, P0 = band * a.BH;
    const int nq = a.TP < a.Wp - Q0 ? a.TP : a.Wp - Q0, np = a.BH < a.Hp - P0 ? a.BH : a.Hp - P0;
    const int H = a.H, W = a.W;
    const bool hi = h != 0;

    const int u = l < 16 ? 2 * l : 2 * (l - 16) + 1;
    int cx = 2 * Q0 - 1 + u;
    cx = cx < 0 ? 0 : (cx >= a.Wc ? a.Wc - 1 : cx);
    unsigned vcol[4];                                     // byte offsets of the lane's four columns inside a row, or out of range
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = q < 3 ? 2 * cx - 3 + 2 * q + h : 2 * cx + 3;
        vcol[q] = (unsigned)col < (unsigned)W ? 4u * (unsigned)col : SENT;
    }
    const size_t img_elems = (size_t)3 * H * W;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)n * img_elems, 0, (unsigned)(img_elems * 4), 0x00020000);

    float win[3][8][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int sl = 0; sl < 8; ++sl)
#pragma unroll
            for (int q = 0; q < 4; ++q) win[c][sl][q] = 0.f;
    float mx[NFH][16];
    const float* const wl_lane = wl + nf * SF_NW * 64 + lane;
    float* const out_n = a.out + (size_t)n * a.Hp * a.Wp * 64 + 32 * nf + 4 * h;
    auto emit = [&](int py) {
        if (py < P0 || py >= P0 + np) return;
#pragma unroll
        for (int f = 0; f < NFH; ++f) {
            float r[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int mi = __builtin_bit_cast(int, mx[f][e]);
                const float right = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(mi, mi, 0x101, 0xf, 0xf, false));   // lane i <- lane i + 1 (row of 16)
                const float odd = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(mi, 0x401f));                        // lane i <- lane i ^ 16
                r[e] = fmaxf(fmaxf(mx[f][e], right), odd);
            }
            if (l < nq) {
                float* o = out_n + ((size_t)py * a.Wp + Q0 + l) * 64 + 32 * f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 sh = *reinterpret_cast<const float4*>(shl + 32 * (nf + f) + 8 * g + 4 * h);
                    float4 v;
                    v.x = fmaxf(r[4 * g + 0] + sh.x, 0.f); v.y = fmaxf(r[4 * g + 1] + sh.y, 0.f);
                    v.z = fmaxf(r[4 * g + 2] + sh.z, 0.f); v.w = fmaxf(r[4 * g + 3] + sh.w, 0.f);
                    *reinterpret_cast<float4*>(o + 8 * g) = v;
                }
            }
        }
    };

    const int oy_first = P0 > 0 ? 2 * P0 - 1 : 0;
    int oy_last = 2 * (P0 + np) - 1;
    oy_last = oy_last < a.Hc ? oy_last : a.Hc - 1;
    // input row r -> window slot SL (= r & 7, a literal in the unrolled body)
#define RPG_SF_LOAD(SL, ROW)                                                                                                   \
    do {                                                                                                                       \
        const int r_ = (ROW);                                                                                                  \
        if ((unsigned)r_ < (unsigned)H) {                                                                                      \
            _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                                                    \
                const unsigned ro_ = (unsigned)(((c * H + r_) * W) * 4);                                                       \
                _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                                  \
                    win[c][SL][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vcol[q] == SENT ? SENT : vcol[q] + ro_, 0, 0)); \
            }                                                                                                                  \
        } else {                                                                                                               \
            _Pragma("unroll") for (int c = 0; c < 3; ++c)                                                                      \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) win[c][SL][q] = 0.f;                                             \
        }                                                                                                                      \
    } while (0)
    // MFMA K of the row's 75 on operand B (both channel halves); the weight of MFMA K + 2 is read behind it
#define RPG_SF_MFMA(K, B)                                                                                                      \
    do {                                                                                                                       \
        _Pragma("unroll") for (int f = 0; f < NFH; ++f) {                                                                      \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[f][(K) % 3], (B), acc[f], 0, 0, 0);                               \
            if ((K) + 2 < SF_NW) wf[f][((K) + 2) % 3] = wl_lane[(f * SF_NW + (K) + 2) * 64];                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                                 \
        }                                                                                                                      \
    } while (0)
#define RPG_SF_ROW(R)                                                                                                          \
    do {                                                                                                                       \
        RPG_SF_LOAD((2 * (R) + 3) & 7, 2 * oy + 3);        /* kernel row 6 of THIS step: used last */                          \
        RPG_SF_LOAD((2 * (R) + 4) & 7, 2 * oy + 4);        /* kernel row 5 of the next step */                                  \
        if (oy >= oy_first) {                                                                                                  \
            f32x16 acc[NFH];        /* (two chains summed at the end were measured: 620 us against 599 -- 256 VGPRs --, same noise) */ \
            float wf[NFH][3];                                                                                                  \
            _Pragma("unroll") for (int f = 0; f < NFH; ++f) {                                                                  \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) acc[f][e] = 0.f;                                                \
                wf[f][0] = wl_lane[(f * SF_NW + 0) * 64];                                                                      \
                wf[f][1] = wl_lane[(f * SF_NW + 1) * 64];                                                                      \
            }                                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                                 \
            _Pragma("unroll") for (int kh = 0; kh < 7; ++kh) {                                                                 \
                const int sl_ = (2 * (R) + 5 + kh) & 7, kb_ = 9 * kh + 3 * (kh / 2);                                           \
                _Pragma("unroll") for (int c = 0; c < 3; ++c)                                                                  \
                    _Pragma("unroll") for (int p = 0; p < 3; ++p) RPG_SF_MFMA(kb_ + 3 * c + p, win[c][sl_][p]);                \
                if (kh & 1) {                                                                                                  \
                    _Pragma("unroll") for (int c = 0; c < 3; ++c)                                                              \
                        RPG_SF_MFMA(kb_ + 9 + c, hi ? win[c][sl_][3] : win[c][(sl_ + 7) & 7][3]);                              \
                }                                                                                                              \
                if (kh == 6) {                                                                                                 \
                    _Pragma("unroll") for (int c = 0; c < 3; ++c) RPG_SF_MFMA(kb_ + 9 + c, win[c][sl_][3]);                    \
                }                                                                                                              \
            }                                                                                                                  \
            if (oy == oy_first) {                                                                                              \
                _Pragma("unroll") for (int f = 0; f < NFH; ++f)                                                                \
                    _Pragma("unroll") for (int e = 0; e < 16; ++e) mx[f][e] = acc[f][e];                                    \
            } else {                                                                                                           \
                _Pragma("unroll") for (int f = 0; f < NFH; ++f)                                                                \
                    _Pragma("unroll") for (int e = 0; e < 16; ++e) mx[f][e] = fmaxf(mx[f][e], acc[f][e]);                   \
            }                                                                                                                  \
            if ((R) & 1) {                                                                                                     \
                emit((oy - 1) >> 1);                                                                                           \
                _Pragma("unroll") for (int f = 0; f < NFH; ++f)                                                                \
                    _Pragma("unroll") for (int e = 0; e < 16; ++e) mx[f][e] = acc[f][e];                                    \
            }                                                                                                                  \
        }                                                                                                                      \
        ++oy;                                                                                                                  \
    } while (0)

    int oy = (oy_first - 3) & ~3;
    while (oy <= oy_last) {
        RPG_SF_ROW(0);
        if (oy > oy_last) break;
        RPG_SF_ROW(1);
        if (oy > oy_last) break;
        RPG_SF_ROW(2);
        if (oy > oy_last) break;
        RPG_SF_ROW(3);
    }
#undef RPG_SF_ROW
#undef RPG_SF_MFMA
#undef RPG_SF_LOAD
    if (!(oy_last & 1)) emit(oy_last >> 1);               // odd Hc: the last pooled row ends on an even convolution row
}

}  // namespace

namespace rpg {

static int g_fused_stem = 1;                     // RPG_TUNE_FUSED_STEM
void stem_pool_set(int on) { g_fused_stem = on; }
bool stem_pool_supported(int h, int w, int cout) { return g_fused_stem && cout == 64 && h >= 1 && w >= 1; }

static int g_stem_f32_strip = 0;                 // RPG_TUNE_FUSED_STEM bit 7 sets it: the strip-march kernel of round 6 (measured -6..-12 % on the kernel, +0.3..0.5 % on
                                                 // the step; off by default: see the comment at its head)
static int g_stem_f32_bh = 0;                    // pooled rows per band (0: by the launch's size)
void stem_pool_set_strip(int on, int bh) { g_stem_f32_strip = on; g_stem_f32_bh = bh > 0 ? bh : 0; }

// Pooled rows per band of a strip-march launch: a band costs one extra convolution row and three warm-up row loads (~1.5 rows), so
// long bands are cheap -- but the launch should fill whole rounds of the resident workgroup slots (`slots` = CUs x workgroups per CU;
// measured round 6, both-halves form: 512 images best with 1 band, 256 with 2, 128 with 4, 64 with 8 -- always exactly one round).
// Picks the band count with the best (round occupancy) x (band overhead) x (ragged last band), fewest bands on ties.
static int strip_band_rows(long items_per_band, int hp, long slots) {
    int best_bh = hp;
    double best = -1.0;
    for (int bands = 1; bands <= hp; ++bands) {
        const int bh = (hp + bands - 1) / bands;
        if (bh < 7 && bands > 1) break;
        const int nb = (hp + bh - 1) / bh;
        if (nb != bands) continue;                          // (the same band height as a smaller count already tried)
        const long wgs = (items_per_band * nb + 3) / 4;
        const long rounds = (wgs + slots - 1) / slots;
        const double eff = (double)wgs / (double)(rounds * slots) * (2.0 * bh / (2.0 * bh + 2.5)) * ((double)hp / (double)(nb * bh));
        if (eff > best + 1e-9) { best = eff; best_bh = bh; }
    }
    return best_bh;
}

int launch_stem_pool(const float* x_nchw, const float* wpack, const float* shift, float* out, int n, int h, int w,
                     hipStream_t s) {
    if (!x_nchw || !wpack || !shift || !out || n <= 0 || h <= 0 || w <= 0 || !aligned16(out)) return RPG_ERR_BAD_ARG;
    if (g_stem_f32_strip && (long)3 * h * w * 4 < (1L << 31)) {
        StemSFArgs sa{};
        sa.x = x_nchw; sa.wpack = wpack + KP * NF * 64; sa.shift = shift; sa.out = out;      // second part of params.pack_stem_pairs
        sa.N = n; sa.H = h; sa.W = w;
        sa.Hc = (h + 6 - 7) / 2 + 1; sa.Wc = (w + 6 - 7) / 2 + 1;
        sa.Hp = (sa.Hc + 2 - 3) / 2 + 1; sa.Wp = (sa.Wc + 2 - 3) / 2 + 1;
        sa.tiles_x = (sa.Wp + SF_TP - 1) / SF_TP;
        sa.TP = (sa.Wp + sa.tiles_x - 1) / sa.tiles_x;
        if (g_stem_f32_bh > 0) {
            sa.BH = g_stem_f32_bh < sa.Hp ? g_stem_f32_bh : sa.Hp;
        } else {
            sa.BH = strip_band_rows((long)n * sa.tiles_x * 2, sa.Hp, 2L * num_cus());
        }
        sa.bands = (sa.Hp + sa.BH - 1) / sa.BH;
        const long items = (long)n * sa.bands * sa.tiles_x * 2;
        if (items + 64 < (1L << 32)) {
            sa.items = (unsigned)items;
            const int slot = timing_begin(RPG_TIMER_CONV, s);
            hipLaunchKernelGGL(stem_strip_f32_kernel, dim3((unsigned)((items + 3) / 4)), dim3(SF_NT), 0, s, sa);
            // executed: per item 2 np + 1 convolution rows x 150 MFMAs of 32 x 32 x 2
            timing_end(slot, 2.0 * (double)n * sa.Hc * sa.Wc * 64.0 * 147.0, s,
                       (double)items * (2.0 * sa.Hp / sa.bands + 1.0) * SF_NW * 4096.0);
            RPG_CHECK_LAUNCH("stem_conv_bn_relu_maxpool (strips)");
            return RPG_OK;
        }
    }
    StemArgs a{};
    a.x = x_nchw; a.wpack = wpack; a.shift = shift; a.out = out;
    a.N = n; a.H = h; a.W = w;
    a.Hc = (h + 6 - 7) / 2 + 1; a.Wc = (w + 6 - 7) / 2 + 1;
    a.Hp = (a.Hc + 2 - 3) / 2 + 1; a.Wp = (a.Wc + 2 - 3) / 2 + 1;
    a.tiles_x = (a.Wp + TWP_MAX - 1) / TWP_MAX;
    a.TWp = (a.Wp + a.tiles_x - 1) / a.tiles_x;
    a.tiles_y = (a.Hp + PH - 1) / PH;
    a.RW = 2 * a.TWp + 1;
    a.nbx = a.RW / 8;                          // block fragments: rows 0..7 x columns 0..8 nbx - 1
    a.nblk = 2 * a.nbx;
    a.ncl = a.RW - 8 * a.nbx;                  // leftover columns (1..7): with the 9th row they go through 1 x 32 fragments
    a.nfrag = a.nblk + (a.RW + 8 * a.ncl + 31) / 32;
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
