// The ResNet stem of the bf16 encoder as ONE kernel for gfx950 (round 3): conv 7x7 / stride 2 / pad 3 (3 -> 64 channels) on
// v_mfma_f32_32x32x16_bf16 + BatchNorm (eval) + ReLU + max-pool 3x3 / stride 2 / pad 1, reading the reference's fp32 NCHW
// input directly and writing the pooled NHWC tensor in bf16.
//
// Replaces (reference: torchvision resnet34 conv1 / bn1 / relu / maxpool, reached from
// /root/reference/python/niantic/modules/posenet.py:1037 after the view of :1035) what the bf16 encoder ran as three kernels up
// to round 2: fp32 NCHW -> bf16 NHWC8 re-layout (120 us at 512 images), the generic implicit-GEMM kernel on an 8-channel image
// (721 us: K = 7*7*8 = 392, five eighths of the MACs on zero channels) and a max-pool pass (228 us).
//
// Same tiling and pooling as the fp32 kernel (stem.hip): a persistent 8-wave workgroup owns 4 pooled rows x TWp <= 28 pooled
// columns of one image; what differs is the operand path, because the bf16 MFMA wants 8 consecutive k per lane:
//   * K is ordered (channel, kernel row | kernel column 0..7): one MFMA = two (c, kh) rows of 8 kernel columns (the 8th has
//     zero weights), 21 rows -> 11 MFMAs per 32-pixel x 32-channel fragment (K = 176 issued for 147 taps);
//   * the input patch is staged in LDS as bf16 WINDOWS: for every (plane, input row, convolution column ox) the 8 input
//     columns 2 ox - 3 .. 2 ox + 4 as one aligned 16-byte chunk, so a lane's A operand is ONE ds_read_b128 (stride-2
//     windows overlap: the patch is stored four-fold; a thread converts a PAIR of input columns and stores it into the four
//     windows that contain it).  The window rows are 60 chunks long (= 4 mod 8): the four image rows of a 4 x 8-pixel
//     fragment then fall on distinct 64-byte quarters of the bank row (conflict-free ds_read_b128 lane groups);
//   * the weights stay in registers for the kernel's life (11 x 2 fragments x 4 VGPRs); BatchNorm scale and shift are applied
//     in fp32 to the accumulators (weights are the plain bf16 roundings, like every other bf16 convolution of the encoder).
// The pooling (in-register 4 x 4 -> 3 x 3 max, LDS atomic max on the bit patterns of the non-negative fp32 values, leftover
// pixels through a per-tile table) is the fp32 kernel's, unchanged: the MFMA C layout does not depend on the input type.
#include "rpg_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int SB_NT = 512;                 // 8 waves, one workgroup per CU
constexpr int PH = 4;                      // pooled rows per tile
constexpr int CR = 2 * PH + 1;             // convolution rows per tile (9)
constexpr int IR = 2 * CR + 5;             // input rows per tile (23)
constexpr int TWP_MAX = 28;                // pooled columns per tile
constexpr int RW_MAX = 2 * TWP_MAX + 1;    // convolution columns per tile (57)
constexpr int NWIN = 60;                   // 16-byte windows per patch row: >= RW_MAX and = 4 (mod 8), see above
constexpr int ROWB = NWIN * 16;            // bytes per patch row (960)
constexpr int NROW = 3 * IR;               // patch rows (69)
constexpr int PATCH_B = NROW * ROWB;       // 66,240
constexpr int POOL_B = (PH * TWP_MAX + 1) * 256;                  // 112 cells + 1 trash cell, 64 x fp32 bit patterns
constexpr int NLEFT_MAX = (RW_MAX + 8 * 7 + 31) / 32;             // leftover fragments (9th row + right-hand columns)
constexpr int TAB_B = NLEFT_MAX * 32 * 16;
constexpr int SB_LDS_BYTES = PATCH_B + POOL_B + TAB_B;            // 97,216
constexpr int KS = 11;                     // MFMAs per fragment and 32 channels: (c, kh) rows 2s and 2s + 1
constexpr int NF = 2;                      // 32-channel fragments

struct StemBArgs {
    const float* x;        // [N][3][H][W] fp32
    const uint4* wpack;    // [KS][NF][64] x 8 bf16: lane l of fragment nf, step s: W[ch = 32 nf + (l & 31)][(c, kh) = row 2s + (l >> 5)][kw 0..7]
    const float* scale;    // [64] folded BatchNorm scale
    const float* shift;    // [64] folded BatchNorm shift
    __bf16* out;           // [N][Hp][Wp][64] bf16
    int N, H, W, Hc, Wc, Hp, Wp;
    int TWp, tiles_x, tiles_y, RW, nfrag, total_tiles;
    int nbx, nblk, ncl;
    // fragments of every wave in processing order (0xFF = none): balanced on the host -- a leftover fragment (table-driven pooling,
    // up to 128 LDS atomics) costs about 2.5 block fragments, and dealt round-robin one wave ended up with 4.5 units against 2.7
    unsigned char wl[8][8];
};

// byte offset of patch row (c, kh) relative to a pixel's first row
__host__ __device__ constexpr int krow_off(int idx) { return ((idx / 7) * IR + idx % 7) * ROWB; }

__global__ __launch_bounds__(SB_NT) void stem_pool_bf16_kernel(StemBArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* patch = lds;
    unsigned* pooled = reinterpret_cast<unsigned*>(lds + PATCH_B);
    uint4* tab = reinterpret_cast<uint4*>(lds + PATCH_B + POOL_B);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    bf16x8 breg[KS][NF];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) breg[s][nf] = __builtin_bit_cast(bf16x8, a.wpack[(s * NF + nf) * 64 + lane]);
    float sc[NF], sh[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        sc[nf] = a.scale[32 * nf + (lane & 31)];
        sh[nf] = a.shift[32 * nf + (lane & 31)];
    }
    // see stem.hip: the operands must not look like pending VMEM results inside the tile loop
    __builtin_amdgcn_s_waitcnt(0x0F70);                                    // vmcnt(0)
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            uint4 t = __builtin_bit_cast(uint4, breg[s][nf]);
            asm volatile("" : "+v"(t.x), "+v"(t.y), "+v"(t.z), "+v"(t.w));
            breg[s][nf] = __builtin_bit_cast(bf16x8, t);
        }
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) asm volatile("" : "+v"(sc[nf]), "+v"(sh[nf]));
    const unsigned trash = (unsigned)(PH * a.TWp) * 256u;
    const unsigned lane_b = 4u * (unsigned)(lane & 31);
    constexpr int NIT = (NROW + 7) / 8;                                    // patch rows per wave (9)
    for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        const int n = tile / (a.tiles_y * a.tiles_x);
        const int tr = tile - n * (a.tiles_y * a.tiles_x);
        const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
        const int P0 = ty * PH, Q0 = tx * a.TWp;
        const int cy0 = 2 * P0 - 1, cx0 = 2 * Q0 - 1;            // first convolution row / column of the tile
        const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;          // first input row / column of the patch
        // ---- phase 1: stage the patch as bf16 windows, clear the pooled tile, build the leftover-pixel table.
        // Wave w converts patch rows w, w + 8, ...; lane m the input-column pair (2m, 2m + 1) of the row, which belongs to the
        // windows m - 3 .. m (at dword 3 .. 0).  Loads first, unconditional on clamped addresses, all in flight together.
        {
            float p0[NIT], p1[NIT];
            const float* img = a.x + (size_t)n * 3 * a.H * a.W;
            const int ix = ix0 + 2 * lane;
            const int ixa = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
            const int ixb = ix + 1 < 0 ? 0 : (ix + 1 >= a.W ? a.W - 1 : ix + 1);
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                int rr = wave + 8 * u;
                rr = rr < NROW ? rr : NROW - 1;
                const int c = rr / IR, r = rr - c * IR;
                int iy = iy0 + r;
                iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
                const float* rowp = img + ((size_t)c * a.H + iy) * a.W;
                p0[u] = rowp[ixa];
                p1[u] = rowp[ixb];
            }
            for (int i = tid; i < (PH * a.TWp + 1) * 16; i += SB_NT) reinterpret_cast<uint4*>(pooled)[i] = make_uint4(0u, 0u, 0u, 0u);
            for (int l = tid; l < (a.nfrag - a.nblk) * 32; l += SB_NT) {
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
            const bool oka = (unsigned)ix < (unsigned)a.W, okb = (unsigned)(ix + 1) < (unsigned)a.W;
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int rr = wave + 8 * u;                     // wave-uniform
                if (rr < NROW) {
                    const int r = rr - (rr / IR) * IR;
                    const bool rok = (unsigned)(iy0 + r) < (unsigned)a.H;
                    const bf16x2 pk = {(__bf16)(rok && oka ? p0[u] : 0.f), (__bf16)(rok && okb ? p1[u] : 0.f)};
                    const unsigned v = __builtin_bit_cast(unsigned, pk);
                    unsigned char* rowb = patch + rr * ROWB;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int ex = lane - t;
                        if (ex >= 0 && ex < a.RW) *reinterpret_cast<unsigned*>(rowb + ex * 16 + 4 * t) = v;
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase 2: 32 convolution pixels x 64 channels per fragment, fragments dealt round-robin to the waves (block
        // fragments = 4 rows x 8 columns, leftover pixels 32 in a row: stem.hip)
        for (int fi = 0; fi < 8; ++fi) {
            const int f = a.wl[wave][fi];
            if (f == 0xFF) break;
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
            const unsigned char* q = patch + (2 * oy * ROWB + 16 * ox);
            const unsigned char* qa = q + (h ? ROWB : 0);                 // half-wave 1 holds the next kernel row,
            const unsigned char* qb = q + (h ? (IR - 6) * ROWB : 0);      // the first row of the next plane (after kernel row 6),
            const unsigned char* qc = q;                                  // or nothing (the 22nd row: zero weights)
            f32x16 acc[NF];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nf][e] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int idx = 2 * s;
                const unsigned char* qq = s == KS - 1 ? qc : (idx % 7 == 6 ? qb : qa);
                const bf16x8 av = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qq + krow_off(idx)));
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) acc[nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, breg[s][nf], acc[nf], 0, 0, 0);
            }
            if (blk) {
                const int cyb = cy0 + 4 * by, cxs = cx0 + 8 * bx + 4 * h;
                bool rok[4], cok[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    rok[t] = (unsigned)(cyb + t) < (unsigned)a.Hc;
                    cok[t] = (unsigned)(cxs + t) < (unsigned)a.Wc;
                }
                const int ra = 2 * by, qa2 = 4 * bx + 2 * h;
                unsigned cell[3][3];
#pragma unroll
                for (int ri = 0; ri < 3; ++ri)
#pragma unroll
                    for (int qi = 0; qi < 3; ++qi) {
                        const int r = ra - 1 + ri, qq = qa2 - 1 + qi;
                        const bool ok = (unsigned)r < (unsigned)PH && P0 + r < a.Hp && (unsigned)qq < (unsigned)a.TWp && Q0 + qq < a.Wp;
                        cell[ri][qi] = (ok ? (unsigned)(r * a.TWp + qq) * 256u : trash) + lane_b;
                    }
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    float c[4][3];
#pragma unroll
                    for (int y = 0; y < 4; ++y) {
                        float v[4];
#pragma unroll
                        for (int xx = 0; xx < 4; ++xx) {
                            const float t = fmaxf(fmaf(acc[nf][4 * y + xx], sc[nf], sh[nf]), 0.f);
                            v[xx] = rok[y] && cok[xx] ? t : 0.f;
                        }
                        c[y][0] = v[0];
                        c[y][1] = fmaxf(fmaxf(v[0], v[1]), v[2]);
                        c[y][2] = fmaxf(v[2], v[3]);
                    }
#pragma unroll
                    for (int qi = 0; qi < 3; ++qi) {
                        const float m0 = c[0][qi];
                        const float m1 = fmaxf(fmaxf(c[0][qi], c[1][qi]), c[2][qi]);
                        const float m2 = fmaxf(c[2][qi], c[3][qi]);
                        const float m[3] = {m0, m1, m2};
#pragma unroll
                        for (int ri = 0; ri < 3; ++ri) {
                            if ((ri == 0 || qi == 0) && __builtin_amdgcn_ballot_w64(cell[ri][qi] != trash + lane_b) == 0) continue;
                            atomicMax(reinterpret_cast<unsigned*>(reinterpret_cast<char*>(pooled) + cell[ri][qi]) + 32 * nf,
                                      __float_as_uint(m[ri]));
                        }
                    }
                }
            } else {
                const int l0 = (f - a.nblk) * 32 + 4 * h;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint4 cells = tab[l0 + (e & 3) + 8 * (e >> 2)];
                    const unsigned v0 = __float_as_uint(fmaxf(fmaf(acc[0][e], sc[0], sh[0]), 0.f));
                    const unsigned v1 = __float_as_uint(fmaxf(fmaf(acc[1][e], sc[1], sh[1]), 0.f));
                    const unsigned co[4] = {cells.x, cells.y, cells.z, cells.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (k > 0 && __builtin_amdgcn_ballot_w64(co[k] != trash) == 0) continue;
                        unsigned* cellp = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(pooled) + (co[k] + lane_b));
                        atomicMax(cellp, v0);
                        atomicMax(cellp + 32, v1);
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase 3: flush the pooled tile as bf16 (8 channels = 16 bytes per thread)
        for (int i = tid; i < PH * a.TWp * 8; i += SB_NT) {
            const int c8 = i & 7, cell = i >> 3;
            const int r = cell / a.TWp, qq = cell - r * a.TWp;
            const int py = P0 + r, px = Q0 + qq;
            if (py < a.Hp && px < a.Wp) {
                const float4 lo = *reinterpret_cast<const float4*>(pooled + cell * 64 + 8 * c8);
                const float4 hi = *reinterpret_cast<const float4*>(pooled + cell * 64 + 8 * c8 + 4);
                const bf16x8 o = {(__bf16)lo.x, (__bf16)lo.y, (__bf16)lo.z, (__bf16)lo.w, (__bf16)hi.x, (__bf16)hi.y, (__bf16)hi.z, (__bf16)hi.w};
                *reinterpret_cast<uint4*>(a.out + ((((size_t)n * a.Hp + py) * a.Wp + px) * 64 + 8 * c8)) = __builtin_bit_cast(uint4, o);
            }
        }
        __syncthreads();
    }
}

}  // namespace

namespace rpg {

bool stem_pool_bf16_supported(int h, int w, int cout) { return cout == 64 && h >= 1 && w >= 1; }

// wpack: [11][2][64] x 8 bf16 (params.pack_stem_bf16); scale / shift: the folded BatchNorm affine (fp32)
int launch_stem_pool_bf16(const float* x_nchw, const void* wpack, const float* scale, const float* shift, void* out, int n, int h,
                          int w, hipStream_t s) {
    if (!x_nchw || !wpack || !scale || !shift || !out || n <= 0 || h <= 0 || w <= 0 || !aligned16(out) || !aligned16(wpack))
        return RPG_ERR_BAD_ARG;
    StemBArgs a{};
    a.x = x_nchw; a.wpack = reinterpret_cast<const uint4*>(wpack); a.scale = scale; a.shift = shift;
    a.out = reinterpret_cast<__bf16*>(out);
    a.N = n; a.H = h; a.W = w;
    a.Hc = (h + 6 - 7) / 2 + 1; a.Wc = (w + 6 - 7) / 2 + 1;
    a.Hp = (a.Hc + 2 - 3) / 2 + 1; a.Wp = (a.Wc + 2 - 3) / 2 + 1;
    a.tiles_x = (a.Wp + TWP_MAX - 1) / TWP_MAX;
    a.TWp = (a.Wp + a.tiles_x - 1) / a.tiles_x;
    a.tiles_y = (a.Hp + PH - 1) / PH;
    a.RW = 2 * a.TWp + 1;
    a.nbx = a.RW / 8;
    a.nblk = 2 * a.nbx;
    a.ncl = a.RW - 8 * a.nbx;
    a.nfrag = a.nblk + (a.RW + 8 * a.ncl + 31) / 32;
    const long total = (long)n * a.tiles_y * a.tiles_x;
    if (total >= (1L << 31) || (long)n * 3 * h * w >= (1L << 40)) return RPG_ERR_BAD_ARG;
    a.total_tiles = (int)total;
    {   // longest-processing-time-first: leftover fragments (cost 5) first, then block fragments (cost 2), each to the least loaded wave
        if (a.nfrag > 64) return RPG_ERR_BAD_ARG;
        int load[8] = {0}, cnt[8] = {0};
        for (int w8 = 0; w8 < 8; ++w8)
            for (int k = 0; k < 8; ++k) a.wl[w8][k] = 0xFF;
        for (int pass = 0; pass < 2; ++pass)
            for (int f = pass == 0 ? a.nblk : 0; f < (pass == 0 ? a.nfrag : a.nblk); ++f) {
                int best = 0;
                for (int w8 = 1; w8 < 8; ++w8)
                    if (load[w8] < load[best]) best = w8;
                if (cnt[best] >= 8) return RPG_ERR_BAD_ARG;
                a.wl[best][cnt[best]++] = (unsigned char)f;
                load[best] += pass == 0 ? 5 : 2;
            }
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    static bool attr[64] = {};
    if (!attr[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_pool_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  SB_LDS_BYTES);
        attr[dev] = true;
    }
    const int slot = timing_begin(RPG_TIMER_CONV, s);
    const int grid = (int)(total < num_cus() ? total : num_cus());
    hipLaunchKernelGGL(stem_pool_bf16_kernel, dim3(grid), dim3(SB_NT), SB_LDS_BYTES, s, a);
    // algorithmic: the 7x7x3 convolution on every output pixel; executed: fragments x 11 steps x 2 MFMAs x 32*32*16*2
    timing_end(slot, 2.0 * (double)n * a.Hc * a.Wc * 64.0 * 147.0, s, (double)total * a.nfrag * KS * NF * 32768.0);
    RPG_CHECK_LAUNCH("stem_conv_bn_relu_maxpool_bf16");
    return RPG_OK;
}

}  // namespace rpg

extern "C" int rpg_stem_conv7x7s2_bn_relu_maxpool_bf16(const float* x_nchw, const void* wpack_bf16, const float* scale,
                                                       const float* shift, void* y_nhwc_bf16, int n, int h, int w, void* stream) {
    return rpg::launch_stem_pool_bf16(x_nchw, wpack_bf16, scale, shift, y_nhwc_bf16, n, h, w, rpg::as_stream(stream));
}
