// The ResNet stem of the bf16 encoder as ONE kernel for gfx950 (round 3): conv 7x7 / stride 2 / pad 3 (3 -> 64 channels) on
// v_mfma_f32_32x32x16_bf16 + BatchNorm (eval) + ReLU + max-pool 3x3 / stride 2 / pad 1, reading the reference's fp32 NCHW
// input directly and writing the pooled NHWC tensor in bf16.
//
// Replaces (reference: torchvision resnet34 conv1 / bn1 / relu / maxpool, reached from
// /root/reference/python/niantic/modules/posenet.py:1037 after the view of :1035) what the bf16 encoder ran as three kernels up
// to round 2: fp32 NCHW -> bf16 NHWC8 re-layout (120 us at 512 images), the generic implicit-GEMM kernel on an 8-channel image
// (721 us: K = 7*7*8 = 392, five eighths of the MACs on zero channels) and a max-pool pass (228 us).
//
// A persistent 4-wave workgroup owns 2 pooled rows x TWp <= 28 pooled columns of one image at a time, i.e. 5 x (2 TWp + 1)
// convolution pixels; three workgroups share a CU (50 KB of LDS and <= 168 VGPRs each).
//   * K is ordered (channel, kernel row | kernel column 0..7): one MFMA = two (c, kh) rows of 8 kernel columns (the 8th has
//     zero weights), 21 rows -> 11 MFMAs per 32-pixel x 32-channel fragment (K = 176 issued for 147 taps);
//   * the input patch (3 planes x 15 rows x 119 columns) is staged in LDS as plain bf16 rows; a lane's pixel operand for a
//     kernel row is 8 consecutive columns starting at an even one: ONE ds_read_b128 at a 4-byte-aligned address;
//   * the weights stay in registers for the kernel's life (11 x 2 fragments x 4 VGPRs) and are the MFMA's A operand, so the
//     accumulators come out pixel-major (4 consecutive channels of one pixel per quad); BatchNorm scale and shift are applied in
//     fp32 to the accumulators (weights are the plain bf16 roundings, like every other bf16 convolution of the encoder);
//   * pooling is a separate pass over an LDS copy of the tile's bf16 convolution outputs.  bf16 rounding is monotone, so
//     rounding before the maximum gives the same bits as rounding after it.
// How it got here (512 images of 224 x 224, one launch; profiles/r3_stem_bf16_versions.txt): 524 us for the fp32 kernel's
// structure with bf16 operands (8 waves, 4 pooled rows, in-register pooling + LDS atomics, every 8-column window stored as its
// own aligned 16-byte chunk); 476 us with the pooling pass; 450 us with conflict-free window stores and the next tile's loads in
// flight during the MFMAs; 440 us as two 4-wave workgroups per CU.  Cycle counters per phase then showed a wave spending 17 k
// cycles per tile of which 1.6 k in MFMAs: every phase is a serial chain of a few hundred instructions on 2 waves per SIMD, and
// only occupancy hides it -- hence the plain-row patch (a quarter of the LDS, no lane shifts in phase 1), scale / shift in LDS
// instead of 64 VGPRs, three workgroups per CU: 403 us.
#include "rpg_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4u __attribute__((aligned(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SB_NT = 256;                 // 4 waves; THREE workgroups per CU (each other's loads, stores and barriers are hidden)
constexpr int SB_WGS_PER_CU = 3;
constexpr int NWV = SB_NT / 64;
constexpr int PH = 2;                      // pooled rows per tile
constexpr int CR = 2 * PH + 1;             // convolution rows per tile (5)
constexpr int IR = 2 * CR + 5;             // input rows per tile (15)
constexpr int TWP_MAX = 28;                // pooled columns per tile
constexpr int RW_MAX = 2 * TWP_MAX + 1;    // convolution columns per tile (57)
constexpr int ROWB = 128 * 2 + 16;         // bytes per patch row: 128 bf16 columns (>= 2 RW_MAX + 5 = 119), rows 4 banks apart
constexpr int NROW = 3 * IR;               // patch rows (45)
constexpr int NIT = (NROW + NWV - 1) / NWV;                       // patch rows per wave (12; rows 45..47 are written, never read)
constexpr int PATCH_B = NIT * NWV * ROWB;  // 13,056
constexpr int CONV_B = CR * RW_MAX * 128;  // the tile's convolution outputs after BN + ReLU, [row][column][64 ch] bf16: 36,480
constexpr int AFF_B = 2 * 64 * 4;          // BatchNorm scale | shift, fp32
constexpr int SB_LDS_BYTES = PATCH_B + CONV_B + AFF_B;            // 50,048
static_assert(SB_WGS_PER_CU * SB_LDS_BYTES <= 160 * 1024, "three workgroups per CU");
constexpr int KS = 11;                     // MFMAs per fragment and 32 channels: (c, kh) rows 2s and 2s + 1
constexpr int NF = 2;                      // 32-channel fragments

struct StemBArgs {
    const void* x;         // [N][3][H][W] fp32, or bf16 (kernel template parameter: the values a host-side rounding of the fp32 input gives)
    const uint4* wpack;    // [KS][NF][64] x 8 bf16: lane l of fragment nf, step s: W[ch = 32 nf + (l & 31)][(c, kh) = row 2s + (l >> 5)][kw 0..7]
    const float* scale;    // [64] folded BatchNorm scale
    const float* shift;    // [64] folded BatchNorm shift
    __bf16* out;           // [N][Hp][Wp][64] bf16
    int N, H, W, Hc, Wc, Hp, Wp;
    int TWp, tiles_x, tiles_y, RW, nfrag;
    int nxcd;              // 8: block b works on the images n = b % 8 (mod 8) -- see the tile loop; 1: tiles in plain order
    unsigned mg_rw, mg_tpi, mg_tx;      // floor(2^32 / d) + 1 for d = RW, tiles per image, tiles_x: the tile loop's divisions as one v_mul_hi
};                                      // each (numerators: a pixel of the tile < 2^12, a tile index < 2^22; round 4)

// byte offset of patch row (c, kh) relative to a pixel's first row
__host__ __device__ constexpr int krow_off(int idx) { return ((idx / 7) * IR + idx % 7) * ROWB; }

template <typename TIn>
__global__ __launch_bounds__(SB_NT) __attribute__((amdgpu_waves_per_eu(3, 3))) void stem_pool_bf16_kernel(StemBArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* patch = lds;
    unsigned char* convt = lds + PATCH_B;
    float* aff = reinterpret_cast<float*>(lds + PATCH_B + CONV_B);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    bf16x8 breg[KS][NF];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) breg[s][nf] = __builtin_bit_cast(bf16x8, a.wpack[(s * NF + nf) * 64 + lane]);
    if (tid < 128) aff[tid] = tid < 64 ? a.scale[tid] : a.shift[tid - 64];
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
    // Tile order.  Neighbouring tiles of an image share input rows (15 rows read per 8 rows owned), and the hardware deals
    // consecutive workgroups to the 8 XCDs round-robin, each with its own L2.  Block b works only on the images n = b (mod 8),
    // its XCD's, in order: the rows a tile shares with the previous one are then still in that L2.
    const int tpi = a.tiles_y * a.tiles_x;
    const int xcd = blockIdx.x % a.nxcd, nslot = gridDim.x / a.nxcd;
    for (int lt = blockIdx.x / a.nxcd;; lt += nslot) {
        const int li = a.mg_tpi ? (int)__umulhi((unsigned)lt, a.mg_tpi) : lt, tr = lt - li * tpi;       // (magic 0 = divisor 1)
        const int n = xcd + a.nxcd * li;
        if (n >= a.N) break;
        const int ty = a.mg_tx ? (int)__umulhi((unsigned)tr, a.mg_tx) : tr, tx = tr - ty * a.tiles_x;
        const int P0 = ty * PH, Q0 = tx * a.TWp;
        const int cy0 = 2 * P0 - 1, cx0 = 2 * Q0 - 1;            // first convolution row / column of the tile
        const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;          // first input row / column of the patch
        // ---- phase 1: the patch, [plane x input row][128 input columns] in bf16.  Wave w stages patch rows w, w + 4, ...; lane m
        // loads the columns m and 64 + m of the row (unconditional, fully coalesced loads on clamped addresses, all in flight
        // together) and stores them as bf16.
        {
            float p0[NIT], p1[NIT];
            const TIn* img = reinterpret_cast<const TIn*>(a.x) + (size_t)n * 3 * a.H * a.W;
            const int ix = ix0 + lane;
            const int ixa = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
            const int ixb = ix + 64 < 0 ? 0 : (ix + 64 >= a.W ? a.W - 1 : ix + 64);
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                int rr = wave + NWV * u;
                rr = rr < NROW ? rr : NROW - 1;
                const int c = rr / IR, r = rr - c * IR;
                int iy = iy0 + r;
                iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
                const TIn* rowp = img + ((size_t)c * a.H + iy) * a.W;
                p0[u] = (float)rowp[ixa];
                p1[u] = (float)rowp[ixb];
            }
            const bool oka = (unsigned)ix < (unsigned)a.W, okb = (unsigned)(ix + 64) < (unsigned)a.W;
            unsigned char* wp = patch + (wave * ROWB + lane * 2);
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int rr = wave + NWV * u;                   // wave-uniform; rows >= NROW hold junk nobody reads
                const int r = rr - (rr / IR) * IR;
                const bool rok = (unsigned)(iy0 + r) < (unsigned)a.H;
                *reinterpret_cast<__bf16*>(wp + u * (NWV * ROWB)) = (__bf16)(rok && oka ? p0[u] : 0.f);
                *reinterpret_cast<__bf16*>(wp + u * (NWV * ROWB) + 128) = (__bf16)(rok && okb ? p1[u] : 0.f);
            }
        }
        __syncthreads();
        // ---- phase 2: convolution + BN + ReLU of the CR x RW pixels under the tile, 32 consecutive pixels (row-major) x 64
        // channels per fragment; results go to the LDS tile as bf16.
        //   * A lane's pixel operand for kernel row (c, kh) is the 8 input columns 2 ox - 3 .. 2 ox + 4 of that patch row: 16 bytes
        //     at a 4-BYTE-aligned address (ds_read_b128 takes it; neighbouring lanes overlap in 12 of their 16 bytes, which the
        //     LDS serves as broadcasts).  The first versions stored every such window as its own aligned 16-byte chunk -- a patch
        //     four times the size, and a phase 1 of 3 lane shifts + a 16-byte store per row that cost as much as the MFMAs.
        //   * The weights are the MFMA's A operand, so a lane ends up with 4 consecutive channels of ONE pixel per accumulator
        //     quad: 8-byte stores.  The 16-byte chunks of a pixel's 128-byte row are XOR-swizzled with the pixel index (the 32
        //     lanes of a store are 32 different pixels; unswizzled they would all hit the same banks).
        //   * 9 fragments over 4 waves: the wave with three changes from tile to tile (the workgroups of a CU put their waves w on
        //     the same SIMD).
        const int npix = CR * a.RW;
        for (int f = (wave - lt) & (NWV - 1); f < a.nfrag; f += NWV) {
            const int pl = f * 32 + (lane & 31);
            const int p = pl < npix ? pl : npix - 1;             // padding lanes of the last fragment: a valid pixel, not stored
            const int oy = a.mg_rw ? (int)__umulhi((unsigned)p, a.mg_rw) : p, ox = p - oy * a.RW;
            const unsigned char* q = patch + (2 * oy * ROWB + 4 * ox);
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
                const bf16x8 av = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4u*>(qq + krow_off(idx)));
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) acc[nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(breg[s][nf], av, acc[nf], 0, 0, 0);
            }
            if (pl < npix) {
                // accumulator element 4 g + k of fragment nf = channel 32 nf + 8 g + 4 h + k
                unsigned char* cp = convt + pl * 128 + 8 * h;
#pragma unroll
                for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(aff + 32 * nf + 8 * g + 4 * h);
                        const f32x4 sh = *reinterpret_cast<const f32x4*>(aff + 64 + 32 * nf + 8 * g + 4 * h);
                        bf16x4 v;
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = (__bf16)fmaxf(fmaf(acc[nf][4 * g + k], sc[k], sh[k]), 0.f);
                        *reinterpret_cast<bf16x4*>(cp + (((4 * nf + g) ^ (pl & 7)) << 4)) = v;
                    }
            }
        }
        __syncthreads();
        // ---- phase 3: max-pool 3x3 / 2 from the LDS tile straight to global memory, 8 channels (16 bytes) per thread and cell.
        // Pooled cell (r, q) of the tile covers convolution rows 2r .. 2r + 2, columns 2q .. 2q + 2 (tile-local); positions outside
        // the image are replaced by a valid neighbour of the same window (a duplicate does not change a maximum), so no masks.
        // Post-ReLU values are >= 0: the unsigned 16-bit maximum of the bit patterns is the bf16 maximum.
        {
            // valid range of tile-local convolution rows / columns: the image's [0, Hc) x [0, Wc)
            const int ylo = cy0 < 0 ? -cy0 : 0, yhi = (a.Hc - 1 - cy0) < (CR - 1) ? (a.Hc - 1 - cy0) : (CR - 1);
            const int xlo = cx0 < 0 ? -cx0 : 0, xhi = (a.Wc - 1 - cx0) < (a.RW - 1) ? (a.Wc - 1 - cx0) : (a.RW - 1);
            for (int i = tid; i < PH * a.TWp * 8; i += SB_NT) {
                const int c8 = i & 7, cell = i >> 3;
                const int r = cell >= a.TWp ? 1 : 0, qq = cell - r * a.TWp;
                static_assert(PH == 2, "row of a cell");
                const int py = P0 + r, px = Q0 + qq;
                if (py < a.Hp && px < a.Wp) {
                    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
                    u16x8 m = {0, 0, 0, 0, 0, 0, 0, 0};
                    int xs[3];
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int xq = 2 * qq + dx;
                        xs[dx] = xq < xlo ? xlo : (xq > xhi ? xhi : xq);
                    }
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        int y = 2 * r + dy;
                        y = y < ylo ? ylo : (y > yhi ? yhi : y);
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const int pix = y * a.RW + xs[dx];
                            const u16x8 v = __builtin_bit_cast(u16x8, *reinterpret_cast<const uint4*>(convt + pix * 128 + ((c8 ^ (pix & 7)) << 4)));
                            m = __builtin_elementwise_max(m, v);
                        }
                    }
                    *reinterpret_cast<uint4*>(a.out + ((((size_t)n * a.Hp + py) * a.Wp + px) * 64 + 8 * c8)) = __builtin_bit_cast(uint4, m);
                }
            }
        }
        __syncthreads();      // the next tile's phase 2 writes the convolution tile (its phase 1 only the patch, but waves drift)
    }
}


// ================================================================================================================================
// Round 6: the same stem as STRIPS MARCHING DOWN THE IMAGE -- no input patch in LDS, no convolution tile in LDS, no barrier.
//
// Why: the tile kernel above spends a wave's life in serial phases (stage the patch -> barrier -> 2-3 fragments of 22 MFMAs +
// BN / rounding / LDS stores -> barrier -> pooling pass -> barrier): 21 k cycles per tile of which 1.6 k are MFMAs, and only
// three resident workgroups per CU hide it (r3-r5: five versions, every one bound by that chain, MFMA-busy 0.24).  Here a WAVE
// owns 32 convolution columns x 32 output channels of one band of an image and walks down the convolution rows:
//   * K is ordered (channel, kernel-row PAIR j, | 8 input columns): 3 x 4 MFMAs per convolution row (kernel row 7 and input
//     column -1 of the window carry zero weights: K = 192 issued for 147 taps).  The pixel operand of (c, j) for convolution row
//     oy is the input rows 2 oy - 3 + 2 j (+ 1 in lanes 32-63), columns 2 ox - 4 .. 2 ox + 3 -- which is exactly the operand of
//     (c, j + 1) of row oy - 1: the 12 operands live in REGISTERS and rotate; a convolution row loads ONE new row pair per
//     channel (3 operands instead of 11), straight from global memory into the lanes that need it (raw buffer loads, 8 consecutive
//     columns per lane; neighbouring lanes overlap and are served by the vector L1), one row ahead of its use;
//   * the weights are the MFMA's A operand and stay in registers (12 x 4 VGPRs per 32 channels): accumulators come out
//     pixel-major, BatchNorm in fp32, rounded to bf16, and the 3 x 3 / 2 max-pool happens IN REGISTERS: vertically a running
//     maximum over the rows of a pooled row (v_pk_max_i16 on the bf16 bits against a maximum that starts at +0: that is the ReLU,
//     and bf16 rounding is monotone, so the bits equal round(max)); horizontally the lanes are ordered even columns | odd columns,
//     so the three columns of a pooled cell are this lane, its right neighbour (DPP row_shl:1) and lane ^ 16 (ds_swizzle);
//   * edges: image columns outside [0, W) are masked to zero after the conversion in the first / last strip only (lane-constant
//     masks), rows outside [0, H) in a wave-uniform slow path; convolution columns outside [0, Wc) are clamped duplicates of a
//     column of the same pooling window (a duplicate does not change a maximum).
// Work item = (image, band of BH pooled rows, strip of <= 15 pooled columns, 32-channel half); four items per workgroup, no LDS
// beyond the BatchNorm table.
// ================================================================================================================================
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two floats -> one register of two bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32 (element-wise casts compile to two
// conversions and a v_perm)
__device__ __forceinline__ unsigned pk2_bf16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
constexpr int SS_NT = 256;
constexpr int SS_TP = 15;                  // pooled columns per strip: 2 * 15 + 1 = 31 of the 32 convolution columns
struct StemSArgs {
    const void* x;
    const uint4* wpack;    // [2 nf][3 c][4 j][64 lanes] x 8 bf16 (params.pack_stem_bf16, second part)
    const float* scale;
    const float* shift;
    __bf16* out;
    int N, H, W, Hc, Wc, Hp, Wp;
    int TP, tiles_x, BH, bands;
    unsigned items;        // N * bands * tiles_x * (2 / NF)
};

// NF: 32-channel halves per wave (2, the default: one work item per strip -- half the input loads and conversions per MFMA, two
// accumulator chains, two waves per SIMD; 1: two work items per strip, three waves per SIMD).  The weights live in LDS (24 KB per
// workgroup, one ds_read_b128 per MFMA, read two MFMAs ahead).  Measured and removed again (profiles/r6_stem_bf16_variants.txt, 512
// images): the weights in 48 registers instead of LDS (264 us against 255 for NF = 1), four waves per SIMD (260).
template <typename TIn, int NF>
__global__ __launch_bounds__(SS_NT) __attribute__((amdgpu_waves_per_eu(NF == 1 ? 3 : 2, NF == 1 ? 3 : 2))) void stem_strip_bf16_kernel(StemSArgs a) {
    constexpr int ESZ = (int)sizeof(TIn);
    constexpr int NLD = ESZ == 4 ? 2 : 1;                 // 16-byte loads per 8-column window
    constexpr unsigned SENT = 0x80000000u;
    __shared__ __attribute__((aligned(16))) float aff[128];
    const int tid = threadIdx.x, lane = tid & 63, l = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ uint4 wl[2 * 12 * 64];
    if (tid < 128) aff[tid] = tid < 64 ? a.scale[tid] : a.shift[tid - 64];
#pragma unroll
    for (int i = 0; i < 2 * 12 * 64 / SS_NT; ++i) wl[tid + i * SS_NT] = a.wpack[tid + i * SS_NT];
    __syncthreads();
    // workgroups b and b + 8 sit on the same XCD: give them neighbouring items (the strips of one band share their input rows)
    const unsigned b = blockIdx.x, nb = gridDim.x;
    unsigned wg = b;
    if ((nb & 15u) == 0) { const unsigned xcd = b & 7u, k = b >> 3; wg = ((k >> 1) * 8u + xcd) * 2u + (k & 1u); }
    const unsigned item = wg * 4u + (unsigned)wave;
    if (item >= a.items) return;
    const int nf = NF == 1 ? (int)(item & 1u) : 0;        // first 32-channel half of this wave
    unsigned rest = NF == 1 ? item >> 1 : item;
    const int strip = (int)(rest % (unsigned)a.tiles_x); rest /= (unsigned)a.tiles_x;
    const int band = (int)(rest % (unsigned)a.bands);
    const int n = (int)(rest / (unsigned)a.bands);
    const int Q0 = strip * a.TP, P0 = band * a.BH;
    const int nq = a.TP < a.Wp - Q0 ? a.TP : a.Wp - Q0, np = a.BH < a.Hp - P0 ? a.BH : a.Hp - P0;
    const int H = a.H, W = a.W;

    // ---- lane geometry: lanes 0-15 even convolution columns of the strip, 16-31 odd ones (same in both half-waves)
    const int u = l < 16 ? 2 * l : 2 * (l - 16) + 1;
    int cx = 2 * Q0 - 1 + u;
    cx = cx < 0 ? 0 : (cx >= a.Wc ? a.Wc - 1 : cx);
    const int col0 = 2 * cx - 4;                          // first input column of the lane's 8-column window (zero-weight tap)
    // byte offset of the window inside an input row; lanes 32-63 read the NEXT row.  The row's own offset is added per load (in the
    // vector offset: the descriptor's range check covers vector offset only, and the records end with the image)
    const size_t img_elems = (size_t)3 * H * W;
    // bf16 images of odd size start at a 2-byte boundary for odd n: the descriptor starts at the dword below (`mis` elements early)
    const int mis = ESZ == 2 ? (int)(((size_t)n * img_elems) & 1) : 0;
    const unsigned vbase = (unsigned)((col0 + h * W + mis) * ESZ);
    // The one place where such a window starts in front of the descriptor's base is row 0 of plane 0 (columns -4 .. -1 of the first
    // strip; with images narrower than 4 pixels the first rows of the other planes too): a negative vector offset is out of range as
    // a whole, also for the part of a 16-byte load that lies inside the row (offset + instruction offset is not evaluated modulo
    // 2^32).  Loads of a row pair that starts less than 4 elements into the image begin at column 0 instead and the registers are
    // moved `lsh` bf16 pairs up afterwards -- exact for any row (wave-uniform slow path, once per image).
    const int lsh = col0 < 0 ? (-col0) >> 1 : 0;          // 0, 1 or 2 pairs
    unsigned cmask[4];                                    // column validity of the four bf16 pairs of the window
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c0 = col0 + 2 * i;
        cmask[i] = ((unsigned)c0 < (unsigned)W ? 0x0000ffffu : 0u) | ((unsigned)(c0 + 1) < (unsigned)W ? 0xffff0000u : 0u);
    }
    const int cfirst = 2 * (2 * Q0 - 1) - 4, clast = 2 * (2 * Q0 - 1 + 31) + 3;
    const bool edge_cols = cfirst < 0 || clast >= W;      // wave-uniform: only the first / last strip mask
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<TIn*>(reinterpret_cast<const TIn*>(a.x) + (size_t)n * img_elems - mis), 0, (unsigned)(((img_elems + mis) * ESZ + 3) & ~(size_t)3), 0x00020000);       // (whole dwords: the last bf16 of an odd-sized image shares its dword with 2 bytes
                                                                                     //  of the next image / of the allocation's padding)

    // ---- weights: A operands, resident
    const uint4* const wl_lane = wl + nf * 12 * 64 + lane;
    auto weight = [&](int f, int k) -> bf16x8 {           // fragment of half nf + f, MFMA k = 4 c + j
        return __builtin_bit_cast(bf16x8, wl_lane[(f * 12 + k) * 64]);
    };

    bf16x8 op[3][4];                                      // pixel operands; logical pair j of row oy lives in slot (j + oy) & 3
    uint4 raw[3][NLD];                                    // the row pair in flight
    // bf16 input of ODD width: every other row starts at a 2-byte boundary, and dword loads need 4: such launches load 5 dwords
    // from the aligned address below and funnel-shift them by 16 bits in the lanes whose row is the misaligned one
    unsigned raw5[3];
    const bool odd_w = ESZ == 2 && (W & 1);
    auto issue = [&](int oy) {                            // the new pair of convolution row oy: input rows 2 oy + 3 | 2 oy + 4
        const int ra = 2 * oy + 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            unsigned voff = vbase + (unsigned)(((c * H + ra) * W) * ESZ);
            if ((c * H + ra) * W < 4) voff += (unsigned)(lsh * 2 * ESZ);
            if (odd_w) {
                voff &= ~3u;
                raw5[c] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff + 16u, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < NLD; ++q) raw[c][q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16u * q, 0, 0));
        }
    };
    auto pack_pair = [&](int c, int oy) -> bf16x8 {       // raw -> the MFMA operand (bf16 x 8), edges zeroed
        uint4 p;
        if constexpr (ESZ == 4) {
            const float4 f0 = __builtin_bit_cast(float4, raw[c][0]), f1 = __builtin_bit_cast(float4, raw[c][NLD - 1]);
            p.x = pk2_bf16(f0.x, f0.y); p.y = pk2_bf16(f0.z, f0.w);
            p.z = pk2_bf16(f1.x, f1.y); p.w = pk2_bf16(f1.z, f1.w);
        } else {
            p = raw[c][0];
            if (odd_w) {                                  // element offset of the row odd <=> it starts 2 bytes past a dword
                const unsigned sh = (unsigned)(((mis + (c * H + 2 * oy + 3 + h) * W) & 1) << 4);
                p.x = __builtin_amdgcn_alignbit(p.y, p.x, sh);
                p.y = __builtin_amdgcn_alignbit(p.z, p.y, sh);
                p.z = __builtin_amdgcn_alignbit(p.w, p.z, sh);
                p.w = __builtin_amdgcn_alignbit(raw5[c], p.w, sh);
            }
        }
        const int ra = 2 * oy + 3;
        if ((c * H + ra) * W < 4) {                        // the shifted loads of the image's first row(s): pairs back to their places
            uint4 q;
            q.x = lsh == 0 ? p.x : 0u;
            q.y = lsh == 0 ? p.y : (lsh == 1 ? p.x : 0u);
            q.z = lsh == 0 ? p.z : (lsh == 1 ? p.y : p.x);
            q.w = lsh == 0 ? p.w : (lsh == 1 ? p.z : p.y);
            p = q;
        }
        if (edge_cols) { p.x &= cmask[0]; p.y &= cmask[1]; p.z &= cmask[2]; p.w &= cmask[3]; }
        if (ra < 0 || ra + 1 >= H) {                      // a row of the pair lies outside the image (first / last rows only)
            const bool keep = (unsigned)(ra + h) < (unsigned)H;
            p.x = keep ? p.x : 0u; p.y = keep ? p.y : 0u; p.z = keep ? p.z : 0u; p.w = keep ? p.w : 0u;
        }
        return __builtin_bit_cast(bf16x8, p);
    };

    typedef unsigned u32;
    u32 mx[NF][8];                                        // running maximum of the pooled row in progress: bf16 pairs, >= +0
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int i = 0; i < 8; ++i) mx[f][i] = 0u;
    auto pk_max = [](u32 x, u32 y) -> u32 {
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(s16x2, x), __builtin_bit_cast(s16x2, y)));
    };
    __bf16* const out_n = a.out + (size_t)n * a.Hp * a.Wp * 64 + 32 * nf + 4 * h;
    auto emit = [&](int py) {                             // horizontal 3-maximum + store of pooled row py (lanes l < nq hold column Q0 + l)
        if (py < P0 || py >= P0 + np) return;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            u32 r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const u32 right = (u32)__builtin_amdgcn_update_dpp(0, (int)mx[f][i], 0x101, 0xf, 0xf, true);       // row_shl:1: lane i <- lane i + 1
                const u32 odd = (u32)__builtin_amdgcn_ds_swizzle((int)mx[f][i], 0x401f);                            // lane i <- lane i ^ 16
                r[i] = pk_max(pk_max(mx[f][i], right), odd);
            }
            if (l < nq) {
                __bf16* o = out_n + ((size_t)py * a.Wp + Q0 + l) * 64 + 32 * f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 v; v.x = r[2 * g]; v.y = r[2 * g + 1];
                    *reinterpret_cast<uint2*>(o + 8 * g) = v;
                }
            }
        }
    };

    const int oy_first = P0 > 0 ? 2 * P0 - 1 : 0;
    int oy_last = 2 * (P0 + np) - 1;
    oy_last = oy_last < a.Hc ? oy_last : a.Hc - 1;
    // One convolution row, rotation phase R = oy & 3 (static register names: the loop below is unrolled four rows deep and starts
    // at a multiple of four).  Rows before oy_first only bring the first row's pairs j = 0..2 into their slots.
    // MFMA k = 4 c + j; the weight fragment of MFMA k + 2 is read behind MFMA k (three fragment registers in rotation).
#define RPG_SS_MFMA(K, R)                                                                                                      \
    do {                                                                                                                       \
        _Pragma("unroll") for (int f = 0; f < NF; ++f) {                                                                       \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[f][(K) % 3], op[(K) / 4][(((K) % 4) + (R)) & 3], acc[f], 0, 0, 0); \
            if ((K) + 2 < 12) wf[f][((K) + 2) % 3] = weight(f, (K) + 2);                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                                                 \
        }                                                                                                                      \
    } while (0)
#define RPG_SS_ROW(R)                                                                                                          \
    do {                                                                                                                       \
        _Pragma("unroll") for (int c = 0; c < 3; ++c) op[c][(3 + (R)) & 3] = pack_pair(c, oy);                                 \
        issue(oy + 1);         /* (past the last row too: range-checked, never used) */                                        \
        if (oy >= oy_first) {                                                                                                  \
            f32x16 acc[NF];                                                                                                    \
            bf16x8 wf[NF][3];                                                                                                  \
            _Pragma("unroll") for (int f = 0; f < NF; ++f) {                                                                   \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) acc[f][e] = 0.f;                                                \
                wf[f][0] = weight(f, 0); wf[f][1] = weight(f, 1);                                                              \
            }                                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                                 \
            RPG_SS_MFMA(0, R); RPG_SS_MFMA(1, R); RPG_SS_MFMA(2, R); RPG_SS_MFMA(3, R);                                        \
            RPG_SS_MFMA(4, R); RPG_SS_MFMA(5, R); RPG_SS_MFMA(6, R); RPG_SS_MFMA(7, R);                                        \
            RPG_SS_MFMA(8, R); RPG_SS_MFMA(9, R); RPG_SS_MFMA(10, R); RPG_SS_MFMA(11, R);                                      \
            u32 pv[NF][8];                                                                                                     \
            _Pragma("unroll") for (int f = 0; f < NF; ++f) {                                                                   \
                _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                \
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(aff + 32 * (nf + f) + 8 * g + 4 * h);                     \
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(aff + 64 + 32 * (nf + f) + 8 * g + 4 * h);                \
                    pv[f][2 * g] = pk2_bf16(fmaf(acc[f][4 * g + 0], sc[0], sh[0]), fmaf(acc[f][4 * g + 1], sc[1], sh[1]));     \
                    pv[f][2 * g + 1] = pk2_bf16(fmaf(acc[f][4 * g + 2], sc[2], sh[2]), fmaf(acc[f][4 * g + 3], sc[3], sh[3])); \
                }                                                                                                              \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) mx[f][i] = pk_max(mx[f][i], pv[f][i]);                           \
            }                                                                                                                  \
            if ((R) & 1) {                                                                                                     \
                emit((oy - 1) >> 1);                                                                                           \
                _Pragma("unroll") for (int f = 0; f < NF; ++f)                                                                 \
                    _Pragma("unroll") for (int i = 0; i < 8; ++i) mx[f][i] = pk_max(pv[f][i], 0u);                             \
            }                                                                                                                  \
        }                                                                                                                      \
        ++oy;                                                                                                                  \
    } while (0)

    int oy = (oy_first - 3) & ~3;                         // a multiple of four at or below the first warm-up row (two's complement)
    issue(oy);
    while (oy <= oy_last) {
        RPG_SS_ROW(0);
        if (oy > oy_last) break;
        RPG_SS_ROW(1);
        if (oy > oy_last) break;
        RPG_SS_ROW(2);
        if (oy > oy_last) break;
        RPG_SS_ROW(3);
    }
#undef RPG_SS_ROW
#undef RPG_SS_MFMA
    if (!(oy_last & 1)) emit(oy_last >> 1);               // odd Hc: the last pooled row ends on an even convolution row
}

}  // namespace

namespace rpg {

// tile geometry + the ranges the kernel's magic divisions are exact for; false = this (n, h, w) is not for the fused kernel
// (the composite forward then takes the three-kernel stem: ADVICE r4 -- the launcher used to fail with RPG_ERR_BAD_ARG instead)
static bool stem_pool_bf16_geometry(int n, int h, int w, StemBArgs& a, int& grid) {
    if (n <= 0 || h <= 0 || w <= 0) return false;
    a.N = n; a.H = h; a.W = w;
    a.Hc = (h + 6 - 7) / 2 + 1; a.Wc = (w + 6 - 7) / 2 + 1;
    a.Hp = (a.Hc + 2 - 3) / 2 + 1; a.Wp = (a.Wc + 2 - 3) / 2 + 1;
    a.tiles_x = (a.Wp + TWP_MAX - 1) / TWP_MAX;
    a.TWp = (a.Wp + a.tiles_x - 1) / a.tiles_x;
    a.tiles_y = (a.Hp + PH - 1) / PH;
    a.RW = 2 * a.TWp + 1;
    a.nfrag = (CR * a.RW + 31) / 32;
    const long total = (long)n * a.tiles_y * a.tiles_x;
    if (total >= (1L << 31) || (long)n * 3 * h * w >= (1L << 40)) return false;
    grid = (int)(total < (long)SB_WGS_PER_CU * num_cus() ? total : (long)SB_WGS_PER_CU * num_cus());
    a.nxcd = (grid % 8 == 0 && n >= 64) ? 8 : 1;
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((1ULL << 32) / (unsigned)d) + 1u; };      // 0: divisor 1 (2^32 + 1 does not fit)
    a.mg_rw = magic(a.RW); a.mg_tpi = magic(a.tiles_y * a.tiles_x); a.mg_tx = magic(a.tiles_x);
    // v_mul_hi exactness needs numerator * divisor < 2^32: the tile index lt < (images + grid) * tiles per image
    return (long)((long)n / a.nxcd + 1 + grid) * a.tiles_y * a.tiles_x * a.tiles_y * a.tiles_x < (1L << 32);
}

int g_stem_strip = 9;          // 0: the tile kernel (rounds 3-5) | bit 0: the strip-march kernel of round 6, + bit 3 (default): both 32-channel halves in
                               // one wave (measured at 512 images, profiles/r6_stem_bf16_variants.txt: 227 us against 255-265 with one half per wave
                               // and three waves per SIMD -- the texture addresser was 72 % busy with every input window loaded by two waves --,
                               // 375 for the tile kernel)
int g_stem_strip_bh = 0;       // pooled rows per band (RPG_TUNE_FUSED_STEM value >> 4, experiments); 0 = by the launch's size
void bf16_set_stem_strip(int mode, int bh) { g_stem_strip = mode; g_stem_strip_bh = bh > 0 ? bh : 0; }

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

static bool stem_strip_geometry(int n, int h, int w, int esz, StemSArgs& a, int& grid) {
    if (n <= 0 || h <= 0 || w <= 0) return false;
    a.N = n; a.H = h; a.W = w;
    a.Hc = (h + 6 - 7) / 2 + 1; a.Wc = (w + 6 - 7) / 2 + 1;
    a.Hp = (a.Hc + 2 - 3) / 2 + 1; a.Wp = (a.Wc + 2 - 3) / 2 + 1;
    a.tiles_x = (a.Wp + SS_TP - 1) / SS_TP;
    a.TP = (a.Wp + a.tiles_x - 1) / a.tiles_x;
    if (g_stem_strip_bh > 0) {
        a.BH = g_stem_strip_bh < a.Hp ? g_stem_strip_bh : a.Hp;
    } else {
        a.BH = strip_band_rows((long)n * a.tiles_x * (g_stem_strip & 8 ? 1 : 2), a.Hp, (g_stem_strip & 8 ? 2L : 3L) * num_cus());
    }
    a.bands = (a.Hp + a.BH - 1) / a.BH;
    const long items = (long)n * a.bands * a.tiles_x * (g_stem_strip & 8 ? 1 : 2);        // bit 3: both channel halves in one wave
    // 32-bit byte offsets inside an image (incl. the rows read past its ends), 32-bit item index
    if ((long)3 * h * w * esz + (long)16 * w * esz >= (1L << 31) || items + 64 >= (1L << 32)) return false;
    a.items = (unsigned)items;
    grid = (int)((items + 3) / 4);
    return true;
}

bool stem_pool_bf16_supported(int n, int h, int w, int cout) {
    StemBArgs a{};
    int grid = 0;
    return cout == 64 && stem_pool_bf16_geometry(n, h, w, a, grid);
}

// wpack: [11][2][64] x 8 bf16 (params.pack_stem_bf16); scale / shift: the folded BatchNorm affine (fp32)
int launch_stem_pool_bf16(const void* x_nchw, int x_is_bf16, const void* wpack, const float* scale, const float* shift, void* out,
                          int n, int h, int w, hipStream_t s) {
    if (!x_nchw || !wpack || !scale || !shift || !out || n <= 0 || h <= 0 || w <= 0 || !aligned16(out) || !aligned16(wpack))
        return RPG_ERR_BAD_ARG;
    if (g_stem_strip) {
        StemSArgs sa{};
        int sgrid = 0;
        if (stem_strip_geometry(n, h, w, x_is_bf16 ? 2 : 4, sa, sgrid)) {
            sa.x = x_nchw; sa.scale = scale; sa.shift = shift; sa.out = reinterpret_cast<__bf16*>(out);
            sa.wpack = reinterpret_cast<const uint4*>(wpack) + KS * NF * 64;       // second part of params.pack_stem_bf16
            const int slot = timing_begin(RPG_TIMER_CONV, s);
            if (g_stem_strip & 8) {
                if (x_is_bf16) hipLaunchKernelGGL((stem_strip_bf16_kernel<__bf16, 2>), dim3(sgrid), dim3(SS_NT), 0, s, sa);
                else hipLaunchKernelGGL((stem_strip_bf16_kernel<float, 2>), dim3(sgrid), dim3(SS_NT), 0, s, sa);
            } else {
                if (x_is_bf16) hipLaunchKernelGGL((stem_strip_bf16_kernel<__bf16, 1>), dim3(sgrid), dim3(SS_NT), 0, s, sa);
                else hipLaunchKernelGGL((stem_strip_bf16_kernel<float, 1>), dim3(sgrid), dim3(SS_NT), 0, s, sa);
            }
            // executed: per item 2 np + 1 convolution rows x 12 MFMAs of 32 x 32 x 16
            timing_end(slot, 2.0 * (double)n * sa.Hc * sa.Wc * 64.0 * 147.0, s,
                       (double)sa.items * (2.0 * sa.Hp / sa.bands + 1.0) * 12.0 * 32768.0);
            RPG_CHECK_LAUNCH("stem_conv_bn_relu_maxpool_bf16 (strips)");
            return RPG_OK;
        }
    }
    StemBArgs a{};
    int grid = 0;
    if (!stem_pool_bf16_geometry(n, h, w, a, grid)) return RPG_ERR_BAD_ARG;
    const long total = (long)n * a.tiles_y * a.tiles_x;
    a.x = x_nchw; a.wpack = reinterpret_cast<const uint4*>(wpack); a.scale = scale; a.shift = shift;
    a.out = reinterpret_cast<__bf16*>(out);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    static bool attr[64] = {};
    if (!attr[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_pool_bf16_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  SB_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_pool_bf16_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  SB_LDS_BYTES);
        attr[dev] = true;
    }
    const int slot = timing_begin(RPG_TIMER_CONV, s);
    if (x_is_bf16) hipLaunchKernelGGL(stem_pool_bf16_kernel<__bf16>, dim3(grid), dim3(SB_NT), SB_LDS_BYTES, s, a);
    else hipLaunchKernelGGL(stem_pool_bf16_kernel<float>, dim3(grid), dim3(SB_NT), SB_LDS_BYTES, s, a);
    // algorithmic: the 7x7x3 convolution on every output pixel; executed: fragments x 11 steps x 2 MFMAs x 32*32*16*2
    timing_end(slot, 2.0 * (double)n * a.Hc * a.Wc * 64.0 * 147.0, s, (double)total * a.nfrag * KS * NF * 32768.0);
    RPG_CHECK_LAUNCH("stem_conv_bn_relu_maxpool_bf16");
    return RPG_OK;
}

}  // namespace rpg

extern "C" int rpg_stem_conv7x7s2_bn_relu_maxpool_bf16(const float* x_nchw, const void* wpack_bf16, const float* scale,
                                                       const float* shift, void* y_nhwc_bf16, int n, int h, int w, void* stream) {
    return rpg::launch_stem_pool_bf16(x_nchw, 0, wpack_bf16, scale, shift, y_nhwc_bf16, n, h, w, rpg::as_stream(stream));
}

// the same kernel on an input that is ALREADY bf16 (the fp32 images rounded on the host before the H2D copy: the kernel rounds
// them to bf16 first thing anyway, so the result is bit-identical and the PCIe transfer is half the size)
extern "C" int rpg_stem_conv7x7s2_bn_relu_maxpool_bf16_xbf16(const void* x_nchw_bf16, const void* wpack_bf16, const float* scale,
                                                             const float* shift, void* y_nhwc_bf16, int n, int h, int w, void* stream) {
    return rpg::launch_stem_pool_bf16(x_nchw_bf16, 1, wpack_bf16, scale, shift, y_nhwc_bf16, n, h, w, rpg::as_stream(stream));
}
