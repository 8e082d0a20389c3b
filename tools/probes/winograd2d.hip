// PROBE, not part of the product library (round 4): correct, parity-tested, and 13-34 % SLOWER than the 1-D kernels of
// csrc/winograd.hip on every ResNet34 shape -- profiles/r4_wino2d_nested_kernel.txt has the A/B and the ablation.  Build it in with
//     RPG_BUILD_DEFINES="-DRPG_PROBE_WINO2D" python relpose-gnn_amd/build.py --force
// (csrc/winograd.hip then includes this file, the weight buffer of a convolution grows from 18 to 42 Cout Cin floats and
// RPG_TUNE_WINO2D = 1 / 2 selects the kernel); tests: tools/probes/test_wino2d_probe.py; A/B: tools/conv_bench.py --wino2d-ab.
//
// 3x3 / stride 1 / pad 1 convolution as the NESTED 2-D Winograd F(4x2, 3x3) on f32 MFMA (gfx950).
//
// csrc/winograd.hip computes F(4,3) along the width only: 6 positions x 3 kernel rows per 4 outputs = 4.5 multiplies per
// output and input channel.  Nesting F(2,3) along the height -- a tile is 2 rows x 4 columns of outputs from 4 x 6 inputs,
// 24 positions, each its own GEMM over the input channels -- needs 3: two thirds of the matrix-pipe work, and the f32 matrix
// pipe is what bounds the encoder.
//
//   V[j][xi]        = sum_{a,b} BhT[j][a] BwT[xi][b] d[a][b]            d = input rows 2hp-1..2hp+2, columns 4tw-1..4tw+4
//   U[j][xi][co][c] = sum_{kh,kw} Gh[j][kh] Gw[xi][kw] w[co][kh][kw][c] (once per weight load, in double)
//   M[j][xi]        = V[j][xi] [tiles x Cin] * U[j][xi]^T [Cin x Cout]  24 GEMMs, K = Cin
//   y[i][o]         = sum_{j,xi} AhT[i][j] AwT[o][xi] M[j][xi]           2 x 4 outputs, lane-local
//   F(2,3): BhT = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1], Gh = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], AhT = [1 1 1 0; 0 1 -1 -1]
//   F(4,3): the matrices of csrc/winograd.hip (points 0, +-1, +-2, inf).
//
// What it costs, and why the kernel looks the way it does (tools/probes/wino_nested_probe.hip measured the K loop first:
// 0.78 of the matrix pipe against 0.91 for the 1-D kernel's, i.e. 1.5 x 0.78 / 0.91 = 1.29x per output):
//   * 24 accumulator positions do not fit a wave as 32 x 32 tiles (24 x 16 registers), so the products are
//     v_mfma_f32_16x16x4_f32 on 16-tile x 16-channel wave tiles: 24 x 4 = 96 accumulator registers, as many as the 1-D
//     kernel, two waves per SIMD.  An operand fragment then feeds ONE MFMA instead of two: 0.5 KB of LDS reads per
//     32-cycle MFMA (50 % of the LDS bandwidth) against 0.5 KB per 64-cycle MFMA.
//   * workgroup = 8 waves on 64 tiles (of 4 x 2 pixels = 512 output pixels) x 32 channels; K step = 8 channels: one LDS
//     image is 24 x (64 + 32) rows of 32 bytes = 72 KB, two of them = 144 KB, one workgroup per CU.
//   * staging: thread = (height position j, tile, 4-channel slot).  A thread loads the TWO input rows its height position
//     combines (BhT rows have two non-zeros: 12 pixels), adds / subtracts them, runs the six-point width transform of
//     csrc/winograd.hip on the result and writes six 16-byte pieces; a wave has one j and 32 consecutive tiles, so its
//     ds_write_b128 cover 1 KB contiguously and its ds_read_b64 512 B contiguously: no swizzle needed, no bank conflicts.
//   * the same rules as the 1-D kernel (tools/probes/mfma_shadow_probe.hip): nothing runs in a phase of its own -- the
//     48 fragment reads, 9 stage pieces (+ 36 packed transform instructions) and 15 raw buffer loads of a K step sit singly
//     behind its 48 MFMAs, pinned by scheduling barriers; one barrier per step; lane-dependent addressing is K-invariant.
// Odd heights (7 x 7 maps) are handled by masking (the second row of the last tile row reads zeros and is not stored).
//
// Reference op replaced: nn.Conv2d(3x3, stride 1, pad 1) + nn.BatchNorm2d (eval) (+ identity) + ReLU of a torchvision
// BasicBlock, reached from /root/reference/python/niantic/modules/posenet.py:1037.
#ifndef RPG_W2_ABL
#define RPG_W2_ABL 0                       // diagnostic builds (tools/probes/wino2d_ablate.sh): 1 no input loads | 2 no weight loads |
#endif                                     // 4 no stage writes | 8 no barrier | 16 no transform arithmetic   (wrong results, timing only)

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace w2d {

constexpr int NT = 512;
constexpr int NPOS = 24;                   // 4 height positions x 6 width positions
constexpr int TT = 64, TC = 32;            // tiles x output channels per workgroup
constexpr int KS = 8;                      // input channels per K step
constexpr int ROWB = KS * 4;               // bytes per LDS row (one tile / one output channel of one position)
constexpr int A_BYTES = NPOS * TT * ROWB;  // 49,152
constexpr int B_BYTES = NPOS * TC * ROWB;  // 24,576
constexpr int IMG_BYTES = A_BYTES + B_BYTES;
constexpr int LDS2_BYTES = 2 * IMG_BYTES;  // 147,456
constexpr unsigned OOB = 0x80000000u;

struct Epi2 {
    const float* scale;
    const float* shift;
    const float* residual;
    float* out;
    int relu;
};

struct F4 { f32x2 lo, hi; };
__device__ __forceinline__ F4 to_f4(const float4& v) { return F4{f32x2{v.x, v.y}, f32x2{v.z, v.w}}; }
__device__ __forceinline__ float4 to_float4(const F4& v) { return make_float4(v.lo.x, v.lo.y, v.hi.x, v.hi.y); }
// packed f32 arithmetic written as instructions (beside MFMAs hipcc splits packed f32 math into scalar halves)
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
RPG_PK_FMA(pk_fma_p1, "1.0")
RPG_PK_FMA(pk_fma_m1, "-1.0")
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

// x [n][H][W][Cin] -> y [n][H][W][Cout];  U2 [24][Cout][Cin];  tiles M = n * H2 * Tw, H2 = ceil(H / 2), Tw = ceil(W / 4)
__global__ __launch_bounds__(NT) void wino2d_conv_kernel(const float* __restrict__ x, const float* __restrict__ U2, int H, int W,
                                                         int Cin, int Cout, int H2, int Tw, int M, Epi2 ep, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // XCD-aware tile order (workgroup b runs on XCD b % 8: contiguous runs of tiles per XCD, channel tile fastest)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3, q8 = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int m0 = (tile / tiles_n) * TT;
    const int n0 = (tile % tiles_n) * TC;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                 // MFMA roles: tiles 16 wm .., channels 16 wn ..
    // staging roles: height position j (wave-uniform), tile t, 4-channel slot s
    const int j = wave >> 1, t = (tid & 127) >> 1, s = tid & 1;
    const int ra = j == 0 ? 0 : j == 2 ? 2 : 1;               // V_j = d[ra] + sgn * d[rb]: BhT rows (1,0,-1,0) (0,1,1,0) (0,-1,1,0) (0,1,0,-1)
    const int rb = j == 2 ? 1 : j == 3 ? 3 : 2;
    const float sgn = j == 1 ? 1.f : -1.f;

    const int HW = H * W;
    const int n_first = m0 / (H2 * Tw);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + (size_t)n_first * HW * Cin, 0,
                                                                          0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U2), 0, 0x7fffffff, 0x00020000);
    unsigned va[6], vb[6];                                    // the six pixels of input rows ra / rb of this thread's tile
    {
        const int m = m0 + t;
        int img_off = 0, row_a = -1, row_b = -1, wi0 = -(1 << 24);
        if (m < M) {
            const int tw = m % Tw, q = m / Tw, hp = q % H2, n = q / H2;
            img_off = (n - n_first) * HW;
            row_a = 2 * hp - 1 + ra;
            row_b = 2 * hp - 1 + rb;
            wi0 = 4 * tw - 1;
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int wi = wi0 + b;
            const bool cv = (unsigned)wi < (unsigned)W;
            va[b] = cv && (unsigned)row_a < (unsigned)H ? 4u * (unsigned)((img_off + row_a * W + wi) * Cin + 4 * s) : OOB;
            vb[b] = cv && (unsigned)row_b < (unsigned)H ? 4u * (unsigned)((img_off + row_b * W + wi) * Cin + 4 * s) : OOB;
        }
    }
    // U pieces: piece q = tid + 512 k (k = 0..2) = (position q >> 6, channel (q & 63) >> 1, slot q & 1) -> LDS A_BYTES + 16 q
    unsigned vu[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int q = tid + NT * k, p = q >> 6, co = n0 + ((q & 63) >> 1);
        vu[k] = co < Cout ? 4u * (unsigned)((p * Cout + co) * Cin + 4 * (q & 1)) : OOB;
    }
    const int nk = Cin / KS;
    int f_kt = 0;                                             // the K step the next fetches load
    float4 da[6], db[6], ub[3];
    if (RPG_W2_ABL & 3) {
#pragma unroll
        for (int i = 0; i < 6; ++i) da[i] = db[i] = make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
        for (int i = 0; i < 3; ++i) ub[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
    // fetch_one(i), i = 0..14: rows a (0..5), rows b (6..11), U pieces (12..14) of K step f_kt.  Past the last step the
    // scalar offset is clamped to the last channel block: those registers are staged into an image nobody reads.
    auto fetch_one = [&](int i) {
        if ((RPG_W2_ABL & 1) && i < 12) return;
        if ((RPG_W2_ABL & 2) && i >= 12) return;
        const int kc = f_kt < nk ? f_kt : nk - 1;
        const unsigned so = 4u * (unsigned)(kc * KS);
        if (i < 6) da[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsx, va[i], so, 0));
        else if (i < 12) db[i - 6] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsx, vb[i - 6], so, 0));
        else ub[i - 12] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsu, vu[i - 12], so, 0));
    };
    // staging: e[b] = d_a[b] + sgn d_b[b] (height transform), then the six-point width transform, then 6 + 3 writes
    const unsigned st_a = (unsigned)((6 * j) * (TT * ROWB) + t * ROWB + 16 * s);
    const unsigned st_b = (unsigned)(A_BYTES + 16 * tid);
    F4 e[6];
    const f32x2 sg = {sgn, sgn};
    auto pre_add = [&](int b) {
        if (RPG_W2_ABL & 16) { e[b] = to_f4(da[b]); return; }
        const F4 a4 = to_f4(da[b]), b4 = to_f4(db[b]);
        f32x2 lo, hi;                                         // a + sgn * b as one packed FMA per half (sgn = +-1: exact)
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(b4.lo), "v"(sg), "v"(a4.lo));
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(b4.hi), "v"(sg), "v"(a4.hi));
        e[b] = F4{lo, hi};
    };
    auto stage_a = [&](int p, int img) {                     // width position p of this thread's height position
        if (RPG_W2_ABL & 4) return;
        if (RPG_W2_ABL & 16) { *reinterpret_cast<float4*>(lds + img + st_a + p * (TT * ROWB)) = to_float4(e[p]); return; }
        F4 v;
        if (p == 0) v = fma4_p4(sub4(e[0], e[2]), sub4(e[4], e[2]));               // 4 d0 - 5 d2 + d4
        else if (p == 5) v = fma4_m4(sub4(e[3], e[1]), sub4(e[5], e[3]));          // 4 d1 - 5 d3 + d5
        else if (p <= 2) {
            const F4 sx = fma4_m4(e[2], e[4]);                                     // d4 - 4 d2
            const F4 tx = fma4_m4(e[1], e[3]);                                     // d3 - 4 d1
            v = p == 1 ? add4(sx, tx) : sub4(sx, tx);
        } else {
            const F4 r = sub4(e[4], e[2]), tq = sub4(e[3], e[1]);
            v = p == 3 ? fma4_p2(tq, r) : fma4_m2(tq, r);                          // r +- 2 t
        }
        *reinterpret_cast<float4*>(lds + img + st_a + p * (TT * ROWB)) = to_float4(v);
    };
    auto stage_b = [&](int k, int img) {
        if (RPG_W2_ABL & 4) return;
        *reinterpret_cast<float4*>(lds + img + st_b + k * (NT * 16)) = ub[k];
    };

    f32x4v acc[NPOS];
#pragma unroll
    for (int p = 0; p < NPOS; ++p) acc[p] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // operand fragments: lane (row = lane & 15, k pair = lane >> 4) reads 8 bytes: MFMA 0 of a position takes .x, MFMA 1 .y
    // (the same permutation of k on both operands)
    const unsigned a_off = (unsigned)((wm * 16 + (lane & 15)) * ROWB + 8 * (lane >> 4));
    const unsigned b_off = (unsigned)(A_BYTES + (wn * 16 + (lane & 15)) * ROWB + 8 * (lane >> 4));
    f32x2 fa[2][4], fb[2][4];
    auto frag = [&](int g, int set, int i, int img) {        // position 4 g + i: its A and its B fragment
        const int p = 4 * g + i;
        fa[set][i] = *reinterpret_cast<const f32x2*>(lds + img + p * (TT * ROWB) + a_off);
        fb[set][i] = *reinterpret_cast<const f32x2*>(lds + img + p * (TC * ROWB) + b_off);
    };
    // One K step on image `cur`: 6 groups of 8 MFMAs (4 positions x 2 k halves).  Behind MFMAs 0-3 of a group: the fragment
    // pairs of the next group (group 5: of the next step's group 0, from `nxt`, after the barrier).  Behind MFMAs 4-7 of groups
    // 0-2: the staging of step kt+1 into `nxt` (two slots of height pre-adds, six width pieces, three U pieces); of groups
    // 3-5: the 15 loads of step kt+2.  The barrier sits after group 4: every wave has written `nxt` and issued its last
    // reads of `cur`.
    auto kstep = [&](int cur, int nxt) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int set = g & 1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pos = i & 3, kk = i >> 2;
                acc[4 * g + pos] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][pos][kk], fb[set][pos][kk], acc[4 * g + pos], 0, 0, 0);
                if (i < 4) {
                    if (g < 5) frag(g + 1, set ^ 1, i, cur);
                    else frag(0, 0, i, nxt);
                } else {
                    const int sl = 4 * (g % 3) + (i - 4);     // 0..11
                    if (g < 3) {
                        if (sl == 0) { pre_add(0); pre_add(1); pre_add(2); }
                        else if (sl == 1) { pre_add(3); pre_add(4); pre_add(5); }
                        else if (sl < 8) stage_a(sl - 2, nxt);
                        else if (sl < 11) stage_b(sl - 8, nxt);
                    } else {
                        if (sl < 3) { fetch_one(2 * sl); fetch_one(2 * sl + 1); }
                        else fetch_one(sl + 3);               // 6 .. 14
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g == 4 && !(RPG_W2_ABL & 8)) {
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        ++f_kt;
    };

    // prologue: step 0 -> image 0, step 1 in registers
#pragma unroll
    for (int i = 0; i < 15; ++i) fetch_one(i);
    ++f_kt;
#pragma unroll
    for (int b = 0; b < 6; ++b) pre_add(b);
#pragma unroll
    for (int p = 0; p < 6; ++p) stage_a(p, 0);
#pragma unroll
    for (int k = 0; k < 3; ++k) stage_b(k, 0);
#pragma unroll
    for (int i = 0; i < 15; ++i) fetch_one(i);
    ++f_kt;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) frag(0, 0, i, 0);
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        kstep(0, IMG_BYTES);
        kstep(IMG_BYTES, 0);
    }
    if (kt < nk) kstep(0, IMG_BYTES);

    // ---- epilogue: lane = (channel n0 + 16 wn + (lane & 15), tiles m0 + 16 wm + 4 (lane >> 4) + e, e = 0..3).  Output
    // transform per tile (lane-local), BatchNorm / residual / ReLU, 4-byte stores (16 lanes = 64 contiguous bytes).
    const int ch = n0 + wn * 16 + (lane & 15);
    const bool ch_ok = ch < Cout;
    const float sc = ch_ok && ep.scale ? ep.scale[ch] : 1.f;
    const float sh = ch_ok && ep.shift ? ep.shift[ch] : 0.f;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(ep.out + (size_t)n_first * HW * Cout, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(ep.residual ? ep.residual : ep.out) + (size_t)n_first * HW * Cout, 0, 0x7fffffff, 0x00020000);
    const float floor_v = ep.relu ? 0.f : -INFINITY;
#pragma unroll
    for (int el = 0; el < 4; ++el) {
        const int m = m0 + wm * 16 + 4 * (lane >> 4) + el;
        const bool ok = ch_ok && m < M;
        const int tw = m % Tw, q = m / Tw, hp = q % H2, n = q / H2;
        // width transform of the four height positions, then the height transform
        float yw[4][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const float m0_ = acc[6 * jj + 0][el], m1 = acc[6 * jj + 1][el], m2 = acc[6 * jj + 2][el];
            const float m3 = acc[6 * jj + 3][el], m4 = acc[6 * jj + 4][el], m5 = acc[6 * jj + 5][el];
            const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
            yw[jj][0] = (m0_ + s12) + s34;
            yw[jj][1] = d12 + 2.f * d34;
            yw[jj][2] = s12 + 4.f * s34;
            yw[jj][3] = (d12 + 8.f * d34) + m5;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ho = 2 * hp + i;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const float y = i == 0 ? (yw[0][o] + yw[1][o]) + yw[2][o] : (yw[1][o] - yw[2][o]) - yw[3][o];
                const int wo = 4 * tw + o;
                const unsigned off = ok && ho < H && wo < W ? 4u * (unsigned)((((n - n_first) * H + ho) * W + wo) * Cout + ch) : OOB;
                float v = y * sc + sh;
                if (ep.residual) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, off, 0, 0));
                v = fmaxf(v, floor_v);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, off, 0, 0);
            }
        }
    }
}

// U2[j][xi][co][c] = sum_{kh,kw} Gh[j][kh] Gw[xi][kw] w[co][kh][kw][c], evaluated in double and rounded once.
__global__ __launch_bounds__(256) void wino2d_weights_kernel(const float* __restrict__ w, float* __restrict__ U2, int Cout, int Cin,
                                                             long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;      // over [co][c]
    if (i >= total) return;
    const int c = (int)(i % Cin);
    const long co = i / Cin;
    double g[3][3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) g[kh][kw] = w[((co * 3 + kh) * 3 + kw) * (long)Cin + c];
    double gw[3][6];                                          // width: Gw of F(4,3) applied to every kernel row
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const double g0 = g[kh][0], g1 = g[kh][1], g2 = g[kh][2];
        gw[kh][0] = g0 / 4.0;
        gw[kh][1] = -(g0 + g1 + g2) / 6.0;
        gw[kh][2] = -(g0 - g1 + g2) / 6.0;
        gw[kh][3] = g0 / 24.0 + g1 / 12.0 + g2 / 6.0;
        gw[kh][4] = g0 / 24.0 - g1 / 12.0 + g2 / 6.0;
        gw[kh][5] = g2;
    }
    const long plane = (long)Cout * Cin;
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) {
        const double a = gw[0][xi], b = gw[1][xi], cc = gw[2][xi];
        const double u[4] = {a, 0.5 * (a + b + cc), 0.5 * (a - b + cc), cc};          // Gh of F(2,3)
#pragma unroll
        for (int j = 0; j < 4; ++j) U2[(6 * j + xi) * plane + i] = (float)u[j];
    }
}

int g_wino2d = 0;                          // RPG_TUNE_WINO2D: 0 off (default: measured slower) | 1 by shape | 2 wherever eligible

}  // namespace w2d

namespace rpg {

void wino2d_set(int v) { w2d::g_wino2d = v; }

// floats of the transformed-weight buffer of one convolution: the 1-D image [6][Cout][3][Cin] followed by the nested one
// [24][Cout][Cin] (both are written by rpg_wino43_transform_weights_f32; the launcher picks per shape)
size_t wino_weight_floats(int cout, int cin) { return (size_t)(18 + 24) * cout * cin; }

int launch_wino2d_weights(const float* w_ohwi, float* u2, int cout, int cin, hipStream_t s) {
    const long total = (long)cout * cin;
    hipLaunchKernelGGL(w2d::wino2d_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_ohwi, u2, cout, cin, total);
    RPG_CHECK_LAUNCH("wino2d_transform_weights");
    return RPG_OK;
}

// true if the nested kernel takes this convolution (eligible, enabled, and measured faster for the shape class)
bool wino2d_takes(int n, int h, int w, int cin, int cout) {
    if (!w2d::g_wino2d || (cin % w2d::KS) || (cout % w2d::TC)) return false;
    const int h2 = (h + 1) / 2, tw = (w + 3) / 4;
    const long M = (long)n * h2 * tw;
    // 32-bit buffer offsets: a workgroup's 64 tiles span at most 64 / (h2 tw) + 2 images; U2 is addressed from its base
    if (M >= (1L << 31) || (64L / ((long)h2 * tw) + 2) * h * w * (cin > cout ? cin : cout) * 4 >= (1L << 31) ||
        24L * cout * cin * 4 >= (1L << 31))
        return false;
    if (w2d::g_wino2d >= 2) return true;
    const long tiles = ((M + w2d::TT - 1) / w2d::TT) * (cout / w2d::TC);
    return tiles >= 2L * num_cus() && h >= 8;   // enough workgroups to fill the chip twice; 7 x 7 maps pad 64 / 49
}

int launch_conv_wino2d(const float* x, const float* u2, const float* scale, const float* shift, const float* residual, float* y,
                       int n, int h, int w, int cin, int cout, int relu, hipStream_t s) {
    if (!x || !u2 || !y || n <= 0 || h <= 0 || w <= 0 || (cin % w2d::KS) || (cout % w2d::TC) || !aligned16(x) || !aligned16(u2))
        return RPG_ERR_BAD_ARG;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    static bool attr[64] = {};
    if (!attr[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(w2d::wino2d_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, w2d::LDS2_BYTES);
        attr[dev] = true;
    }
    const int h2 = (h + 1) / 2, tw = (w + 3) / 4;
    const long M = (long)n * h2 * tw;
    const int tm = (int)((M + w2d::TT - 1) / w2d::TT), tn = cout / w2d::TC;
    w2d::Epi2 ep{scale, shift, residual, y, relu};
    const int slot = timing_begin(RPG_TIMER_CONV_WINO, s);
    hipLaunchKernelGGL(w2d::wino2d_conv_kernel, dim3((unsigned)(tm * tn)), dim3(w2d::NT), w2d::LDS2_BYTES, s, x, u2, h, w, cin, cout, h2, tw, (int)M, ep, tn);
    // executed matrix-pipe FLOP: workgroups x K steps x 48 MFMAs x 8 waves x 2048
    timing_end(slot, 2.0 * (double)n * h * w * cout * 9.0 * cin, s, (double)tm * tn * (cin / w2d::KS) * 48.0 * 8.0 * 2048.0);
    RPG_CHECK_LAUNCH("conv3x3_wino2d");
    return RPG_OK;
}

}  // namespace rpg
