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
#include <type_traits>
#include <utility>
#include <vector>

#include "rpg_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

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
    float* out_relu = nullptr;            // optional second output [M][ldc] = max(out, 0) (the same values, rectified)
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

struct GatherArgs {
    const float* a[3];
    const int64_t* idx[3];
    int ld[3];
    int w0, w01, K;    // segment boundaries along k: [0,w0) [w0,w01) [w01,K)
    int fold_k;        // two-level accumulation: fold the accumulator every fold_k of K (0 = never); buffer-loader kernels on 32x32 wave tiles
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

enum TileShape { TILE_128x128 = 0, TILE_256x64 = 1, TILE_64x64 = 2, TILE_128x64 = 3 };

// Tuning knobs (rpg_set_tuning): -1 = automatic.
int g_force_tile = -1;
int g_tile_128x64 = 1;                // prefer 128x64 over 128x128 tiles for problems smaller than 1.5 big tiles per CU
int g_bk = 0;                        // 0 = automatic: 32 for the 128x128 tile (2 workgroups/CU), else 16
int g_epi_lds = 1;
int g_streamk = 1;
int g_gnn_split = 1;
int g_gnn_fuse_agg = 1;
int g_fold_k = 256;                  // RPG_TUNE_FOLD_K: two-level accumulation of the Linears (gemm_engine.inc), 0 = one sequential chain
int SK_MIN_ITS = 8;                  // at least this many k-steps per stream-K workgroup (RPG_TUNE_SK_MIN_ITS)
constexpr int SK_MIN_NK = 32;        // tiles with fewer k-steps are not worth splitting (fix-up traffic dominates)

// Library-owned scratch pool of the fine-grained entry points (see rpg_common.h): one block per (device, stream) that
// only grows; a grown block's predecessor is retired, not freed (a captured HIP graph or work still in flight may
// hold its address), and nothing is allocated while the stream is being captured.
// Every scratch block starts with kCounterBytes of ARRIVAL COUNTERS (round 4): one unsigned per split tile of the launch in
// flight, zero between launches -- the workgroup that finds itself the last contributor of a tile combines the partial slabs
// itself (in k order) and resets the counter, so no fix-up kernel follows.  The block's owner zeroes the header once (pool:
// at allocation; caller workspace: ScratchScope's constructor, one memset node per composite call).
constexpr size_t kCounterBytes = 16384;          // 4096 counters: > the split tiles of any launch (< 3 x CUs)
struct Scratch { float* p = nullptr; size_t bytes = 0; };
std::mutex g_scratch_mu;
std::map<std::pair<int, hipStream_t>, Scratch> g_scratch;
std::vector<float*> g_retired;
struct ScratchTls { float* p = nullptr; size_t bytes = 0; bool on = false; };
thread_local ScratchTls t_scratch;

float* get_scratch(hipStream_t s, size_t bytes, unsigned** counters = nullptr) {
    bytes += kCounterBytes;
    if (t_scratch.on) {                                                             // caller workspace (composite forwards)
        if (bytes > t_scratch.bytes) return nullptr;
        if (counters) *counters = reinterpret_cast<unsigned*>(t_scratch.p);
        return t_scratch.p + kCounterBytes / sizeof(float);
    }
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
        // arrival counters start at zero (stream-ordered before the first kernel that uses the block; the retired block's
        // counters are all back at zero by then: every launch resets what it counted)
        if (hipMemsetAsync(np, 0, kCounterBytes, s) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(np); return nullptr; }
        if (sc.p) g_retired.push_back(sc.p);
        sc.p = np;
        sc.bytes = want;
    }
    if (counters) *counters = reinterpret_cast<unsigned*>(sc.p);
    return sc.p + kCounterBytes / sizeof(float);
}
// RPG_TUNE_FIXUP_PRIO (round 6 experiment): the fix-up launches of split tiles (Winograd tail, stream-K) go to a
// HIGH-PRIORITY companion of the launch stream -- event hand-off there and back -- so that, with two batch halves on two streams,
// a 6-us fix-up is dispatched ahead of the other stream's queued convolution workgroups instead of behind them (measured r5:
// 22 us per Winograd fix-up under two streams against 6 alone).  One companion (stream + two events) per (device, stream), created
// on first use outside of any capture; never freed (streams are few and live as long as the process).
int g_fixup_prio = 0;
struct Companion { hipStream_t hi = nullptr; hipEvent_t there = nullptr, back = nullptr; };
std::map<std::pair<int, hipStream_t>, Companion> g_companion;
hipStream_t fixup_hop_begin_impl(hipStream_t s) {
    if (!g_fixup_prio) return s;
    int dev = 0;
    (void)hipGetDevice(&dev);
    Companion* c;
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        c = &g_companion[{dev, s}];
        if (!c->hi) {
            int lo = 0, hi = 0;                                                        // (numerically lowest = greatest priority)
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); return s; }
            if (hipStreamCreateWithPriority(&c->hi, hipStreamNonBlocking, hi) != hipSuccess ||
                hipEventCreateWithFlags(&c->there, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->back, hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                c->hi = nullptr;
                return s;
            }
        }
    }
    if (hipEventRecord(c->there, s) != hipSuccess || hipStreamWaitEvent(c->hi, c->there, 0) != hipSuccess) { (void)hipGetLastError(); return s; }
    return c->hi;
}
void fixup_hop_end_impl(hipStream_t s, hipStream_t used) {
    if (used == s) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    Companion c;
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        c = g_companion[{dev, s}];
    }
    (void)hipEventRecord(c.back, used);
    (void)hipStreamWaitEvent(s, c.back, 0);
}
int g_inkernel_fixup = 0;            // RPG_TUNE_INKERNEL_FIXUP (bit 0: stream-K, bit 1: Winograd): partial tiles combined by the last-arriving
                                     // workgroup instead of a fix-up launch.  OFF by default: measured slower

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
    // a K loop of <= 16 steps is all prologue and epilogue: small tiles, several workgroups per CU hiding each other's (the 1x1 / stride 2
    // downsample convolutions, r5, tools/conv_bench.py --only ds --tile t: K = 64 / 128 / 256: 64.9 / 55.4 / 49.0 us on 128x128 tiles,
    // 55.7 / 48.7 / 47.0 on 64x64)
    if (K <= 256 && t128 >= 384) return TILE_64x64;
    if (t128 >= 384) return TILE_128x128;                 // >= 1.5 big tiles per CU
    // fewer big tiles than CUs can balance: fine with stream-K when K is long enough to split, else go small
    // (128x64 halves the tile so that twice as many workgroups share the work before stream-K has to split K: measured
    // on the GNN Linears, M = 1792: N = 768 85 -> 68 us, N = 2048 142 -> 136 us; M = 256, N = 2048 63 -> 57 us)
    if (g_streamk && K >= 32 * SK_MIN_NK && (long)M * N >= 128L * 128 * 8) return g_tile_128x64 ? TILE_128x64 : TILE_128x128;
    return TILE_64x64;
}

thread_local double t_executed = 0.0;      // FLOP the matrix pipe issues for the last launch_one (whole padded tiles)

#define RPG_ENGINE_NS eng4
#define RPG_ENGINE_NT 256
#include "gemm_engine.inc"
#undef RPG_ENGINE_NS
#undef RPG_ENGINE_NT
#define RPG_ENGINE_NS eng8
#define RPG_ENGINE_NT 512
#include "gemm_engine.inc"
#undef RPG_ENGINE_NS
#undef RPG_ENGINE_NT

// ------------------------------------------------------------------------------------------------
// Exact-fit Linear for M = 7 * 2^k rows (round 5).  The GNN's edge GEMMs have M = 56 edges x graphs = 1792 (32 graphs) or 896
// (16 per stream) rows: 7 * 256 / 7 * 128 -- on 128-row tiles any equal split of the work leaves 7/8 of a power-of-two machine
// busy, and the stream-K split that balances it costs a fix-up launch per Linear (every tiling of the 32x32x2 engine measured
// within 4 % of 0.70 of the f32 matrix peak on the M = 1792, N = K = 2048 Linear: tools/conv_bench.py --only e).  Tiles that
// DIVIDE it: 112 x 64 -- 16 x 32 = 512 of them for M = 1792, N = 2048: two per CU, no remainder, no split, no fix-up.  112 = 7 x 16,
// so the fragments are those of v_mfma_f32_16x16x4_f32 (same 64 FLOP / clk / SIMD as the 32x32x2): a workgroup of 4 waves, one
// per SIMD, wave w owning all 7 row fragments x the 16 columns 16 w .. (7 x 4 = 28 accumulator registers, + 28 of the two-level
// sum); two workgroups per CU (51 KB of LDS each).  K advances in steps of 32 through a double-buffered LDS image [112 + 64 rows]
// [32 + 4] as in the tile engine; a lane reads 4 consecutive k of its row at offset 4 (lane >> 4) and feeds 4 MFMAs (the same
// k permutation for A and W).  Staging by raw buffer loads with the K position in the scalar offset (rows past 112 carry an
// out-of-range offset), one barrier per step, every load / LDS access placed behind an MFMA.  The epilogue transposes the
// WHOLE tile through LDS (112 x 64 floats) so that global accesses are 256-byte row segments, and finishes 16-byte column
// groups with the engine's own code (streamk_finish_quad: bias, gathered residual rows, ReLU, second rectified output).
// Plain A operand (one ungathered source), K % 32 == 0, N % 64 == 0, M % 112 == 0 (launcher).
// ------------------------------------------------------------------------------------------------
typedef float f32x4m __attribute__((ext_vector_type(4)));
constexpr int L112_BM = 112, L112_BN = 64, L112_BK = 32, L112_LD = L112_BK + 4, L112_NT = 256;
constexpr int L112_STAGE = (L112_BM + L112_BN) * L112_LD;                 // floats per LDS image
constexpr int L112_SLAB = L112_BM * (L112_BN + 4);                        // floats of the epilogue slab
constexpr int L112_LDS_BYTES = (2 * L112_STAGE > L112_SLAB ? 2 * L112_STAGE : L112_SLAB) * 4;
int g_lin112 = 3;                    // RPG_TUNE_LIN112: bit 0 the exact-fit kernel, bit 1 its eight-wave form for launches of about one tile per CU

__global__ __launch_bounds__(L112_NT, 2) void linear112_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Wt, int ldw,
                                                               int M, int N, int K, Epilogue ep, int tiles_n, int fold_k) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3, q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int m0 = (tile / tiles_n) * L112_BM, n0 = (tile % tiles_n) * L112_BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- staging: thread (slot = tid % 8, row = tid / 8): rows row + 32 j; A rows j = 0..3 (the last pass half empty), W rows j = 0..1
    const int slot = tid & 7, srow = tid >> 3;
    const __amdgpu_buffer_rsrc_t rsa = make_rsrc(A + (size_t)m0 * lda);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(Wt + (size_t)n0 * ldw);
    unsigned aoff[4], woff[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = srow + 32 * j;
        aoff[j] = row < L112_BM ? 4u * (unsigned)(row * lda + 4 * slot) : OOB;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) woff[j] = 4u * (unsigned)((srow + 32 * j) * ldw + 4 * slot);
    const int st_off = srow * L112_LD + 4 * slot;
    float4 rr[2][6];                                    // staging registers: the data of step s lives in set s & 1
    auto load_job = [&](int set, int qj, int kpos) {    // past K: out-of-range offsets (zeros nobody uses; never an address past the matrices)
        const bool live = kpos < K;
        if (qj < 4) rr[set][qj] = buf_ld4(rsa, live ? aoff[qj] : OOB, 4u * (unsigned)kpos);
        else rr[set][qj] = buf_ld4(rsw, live ? woff[qj - 4] : OOB, 4u * (unsigned)kpos);
    };
    auto write_job = [&](int set, int qj, int img) {
        const int row = qj < 4 ? 32 * qj : L112_BM + 32 * (qj - 4);
        if (qj == 3) {                                   // rows 96 .. 127: only 96 .. 111 exist in the image
            if (srow < 16) *reinterpret_cast<float4*>(&lds[img + st_off + row * L112_LD]) = rr[set][qj];
        } else {
            *reinterpret_cast<float4*>(&lds[img + st_off + row * L112_LD]) = rr[set][qj];
        }
    };
    // ---- fragments: lane (i16 = lane & 15, g = lane >> 4) reads 4 consecutive k at offset 4 g of row i16 (+ 16 f) / of W row 16 wave + i16
    const int a_off = (lane & 15) * L112_LD + 4 * (lane >> 4);
    const int b_off = (L112_BM + 16 * wave + (lane & 15)) * L112_LD + 4 * (lane >> 4);
    float4 fa[2][7], fb[2];
    auto read_job = [&](int set, int rj, int kb, int img) {
        if (rj < 7) fa[set][rj] = *reinterpret_cast<const float4*>(&lds[img + a_off + rj * 16 * L112_LD + kb]);
        else fb[set] = *reinterpret_cast<const float4*>(&lds[img + b_off + kb]);
    };
    auto comp = [](const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; };
    f32x4m acc[7], acc2[7];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[i][e] = 0.f; acc2[i][e] = 0.f; }

    // One K step t = 2 groups of 16 k = 2 x 28 MFMAs.  Behind the MFMAs of group 0: the 8 fragment reads of group 1, the 6 stage
    // writes of step t + 1 (register set (t + 1) & 1) and, into the registers those writes just freed, the 6 loads of step t + 3
    // -- TWO steps (~3.4 k cycles) before they are written to LDS: one step is shorter than a loaded L2 / HBM round trip, and the
    // barrier's vmcnt(0) waited for them every step (one-step prefetch: 0.72 of the f32 matrix peak in the model, two-step: see
    // DESIGN.md); behind group 1's: the fragment reads of the next step's group 0 from the other image.  The barrier sits after
    // group 0: by then every wave has written image t + 1.
    const int nsteps = K / L112_BK;
    auto kstep = [&](int cur, int nxt, int wset, int kload) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int ms = 0; ms < 28; ++ms) {
                const int c = ms / 7, i = ms % 7;
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(fa[g][i], c), comp(fb[g], c), acc[i], 0, 0, 0);
                if (g == 0) {
                    if (ms < 8) read_job(1, ms, 16, cur);
                    else if (ms < 14) write_job(wset, ms - 8, nxt);
                    else if (ms < 20) load_job(wset, ms - 14, kload);
                } else {
                    if (ms < 8) read_job(0, ms, 0, nxt);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g == 0) {
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // prologue: stage step 0, load step 1, first fragments
#pragma unroll
    for (int qj = 0; qj < 6; ++qj) load_job(0, qj, 0);
#pragma unroll
    for (int qj = 0; qj < 6; ++qj) load_job(1, qj, L112_BK);
#pragma unroll
    for (int qj = 0; qj < 6; ++qj) write_job(0, qj, 0);
#pragma unroll
    for (int qj = 0; qj < 6; ++qj) load_job(0, qj, 2 * L112_BK);
    __syncthreads();
#pragma unroll
    for (int rj = 0; rj < 8; ++rj) read_job(0, rj, 0, 0);
    const int fold_steps = fold_k >= 2 * L112_BK ? (fold_k / L112_BK) & ~1 : 0x7ffffffe;
    int kt = 0;
    while (kt + 1 < nsteps) {
        int kstop = kt + fold_steps;
        if (kstop > nsteps) kstop = nsteps;
        for (; kt + 1 < kstop; kt += 2) {
            kstep(0, L112_STAGE, 1, (kt + 3) * L112_BK);
            kstep(L112_STAGE, 0, 0, (kt + 4) * L112_BK);
        }
        if (kt + 1 < nsteps) {
            asm volatile("");
#pragma unroll
            for (int i = 0; i < 7; ++i) { acc2[i] += acc[i]; acc[i] = f32x4m{0.f, 0.f, 0.f, 0.f}; }
        }
    }
    if (kt < nsteps) kstep(0, L112_STAGE, 1, (kt + 3) * L112_BK);
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] += acc2[i];
    __syncthreads();
    // ---- epilogue: D[row = 4 (lane >> 4) + e][col = lane & 15] of fragment i -> slab [112][64 + 4]; then 16-byte column groups
    constexpr int SP = L112_BN + 4;
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[(16 * i + 4 * (lane >> 4) + e) * SP + 16 * wave + (lane & 15)] = acc[i][e];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < L112_BM * (L112_BN / 4) / L112_NT; ++t) {
        const int idx4 = t * L112_NT + tid;
        const int row = idx4 / (L112_BN / 4), c4 = idx4 % (L112_BN / 4);
        const float4 v = *reinterpret_cast<const float4*>(&lds[row * SP + 4 * c4]);
        eng4::streamk_finish_quad<L112_BM, L112_BN>(v, ep, M, N, tiles_n, tile, idx4, 1);
    }
}

// The same tile on EIGHT waves for launches with about one tile per CU (M = 896: the default two-stream schedule's edge GEMMs):
// with one 4-wave workgroup per CU every SIMD holds a lone wave, nothing hides its barrier and LDS latencies (measured: ~0.62 of
// the matrix rate inside the K loop against ~0.8 with two waves per SIMD).  Waves 0..3 take the first 16 k of every 32-deep step,
// waves 4..7 the second 16 (same column blocks); the two partial tiles meet in the LDS slab in front of the epilogue (waves 4..7
// first, then 0..3 add theirs: (k 0..15 + k 32..47 + ...) + (k 16..31 + ...) -- a summation order of its own, like every tiling).
constexpr int L112_NT2 = 512;
__global__ __launch_bounds__(L112_NT2, 1) void linear112k2_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Wt, int ldw,
                                                                 int M, int N, int K, Epilogue ep, int tiles_n, int fold_k) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3, q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int m0 = (tile / tiles_n) * L112_BM, n0 = (tile % tiles_n) * L112_BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kw = wave >> 2, wv = wave & 3;                 // k half of a step / column block of this wave
    // staging: thread (slot = tid % 8, row = tid / 8 in 0..63): A rows row, row + 64 (>= 112: nothing), W row `row`
    const int slot = tid & 7, srow = tid >> 3;
    const __amdgpu_buffer_rsrc_t rsa = make_rsrc(A + (size_t)m0 * lda);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(Wt + (size_t)n0 * ldw);
    unsigned soff[3];
    soff[0] = 4u * (unsigned)(srow * lda + 4 * slot);
    soff[1] = srow + 64 < L112_BM ? 4u * (unsigned)((srow + 64) * lda + 4 * slot) : OOB;
    soff[2] = 4u * (unsigned)(srow * ldw + 4 * slot);
    const int st_off = srow * L112_LD + 4 * slot;
    float4 rr[2][3];
    auto load_job = [&](int set, int qj, int kpos) {
        const bool live = kpos < K;
        rr[set][qj] = buf_ld4(qj < 2 ? rsa : rsw, live ? soff[qj] : OOB, 4u * (unsigned)kpos);
    };
    auto write_job = [&](int set, int qj, int img) {
        const int row = qj == 0 ? 0 : (qj == 1 ? 64 : L112_BM);
        if (qj == 1) {
            if (srow + 64 < L112_BM) *reinterpret_cast<float4*>(&lds[img + st_off + row * L112_LD]) = rr[set][qj];
        } else {
            *reinterpret_cast<float4*>(&lds[img + st_off + row * L112_LD]) = rr[set][qj];
        }
    };
    const int a_off = (lane & 15) * L112_LD + 4 * (lane >> 4) + 16 * kw;
    const int b_off = (L112_BM + 16 * wv + (lane & 15)) * L112_LD + 4 * (lane >> 4) + 16 * kw;
    float4 fa[2][7], fb[2];
    auto read_job = [&](int set, int rj, int img) {
        if (rj < 7) fa[set][rj] = *reinterpret_cast<const float4*>(&lds[img + a_off + rj * 16 * L112_LD]);
        else fb[set] = *reinterpret_cast<const float4*>(&lds[img + b_off]);
    };
    auto comp = [](const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; };
    f32x4m acc[7], acc2[7];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[i][e] = 0.f; acc2[i][e] = 0.f; }
    // step t: 28 MFMAs per wave on fragment set t & 1; behind them the 3 stage writes of step t + 1 and the 3 loads of step t + 3,
    // then the barrier, then the 8 fragment reads of step t + 1 from the image just completed
    const int nsteps = K / L112_BK;
    auto kstep = [&](int fset, int nxt, int wset, int kload) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ms = 0; ms < 28; ++ms) {
            const int c = ms / 7, i = ms % 7;
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(fa[fset][i], c), comp(fb[fset], c), acc[i], 0, 0, 0);
            if (ms < 3) write_job(wset, ms, nxt);
            else if (ms < 6) load_job(wset, ms - 3, kload);
            else if (ms >= 14 && ms < 22) read_job(fset ^ 1, ms - 14, nxt);
            __builtin_amdgcn_sched_barrier(0);
            if (ms == 13) {
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
#pragma unroll
    for (int qj = 0; qj < 3; ++qj) load_job(0, qj, 0);
#pragma unroll
    for (int qj = 0; qj < 3; ++qj) load_job(1, qj, L112_BK);
#pragma unroll
    for (int qj = 0; qj < 3; ++qj) write_job(0, qj, 0);
#pragma unroll
    for (int qj = 0; qj < 3; ++qj) load_job(0, qj, 2 * L112_BK);
    __syncthreads();
#pragma unroll
    for (int rj = 0; rj < 8; ++rj) read_job(0, rj, 0);
    const int fold_steps = fold_k >= 2 * L112_BK ? (fold_k / L112_BK) & ~1 : 0x7ffffffe;
    int kt = 0;
    while (kt + 1 < nsteps) {
        int kstop = kt + fold_steps;
        if (kstop > nsteps) kstop = nsteps;
        for (; kt + 1 < kstop; kt += 2) {
            kstep(0, L112_STAGE, 1, (kt + 3) * L112_BK);
            kstep(1, 0, 0, (kt + 4) * L112_BK);
        }
        if (kt + 1 < nsteps) {
            asm volatile("");
#pragma unroll
            for (int i = 0; i < 7; ++i) { acc2[i] += acc[i]; acc[i] = f32x4m{0.f, 0.f, 0.f, 0.f}; }
        }
    }
    if (kt < nsteps) kstep(0, L112_STAGE, 1, (kt + 3) * L112_BK);
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] += acc2[i];
    __syncthreads();
    constexpr int SP = L112_BN + 4;
    if (kw == 1) {
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) lds[(16 * i + 4 * (lane >> 4) + e) * SP + 16 * wv + (lane & 15)] = acc[i][e];
    }
    __syncthreads();
    if (kw == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float* p = &lds[(16 * i + 4 * (lane >> 4) + e) * SP + 16 * wv + (lane & 15)];
                *p = acc[i][e] + *p;
            }
    }
    __syncthreads();
    for (int idx4 = tid; idx4 < L112_BM * (L112_BN / 4); idx4 += L112_NT2) {
        const int row = idx4 / (L112_BN / 4), c4 = idx4 % (L112_BN / 4);
        const float4 v = *reinterpret_cast<const float4*>(&lds[row * SP + 4 * c4]);
        eng4::streamk_finish_quad<L112_BM, L112_BN>(v, ep, M, N, tiles_n, tile, idx4, 1);
    }
}

// true if launched (shape eligible)
bool launch_linear112(const float* A, int lda, const float* Wt, int ldw, int M, int N, int K, const Epilogue& ep, hipStream_t s) {
    if (!(g_lin112 & 1) || M % L112_BM || N % L112_BN || K % L112_BK || K < 2 * L112_BK) return false;
    const int tm = M / L112_BM, tn = N / L112_BN;
    const long tiles = (long)tm * tn;
    // worth it where the tiles fill the machine: >= 1 per CU (N = 2048: M = 896 -> 256 tiles, M = 1792 -> 512).  Measured below that:
    // the attention projections' N = 768 at M = 1792 (192 tiles, one 4-wave workgroup = one wave per SIMD on three quarters of the
    // CUs) take 79.7 us here against 75-78 us on the stream-K engine
    if (tiles < cu_count() || (long)L112_BM * lda * 4 >= (1L << 31) || (long)L112_BN * ldw * 4 >= (1L << 31)) return false;
    const bool vec_ok = (ep.ldc % 4 == 0) && rpg::aligned16(ep.out) && (!ep.residual || rpg::aligned16(ep.residual)) &&
                        (!ep.residual2 || rpg::aligned16(ep.residual2)) && (ep.ldr % 4 == 0) && (!ep.scale || rpg::aligned16(ep.scale)) &&
                        (!ep.shift || rpg::aligned16(ep.shift)) && (!ep.out_relu || rpg::aligned16(ep.out_relu)) && rpg::aligned16(A) &&
                        rpg::aligned16(Wt) && (lda % 4 == 0) && (ldw % 4 == 0);
    if (!vec_ok) return false;
    t_executed = 2.0 * (double)M * N * (double)K;
    // about one tile per CU: eight waves per tile (two per SIMD); from two tiles per CU on: four waves, two workgroups per CU
    if ((g_lin112 & 2) && 2 * tiles < 3L * cu_count())
        hipLaunchKernelGGL(linear112k2_kernel, dim3((unsigned)tiles), dim3(L112_NT2), L112_LDS_BYTES, s, A, lda, Wt, ldw, M, N, K, ep, tn, g_fold_k);
    else
        hipLaunchKernelGGL(linear112_kernel, dim3((unsigned)tiles), dim3(L112_NT), L112_LDS_BYTES, s, A, lda, Wt, ldw, M, N, K, ep, tn, g_fold_k);
    return true;
}

template <int BK, bool EPI, template <int, int> class Loader, class Args>
void launch_shape(TileShape t, const Args& args, const float* Wt, int ldw, int M, int N, int K, const Epilogue& ep,
                  bool vec_ok, hipStream_t s) {
    switch (t) {
        case TILE_128x128: eng4::launch_one<128, 128, 2, 2, BK, EPI, Loader, Args>(args, Wt, ldw, M, N, K, ep, vec_ok, s); break;
        case TILE_256x64: eng4::launch_one<256, 64, 4, 1, BK, EPI, Loader, Args>(args, Wt, ldw, M, N, K, ep, vec_ok, s); break;
        case TILE_128x64: eng4::launch_one<128, 64, 2, 2, BK, EPI, Loader, Args>(args, Wt, ldw, M, N, K, ep, vec_ok, s); break;
        default: eng4::launch_one<64, 64, 2, 2, BK, EPI, Loader, Args>(args, Wt, ldw, M, N, K, ep, vec_ok, s); break;
    }
}

// The 8-wave workgroups (two waves per SIMD; K step 32, buffer loaders, LDS epilogue only): 128x128 as 2 x 4 waves of
// 64x32, 128x64 as 4 x 2 waves of 32x32.
template <template <int, int> class Loader8, class Args>
void launch_shape8(TileShape t, const Args& args, const float* Wt, int ldw, int M, int N, int K, const Epilogue& ep,
                   hipStream_t s) {
    if (t == TILE_128x128) eng8::launch_one<128, 128, 2, 4, 32, true, Loader8, Args>(args, Wt, ldw, M, N, K, ep, true, s);
    else eng8::launch_one<128, 64, 4, 2, 32, true, Loader8, Args>(args, Wt, ldw, M, N, K, ep, true, s);
}

int g_fast = 1;                      // RPG_TUNE_FAST_LOADER: buffer-load loaders + interleaved main loop where eligible
int g_waves8 = 1;                    // RPG_TUNE_WAVES8: 8-wave workgroups for the 128-row tiles where the fast path applies

// The loaders of the two A-operand kinds, per engine (alias templates: template-template arguments of the launchers).
struct ConvKind {
    template <int R, int BK> using L4 = eng4::ConvLoader<R, BK>;
    template <int R, int BK> using B4 = eng4::ConvLoaderB<R, BK>;
    template <int R, int BK> using T4 = eng4::ConvLoaderTap<R, BK>;
    template <int R, int BK> using B8 = eng8::ConvLoaderB<R, BK>;
};
struct GatherKind {
    template <int R, int BK> using L4 = eng4::GatherLoader<R, BK>;
    template <int R, int BK> using B4 = eng4::GatherLoaderB<R, BK>;
    template <int R, int BK> using T4 = eng4::GatherLoaderB<R, BK>;
    template <int R, int BK> using B8 = eng8::GatherLoaderB<R, BK>;
};

// seg_align: 32 / 16 if every K segment of the A operand (gather widths, Cin) is a multiple of it and all offsets of
// the buffer loaders fit in 32 bits, else 0 (general loaders).
// seg_align == -1: the A operand is a 4-channel convolution input (ConvLoaderTap via T4; any K).
template <class Kind, class Args>
int launch_tiles(const Args& args, const float* Wt, int ldw, int M, int N, int K, const Epilogue& ep, hipStream_t s,
                 int seg_align) {
    TileShape t = pick_tile(M, N, K);
    // two-level accumulation lives in the 32x32 wave tiles (128x64 on 8 waves): a Linear long enough to need it does not take the 128x128 tile
    if (std::is_same<Kind, GatherKind>::value && g_fold_k > 0 && g_force_tile < 0 && t == TILE_128x128 && K >= 2 * g_fold_k) t = TILE_128x64;
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
        if (bk == 32) launch_shape<32, true, Kind::template T4, Args>(t, args, Wt, ldw, M, N, K, ep, vec_ok, s);
        else launch_shape<16, true, Kind::template T4, Args>(t, args, Wt, ldw, M, N, K, ep, vec_ok, s);
        return 0;
    }
    if (fast && bk == 32 && g_waves8 && (t == TILE_128x128 || t == TILE_128x64)) {
        launch_shape8<Kind::template B8, Args>(t, args, Wt, ldw, M, N, K, ep, s);
        return 0;
    }
    if (bk == 32) {
        if (fast) launch_shape<32, true, Kind::template B4, Args>(t, args, Wt, ldw, M, N, K, ep, vec_ok, s);
        else if (epi) launch_shape<32, true, Kind::template L4, Args>(t, args, Wt, ldw, M, N, K, ep, vec_ok, s);
        else launch_shape<32, false, Kind::template L4, Args>(t, args, Wt, ldw, M, N, K, ep, vec_ok, s);
    } else {
        if (fast) launch_shape<16, true, Kind::template B4, Args>(t, args, Wt, ldw, M, N, K, ep, vec_ok, s);
        else if (epi) launch_shape<16, true, Kind::template L4, Args>(t, args, Wt, ldw, M, N, K, ep, vec_ok, s);
        else launch_shape<16, false, Kind::template L4, Args>(t, args, Wt, ldw, M, N, K, ep, vec_ok, s);
    }
    return 0;
}

}  // namespace

namespace rpg {

bool gnn_split_enabled() { return g_gnn_split != 0; }
bool gnn_fuse_agg_enabled() { return g_gnn_fuse_agg != 0; }
float* stream_scratch(hipStream_t s, size_t bytes, unsigned** counters) { return get_scratch(s, bytes, counters); }
hipStream_t fixup_hop_begin(hipStream_t s) { return fixup_hop_begin_impl(s); }
void fixup_hop_end(hipStream_t s, hipStream_t used) { fixup_hop_end_impl(s, used); }
bool inkernel_fixup_enabled() { return (g_inkernel_fixup & 2) != 0; }      // (the Winograd launcher asks)
ScratchScope::ScratchScope(void* p, size_t bytes, hipStream_t s) {
    // the arrival counters at the head of the slice must be zero when the first split launch of this call starts: the
    // slice is a piece of a workspace that other calls use for other things in between
    if (p && bytes >= kCounterBytes && hipMemsetAsync(p, 0, kCounterBytes, s) != hipSuccess) { (void)hipGetLastError(); p = nullptr; bytes = 0; }
    t_scratch.p = static_cast<float*>(p);
    t_scratch.bytes = p ? bytes : 0;
    t_scratch.on = true;
}
ScratchScope::~ScratchScope() { t_scratch = ScratchTls{}; }
size_t split_scratch_bytes() { return (size_t)2 * 3 * cu_count() * 65536 + kCounterBytes; }
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
    launch_tiles<ConvKind, ConvArgs>(a, w, (int)K, (int)M, cout, (int)K, ep, s, seg);
    timing_end(slot, 2.0 * (double)M * cout * (double)kh * kw * (alg_cin > 0 ? alg_cin : cin), s, t_executed);
    RPG_CHECK_LAUNCH("conv2d_bn_act");
    return RPG_OK;
}

int launch_linear(const GatherSrc& src, const float* weight, const float* bias, const float* residual, float* out,
                  int m, int n_out, int relu, hipStream_t s, const GatherRes* gres, float* out_relu) {
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
    a.fold_k = g_fold_k;
    Epilogue ep{nullptr, bias, residual, out, n_out, relu};
    ep.out_relu = out_relu;
    if (out_relu && !aligned16(out_relu)) return RPG_ERR_BAD_ARG;
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
    if (!(src.n == 1 && !src.idx[0] && launch_linear112(src.a[0], src.ld[0], weight, K, m, n_out, K, ep, s)))
        launch_tiles<GatherKind, GatherArgs>(a, weight, K, m, n_out, K, ep, s, seg);
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

extern "C" int rpg_linear_gather_ex_f32(int n_src, const float* const* a, const int64_t* const* idx, const int* ld,
                                        const int* width, const long* rows, const float* weight, const float* bias,
                                        const float* residual, const int64_t* res_idx, const float* residual2,
                                        const int64_t* res2_idx, int ldr, float* out, float* out_relu, int m, int n_out,
                                        int relu, void* stream) {
    if (n_src < 1 || n_src > 3 || !a || !ld || !width) return RPG_ERR_BAD_ARG;
    rpg::GatherSrc src{};
    src.n = n_src;
    for (int i = 0; i < n_src; ++i) {
        src.a[i] = a[i];
        src.idx[i] = idx ? idx[i] : nullptr;
        src.ld[i] = ld[i];
        src.width[i] = width[i];
        src.rows[i] = rows ? rows[i] : 0;
    }
    if (!res_idx && !residual2)          // plain residual rows (pitch n_out), as rpg_linear_gather_f32
        return (res2_idx || (residual && ldr && ldr != n_out)) ? RPG_ERR_BAD_ARG
               : rpg::launch_linear(src, weight, bias, residual, out, m, n_out, relu, rpg::as_stream(stream), nullptr, out_relu);
    rpg::GatherRes gr{residual, res_idx, residual2, res2_idx, ldr};
    return rpg::launch_linear(src, weight, bias, nullptr, out, m, n_out, relu, rpg::as_stream(stream), &gr, out_relu);
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
        case RPG_TUNE_FIXUP_PRIO: g_fixup_prio = value != 0; return RPG_OK;
        case RPG_TUNE_BF16_PAIR: rpg::bf16_set_pair(value != 0); return RPG_OK;
        case RPG_TUNE_TILE: g_force_tile = value; return RPG_OK;
        case RPG_TUNE_BK: if (value != 0 && value != 16 && value != 32) return RPG_ERR_BAD_ARG; g_bk = value; return RPG_OK;
        case RPG_TUNE_EPILOGUE: g_epi_lds = value != 0; return RPG_OK;
        case RPG_TUNE_STREAMK: g_streamk = value != 0; return RPG_OK;
        case RPG_TUNE_BF16_BK: if (value != 32 && value != 64) return RPG_ERR_BAD_ARG; rpg::bf16_set_bk(value); return RPG_OK;
        case RPG_TUNE_INKERNEL_FIXUP:      // bit 0: stream-K tiles of the GEMM engine, bit 1: Winograd tail tiles (<= 8 parts); + 4: <= 32 parts
            if (value < 0 || value > 7) return RPG_ERR_BAD_ARG;
            g_inkernel_fixup = value & 3;
            rpg::wino_combine_max_set((value & 4) ? 32 : 8);
            return RPG_OK;
        case RPG_TUNE_GNN_SPLIT: g_gnn_split = value != 0; return RPG_OK;
        case RPG_TUNE_GNN_FUSE_AGG: g_gnn_fuse_agg = value != 0; return RPG_OK;
        case RPG_TUNE_LIN112: if (value < 0 || value > 3) return RPG_ERR_BAD_ARG; g_lin112 = value; return RPG_OK;
        case RPG_TUNE_FOLD_K: if (value < 0 || (value % 64)) return RPG_ERR_BAD_ARG; g_fold_k = value; return RPG_OK;
        case RPG_TUNE_FAST_LOADER: g_fast = value != 0; return RPG_OK;
        case RPG_TUNE_WAVES8: g_waves8 = value != 0; return RPG_OK;
        case RPG_TUNE_WINO_SPLIT: if (value < 0 || value > 96) return RPG_ERR_BAD_ARG; rpg::wino_split_set(value); return RPG_OK;
        case RPG_TUNE_BF16_TILE: if (value < -1 || value > 3) return RPG_ERR_BAD_ARG; rpg::bf16_set_tile(value); return RPG_OK;
        case RPG_TUNE_BF16_FAST: rpg::bf16_set_fast(value != 0); return RPG_OK;
        case RPG_TUNE_BF16_LEAN_EPI: rpg::bf16_set_lean_epi(value != 0); return RPG_OK;
        case RPG_TUNE_BF16_PERSIST: rpg::bf16_set_persist(value != 0); return RPG_OK;
        case RPG_TUNE_BF16_FUSE_BLOCK: rpg::bf16_set_fuse_block((int)(value & 7)); return RPG_OK;
        case RPG_TUNE_BF16_TAIL: if (value < 0 || value > 7) return RPG_ERR_BAD_ARG; rpg::bf16_set_tail(value); return RPG_OK;
        case RPG_TUNE_BF16_LINEAR_DMA: if (value != 0 && (value < 10 || value > 19)) return RPG_ERR_BAD_ARG; rpg::bf16_set_linear_dma(value); return RPG_OK;
        case RPG_TUNE_WINO2D:              // probe builds only (tools/probes/winograd2d.hip); the product accepts 0
            if (value < 0 || value > 2 || (value != 0 && rpg_wino43_weights_floats(1, 1) != 42)) return RPG_ERR_BAD_ARG;
            rpg::wino2d_set(value);
            return RPG_OK;
        case RPG_TUNE_BF16_CHUNK:          // images per depth-first group (0 = off) + 4096 * (least activation-tensor size in MB; 0 = 64)
            if (value < 0) return RPG_ERR_BAD_ARG;
            rpg::bf16_set_chunk(value & 4095, (value >> 12) ? (value >> 12) : 64);
            return RPG_OK;
        case RPG_TUNE_BF16_PATCH: if (value < 0 || value % 10 > 4 || value > 14) return RPG_ERR_BAD_ARG; rpg::bf16_set_patch(value); return RPG_OK;
        case RPG_TUNE_BF16_DMA: if (value < 0 || value > 40) return RPG_ERR_BAD_ARG; rpg::bf16_set_dma(value); return RPG_OK;
        case RPG_TUNE_SK_MIN_ITS: if (value < 1 || value > 4096) return RPG_ERR_BAD_ARG; SK_MIN_ITS = value; return RPG_OK;
        case RPG_TUNE_BF16_WS64: if (value < 0 || value > 2) return RPG_ERR_BAD_ARG; return rpg::bf16_set_ws64(value);
        case RPG_TUNE_FUSED_STEM: rpg::stem_pool_set((value & 1) != 0); rpg::stem_pool_set_strip((value & 128) != 0, value >> 8); rpg::bf16_set_fused_stem((value & 1) != 0); rpg::bf16_set_stem_strip((value & 2) ? 0 : ((value & 32) ? 1 : 9), value >> 8); return RPG_OK;
        case RPG_TUNE_WINOGRAD: if (value < 0 || (value > 3 && value < 16)) return RPG_ERR_BAD_ARG; rpg::wino_set(value); return RPG_OK;
        case RPG_TUNE_WINO_PERSIST: if (value < 0 || value > 2) return RPG_ERR_BAD_ARG; rpg::wino_persist_set(value); return RPG_OK;
        case RPG_TUNE_WINO_SHORT: if (value < 0) return RPG_ERR_BAD_ARG; rpg::wino_short_set(value); return RPG_OK;
        default: return RPG_ERR_BAD_ARG;
    }
}
