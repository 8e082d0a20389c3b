// Shared host-side helpers for the gfx950 kernels of the relpose-gnn hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/relpose_gnn_hip.h"

#include "rpg_coherent.h"      // agent-scope partial-slab accesses + arrival ticket (device code of winograd.hip / gemm_engine.inc)

namespace rpg {

// Last launch error text (thread local), exposed through rpg_last_error().
void set_last_error(const char* where, hipError_t e);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Kernel timing hooks (timing.hip).  begin() returns a slot (or -1 when timing is disabled).
int timing_begin(int klass, hipStream_t s);
// work = algorithmic work of the launch (FLOP / bytes); executed = FLOP the matrix pipe really issues for it
// (padded tiles, Winograd's reduced multiply count); 0 = same as work
void timing_end(int slot, double work, hipStream_t s, double executed = 0.0);

#define RPG_CHECK_LAUNCH(where)                         \
    do {                                                \
        hipError_t _e = hipGetLastError();              \
        if (_e != hipSuccess) {                         \
            rpg::set_last_error(where, _e);             \
            return RPG_ERR_LAUNCH;                      \
        }                                               \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }


// ---- internal launchers shared between the fine-grained C ABI and the composite forwards ----
struct GatherSrc {
    const float* a[3];
    const int64_t* idx[3];
    int ld[3];
    int width[3];
    int n;
    long rows[3];          // rows of a[i] when idx[i] is given (0 = unknown: the 32-bit-offset fast path is not used)
};

int launch_conv(const float* x, const float* w, const float* scale, const float* shift, const float* residual,
                float* y, int n, int h, int wd, int cin, int cout, int kh, int kw, int stride, int pad, int relu,
                hipStream_t s, int alg_cin = 0 /* channels counted as algorithmic work; 0 = cin */);
int launch_conv_wino(const float* x, const float* u, const float* scale, const float* shift, const float* residual,
                     float* y, int n, int h, int w, int cin, int cout, int relu, hipStream_t s);
// Workspace carving: every sub-buffer is followed by this many bytes of padding.  The activation buffers of a layer are
// whole MiB apart otherwise (256 images x 28 x 28 x 128 x 4 B = 98 MiB), so the input, residual and output streams of
// a convolution hit the same HBM channels in lock-step: measured on the layer-2 Winograd convolution, 323 us with
// 2-MiB-congruent buffers vs 272 us with >= 68 KB of skew between them (tools/probes/alias_probe.py).
constexpr size_t kWorkspaceSkew = 260 * 1024 + 4096;

// Scratch for split-K / stream-K partial tiles (consumers run in stream order).  Inside a composite forward it is a
// slice of the CALLER's workspace (ScratchScope, set for the duration of the host-side call; a request that does not fit
// returns null and the launcher simply does not split).  The fine-grained entry points have no workspace argument:
// they draw on a library-owned pool, one block per (device, stream), which only ever grows, never while the stream is
// being captured into a HIP graph, and whose old blocks stay alive until rpg_release_scratch().
// counters (optional out): the block's arrival counters -- 4096 unsigned, zero between launches (see gemm_f32.hip) -- for the
// in-kernel combine of split tiles by the last-arriving workgroup
float* stream_scratch(hipStream_t s, size_t bytes, unsigned** counters = nullptr);
// RPG_TUNE_FIXUP_PRIO: the stream a split tile's fix-up launch goes to (s itself when off; else the high-priority companion of s,
// already waiting for what s has enqueued) and the hand-back (s waits for the companion)
hipStream_t fixup_hop_begin(hipStream_t s);
void fixup_hop_end(hipStream_t s, hipStream_t used);
bool inkernel_fixup_enabled();            // RPG_TUNE_INKERNEL_FIXUP bit 1 (Winograd tail tiles); OFF by default (measured slower)
struct ScratchScope {
    ScratchScope(void* p, size_t bytes, hipStream_t s);       // zeroes the slice's counter header on s (one memset node per call)
    ~ScratchScope();
    ScratchScope(const ScratchScope&) = delete;
    ScratchScope& operator=(const ScratchScope&) = delete;
};
// upper bound of what one launch asks stream_scratch for (stream-K: < 2 partial tiles of 64 KB per resident workgroup,
// <= 3 workgroups per CU; Winograd split-K tail: <= one 128-KB partial tile per CU)
size_t split_scratch_bytes();
int num_cus();
bool wino_enabled();
// true if the composite forward should route this 3x3/stride-1 convolution to launch_conv_wino (enabled, shape
// addressable, and large enough that the un-split K loop is not latency-bound)
bool wino_pays(int n, int h, int w, int cin, int cout);
void wino_set(int on);
// nested 2-D Winograd F(4x2, 3x3) (csrc/winograd2d.hip)
void wino2d_set(int v);
size_t wino_weight_floats(int cout, int cin);
int launch_wino2d_weights(const float* w_ohwi, float* u2, int cout, int cin, hipStream_t s);
bool wino2d_takes(int n, int h, int w, int cin, int cout);
int launch_conv_wino2d(const float* x, const float* u2, const float* scale, const float* shift, const float* residual, float* y,
                       int n, int h, int w, int cin, int cout, int relu, hipStream_t s);
void stem_pool_set(int on);
void stem_pool_set_strip(int on, int band_rows);      // stem.hip: fp32 strip-march kernel (round 6) | tile kernel (rounds 2-5)
void wino_split_set(int on);
void wino_short_set(int cin);
void wino_persist_set(int on);
void wino_combine_max_set(int v);
void bf16_set_fast(int on);
void bf16_set_tile(int t);
void bf16_set_dma(int v);
void bf16_set_patch(int v);
void bf16_set_fused_stem(int on);
void bf16_set_stem_strip(int on, int band_rows);   // stem_bf16.hip: strip-march kernel (round 6) | tile kernel (rounds 3-5)
void bf16_set_chunk(int images, int min_mb);
void bf16_set_lean_epi(int on);
void bf16_set_persist(int v);
void bf16_set_fuse_block(int v);
void bf16_set_tail(int v);
void bf16_set_pair(int v);
void bf16_set_linear_dma(int v);
int bf16_set_ws64(int v);      // RPG_OK, or RPG_ERR_BAD_ARG for v != 0 in a build without the probe kernel (tools/probes/conv3x3_bf16_ws64.inc)
bool stem_pool_bf16_supported(int n, int h, int w, int cout);      // incl. the ranges the kernel's magic divisions are exact for
int launch_stem_pool_bf16(const void* x_nchw, int x_is_bf16, const void* wpack, const float* scale, const float* shift, void* out, int n, int h,
                          int w, hipStream_t s);
int launch_f32_to_bf16(const float* src, int ld_src, void* dst, int ld_dst, int col_off, long rows, int cols, hipStream_t s);
int launch_linear_bf16(const void* a, const void* w, const float* bias, const float* res, const int64_t* res_idx,
                       const float* res2, const int64_t* res2_idx, int ldr, float* out, int m, int k, int n_out, int relu,
                       hipStream_t s);
struct LinearBf16Out {
    void* out = nullptr;       // primary output [m][n_out]: fp32 (out_f32) or bf16; may be null when out2 is given
    int out_f32 = 1;
    void* out2 = nullptr;      // optional second output, bf16, pitch ld2 elements: bf16(relu2 ? max(y, 0) : y), y = before the primary's ReLU
    int ld2 = 0;
    int relu2 = 0;
};
int launch_linear_bf16_ex(const void* a, int lda, const void* w, const float* bias, const float* res, const int64_t* res_idx,
                          const float* res2, const int64_t* res2_idx, int ldr, const LinearBf16Out& o, int m, int k, int n_out,
                          int relu, hipStream_t s);
// Gathered residual rows added in the epilogue: out[m] += res1[idx1[m]] (+ res2[idx2[m]]), row pitch ld.
struct GatherRes {
    const float* res1;
    const int64_t* idx1;
    const float* res2;
    const int64_t* idx2;
    int ld;
};
// out_relu: optional second output of the same shape, = max(out, 0)
int launch_linear(const GatherSrc& src, const float* weight, const float* bias, const float* residual, float* out,
                  int m, int n_out, int relu, hipStream_t s, const GatherRes* gres = nullptr, float* out_relu = nullptr);
int launch_gather_add2_relu(const float* pq, const int64_t* lo, const int64_t* hi, const float* bias, float* out, int e,
                            int d, hipStream_t s);
bool gnn_split_enabled();
bool gnn_fuse_agg_enabled();
void bf16_set_bk(int bk);

}  // namespace rpg
