// HBM-bound encoder kernels for gfx950: input re-layout, 3x3/2 max-pool, global average pool.
// All are one-float4-per-lane streaming kernels over NHWC tensors (coalesced 16 B/lane, 1 KiB per wave
// instruction); grids are capped and grid-strided so a launch is a few thousand workgroups at most.
// Reference ops replaced: x.view(N,3,H,-1) NCHW input (posenet.py:1035), nn.MaxPool2d(3,2,1) and
// nn.AdaptiveAvgPool2d(1) of the torchvision resnet34 passed in at testing/test.py:151.
#include "rpg_common.h"

namespace {

constexpr int NT = 256;
inline int capped_grid(long work_items) {
    long g = (work_items + NT - 1) / NT;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// [n][3][h][w] -> [n][h][w][4] with channel 3 = 0.  One lane per pixel: three coalesced plane reads, one 16 B store.
__global__ __launch_bounds__(NT) void nchw3_to_nhwc4_kernel(const float* __restrict__ x, float4* __restrict__ y,
                                                            long npix_total, int hw) {
    for (long p = (long)blockIdx.x * NT + threadIdx.x; p < npix_total; p += (long)gridDim.x * NT) {
        const long n = p / hw;
        const int q = (int)(p - n * hw);
        const float* b = x + n * 3 * (long)hw + q;
        y[p] = make_float4(b[0], b[hw], b[2 * (long)hw], 0.f);
    }
}

// 3x3 window, stride 2, padding 1 (padding never wins: out-of-range taps are skipped), floor mode.
__global__ __launch_bounds__(NT) void maxpool3x3s2_kernel(const float4* __restrict__ x, float4* __restrict__ y,
                                                          int h, int w, int c4, int ho, int wo, long total) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i % c4);
        long t = i / c4;
        const int ox = (int)(t % wo);
        t /= wo;
        const int oy = (int)(t % ho);
        const long n = t / ho;
        const float4* img = x + n * (long)h * w * c4;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)w) continue;
                const float4 v = img[((long)iy * w + ix) * c4 + c];
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        y[i] = m;
    }
}

// [n][hw][c] -> [n][c]: one lane per (n, c/4), pixels summed in ascending order then divided by hw.
__global__ __launch_bounds__(NT) void global_avgpool_kernel(const float4* __restrict__ x, float4* __restrict__ y,
                                                            int hw, int c4, long total) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i % c4);
        const long n = i / c4;
        const float4* p = x + n * (long)hw * c4 + c;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = 0; q < hw; ++q) {
            const float4 v = p[(long)q * c4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const float d = (float)hw;
        y[i] = make_float4(s.x / d, s.y / d, s.z / d, s.w / d);
    }
}

}  // namespace

extern "C" int rpg_nchw3_to_nhwc4_f32(const float* x_nchw, float* y_nhwc4, int n, int h, int w, void* stream) {
    if (!x_nchw || !y_nhwc4 || n <= 0 || h <= 0 || w <= 0 || !rpg::aligned16(y_nhwc4)) return RPG_ERR_BAD_ARG;
    const long total = (long)n * h * w;
    hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(capped_grid(total)), dim3(NT), 0, rpg::as_stream(stream), x_nchw,
                       reinterpret_cast<float4*>(y_nhwc4), total, h * w);
    RPG_CHECK_LAUNCH("nchw3_to_nhwc4");
    return RPG_OK;
}

extern "C" int rpg_maxpool3x3s2_nhwc_f32(const float* x, float* y, int n, int h, int w, int c, void* stream) {
    if (!x || !y || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) || !rpg::aligned16(x) || !rpg::aligned16(y))
        return RPG_ERR_BAD_ARG;
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long total = (long)n * ho * wo * (c / 4);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(capped_grid(total)), dim3(NT), 0, rpg::as_stream(stream),
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), h, w, c / 4, ho, wo, total);
    RPG_CHECK_LAUNCH("maxpool3x3s2");
    return RPG_OK;
}

extern "C" int rpg_global_avgpool_nhwc_f32(const float* x, float* y, int n, int hw, int c, void* stream) {
    if (!x || !y || n <= 0 || hw <= 0 || c <= 0 || (c & 3) || !rpg::aligned16(x) || !rpg::aligned16(y))
        return RPG_ERR_BAD_ARG;
    const long total = (long)n * (c / 4);
    hipLaunchKernelGGL(global_avgpool_kernel, dim3(capped_grid(total)), dim3(NT), 0, rpg::as_stream(stream),
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), hw, c / 4, total);
    RPG_CHECK_LAUNCH("global_avgpool");
    return RPG_OK;
}
