// Sustained bf16 MFMA rate of the whole chip (gfx950) as a function of run length and operand data: is the 2.5 PFLOP/s data-sheet
// peak (256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz) reachable by a chip-wide stream of v_mfma_f32_32x32x16_bf16 with NO memory
// traffic at all, or does the power management pull the clock down?  (The f32 counterpart, mfma_power_probe.hip, holds 155.9 of
// 157.3 TFLOP/s.)  Every wave runs `iters` x 16 MFMAs on 4 independent accumulators; HIP events give TFLOP/s, s_memtime the
// average shader clock.  This is the ceiling every bf16 convolution kernel of the encoder is priced against in practice.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mppb tools/probes/mfma_power_probe_bf16.hip && /tmp/mppb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void mfma_loop(const uint4* __restrict__ src, float* sink, long iters, unsigned long long* cyc) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, src[(t * 8 + i) & 65535]);
        b[i] = __builtin_bit_cast(bf16x8, src[(t * 8 + 4 + i) & 65535]);
    }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (long it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[k], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[(k + 1) & 3], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(k + 1) & 3], b[k], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(k + 2) & 3], b[(k + 3) & 3], c3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    std::vector<unsigned short> h(65536 * 8);
    uint4* src;
    float* sink;
    unsigned long long* cyc;
    hipMalloc(&src, 65536 * 16);
    hipMalloc(&sink, 256);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int data = 0; data < 3; ++data) {          // 0: zeros, 1: random bf16 in (-1e-3, 1e-3) (sums stay finite), 2: activation-like: half zeros (ReLU), half |N(0,1)|-ish
        for (auto& v : h) {
            float f = 0.f;
            if (data == 1) f = (rand() / (float)RAND_MAX - 0.5f) * 2e-3f;
            if (data == 2) f = (rand() & 1) ? 0.f : (rand() / (float)RAND_MAX) * 2e-3f;
            unsigned u;
            memcpy(&u, &f, 4);
            v = (unsigned short)(u >> 16);
        }
        hipMemcpy(src, h.data(), 65536 * 16, hipMemcpyHostToDevice);
        for (int wg = 256; wg <= 512; wg *= 2) {    // 256 x 512 threads = 2 waves per SIMD; 512 workgroups = 4 waves per SIMD
            for (long iters : {2000L, 20000L, 200000L, 2000000L}) {
                hipLaunchKernelGGL(mfma_loop, dim3(wg), dim3(512), 0, 0, src, sink, iters / 10 + 1, cyc);    // warm
                hipDeviceSynchronize();
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(mfma_loop, dim3(wg), dim3(512), 0, 0, src, sink, iters, cyc);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                unsigned long long c = 0;
                hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
                const double flop = (double)wg * 8 * iters * 16 * 32768.0;
                printf("data %d  wg %d  iters %8ld  %9.3f ms  %7.1f TFLOP/s  (%.3f of 2516.6)  cycles %llu -> %.3f GHz, %.1f cycles per MFMA per SIMD\n",
                       data, wg, iters, ms, flop / ms / 1e9, flop / ms / 1e9 / 2516.6, c, c / (ms * 1e6),
                       (double)c / (iters * 16.0 * (wg / 256) * 2));
                fflush(stdout);
            }
        }
    }
    return 0;
}
