// The bf16 matrix pipe's sustained chip-wide rate WITH the operand traffic a convolution tile needs (gfx950, power cap): like
// mfma_power_probe_bf16.hip (registers only: 2.5 / 2.1 / 1.8 PFLOP/s on zero / ReLU-like / random operands), but every 16 MFMAs are
// fed by NR ds_read_b128 of fresh operands from LDS (the 256 x 256 patch kernel reads 12 fragments per 16 MFMAs; the 512 x 64 block
// kernel 12 per 10).  What comes out is the ceiling of an ideal kernel whose only overhead is its own LDS operand reads.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mlpb tools/probes/mfma_lds_power_probe_bf16.hip && /tmp/mlpb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NR>
__global__ __launch_bounds__(512) void mfma_lds_loop(const uint4* __restrict__ src, float* sink, long iters) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 512) lds[i] = src[(blockIdx.x * 8192 + i) & 65535];      // 128 KB of operands
    __syncthreads();
    bf16x8 f[12];
    for (int i = 0; i < 12; ++i) f[i] = __builtin_bit_cast(bf16x8, lds[(tid + 512 * i) & 8191]);
    f32x16 c[4] = {};
    unsigned base = tid;
    for (long it = 0; it < iters; ++it) {
        base = (base + 517) & 8191;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
                if (k * 3 + r < NR) f[k * 3 + r] = __builtin_bit_cast(bf16x8, lds[(base + 512 * (k * 3 + r)) & 8191]);
            c[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[(k * 3 + 3) % 12], f[(k * 3 + 4) % 12], c[0], 0, 0, 0);
            c[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[(k * 3 + 3) % 12], f[(k * 3 + 5) % 12], c[1], 0, 0, 0);
            c[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[(k * 3 + 6) % 12], f[(k * 3 + 4) % 12], c[2], 0, 0, 0);
            c[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[(k * 3 + 6) % 12], f[(k * 3 + 5) % 12], c[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c[0][i] + c[1][i] + c[2][i] + c[3][i];
    if (s == 123.456f) sink[0] = s;
}

template <int NR>
static void run(const uint4* src, float* sink, const char* what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_lds_loop<NR>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    for (long iters : {20000L, 200000L}) {
        hipLaunchKernelGGL(mfma_lds_loop<NR>, dim3(256), dim3(512), 128 * 1024, 0, src, sink, iters / 10 + 1);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_lds_loop<NR>, dim3(256), dim3(512), 128 * 1024, 0, src, sink, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 256.0 * 8 * iters * 16 * 32768.0;
        printf("%s  %2d ds_read_b128 per 16 MFMAs  iters %7ld  %9.3f ms  %7.1f TFLOP/s (%.3f of 2516.6)\n", what, NR, iters, ms, flop / ms / 1e9,
               flop / ms / 1e9 / 2516.6);
        fflush(stdout);
    }
}

int main() {
    std::vector<unsigned short> h(65536 * 8);
    uint4* src;
    float* sink;
    hipMalloc(&src, 65536 * 16);
    hipMalloc(&sink, 256);
    for (int data = 1; data < 3; ++data) {          // 1: random bf16, 2: ReLU-like (half zeros)
        for (auto& v : h) {
            float f = 0.f;
            if (data == 1) f = (rand() / (float)RAND_MAX - 0.5f) * 2e-3f;
            if (data == 2) f = (rand() & 1) ? 0.f : (rand() / (float)RAND_MAX) * 2e-3f;
            unsigned u;
            memcpy(&u, &f, 4);
            v = (unsigned short)(u >> 16);
        }
        hipMemcpy(src, h.data(), 65536 * 16, hipMemcpyHostToDevice);
        const char* what = data == 1 ? "random   " : "relu-like";
        run<0>(src, sink, what);
        run<6>(src, sink, what);
        run<12>(src, sink, what);
    }
    return 0;
}
