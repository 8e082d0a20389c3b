// How much other work hides behind v_mfma_f32_32x32x2_f32 on gfx950?  One wave per SIMD (256 threads, 96 KB LDS per
// workgroup, one workgroup per CU); each loop iteration issues 8 independent-accumulator MFMAs, each followed by NV
// v_fma_f32, NDR ds_read_b128, NDW ds_write_b128 and NVM buffer_load_dwordx4 (L2-resident).  Prints shader cycles per
// MFMA (64 = the pipe's own rate).   hipcc --offload-arch=gfx950 -O3 -o mfma_shadow_probe mfma_shadow_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NDR, int NDW, int NVM, int NW, int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ g, float* out, int iters, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 24576; i += 256) lds[i] = (float)i;
    __syncthreads();
    if (NW < 4 && (tid >> 6) >= NW) return;
    f32x16 acc[6];
    for (int a = 0; a < 6; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    float x0 = tid, x1 = 1.0f, x2 = 0.5f;
    float y[8] = {0};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 z[4] = {}, p1 = {1.f, 2.f}, p2 = {0.5f, 0.25f};
    int iy[8] = {0}, i1 = tid, i2 = 3;
    f32x4 r = {0, 0, 0, 0}, w = {1, 2, 3, 4}, v = {0, 0, 0, 0};
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, 1 << 20, 0x00020000);
    const unsigned laddr = (unsigned)(tid * 16);
    float a = 1.0f, b = 2.0f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m % 6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m % 6], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x0) : "v"(x1), "v"(x2));          // dependent chain
                if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(y[k & 7]) : "v"(x1), "v"(x2));     // independent
                if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %1" : "=v"(z[k & 3]) : "v"(p1), "v"(p2));  // independent, packed
                if (MODE == 3) asm volatile("v_add_u32 %0, %1, %2" : "=v"(iy[k & 7]) : "v"(i1), "v"(i2));         // independent int
            }
#pragma unroll
            for (int k = 0; k < NDR; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(laddr), "n"(4096 * k));
#pragma unroll
            for (int k = 0; k < NDW; ++k) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(laddr), "v"(w), "n"(8192 + 4096 * k));
#pragma unroll
            for (int k = 0; k < NVM; ++k) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(laddr), "s"(rs));
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = x0 + r.x + v.x;
    for (int k = 0; k < 8; ++k) s += y[k] + (float)iy[k];
    for (int k = 0; k < 4; ++k) s += z[k].x;
    for (int a2 = 0; a2 < 6; ++a2) s += acc[a2][0];
    if (s == 123.456f) out[tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NV, int NDR, int NDW, int NVM, int NW = 4, int MODE = 0, int BPC = 1>
void run(const char* name, const float* g, float* out, long long* cyc) {
    const int iters = 2000, lds_bytes = BPC == 1 ? 98304 : 65536;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<NV, NDR, NDW, NVM, NW, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NV, NDR, NDW, NVM, NW, MODE>), dim3(256 * BPC), dim3(256), lds_bytes, 0, g, out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NV, NDR, NDW, NVM, NW, MODE>), dim3(256 * BPC), dim3(256), lds_bytes, 0, g, out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long c = 0;
    hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    // s_memtime counts at a fixed 100 MHz on gfx9; report wall time per MFMA in ns and in 2.4 GHz cycles
    const double ns = ms * 1e6 / (iters * 8.0 * BPC);   // per MFMA issued on a SIMD
    printf("%-34s NV=%2d NDR=%d NDW=%d NVM=%d waves=%d  %7.2f ns/MFMA = %6.1f cyc@2.4GHz   (counter %lld)\n", name, NV, NDR, NDW, NVM,
           NW, ns, ns * 2.4, c);
}

int main() {
    float *g, *out;
    long long* cyc;
    hipMalloc(&g, 1 << 20);
    hipMemset(g, 0, 1 << 20);
    hipMalloc(&out, 4096);
    hipMalloc(&cyc, 8);
    run<0, 0, 0, 0>("bare MFMA", g, out, cyc);
    run<0, 0, 0, 0, 1>("bare MFMA, 1 wave/CU", g, out, cyc);
    run<4, 0, 0, 0>("+4 v_fma", g, out, cyc);
    run<8, 0, 0, 0>("+8 v_fma", g, out, cyc);
    run<12, 0, 0, 0>("+12 v_fma", g, out, cyc);
    run<16, 0, 0, 0>("+16 v_fma", g, out, cyc);
    run<24, 0, 0, 0>("+24 v_fma", g, out, cyc);
    run<0, 1, 0, 0>("+1 ds_read_b128", g, out, cyc);
    run<0, 2, 0, 0>("+2 ds_read_b128", g, out, cyc);
    run<0, 4, 0, 0>("+4 ds_read_b128", g, out, cyc);
    run<0, 0, 1, 0>("+1 ds_write_b128", g, out, cyc);
    run<0, 0, 2, 0>("+2 ds_write_b128", g, out, cyc);
    run<0, 0, 0, 1>("+1 buffer_load_dwordx4", g, out, cyc);
    run<0, 0, 0, 2>("+2 buffer_load_dwordx4", g, out, cyc);
    run<4, 1, 0, 0>("+4 v_fma +1 ds_read", g, out, cyc);
    run<8, 1, 1, 0>("+8 v_fma +1 ds_read +1 ds_write", g, out, cyc);
    run<4, 1, 0, 1>("+4 v_fma +1 ds_read +1 vmem", g, out, cyc);
    run<4, 0, 0, 0, 4, 1>("+4 independent v_fma", g, out, cyc);
    run<8, 0, 0, 0, 4, 1>("+8 independent v_fma", g, out, cyc);
    run<16, 0, 0, 0, 4, 1>("+16 independent v_fma", g, out, cyc);
    run<4, 0, 0, 0, 4, 2>("+4 independent v_pk_fma", g, out, cyc);
    run<8, 0, 0, 0, 4, 2>("+8 independent v_pk_fma", g, out, cyc);
    run<8, 0, 0, 0, 4, 3>("+8 independent v_add_u32", g, out, cyc);
    run<16, 0, 0, 0, 4, 3>("+16 independent v_add_u32", g, out, cyc);
    run<0, 0, 0, 0, 4, 0, 2>("bare MFMA, 2 waves/SIMD", g, out, cyc);
    run<8, 0, 0, 0, 4, 1, 2>("+8 indep v_fma, 2 waves/SIMD", g, out, cyc);
    run<16, 0, 0, 0, 4, 1, 2>("+16 indep v_fma, 2 waves/SIMD", g, out, cyc);
    run<8, 0, 0, 0, 4, 3, 2>("+8 indep v_add_u32, 2 waves/SIMD", g, out, cyc);
    run<0, 1, 1, 1, 4, 0, 2>("+1 dsr +1 dsw +1 vmem, 2 waves/SIMD", g, out, cyc);
    return 0;
}
