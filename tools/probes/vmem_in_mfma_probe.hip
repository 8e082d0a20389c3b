// Probe (round 3): what does one vector-memory instruction cost a wave that has a SIMD to itself and is otherwise issuing MFMAs?
// 4 waves per workgroup, one workgroup per CU (LDS 100 KB), each wave: `steps` x { 6 MFMAs 32x32x16 bf16 ; one VMEM op }.
// mode 0: no VMEM | 1: global_store_dwordx4 (1 KB per wave, fresh lines) | 2: buffer_load_dwordx4 ... lds (1 KB per wave) | 3: both
// | 4: global_load_dwordx4 into registers (consumed at the end).  Prints cycles per step (s_memtime) and the aggregate rates.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* dst, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)dst, 16, voff, soff, 0, 0);
}

__global__ __launch_bounds__(1024) void probe(const uint4* __restrict__ src, uint4* __restrict__ dst, int steps, int mode, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a = __builtin_bit_cast(bf16x8, src[lane]), b = __builtin_bit_cast(bf16x8, src[64 + lane]);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(src), 0, 0x7fffffff, 0x00020000);
    const int nwv = blockDim.x >> 6;
    const size_t wbase = ((size_t)blockIdx.x * nwv + wave) * (size_t)steps * 64;
    uint4 keep = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const long long t0 = clock64();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            if (i == 1) {
                if (mode & 1) dst[wbase + (size_t)s * 64 + lane] = make_uint4(s, lane, wave, 7);
                if (mode & 2) dma16(rs, lds + (wave & 3) * 16384 + (s & 15) * 1024, (unsigned)(((wbase + (size_t)s * 64) % (1u << 24) + lane) * 16), 0);
                if (mode & 4) { const uint4 v = src[(wbase + (size_t)s * 64) % (1u << 24) + lane]; keep.x ^= v.x; keep.y ^= v.y; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.f;
    for (int i = 0; i < 6; ++i) sum += acc[i][0];
    if (sum == 12345.f || keep.x == 0x12345u) dst[0] = make_uint4(1, 2, 3, 4);
    if (lane == 0) atomicAdd((unsigned long long*)cyc, (unsigned long long)(t1 - t0));
}

// the fp32 matrix pipe alone (v_mfma_f32_32x32x2f32, the Winograd kernel's instruction): what does it sustain?
typedef float f32x16b __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(1024) void probe_f32(const float* __restrict__ src, float* __restrict__ dst, int steps) {
    const int lane = threadIdx.x & 63;
    f32x16b acc[6];
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const float a = src[lane], b = src[64 + lane];
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sum = 0.f;
    for (int i = 0; i < 6; ++i) sum += acc[i][0];
    if (sum == 12345.f) dst[0] = sum;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 512;
    int dev = 0; hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, dev);
    const int cus = pr.multiProcessorCount;
    uint4 *src, *dst; long long* cyc;
    const size_t nsrc = (1u << 24) + 4096, ndst = (size_t)cus * 16 * steps * 64 + 64;
    (void)hipMalloc(&src, nsrc * 16); (void)hipMalloc(&dst, ndst * 16); (void)hipMalloc(&cyc, 8);
    (void)hipMemset(src, 0, nsrc * 16);
    (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    // the matrix pipe alone, 1 / 2 / 4 waves per SIMD: is 32 cycles per 32x32x16 bf16 MFMA (2.5 PFLOP/s) reachable at all?
    for (int nw : {4, 8, 16}) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipMemset(cyc, 0, 8);
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(cus), dim3(64 * nw), 100 * 1024, 0, src, dst, steps, 0, cyc);
            (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 1)
                printf("MFMA only, %2d waves per CU: %.1f us -> %.3f PFLOP/s (32x32x16 bf16, 6 independent accumulators per wave)\n", nw, ms * 1e3,
                       (double)cus * nw * steps * 6 * 32768.0 / (ms * 1e-3) / 1e15);
        }
    }
    for (int nw : {4, 8, 16}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(probe_f32, dim3(cus), dim3(64 * nw), 0, 0, (const float*)src, (float*)dst, steps * 2);
            (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 1)
                printf("fp32 MFMA only, %2d waves per CU: %.1f us -> %.1f TFLOP/s (32x32x2 f32, 6 independent accumulators per wave; data sheet 157.3)\n", nw,
                       ms * 1e3, (double)cus * nw * steps * 2 * 6 * 4096.0 / (ms * 1e-3) / 1e12);
        }
    }
    for (int mode : {0, 1, 2, 3, 4, 0}) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipMemset(cyc, 0, 8);
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(cus), dim3(256), 100 * 1024, 0, src, dst, steps, mode, cyc);
            (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            long long h = 0; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            if (rep == 1)
                printf("mode %d: %.1f cycles per step and wave (6 MFMAs = 192 ideal), %.1f us, per op 1 KB x %d waves -> %.2f TB/s\n", mode,
                       (double)h / ((double)cus * 4 * steps), ms * 1e3, cus * 4, (double)cus * 4 * steps * 1024.0 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
