// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes the Winograd kernel uses.
// MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read ("double it"), but
// "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".
// Every kernel below touches EXACTLY `bytes` bytes of a buffer larger than L2 + MALL (1 GiB), once:
//   stream_b128     64 lanes x 16 B contiguous per instruction (the guide's calibrated case)
//   seg64_b128      the Winograd A-operand shape: 4 lanes x 16 B = one 64-byte segment (16 channels of a pixel) per row,
//                   rows `pitch` bytes apart (pitch = 256: Cin = 64; the other three 64-byte segments of a row are read
//                   by the following three "K steps" of the same workgroup, like the kernel does)
//   seg64_once      the same, but only the FIRST 64-byte segment of every 256-byte row is ever read (what a 64-byte
//                   request costs on its own)
//   store_b128      64 lanes x 16 B contiguous stores
// Run under rocprofv3 --pmc FETCH_SIZE and, separately, --pmc WRITE_SIZE; compare counter x 1024 with the printed bytes.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fc tools/probes/fetch_calib_probe.hip && rocprofv3 --kernel-trace --pmc FETCH_SIZE -- /tmp/fc
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void stream_b128(const float4* __restrict__ p, float4* sink, long n4) {
    float4 acc = make_float4(0, 0, 0, 0);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = p[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x == 123.456f) sink[0] = acc;
}

// rows of `pitch4` float4; a workgroup owns 64 consecutive rows and walks their 64-byte segments seg = 0 .. nseg-1 in
// turn (thread = (row, 16-byte slot)), like the K steps of the Winograd kernel walk the channels of its pixels
__global__ __launch_bounds__(256) void seg64_b128(const float4* __restrict__ p, float4* sink, long rows, int pitch4, int nseg) {
    float4 acc = make_float4(0, 0, 0, 0);
    const int r = threadIdx.x >> 2, slot = threadIdx.x & 3;
    for (long r0 = (long)blockIdx.x * 64; r0 < rows; r0 += (long)gridDim.x * 64) {
        for (int seg = 0; seg < nseg; ++seg) {
            const float4 v = p[(r0 + r) * pitch4 + seg * 4 + slot];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            __syncthreads();
        }
    }
    if (acc.x == 123.456f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void store_b128(float4* __restrict__ p, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
        p[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}

int main() {
    const long bytes = 1L << 30;
    float4 *buf, *sink;
    hipMalloc(&buf, bytes);
    hipMalloc(&sink, 256);
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    const long n4 = bytes / 16;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream_b128, dim3(4096), dim3(256), 0, 0, buf, sink, n4);
        hipLaunchKernelGGL(seg64_b128, dim3(4096), dim3(256), 0, 0, buf, sink, bytes / 256, 16, 4);     // all 4 segments: 1 GiB
        hipLaunchKernelGGL(seg64_b128, dim3(4096), dim3(256), 0, 0, buf, sink, bytes / 256, 16, 1);     // first segment only: 256 MiB
        hipLaunchKernelGGL(seg64_b128, dim3(4096), dim3(256), 0, 0, buf, sink, bytes / 512, 32, 8);     // Cin = 128 rows: 1 GiB
        hipLaunchKernelGGL(store_b128, dim3(4096), dim3(256), 0, 0, buf, n4);
    }
    hipDeviceSynchronize();
    printf("bytes touched per launch: stream_b128 %ld, seg64_b128(pitch 256, 4 segs) %ld, seg64_b128(1 seg) %ld, "
           "seg64_b128(pitch 512, 8 segs) %ld, store_b128 %ld\n", bytes, bytes, bytes / 4, bytes, bytes);
    return 0;
}
