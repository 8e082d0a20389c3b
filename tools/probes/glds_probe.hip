// Probe of `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer resource) on gfx950, for the bf16 convolution kernel:
//   1. where lane l's 16 bytes land (M0 base + 16 * l, wave-uniform base);
//   2. what an OUT-OF-RANGE lane (voffset >= num_records) does to its 16 bytes of LDS: zero fill (needed for the image
//      borders / ragged rows of the implicit GEMM) or left untouched;
//   3. soffset (scalar) is added to the address but NOT to the range check?  (checked: voffset + soffset >= num_records)
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/glds_probe.hip -o gpurun_out/glds_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;

__global__ void probe(const unsigned* x, unsigned* y, int n_records_bytes, int soff) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 64 * 4 * 2; i += blockDim.x) lds[i] = 0xDEADBEEFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(x), 0, n_records_bytes, 0x00020000);
    // lane l reads 16 bytes at byte offset 32 * (63 - l) (reversed, stride 2 chunks): a gather; odd lanes of wave 1 are out of range
    unsigned voff = 32u * (unsigned)(63 - lane);
    if (wave == 1 && (lane & 1)) voff = 0x80000000u;
    if (wave == 1 && lane == 2) voff = (unsigned)n_records_bytes - 8u;      // straddles the end: partially out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(lds + wave * 256), 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 2 * 256; i += blockDim.x) y[i] = lds[i];
}

int main() {
    const int n = 4096;
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; ++i) h[i] = 0x1000u + i;
    unsigned *dx, *dy;
    hipMalloc(&dx, n * 4);
    hipMalloc(&dy, 512 * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int soff : {0, 64}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(128), 4096 * 4, 0, dx, dy, 2048 * 4, soff);
        std::vector<unsigned> o(512);
        hipMemcpy(o.data(), dy, 512 * 4, hipMemcpyDeviceToHost);
        printf("soffset %d bytes (%s)\n", soff, hipGetErrorString(hipGetLastError()));
        int ok_layout = 1;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j)
                if (o[4 * l + j] != 0x1000u + (unsigned)(8 * (63 - l) + j + soff / 4)) ok_layout = 0;
        printf("  wave 0: lane l -> LDS dwords 4l..4l+3 = x[8*(63-l) + soff/4 ..]: %s\n", ok_layout ? "YES" : "NO");
        printf("  wave 1 lane 0 (in range) : %08x %08x %08x %08x\n", o[256], o[257], o[258], o[259]);
        printf("  wave 1 lane 1 (OOB)      : %08x %08x %08x %08x   <- 0 = zero fill, deadbeef = untouched\n", o[260], o[261], o[262], o[263]);
        printf("  wave 1 lane 2 (straddle) : %08x %08x %08x %08x\n", o[264], o[265], o[266], o[267]);
        printf("  wave 1 lane 3 (OOB)      : %08x %08x %08x %08x\n", o[268], o[269], o[270], o[271]);
    }
    return 0;
}
