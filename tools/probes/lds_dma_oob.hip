// Probe (gfx950): semantics of buffer_load_dwordx4 ... offen lds for (a) lanes whose offset is out of the descriptor's range,
// (b) exec-masked lanes, (c) an M0 base that is 16-byte but not 1-KiB aligned, (d) soffset excluded/included in the range check.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/lds_dma_oob.hip -o tools/probes/lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* src, float* out, int nbytes, int soff, int nact, int ldsbase) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 1024; i += 64) ((float*)smem)[i] = -7.f;   // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    int voff = threadIdx.x * 16;
    if ((threadIdx.x & 3) == 1) voff = 0x7fffff00;          // far out of range
    if ((threadIdx.x & 3) == 2) voff = nbytes - soff + 16 * (threadIdx.x >> 2);   // in range only if soffset is NOT part of the check
    if (threadIdx.x < nact)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + ldsbase), 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = ((float*)smem)[i];
}
int main() {
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, n * 4 * 2); hipMalloc(&o, 1024 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset((char*)d + n * 4, 0x7f, n * 4);   // memory beyond the descriptor: must never show up
    const int nbytes = 2048, soff = 512, nact = 60, ldsbase = 80;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096 + 128, 0, d, o, nbytes, soff, nact, ldsbase);
    std::vector<float> r(1024);
    hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    printf("lane : first float written at smem[ldsbase + 16*lane]\n");
    for (int l = 0; l < 64; ++l) printf("lane %2d (kind %d): %g %g\n", l, l & 3, r[ldsbase / 4 + 4 * l], r[ldsbase / 4 + 4 * l + 3]);
    printf("before base: %g, after last lane: %g\n", r[ldsbase / 4 - 1], r[ldsbase / 4 + 256]);
    return 0;
}
