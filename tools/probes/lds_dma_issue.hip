// Probe (gfx950): what one `buffer_load_dwordx4 ... lds` costs the ISSUING wave (s_memtime around a burst of 16, nothing waited
// for inside the bracket), as a function of (a) whether consecutive instructions change M0 (the LDS destination), (b) how many
// waves of the CU issue bursts at the same time, (c) whether the source lines are hot in L2.  conv_f16x3's stamps charge a halo
// piece (two such instructions) ~285 cycles; this separates the instruction's own issue cost from queueing behind other waves.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/lds_dma_issue.hip -o /tmp/lds_dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>   // 0: one LDS destination (M0 constant), 1: a new destination per instruction, 2: new destination via the 12-bit immediate
__global__ __launch_bounds__(1024) void k(const unsigned char* src, size_t span, long long* out, int reps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    unsigned char* const mine = smem + wave * 8192;
    long long total = 0;
    for (int r = 0; r < reps; ++r) {
        const size_t base = (((size_t)blockIdx.x * nw + wave) * reps + r) * 16384 % span;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + base), 0, 16384, 0x00020000);
        __builtin_amdgcn_s_barrier();
        const long long t0 = __builtin_amdgcn_s_memtime();
#define ONE(i)                                                                                                                      \
    do {                                                                                                                            \
        if (MODE == 0)                                                                                                              \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(mine), 16, lane * 16 + (i) * 1024, 0, 0, 0); \
        else if (MODE == 1)                                                                                                         \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(mine + ((i) & 7) * 1024), 16,       \
                                                     lane * 16 + (i) * 1024, 0, 0, 0);                                              \
        else /* the immediate moves the LDS and the memory address together */                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(mine + (((i) >> 2) & 1) * 4096), 16, \
                                                     lane * 16 + ((i) >> 2) * 4096, 0, ((i) & 3) * 1008, 0);                        \
    } while (0)
        ONE(0); ONE(1); ONE(2); ONE(3); ONE(4); ONE(5); ONE(6); ONE(7);
        ONE(8); ONE(9); ONE(10); ONE(11); ONE(12); ONE(13); ONE(14); ONE(15);
#undef ONE
        const long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        total += t1 - t0;
    }
    if (lane == 0) out[blockIdx.x * nw + wave] = total;
}

int main() {
    const size_t span = (size_t)1 << 30;
    unsigned char* src;
    long long* out;
    hipMalloc(&src, span + 65536);
    hipMemset(src, 1, span + 65536);
    hipMalloc(&out, 256 * 16 * sizeof(long long));
    const int reps = 64;
    printf("cycles per LDS-DMA instruction seen by the issuing wave (16 per burst, %d bursts, 256 workgroups)\n", reps);
    for (int waves : {1, 4, 8, 16}) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int hot = 0; hot < 2; ++hot) {
                const size_t sp = hot ? (size_t)16 << 20 : span;   // 16 MB: L2 / Infinity Cache resident after the first pass
                for (int pass = 0; pass < 2; ++pass) {
                    const int lds = 8192 * waves;
                    hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    hipFuncSetAttribute(reinterpret_cast<const void*>(k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), lds, 0, src, sp, out, reps);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), lds, 0, src, sp, out, reps);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), lds, 0, src, sp, out, reps);
                    hipDeviceSynchronize();
                }
                std::vector<long long> h(256 * waves);
                hipMemcpy(h.data(), out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
                double s = 0;
                for (long long v : h) s += (double)v;
                printf("%2d wave(s)/CU  %-28s %-8s %7.1f\n", waves,
                       mode == 0 ? "M0 constant" : mode == 1 ? "new M0 per instruction" : "M0 per 4, immediate offsets", hot ? "hot" : "HBM",
                       s / h.size() / reps / 16.0);
            }
        }
    }
    return 0;
}
