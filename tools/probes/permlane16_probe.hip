// Probe (gfx950): which lanes v_permlane16_swap_b32 exchanges, through the hipcc builtin with two DIFFERENT operands.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/permlane16_probe.hip -o /tmp/permlane16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    unsigned a = 1000 + l, b = 2000 + l;
    asm volatile("" : "+v"(a), "+v"(b));
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[l] = r[0];
    out[64 + l] = r[1];
}
int main() {
    unsigned* d;
    hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[128];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int row = 0; row < 4; ++row)
        printf("row %d (lanes %2d..%2d): r0 = %u..%u   r1 = %u..%u\n", row, 16 * row, 16 * row + 15, h[16 * row], h[16 * row + 15],
               h[64 + 16 * row], h[64 + 16 * row + 15]);
    return 0;
}
