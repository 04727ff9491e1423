// Probe (gfx950): write-only bandwidth of the (hi, lo) epilogue's store shapes on a tensor the size of lu0.convT's output
// (237 images x 5 octets x 256 x 256 pixels x 16 bytes, two planes = 2.49 GB), 16-byte units, one `global_store_dwordx4` per lane:
//   contig : a wave instruction writes 1 KB contiguous                                    (the ceiling)
//   plain  : 4 runs of 256 B (2 octets x 2 pixel rows)                                    (plain convolution, planar output)
//   nph4   : 2 runs of 512 B, each written by two 16-lane rows at a 32-byte unit stride    (fused transposed convolution)
//   nph4c  : the same 2 runs of 512 B with lanes in address order                          (what a lane permutation would buy)
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/store_patterns.hip -o /tmp/store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kImgs = 237, kOct = 5, kS = 256;          // output tensor: [img][oct][y][x][8 halves]
constexpr size_t kPlane = (size_t)kImgs * kOct * kS * kS * 16;

// one workgroup = 4 waves x 2 M-tiles = 8 input rows segments of 16 pixels -> here: a 16-wide, 8-high block of input pixels
// = 32 x 16 output pixels, all 5 octets, both planes: 2 (m) x 2 (pu) x 3 (n) x 2 (planes) stores per wave
template <int PAT>
__global__ __launch_bounds__(256) void k(unsigned char* hi, unsigned char* lo, uint4 v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, li = lane & 15;
    const int wg = blockIdx.x;
    if constexpr (PAT == 0) {
        // 24 KB-instructions per wave, 20 of them live (5 of 6 octets): contiguous
        const size_t base = ((size_t)wg * 4 + wave) * 10 * 1024;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            *reinterpret_cast<uint4*>(hi + base + j * 1024 + lane * 16) = v;
            *reinterpret_cast<uint4*>(lo + base + j * 1024 + lane * 16) = v;
        }
        return;
    }
    // an image has 128 x 128 input pixels = 8 x 16 blocks of (16 wide, 8 high); output row 2y+pu holds 32 pixels of the block
    const int x0 = (wg & 7) * 16, y0 = ((wg >> 3) & 15) * 8, im = wg >> 7;
    if (im >= kImgs) return;
    const size_t ibase = (size_t)im * kOct * kS * kS * 16;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int pu = 0; pu < 2; ++pu)
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const int y = y0 + wave * 2 + m, oct = n * 2 + (q >> 1);
                int pix;
                if constexpr (PAT == 1)            // 4 runs of 256 B: q even = left half of row (m), q odd = right half of row (m ^ 1)
                    pix = (q & 1) ? ((y0 + wave * 2 + (m ^ 1)) * 2 + pu) * kS + x0 * 2 + 16 + li : (y * 2 + pu) * kS + x0 * 2 + li;
                else if constexpr (PAT == 2)       // nph4: pixel 2 li + (q & 1)
                    pix = (y * 2 + pu) * kS + x0 * 2 + 2 * li + (q & 1);
                else                               // nph4c: address order
                    pix = (y * 2 + pu) * kS + x0 * 2 + (q & 1) * 16 + li;
                if (oct < kOct) {
                    const size_t o = ibase + ((size_t)oct * kS * kS + pix) * 16;
                    *reinterpret_cast<uint4*>(hi + o) = v;
                    *reinterpret_cast<uint4*>(lo + o) = v;
                }
            }
}

int main() {
    unsigned char *hi, *lo;
    if (hipMalloc(&hi, kPlane) != hipSuccess || hipMalloc(&lo, kPlane) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const uint4 v = make_uint4(1, 2, 3, 4);
    const int wgs = kImgs * 128;
    const char* names[4] = {"contig", "plain (4 x 256 B)", "nph4 (2 x 512 B, interleaved rows)", "nph4c (2 x 512 B, address order)"};
    for (int rep = 0; rep < 2; ++rep)
        for (int pat = 0; pat < 4; ++pat) {
            float best = 1e9f;
            for (int it = 0; it < 6; ++it) {
                hipEventRecord(e0, 0);
                if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, hi, lo, v);
                if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, hi, lo, v);
                if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, hi, lo, v);
                if (pat == 3) hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(256), 0, 0, hi, lo, v);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (it > 0 && ms < best) best = ms;
            }
            const double bytes = (double)wgs * 4 * 20 * 1024;
            printf("%-40s %.3f ms  %.2f TB/s  (%.2f GB)\n", names[pat], best, bytes / best * 1e-9, bytes * 1e-9);
        }
    return 0;
}
