// Bare-loop probes (gfx950, MI355X), LDS-fed operands, random binary16 data, fp32 accumulate:
//   shape A/B   v_mfma_f32_16x16x32_f16 vs v_mfma_f32_32x32x16_f16 at the SAME 64 x 64 output tile per wave and the same
//               ds_read_b128 traffic per K = 32 (cdna_hip_programming.md section 5.4 rule 28 / MI355X_MICROARCH.md DVFS item 7)
//   loop shapes the inner loop of conv_f16x3<9, 4, 1> (4 pixel tiles x 9 channel tiles, 3 split-precision products: 26 fragment
//               reads per 108 MFMAs) against the inner loop a Winograd F(2x2, 3x3) kernel would run with the output transform
//               in-lane (16 positions x 1 patch tile x 2 channel tiles: 96 fragment reads per 96 MFMAs)
// Each variant: grid = 256 CUs x WGS workgroups of 256 threads, ITERS loop trips, timed with HIP events over REPS launches
// after a 2 s warm-up; in-kernel clock = d(s_memtime) / d(s_memrealtime) * 100 MHz of block 0.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_loops.hip -o /tmp/mfma_loops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LDS_HALVES = 32 * 1024;   // 64 KB of operands

__device__ __forceinline__ void fill_lds(_Float16* lds, const _Float16* src) {
    for (int i = threadIdx.x; i < LDS_HALVES / 8; i += blockDim.x)
        reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(src)[i + blockIdx.x % 7 * 64];
    __syncthreads();
}
struct Stamp { long long c0, r0; };
__device__ __forceinline__ Stamp stamp() { return {(long long)__builtin_amdgcn_s_memtime(), (long long)__builtin_amdgcn_s_memrealtime()}; }

// ---- 16x16x32: 4 x 4 tiles per wave; the fragments of trip it+1 are read in front of the MFMAs of trip it
__global__ void __launch_bounds__(256) k_16(const _Float16* src, float* out, long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    fill_lds(lds, src);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0, 0, 0, 0};
    h8 a[2][4], b[2][4];
    auto load = [&](int it, h8 (&A)[4], h8 (&B)[4]) {
        const _Float16* base = lds + ((it * 8 + wave * 2) & 31) * 512 + lane * 8;   // walks the buffer, 1 KiB per fragment
#pragma unroll
        for (int m = 0; m < 4; ++m) A[m] = *reinterpret_cast<const h8*>(base + m * 512);
#pragma unroll
        for (int n = 0; n < 4; ++n) B[n] = *reinterpret_cast<const h8*>(base + (4 + n) * 512);
    };
    load(0, a[0], b[0]);
    const Stamp s0 = stamp();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            load(it + u + 1, a[u ^ 1], b[u ^ 1]);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u][m], b[u][n], acc[m][n], 0, 0, 0);
        }
    }
    const Stamp s1 = stamp();
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = s1.c0 - s0.c0; clk[1] = s1.r0 - s0.r0; }
}
// ---- 32x32x16: 2 x 2 tiles per wave, two K = 16 sub-steps per trip, the same prefetch
__global__ void __launch_bounds__(256) k_32(const _Float16* src, float* out, long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    fill_lds(lds, src);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0;
    h8 a[2][2][2], b[2][2][2];
    auto load = [&](int it, h8 (&A)[2][2], h8 (&B)[2][2]) {
        const _Float16* base = lds + ((it * 8 + wave * 2) & 31) * 512 + lane * 8;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                A[k][m] = *reinterpret_cast<const h8*>(base + (k * 2 + m) * 512);
                B[k][m] = *reinterpret_cast<const h8*>(base + (4 + k * 2 + m) * 512);
            }
    };
    load(0, a[0], b[0]);
    const Stamp s0 = stamp();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            load(it + u + 1, a[u ^ 1], b[u ^ 1]);
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u][k][m], b[u][k][n], acc[m][n], 0, 0, 0);
        }
    }
    const Stamp s1 = stamp();
    float s = 0;
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) s += acc[m][n][0] + acc[m][n][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = s1.c0 - s0.c0; clk[1] = s1.r0 - s0.r0; }
}
// ---- conv_f16x3<9, 4, 1>-shaped trip: 4 pixel tiles (hi, lo) x 9 channel tiles (hi, lo), 3 products each
__global__ void __launch_bounds__(256, 2) k_conv9(const _Float16* src, float* out, long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    fill_lds(lds, src);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[4][9];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 9; ++n) acc[m][n] = (f32x4){0, 0, 0, 0};
    const Stamp s0 = stamp();
    for (int it = 0; it < iters; ++it) {
        const _Float16* base = lds + ((it * 26 + wave * 8) & 31) * 512 + lane * 8;
        h8 ah[4], al[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            ah[m] = *reinterpret_cast<const h8*>(base + (2 * m) * 512);
            al[m] = *reinterpret_cast<const h8*>(base + (2 * m + 1) * 512);
        }
#pragma unroll
        for (int n = 0; n < 9; ++n) {
            __builtin_amdgcn_iglp_opt(0);
            const h8 bh = *reinterpret_cast<const h8*>(base + (8 + 2 * n) * 512);
            const h8 bl = *reinterpret_cast<const h8*>(base + (9 + 2 * n) * 512);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 c = acc[m][n];
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al[m], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah[m], c, 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah[m], c, 0, 0, 0);
            }
        }
    }
    const Stamp s1 = stamp();
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 9; ++n) s += acc[m][n][0] + acc[m][n][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = s1.c0 - s0.c0; clk[1] = s1.r0 - s0.r0; }
}
// ---- Winograd F(2x2, 3x3)-shaped trip: 16 positions x (1 patch tile (hi, lo) x 2 channel tiles (hi, lo)), 3 products each
__global__ void __launch_bounds__(256, 2) k_wino(const _Float16* src, float* out, long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    fill_lds(lds, src);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[16][2];
    for (int p = 0; p < 16; ++p) for (int n = 0; n < 2; ++n) acc[p][n] = (f32x4){0, 0, 0, 0};
    const Stamp s0 = stamp();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            __builtin_amdgcn_iglp_opt(0);
            const _Float16* bp = lds + ((it * 6 + wave * 8 + p * 6) & 31) * 512 + lane * 8;
            const h8 vh = *reinterpret_cast<const h8*>(bp), vl = *reinterpret_cast<const h8*>(bp + 512);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const h8 uh = *reinterpret_cast<const h8*>(bp + (2 + 2 * n) * 512);
                const h8 ul = *reinterpret_cast<const h8*>(bp + (3 + 2 * n) * 512);
                f32x4 c = acc[p][n];
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(uh, vl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ul, vh, c, 0, 0, 0);
                acc[p][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(uh, vh, c, 0, 0, 0);
            }
        }
    }
    const Stamp s1 = stamp();
    float s = 0;
    for (int p = 0; p < 16; ++p) for (int n = 0; n < 2; ++n) s += acc[p][n][0] + acc[p][n][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = s1.c0 - s0.c0; clk[1] = s1.r0 - s0.r0; }
}

// ---- Winograd-shaped trip, positions split over the waves: wave w owns positions 4w..4w+3 for 4 patch tiles x 2 channel tiles
// (48 fragment reads per 96 MFMAs; the output transform then needs one accumulator exchange through LDS per K loop, not per trip)
__global__ void __launch_bounds__(256, 2) k_wino_split(const _Float16* src, float* out, long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    fill_lds(lds, src);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[4][4][2];
    for (int p = 0; p < 4; ++p) for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) acc[p][m][n] = (f32x4){0, 0, 0, 0};
    const Stamp s0 = stamp();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            __builtin_amdgcn_iglp_opt(0);
            const _Float16* bp = lds + ((it * 12 + wave * 5 + p * 12) & 31) * 512 + lane * 8;
            h8 vh[4], vl[4], uh[2], ul[2];
#pragma unroll
            for (int m = 0; m < 4; ++m) { vh[m] = *reinterpret_cast<const h8*>(bp + (2 * m) * 512); vl[m] = *reinterpret_cast<const h8*>(bp + (2 * m + 1) * 512); }
#pragma unroll
            for (int n = 0; n < 2; ++n) { uh[n] = *reinterpret_cast<const h8*>(bp + (8 + 2 * n) * 512); ul[n] = *reinterpret_cast<const h8*>(bp + (9 + 2 * n) * 512); }
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    f32x4 c = acc[p][m][n];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(uh[n], vl[m], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ul[n], vh[m], c, 0, 0, 0);
                    acc[p][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(uh[n], vh[m], c, 0, 0, 0);
                }
        }
    }
    const Stamp s1 = stamp();
    float s = 0;
    for (int p = 0; p < 4; ++p) for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) s += acc[p][m][n][0] + acc[p][m][n][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = s1.c0 - s0.c0; clk[1] = s1.r0 - s0.r0; }
}

// ---- the MIXED trip of the FP8 / FP6 cross-term plan (accuracy: tests/fp8_cross_term_report.py): per K = 128 and
// (pixel tile, channel tile) pair  4 x v_mfma_f32_16x16x32_f16 (x_hi * w_hi)  +  2 x v_mfma_scale_f32_16x16x128_f8f6f4 whose K axis is the
// concatenation [64 K of cross term 1 | 64 K of cross term 2] (A = [q(w_lo) | q(w_hi)], B = [q(x_hi) | q(x_lo)]): 144 + 72 matrix
// instructions per wave instead of 4 x 108 = 432 binary16 ones.  FMT 0 = e4m3 (8 VGPRs per operand, a scaled MFMA = 2 binary16 MFMAs of
// time), FMT 2 = e2m3 (6 VGPRs, = 1 binary16 MFMA of time).  Operand bytes come from LDS in conflict-free 16-byte planes like the
// production fragments; scales are 2^0.  What it prices: the ceiling of the plan against `conv_f16x3<9,4,1>-shaped` at the same K.
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int FMT>
__global__ void __launch_bounds__(256, 2) k_conv9_mixed(const _Float16* src, float* out, long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    fill_lds(lds, src);
    // (byte patterns of small positive numbers in every format: no NaN / infinity encodings among the 8- and 6-bit operands)
    for (int i = threadIdx.x; i < LDS_HALVES / 2; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] &= 0x37373737u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[4][9];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 9; ++n) acc[m][n] = (f32x4){0, 0, 0, 0};
    constexpr int NR = FMT == 0 ? 2 : 2;            // 16-byte reads per low-precision operand (e2m3: 24 of the 32 bytes are used)
    const Stamp s0 = stamp();
    for (int it = 0; it < iters; ++it) {
        // binary16 part: four k-steps of ONE product each (hi planes only: 13 fragment reads per 36 MFMAs)
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const _Float16* base = lds + ((it * 13 + k4 * 5 + wave * 8) & 31) * 512 + lane * 8;
            h8 ah[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) ah[m] = *reinterpret_cast<const h8*>(base + m * 512);
#pragma unroll
            for (int n = 0; n < 9; ++n) {
                __builtin_amdgcn_iglp_opt(0);
                const h8 bh = *reinterpret_cast<const h8*>(base + (4 + n) * 512);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah[m], acc[m][n], 0, 0, 0);
            }
        }
        // low-precision part: two scaled MFMAs per pair (each: 64 K of both cross terms)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const _Float16* base = lds + ((it * 7 + half * 11 + wave * 8) & 31) * 512 + lane * 8;
            i32x8 xb[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint4 p0 = *reinterpret_cast<const uint4*>(base + (2 * m) * 512), p1 = *reinterpret_cast<const uint4*>(base + (2 * m + 1) * 512);
                xb[m] = (i32x8){(int)p0.x, (int)p0.y, (int)p0.z, (int)p0.w, (int)p1.x, (int)p1.y, FMT == 0 ? (int)p1.z : 0, FMT == 0 ? (int)p1.w : 0};
            }
#pragma unroll
            for (int n = 0; n < 9; ++n) {
                __builtin_amdgcn_iglp_opt(0);
                const uint4 p0 = *reinterpret_cast<const uint4*>(base + (8 + 2 * n) * 512), p1 = *reinterpret_cast<const uint4*>(base + (9 + 2 * n) * 512);
                const i32x8 wa = (i32x8){(int)p0.x, (int)p0.y, (int)p0.z, (int)p0.w, (int)p1.x, (int)p1.y, FMT == 0 ? (int)p1.z : 0, FMT == 0 ? (int)p1.w : 0};
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xb[m], acc[m][n], FMT, FMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
        }
    }
    const Stamp s1 = stamp();
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 9; ++n) s += acc[m][n][0] + acc[m][n][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = s1.c0 - s0.c0; clk[1] = s1.r0 - s0.r0; }
}

// ---- the same trip inside conv_f16x3's STAGE structure, nothing else of the kernel (no prologue index arithmetic, no epilogue,
// no tile loop): per k-step one s_waitcnt vmcnt(0) + barrier, the next k-step's weight block (9 N-tiles x (hi, lo) x 1 KiB = 18 KiB per
// workgroup) streamed from a global slab into the other of two LDS weight buffers by LDS-DMA, one 1-KiB piece per N-tile iteration and
// wave like wq_one(), and every 3rd k-step a halo chunk (13 pieces of 1 KiB: the bytes per k-step of a 3-octet chunk of 324 pixels x
// (hi, lo) every 7 k-steps) into one of two halo slots -- 71 KiB of LDS, two workgroups per CU like the 80 000-byte production plans.  Workgroups that share an N-block walk the same slab (L2 hits, as in a layer); halo chunks are private (HBM / Infinity Cache).
// The weight fragments the MFMAs consume are read from the buffer the previous k-step filled; the pixel fragments from a static
// region.  What it prices: the steady state of the production pipeline against the bare trip above.
__global__ void __launch_bounds__(256, 2) k_conv9_staged(const _Float16* src, const uint4* wslab, const uint4* act, size_t act_units,
                                                         float* out, long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    constexpr int kStatic = 8 * 1024, kWbuf = 64 + 9 * 2048, kHalo = 13 * 1024;   // bytes
    unsigned char* const base8 = reinterpret_cast<unsigned char*>(lds);
    for (int i = threadIdx.x; i < kStatic / 16; i += blockDim.x) reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(src)[i + blockIdx.x % 7 * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* const wb = base8 + kStatic;
    unsigned char* const hb = wb + 2 * kWbuf;
    constexpr int kSteps = 81;                                   // k-steps of one pass over the slab (ld4.conv / lu3.conv have 81)
    const uint4* const wsrc = wslab + (size_t)(blockIdx.x & 3) * kSteps * (9 * 2048 / 16);   // this workgroup's N-block
    const uint4* const asrc = act + ((size_t)blockIdx.x * 97 * 31 * 64) % (act_units - (size_t)31 * 64 * 16);
    f32x4 acc[4][9];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 9; ++n) acc[m][n] = (f32x4){0, 0, 0, 0};
#define GLDS16(g, l) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g), (__attribute__((address_space(3))) void*)(l), 16, 0, 0)
    auto wpiece = [&](int step, int buf, int c) {                // the wave's c-th 1-KiB piece of k-step `step`'s block
        const int pc = wave + 4 * c;
        if (pc < 18) GLDS16(wsrc + ((size_t)(step % kSteps) * 18 + pc) * 64 + lane, wb + buf * kWbuf + 64 + pc * 1024);
    };
    for (int c = 0; c < 5; ++c) wpiece(0, 0, c);
    __syncthreads();
    const Stamp s0 = stamp();
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (it % 3 == 0)                                         // a halo chunk into the slot the previous chunk's k-steps have left
            for (int j = wave; j < 13; j += 4) GLDS16(asrc + ((size_t)(it / 3 % 32) * 13 + j) * 64 + lane, hb + ((it / 3) & 1) * kHalo + j * 1024);
        const unsigned char* const wl = wb + (it & 1) * kWbuf + 64 + lane * 16;
        const _Float16* abase = lds + ((it * 3 + wave) & 3) * 512 + lane * 8;
        h8 ah[4], al[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            ah[m] = *reinterpret_cast<const h8*>(abase + (2 * m % 4) * 512);
            al[m] = *reinterpret_cast<const h8*>(abase + ((2 * m + 1) % 4) * 512);
        }
#pragma unroll
        for (int n = 0; n < 9; ++n) {
            if (n < 5) wpiece(it + 1, (it + 1) & 1, n);
            __builtin_amdgcn_iglp_opt(0);
            const h8 bh = *reinterpret_cast<const h8*>(wl + n * 2048);
            const h8 bl = *reinterpret_cast<const h8*>(wl + n * 2048 + 1024);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 c = acc[m][n];
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al[m], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah[m], c, 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah[m], c, 0, 0, 0);
            }
        }
    }
    const Stamp s1 = stamp();
#undef GLDS16
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 9; ++n) s += acc[m][n][0] + acc[m][n][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = s1.c0 - s0.c0; clk[1] = s1.r0 - s0.r0; }
}

// ---- the stage structure as a RING: three weight buffers, the weight block of k-step s+2 issued under the MFMAs of
// k-step s, a COUNTED s_waitcnt vmcnt(N) in front of a raw s_barrier -- the loads of step s+1 (and a halo chunk issued after them) stay in
// flight across the barrier of step s.  vmcnt retires in issue order, so N = the wave's own loads issued after the block it needs:
// w(s+1) [+ the halo pieces issued behind w(s) or w(s+1)].  Halo: 9 pieces every 2nd k-step (the same bytes per k-step as above), issued
// in the N-tile slots 5..8 BEHIND the step's weight pieces.  LDS: 4 KiB of static pixel fragments + 3 x 18.5 KiB + 2 x 9 KiB = 77.5 KiB (two
// workgroups per CU).  RING = false: the same kernel with vmcnt(0) (what the extra buffer alone is worth).
template <bool RING>
__global__ void __launch_bounds__(256, 2) k_conv9_ring(const _Float16* src, const uint4* wslab, const uint4* act, size_t act_units,
                                                       float* out, long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    constexpr int kStatic = 4 * 1024, kWbuf = 64 + 9 * 2048, kHalo = 9 * 1024;   // bytes
    unsigned char* const base8 = reinterpret_cast<unsigned char*>(lds);
    for (int i = threadIdx.x; i < kStatic / 16; i += blockDim.x) reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(src)[i + blockIdx.x % 7 * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* const wb = base8 + kStatic;
    unsigned char* const hb = wb + 3 * kWbuf;
    constexpr int kSteps = 81;
    const uint4* const wsrc = wslab + (size_t)(blockIdx.x & 3) * kSteps * (9 * 2048 / 16);
    const uint4* const asrc = act + ((size_t)blockIdx.x * 97 * 31 * 64) % (act_units - (size_t)31 * 64 * 16);
    f32x4 acc[4][9];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 9; ++n) acc[m][n] = (f32x4){0, 0, 0, 0};
#define GLDS16(g, l) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g), (__attribute__((address_space(3))) void*)(l), 16, 0, 0)
    auto wpiece = [&](int step, int buf, int c) {
        const int pc = wave + 4 * c;
        if (pc < 18) GLDS16(wsrc + ((size_t)(step % kSteps) * 18 + pc) * 64 + lane, wb + buf * kWbuf + 64 + pc * 1024);
    };
    auto hpiece = [&](int chunk, int c) {
        const int j = wave + 4 * c;
        if (j < 9) GLDS16(asrc + ((size_t)(chunk % 32) * 9 + j) * 64 + lane, hb + (chunk & 1) * kHalo + j * 1024);
    };
    const int nw = wave < 2 ? 5 : 4, nh = wave < 1 ? 3 : 2;       // this wave's pieces per weight block / halo chunk
    auto wait_vm = [&](int n) {                                  // (wave-uniform n: one scalar branch)
        switch (n) {
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    for (int c = 0; c < 5; ++c) wpiece(0, 0, c);
    for (int c = 0; c < 5; ++c) wpiece(1, 1, c);
    __syncthreads();
    const Stamp s0 = stamp();
    // halo steps: even `it`.  At the top of step it the queue holds (oldest first) w(it), [halo(it-2) if it even], w(it+1), [halo(it-1) if it
    // odd]: every step leaves one weight block and one halo chunk in flight.
    for (int it = 0; it < iters; ++it) {
        if (RING) wait_vm(it < 2 ? 0 : nw + nh);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int bsel = it % 3;
        const unsigned char* const wl = wb + bsel * kWbuf + 64 + lane * 16;
        const _Float16* abase = lds + ((it * 3 + wave) & 1) * 512 + lane * 8;
        h8 ah[4], al[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            ah[m] = *reinterpret_cast<const h8*>(abase + (2 * m % 2) * 512);
            al[m] = *reinterpret_cast<const h8*>(abase + ((2 * m + 1) % 2) * 512 + 1024);
        }
        const int nbuf = (it + 2) % 3;
        const bool hstep = (it & 1) == 0;
#pragma unroll
        for (int n = 0; n < 9; ++n) {
            if (n < 5) wpiece(it + 2, nbuf, n);
            else if (hstep && n < 8) hpiece(it >> 1, n - 5);
            __builtin_amdgcn_iglp_opt(0);
            const h8 bh = *reinterpret_cast<const h8*>(wl + n * 2048);
            const h8 bl = *reinterpret_cast<const h8*>(wl + n * 2048 + 1024);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 c = acc[m][n];
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al[m], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah[m], c, 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah[m], c, 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const Stamp s1 = stamp();
#undef GLDS16
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 9; ++n) s += acc[m][n][0] + acc[m][n][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = s1.c0 - s0.c0; clk[1] = s1.r0 - s0.r0; }
}

template <bool RING>
static void run_ring(int wgs_per_cu, int iters, const _Float16* src, const uint4* wslab, const uint4* act, size_t act_units, float* out,
                     long long* clk) {
    const int grid = 256 * wgs_per_cu;
    const int lds = 4 * 1024 + 3 * (64 + 9 * 2048) + 2 * 9 * 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv9_ring<RING>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float ms = 0;
    hipEventRecord(e0);
    do {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_conv9_ring<RING>, dim3(grid), dim3(256), lds, 0, src, wslab, act, act_units, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    } while (ms < 2000.f);
    const int reps = 50;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_conv9_ring<RING>, dim3(grid), dim3(256), lds, 0, src, wslab, act, act_units, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    long long c[2];
    hipMemcpy(c, clk, sizeof c, hipMemcpyDeviceToHost);
    const double mfmas = (double)grid * 4 * iters * 108.0;
    const double tflops = mfmas * 2.0 * 16 * 16 * 32 * reps / (ms * 1e-3) / 1e12;
    printf("%-28s %d WG/CU  %8.3f ms/launch  %8.1f TFLOP/s issued  wave cycles per MFMA %6.2f  in-kernel clock %.2f GHz\n",
           RING ? "stage RING, vmcnt(N)" : "3 weight buffers, vmcnt(0)", wgs_per_cu, ms / reps, tflops, (double)c[0] / ((double)iters * 108.0),
           (double)c[0] / (double)c[1] * 0.1);
}

static void run_staged(int wgs_per_cu, int iters, const _Float16* src, const uint4* wslab, const uint4* act, size_t act_units, float* out,
                       long long* clk) {
    const int grid = 256 * wgs_per_cu;
    const int lds = 8 * 1024 + 2 * (64 + 9 * 2048) + 2 * 13 * 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv9_staged), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float ms = 0;
    hipEventRecord(e0);
    do {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_conv9_staged, dim3(grid), dim3(256), lds, 0, src, wslab, act, act_units, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    } while (ms < 2000.f);
    const int reps = 50;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_conv9_staged, dim3(grid), dim3(256), lds, 0, src, wslab, act, act_units, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    long long c[2];
    hipMemcpy(c, clk, sizeof c, hipMemcpyDeviceToHost);
    const double mfmas = (double)grid * 4 * iters * 108.0;
    const double tflops = mfmas * 2.0 * 16 * 16 * 32 * reps / (ms * 1e-3) / 1e12;
    printf("%-28s %d WG/CU  %8.3f ms/launch  %8.1f TFLOP/s issued  wave cycles per MFMA %6.2f  in-kernel clock %.2f GHz\n",
           "conv_f16x3<9,4,1> stage loop", wgs_per_cu, ms / reps, tflops, (double)c[0] / ((double)iters * 108.0), (double)c[0] / (double)c[1] * 0.1);
}

template <typename K>
static void run(const char* name, K kern, int wgs_per_cu, int iters, double mfma_per_trip_per_wave, double flop_per_mfma,
                const _Float16* src, float* out, long long* clk) {
    const int grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_HALVES * 2);
    // warm-up ~2 s so that the clock the chip holds under this load has settled
    hipEventRecord(e0);
    float ms = 0;
    int warm = 0;
    do {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_HALVES * 2, 0, src, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        warm += 20;
    } while (ms < 2000.f);
    const int reps = 50;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_HALVES * 2, 0, src, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    long long c[2];
    hipMemcpy(c, clk, sizeof c, hipMemcpyDeviceToHost);
    const double mfmas = (double)grid * 4 * iters * mfma_per_trip_per_wave;
    const double tflops = mfmas * flop_per_mfma * reps / (ms * 1e-3) / 1e12;
    const double cyc_per_mfma = (double)c[0] / ((double)iters * mfma_per_trip_per_wave);
    printf("%-28s %d WG/CU  %8.3f ms/launch  %8.1f TFLOP/s issued  wave cycles per MFMA %6.2f  in-kernel clock %.2f GHz\n", name,
           wgs_per_cu, ms / reps, tflops, cyc_per_mfma, (double)c[0] / (double)c[1] * 0.1);
}

int main() {
    std::vector<_Float16> h(LDS_HALVES + 8 * 512);
    srand(1234);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    _Float16* src; float* out; long long* clk;
    hipMalloc(&src, h.size() * 2); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&clk, 64);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 4000;
    for (int wgs = 1; wgs <= 2; ++wgs) {
        run("mfma 16x16x32, 64x64 tile", k_16, wgs, iters, 16, 2.0 * 16 * 16 * 32, src, out, clk);
        run("mfma 32x32x16, 64x64 tile", k_32, wgs, iters, 8, 2.0 * 32 * 32 * 16, src, out, clk);
    }
    for (int wgs = 1; wgs <= 2; ++wgs) {
        run("conv_f16x3<9,4,1>-shaped", k_conv9, wgs, iters / 4, 108, 2.0 * 16 * 16 * 32, src, out, clk);
        run("winograd F(2x2,3x3)-shaped", k_wino, wgs, iters / 4, 96, 2.0 * 16 * 16 * 32, src, out, clk);
        run("winograd, positions per wave", k_wino_split, wgs, iters / 4, 96, 2.0 * 16 * 16 * 32, src, out, clk);
    }
    for (int wgs = 1; wgs <= 2; ++wgs) {
        // per trip K = 128: the 3-product loop would issue 432 binary16 MFMAs per wave; "TFLOP/s issued" below is quoted on THAT count,
        // i.e. it is the 3-product-equivalent rate the plan's arithmetic runs at (ceiling 1.5 x / 2 x the bare loop's)
        run("mixed f16 + MX e4m3 cross", k_conv9_mixed<0>, wgs, iters / 16, 432, 2.0 * 16 * 16 * 32, src, out, clk);
        run("mixed f16 + MX e2m3 cross", k_conv9_mixed<2>, wgs, iters / 16, 432, 2.0 * 16 * 16 * 32, src, out, clk);
    }
    {   // the stage-loop skeleton: weight slab of 4 N-blocks x 81 k-steps x 18 KiB (5.8 MB, random binary16), 256 MB of halo source
        const size_t wunits = (size_t)4 * 81 * 18 * 64, aunits = (size_t)256 * 1024 * 1024 / 16;
        std::vector<_Float16> hw(wunits * 8);
        for (auto& v : hw) v = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
        uint4 *wslab, *act;
        hipMalloc(&wslab, wunits * 16); hipMalloc(&act, aunits * 16);
        hipMemcpy(wslab, hw.data(), wunits * 16, hipMemcpyHostToDevice);
        for (size_t o = 0; o < aunits * 16; o += wunits * 16) hipMemcpy((char*)act + o, hw.data(), std::min(wunits * 16, aunits * 16 - o), hipMemcpyHostToDevice);
        for (int wgs = 1; wgs <= 2; ++wgs) run_staged(wgs, 81 * 12, src, wslab, act, aunits, out, clk);
        for (int wgs = 1; wgs <= 2; ++wgs) run_ring<false>(wgs, 81 * 12, src, wslab, act, aunits, out, clk);
        for (int wgs = 1; wgs <= 2; ++wgs) run_ring<true>(wgs, 81 * 12, src, wslab, act, aunits, out, clk);
    }
    return 0;
}
