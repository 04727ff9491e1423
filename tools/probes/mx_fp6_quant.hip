// The two halves of the F6 form's arithmetic against a host model (gfx950): T3 the kernel's on-the-fly block quantisation
// (quant_block_e2m3 of umx_conv_f16.hip, copied verbatim below) against the MX rule "scale = 2^(floor(log2 max) - 2), elements RNE";
// T4 planner-packed weights (mx_pack_e2m3 of umx_plan.hip, copied verbatim) x kernel-quantised pixels through the scaled MFMA against
// the float64 product of the de-quantised operands (must agree to fp32 rounding) and of the unquantised ones (the plan's error).
// build: hipcc -O3 --offload-arch=gfx950 -w tools/probes/mx_fp6_quant.hip -o /tmp/mx_fp6_quant
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
using std::max;
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
#define UMX_MAX3(dst, a, b, c) asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(dst) : "v"(a), "v"(b), "v"(c))
#define UMX_MAX3N(dst, a, b, c) asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3 neg_lo:[1,1,1] neg_hi:[1,1,1]" : "=v"(dst) : "v"(a), "v"(b), "v"(c))
#define UMX_MAX3N2(dst, a, b, c) asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(dst) : "v"(a), "v"(b), "v"(c))
// NB blocks at once (1 or 2: two independent dependency chains interleaved instruction by instruction -- the reduction is a tree of
// depth 4, one block's chain alone leaves the vector unit waiting for its own results).  volatile: the sequence stays together and in
// this order -- left to the scheduler, the conversions sink behind every block's reduction and all fragment registers stay live.
template <int NB>
__device__ __forceinline__ void quant_blocks_e2m3(const h8 (&v)[NB][4], i32x6 (&out)[NB], int (&scale_e8m0)[NB]) {
    h32 x[NB];
    u32x16 d[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[b][8 * p + j] = v[b][p][j];
        d[b] = __builtin_bit_cast(u32x16, x[b]);
    }
    // per half-word position: the largest value (tree P) and the largest negated value (tree N) of the 16 words, three at a time
    unsigned P[NB][5], N[NB][5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            UMX_MAX3(P[b][t], d[b][3 * t], d[b][3 * t + 1], d[b][3 * t + 2]);
            UMX_MAX3N(N[b][t], d[b][3 * t], d[b][3 * t + 1], d[b][3 * t + 2]);
        }
    unsigned P2[NB][2], N2[NB][2], m2[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        UMX_MAX3(P2[b][0], P[b][0], P[b][1], P[b][2]);
        UMX_MAX3(N2[b][0], N[b][0], N[b][1], N[b][2]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        UMX_MAX3(P2[b][1], P[b][3], P[b][4], d[b][15]);
        UMX_MAX3N2(N2[b][1], N[b][3], N[b][4], d[b][15]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) UMX_MAX3(m2[b], P2[b][0], P2[b][1], N2[b][0]);
#pragma unroll
    for (int b = 0; b < NB; ++b) UMX_MAX3(m2[b], m2[b], N2[b][1], N2[b][1]);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const unsigned m = m2[b] & 0x7fff7fffu;
        const unsigned mm = max(m & 0xffffu, m >> 16);   // bit pattern of the block's largest magnitude (binary16 orders like its bits)
        // its exponent through binary32 -- the lo halves of activations below 0.25 are binary16 SUBNORMALS (taking the exponent field of
        // the binary16 pattern pins their blocks' scale at 2^-17 and leaves them one or two significant bits: the x_lo * w_hi term was
        // as good as dropped, 1.5e-5 .. 3.6e-5 instead of 3e-6 on the random graphs).  2^(exponent - 2): the largest value lands in
        // [4, 8) (e2m3: up to 7.5, saturating); an all-zero block takes the smallest scale
        const unsigned ef = __float_as_uint((float)__builtin_bit_cast(_Float16, (unsigned short)mm)) >> 23;
        const unsigned e = max(ef, 3u) - 2u;
        scale_e8m0[b] = (int)e;
        const float sc = __uint_as_float(e << 23);
        asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(out[b]) : "v"(x[b]), "v"(sc));   // (early clobber: a multi-pass instruction, the 6 result registers must not overlap the 16 + 1 it is still reading)
    }
}
__device__ __forceinline__ void quant_block_e2m3(const h8 (&v)[4], i32x6& out, int& scale_e8m0) {
    h8 vv[1][4] = {{v[0], v[1], v[2], v[3]}};
    i32x6 o[1];
    int s[1];
    quant_blocks_e2m3<1>(vv, o, s);
    out = o[0];
    scale_e8m0 = s[0];
}

__global__ void k_quant(const _Float16* in, unsigned* out, int* sc) {
    const int lane = threadIdx.x;
    h8 v[4];
    for (int p = 0; p < 4; ++p) for (int j = 0; j < 8; ++j) v[p][j] = in[lane * 32 + p * 8 + j];
    i32x6 r; int s;
    quant_block_e2m3(v, r, s);
    for (int i = 0; i < 6; ++i) out[lane * 6 + i] = (unsigned)r[i];
    sc[lane] = s;
}
__global__ void k_dot(const _Float16* xin, const unsigned* a6, const int* sa, float* d) {
    const int lane = threadIdx.x;
    h8 v[4];
    for (int p = 0; p < 4; ++p) for (int j = 0; j < 8; ++j) v[p][j] = xin[lane * 32 + p * 8 + j];
    i32x6 xb; int sx;
    quant_block_e2m3(v, xb, sx);
    i32x6 wa;
    for (int i = 0; i < 6; ++i) wa[i] = (int)a6[lane * 6 + i];
    const int s_a = sa[lane];
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n s_nop 7\n s_nop 7"
                 : "+v"(c) : "v"(wa), "v"(xb), "v"(s_a), "v"(sx));
    for (int i = 0; i < 4; ++i) d[lane * 4 + i] = c[i];
}
static int mx_pack_e2m3(const double (&v)[32], double amax, unsigned char (&out)[24]) {
    memset(out, 0, sizeof out);
    if (!(amax > 0.0) || !std::isfinite(amax)) return 127;
    int e2;
    std::frexp(amax, &e2);                        // amax = m * 2^e2, m in [0.5, 1): floor(log2 amax) = e2 - 1
    int se = std::max(-127, std::min(127, e2 - 1 - 2));
    const double inv = std::ldexp(1.0, -se);
    for (int i = 0; i < 32; ++i) {
        const double a = std::fabs(v[i]) * inv;
        unsigned code;
        if (a >= 7.5) code = 31;
        else {
            int eb = 0;                            // binade: [0, 1) subnormal step 1/8, [1, 2) 1/8, [2, 4) 1/4, [4, 8) 1/2
            if (a >= 4.0) eb = 3; else if (a >= 2.0) eb = 2; else if (a >= 1.0) eb = 1;
            const double step = eb <= 1 ? 0.125 : eb == 2 ? 0.25 : 0.5;
            const double qv = std::nearbyint(a / step) * step;   // (default rounding mode: to nearest even)
            if (qv >= 7.5) code = 31;
            else if (qv < 1.0) code = (unsigned)std::lround(qv * 8.0);
            else {
                int ee = qv >= 4.0 ? 3 : qv >= 2.0 ? 2 : 1;
                code = (unsigned)(ee << 3) | (unsigned)std::lround((qv / std::ldexp(1.0, ee - 1) - 1.0) * 8.0);
            }
        }
        if (std::signbit(v[i]) && code) code |= 32u;
        for (int b = 0; b < 6; ++b)
            if ((code >> b) & 1u) out[(6 * i + b) / 8] |= (unsigned char)(1u << ((6 * i + b) % 8));
    }
    return se + 127;
}


static double e2m3_value(unsigned code) {
    const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
    const double v = e == 0 ? m / 8.0 : std::ldexp(1.0 + m / 8.0, e - 1);
    return s ? -v : v;
}
static unsigned field(const unsigned char* b, int i) {
    unsigned g = 0;
    for (int k = 0; k < 6; ++k) g |= ((b[(6 * i + k) / 8] >> ((6 * i + k) % 8)) & 1u) << k;
    return g;
}
int main() {
    srand(11);
    // pixel operand: lane (col l & 15, block l >> 4), 32 halves each, magnitudes spread over blocks and inside them
    std::vector<_Float16> x(64 * 32);
    for (int l = 0; l < 64; ++l) {
        const double blk = std::ldexp(1.0, (l * 7) % 23 - 12);
        for (int i = 0; i < 32; ++i) x[l * 32 + i] = (_Float16)(((rand() % 20001) - 10000) / 10000.0 * blk * std::ldexp(1.0, -(rand() % 5)));
    }
    for (int l = 40; l < 48; ++l)                                               // blocks of binary16 subnormals (what the lo halves of small activations are)
        for (int i = 0; i < 32; ++i) { const unsigned short bits = (unsigned short)((rand() % (8 << (l - 40))) | ((rand() & 1) << 15)); memcpy(&x[l * 32 + i], &bits, 2); }
    for (int i = 0; i < 32; ++i) x[5 * 32 + i] = (_Float16)0.f;               // an all-zero block
    for (int i = 0; i < 32; ++i) x[6 * 32 + i] = (_Float16)-std::fabs((float)x[6 * 32 + i]);   // an all-negative block
    _Float16* dx; unsigned* dq; int* ds;
    hipMalloc(&dx, x.size() * 2); hipMalloc(&dq, 64 * 24); hipMalloc(&ds, 256);
    hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_quant, dim3(1), dim3(64), 0, 0, dx, dq, ds);
    std::vector<unsigned> q(64 * 6); std::vector<int> sc(64);
    hipMemcpy(q.data(), dq, 64 * 24, hipMemcpyDeviceToHost); hipMemcpy(sc.data(), ds, 256, hipMemcpyDeviceToHost);
    int bad_s = 0, bad_e = 0;
    std::vector<double> xq(64 * 32);   // de-quantised pixel operand as the kernel produced it
    for (int l = 0; l < 64; ++l) {
        double v32[32], amax = 0;
        for (int i = 0; i < 32; ++i) { v32[i] = (double)(float)x[l * 32 + i]; amax = std::max(amax, std::fabs(v32[i])); }
        unsigned char want[24];
        int e8 = mx_pack_e2m3(v32, amax, want);
        if (amax == 0) e8 = sc[l];   // (any scale does for a zero block)
        bad_s += e8 != sc[l];
        for (int i = 0; i < 32; ++i) {
            const unsigned g = field(reinterpret_cast<const unsigned char*>(&q[l * 6]), i);
            if (e8 == sc[l] && g != field(want, i)) {
                if (bad_e < 12) printf("   lane %d elem %d: x %g scale 2^%d x/scale %g: host code %u (%g) device code %u (%g)\n", l, i, v32[i], sc[l] - 127,
                                       v32[i] * std::ldexp(1.0, 127 - sc[l]), field(want, i), e2m3_value(field(want, i)), g, e2m3_value(g));
                ++bad_e;
            }
            xq[l * 32 + i] = e2m3_value(g) * std::ldexp(1.0, sc[l] - 127);
        }
    }
    printf("T3 quant_block_e2m3 vs the host MX rule: %d of 64 block scales differ, %d of 2048 elements differ\n", bad_s, bad_e);
    // weights: lane (row l & 15, block l >> 4)
    std::vector<unsigned> a6(64 * 6); std::vector<int> sa(64);
    std::vector<double> W(64 * 32), Wq(64 * 32);
    for (int l = 0; l < 64; ++l) {
        double v32[32], amax = 0;
        const double blk = std::ldexp(1.0, (l * 5) % 9 - 14);
        for (int i = 0; i < 32; ++i) { v32[i] = (double)(float)(_Float16)(((rand() % 20001) - 10000) / 10000.0 * blk); amax = std::max(amax, std::fabs(v32[i])); W[l * 32 + i] = v32[i]; }
        unsigned char bytes[24];
        sa[l] = mx_pack_e2m3(v32, amax, bytes);
        memcpy(&a6[l * 6], bytes, 24);
        for (int i = 0; i < 32; ++i) Wq[l * 32 + i] = e2m3_value(field(bytes, i)) * std::ldexp(1.0, sa[l] - 127);
    }
    unsigned* da; int* dsa; float* dd;
    hipMalloc(&da, 64 * 24); hipMalloc(&dsa, 256); hipMalloc(&dd, 64 * 16);
    hipMemcpy(da, a6.data(), 64 * 24, hipMemcpyHostToDevice); hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_dot, dim3(1), dim3(64), 0, 0, dx, da, dsa, dd);
    std::vector<float> D(256);
    hipMemcpy(D.data(), dd, 1024, hipMemcpyDeviceToHost);
    double e_q = 0, e_x = 0, mag = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (l >> 4) + r, col = l & 15;
            double wq = 0, wx = 0;
            for (int b = 0; b < 4; ++b)
                for (int i = 0; i < 32; ++i) {
                    wq += Wq[(b * 16 + row) * 32 + i] * xq[(b * 16 + col) * 32 + i];
                    wx += W[(b * 16 + row) * 32 + i] * (double)(float)x[(b * 16 + col) * 32 + i];
                }
            e_q = std::max(e_q, std::fabs(wq - (double)D[l * 4 + r]));
            e_x = std::max(e_x, std::fabs(wx - (double)D[l * 4 + r]));
            mag = std::max(mag, std::fabs(wx));
        }
    printf("T4 packed weights x quantised pixels through the scaled MFMA: max |D - product of de-quantised operands| = %.3g, "
           "max |D - exact product| = %.3g, products up to %.3g\n", e_q, e_x, mag);
    return 0;
}
