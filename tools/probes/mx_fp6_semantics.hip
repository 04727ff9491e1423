// What the hardware does, checked against a host model with exact data (gfx950, MI355X):
//   T1  v_cvt_scalef32_pk32_fp6_f16: 32 binary16 values of a lane -> 32 OCP fp6 e2m3 values in 6 dwords.  Direction of the scale
//       (divide or multiply), rounding, saturation and the bit order of the 6-bit fields.
//   T2  v_mfma_scale_f32_16x16x128_f8f6f4 with e2m3 operands (cbsz = blgp = 2): lane -> (row / column, K block) map of A and B, the
//       order of the 32 K elements inside a lane's 192 bits, the e8m0 scale bytes (op_sel 0 = byte 0 of the scale registers), C/D map.
// Everything the split-precision cross-term plan (umx_conv_f16.hip, F6 form) and the planner's host-side weight packing assume.
// build: hipcc -O3 --offload-arch=gfx950 -w tools/probes/mx_fp6_semantics.hip -o /tmp/mx_fp6_semantics
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_cvt(const _Float16* in, const float* scale, unsigned* out) {
    const int lane = threadIdx.x;
    h32 v;
    for (int i = 0; i < 32; ++i) v[i] = in[lane * 32 + i];
    const i32x6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, scale[lane]);
    for (int i = 0; i < 6; ++i) out[lane * 6 + i] = (unsigned)r[i];
}
__global__ void k_mfma(const unsigned* a6, const unsigned* b6, const unsigned* sa, const unsigned* sb, float* d) {
    const int lane = threadIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 6; ++i) { a[i] = (int)a6[lane * 6 + i]; b[i] = (int)b6[lane * 6 + i]; }
    a[6] = a[7] = b[6] = b[7] = 0;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, (int)sa[lane], 0, (int)sb[lane]);
    for (int i = 0; i < 4; ++i) d[lane * 4 + i] = c[i];
}

// host model of e2m3: value of a 6-bit code, and round-to-nearest-even quantisation with saturation
static double e2m3_value(unsigned code) {
    const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
    const double v = e == 0 ? m / 8.0 : std::ldexp(1.0 + m / 8.0, e - 1);
    return s ? -v : v;
}
static unsigned e2m3_quant(double x) {
    const unsigned s = std::signbit(x) ? 32u : 0u;
    double a = std::fabs(x);
    if (a >= 7.5) return s | 31u;
    unsigned best = 0;
    double bd = 1e9;
    for (unsigned c = 0; c < 32; ++c) {
        const double d = std::fabs(e2m3_value(c) - a);
        if (d < bd || (d == bd && (c & 1) == 0)) { bd = d; best = c; }
    }
    return s | best;
}
static void pack6(const unsigned* codes, unsigned* dw) {   // element i in bits [6 i, 6 i + 6) of the 192-bit string, little endian
    memset(dw, 0, 24);
    for (int i = 0; i < 32; ++i)
        for (int b = 0; b < 6; ++b)
            if ((codes[i] >> b) & 1) dw[(6 * i + b) / 32] |= 1u << ((6 * i + b) % 32);
}

int main() {
    srand(7);
    // ---- T1
    std::vector<_Float16> in(64 * 32);
    std::vector<float> sc(64);
    for (int l = 0; l < 64; ++l) {
        sc[l] = std::ldexp(1.f, (l % 9) - 4);
        for (int i = 0; i < 32; ++i) in[l * 32 + i] = (_Float16)(((rand() % 2001) - 1000) / 1000.f * 9.f * sc[l]);   // (some beyond 7.5 * scale)
    }
    _Float16* din; float* dsc; unsigned* dout;
    hipMalloc(&din, in.size() * 2); hipMalloc(&dsc, 64 * 4); hipMalloc(&dout, 64 * 6 * 4);
    hipMemcpy(din, in.data(), in.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dsc, sc.data(), 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, din, dsc, dout);
    std::vector<unsigned> got(64 * 6);
    hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost);
    for (int hyp = 0; hyp < 2; ++hyp) {
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            unsigned codes[32], dw[6];
            for (int i = 0; i < 32; ++i) {
                const double x = (double)(float)in[l * 32 + i];
                codes[i] = e2m3_quant(hyp == 0 ? x / sc[l] : x * sc[l]);
            }
            pack6(codes, dw);
            for (int i = 0; i < 6; ++i) bad += dw[i] != got[l * 6 + i];
        }
        printf("T1 cvt_scalef32_pk32_fp6_f16: hypothesis %s, RNE, saturating, element i at bits [6i, 6i+6): %d of 384 dwords differ\n",
               hyp == 0 ? "x / scale" : "x * scale", bad);
    }
    {   // element-level report under the divide hypothesis (what differs, if anything)
        int shown = 0;
        for (int l = 0; l < 64 && shown < 6; ++l)
            for (int i = 0; i < 32 && shown < 6; ++i) {
                const double x = (double)(float)in[l * 32 + i] / sc[l];
                const unsigned want = e2m3_quant(x);
                unsigned g = 0;
                for (int b = 0; b < 6; ++b) g |= ((got[l * 6 + (6 * i + b) / 32] >> ((6 * i + b) % 32)) & 1u) << b;
                if (g != want) { printf("   lane %d elem %d: x/scale = %g want code %u (%g) got %u (%g)\n", l, i, x, want, e2m3_value(want), g, e2m3_value(g)); ++shown; }
            }
    }
    {   // ---- T1b: binary16 SUBNORMAL inputs (the lo halves of small activations are): converted, or flushed to zero?
        std::vector<_Float16> in2(64 * 32);
        std::vector<float> sc2(64);
        for (int l = 0; l < 64; ++l) {
            sc2[l] = std::ldexp(1.f, -24 + (l % 4));          // scale 2^-24 .. 2^-21: a subnormal m * 2^-24 lands on m, m/2, m/4, m/8
            for (int i = 0; i < 32; ++i) {
                const unsigned short bits = (unsigned short)(((l * 32 + i) * 7) % 57 | ((i & 1) << 15));   // mantissa-only patterns: subnormals 0 .. 56 * 2^-24
                memcpy(&in2[l * 32 + i], &bits, 2);
            }
        }
        hipMemcpy(din, in2.data(), in2.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dsc, sc2.data(), 64 * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, din, dsc, dout);
        hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0, zeros = 0, nonzero_in = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 32; ++i) {
                const double x = (double)(float)in2[l * 32 + i] / sc2[l];
                const unsigned want = e2m3_quant(x);
                unsigned g = 0;
                for (int b = 0; b < 6; ++b) g |= ((got[l * 6 + (6 * i + b) / 32] >> ((6 * i + b) % 32)) & 1u) << b;
                bad += (g & 31u) != (want & 31u);
                if ((want & 31u) != 0) { ++nonzero_in; zeros += (g & 31u) == 0; }
            }
        printf("T1b binary16 subnormal inputs: %d of 2048 magnitudes differ from the exact model; of %d inputs that should convert to a non-zero "
               "code %d came out zero (flushed)\n", bad, nonzero_in, zeros);
    }
    // ---- T2: exact data (every product and sum representable): A[row][k], B[k][col], k = 0..127; lane l holds row/col l & 15, K block l >> 4
    std::vector<unsigned> a6(64 * 6), b6(64 * 6), sa(64), sb(64);
    std::vector<double> A(16 * 128), Bm(128 * 16);
    for (int l = 0; l < 64; ++l) {
        unsigned ca[32], cb[32];
        const int ea = 120 + rand() % 12, eb = 122 + rand() % 10;   // e8m0 scale bytes (2^(e - 127)); junk in the other bytes
        sa[l] = (unsigned)ea | 0x55aa3300u;
        sb[l] = (unsigned)eb | 0x11227700u;
        for (int i = 0; i < 32; ++i) {
            ca[i] = rand() & 63; cb[i] = rand() & 63;
            A[(l & 15) * 128 + 32 * (l >> 4) + i] = e2m3_value(ca[i]) * std::ldexp(1.0, ea - 127);
            Bm[(32 * (l >> 4) + i) * 16 + (l & 15)] = e2m3_value(cb[i]) * std::ldexp(1.0, eb - 127);
        }
        pack6(ca, &a6[l * 6]); pack6(cb, &b6[l * 6]);
    }
    unsigned *da, *db, *dsa, *dsb; float* dd;
    hipMalloc(&da, 64 * 24); hipMalloc(&db, 64 * 24); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 64 * 16);
    hipMemcpy(da, a6.data(), 64 * 24, hipMemcpyHostToDevice); hipMemcpy(db, b6.data(), 64 * 24, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
    std::vector<float> D(64 * 4);
    hipMemcpy(D.data(), dd, 64 * 16, hipMemcpyDeviceToHost);
    double worst = 0, scale_of = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (l >> 4) + r, col = l & 15;   // C/D: row = 4 * (lane >> 4) + reg, column = lane & 15
            double want = 0;
            for (int k = 0; k < 128; ++k) want += A[row * 128 + k] * Bm[k * 16 + col];
            worst = std::max(worst, std::fabs(want - (double)D[l * 4 + r]));
            scale_of = std::max(scale_of, std::fabs(want));
        }
    printf("T2 mfma_scale 16x16x128 e2m3 x e2m3: A lane = (row l&15, K block l>>4), 32 K elements in bit order, scale byte 0 = e8m0, "
           "D row = 4*(l>>4)+reg, col = l&15: max |D - host| = %.3g (|D| up to %.3g)\n", worst, scale_of);
    return 0;
}
