// First down-sampling layer of the split-precision plan for gfx950 (MI355X, CDNA4): conv ks x ks (main + shortcut filter
// summed) -> BN affine -> (Leaky)ReLU [-> BN affine, legacy graph] -> 2 x 2 max-pool (reference UnMicst1-5.py:83-118,
// UnMicst.py:80-104).
//
// Why a kernel of its own.  conv_f16x3 spends one (tap, octet) pair = 8 K-slots per tap on an input that has 1-2 real
// channels: the duo models' first layer runs 3 k-steps of 32 for K = 18 real values, streams its weights through LDS per
// 256-pixel workgroup and pays a pipeline prologue, a barrier per stage and an LDS-transpose epilogue for 0.1 GFLOP per tile.
// Here
//   * K = taps x channel slots is packed densely (k = tap * CW + c, CW = 1, 2 or 4 channel slots): ONE k-step for the solo /
//     duo / legacy first layers, so a 16-pixel M-tile costs 9 MFMAs (3 N-tiles x 3 split-precision products);
//   * the whole weight set is 2 * NT * NKS A-fragments = 24 registers: loaded once per wave, no weight traffic at all;
//   * a workgroup owns a 16 x 64-pixel region of one tile (4 waves x 8 row pairs), stages its halo once as
//     [pixel]{hi[CW] | lo[CW]} words so that a lane's B-fragment is 8/CW LDS reads and no packing;
//   * the epilogue pools in registers (row pair in the wave, column pair by DPP) and BOTH lanes of a column pair store:
//     the even lane channels 0-1, the odd lane channels 2-3 of the group's four -- every lane writes 4 bytes, a wave
//     instruction covers two whole 128-byte lines of the octet-planar output per plane, no LDS transpose.
// Arithmetic is conv_f16x3's: x*w = x_hi*w_hi + x_hi*w_lo + x_lo*w_hi on v_mfma_f32_16x16x32_f16, fp32 accumulate.
#include "umx_kernels.h"

namespace umx {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT, int CW, int NKS>
__global__ void __launch_bounds__(256, 2) conv_first(const FirstParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TPL = 8 / CW;   // taps per lane and k-step
    constexpr int PXB = 4 * CW;   // LDS bytes per halo pixel: hi[CW] | lo[CW]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4;    // which 8-wide k group of the 16x16x32 MFMA this lane feeds
    const int li = lane & 15;   // pixel (B operand) / output channel (A operand, C/D) inside the tile

    // ---- workgroup -> (tile, region)
    const int P = p.P, pad = (p.ks - 1) >> 1;
    const int nrx = P >> p.rw_log2, nry = P >> 4;
    int bid = blockIdx.x;
    const int rxi = bid % nrx; bid /= nrx;
    const int ryi = bid % nry;
    const int img = bid / nry;
    const int y0 = ryi * 16, x0 = rxi << p.rw_log2;

    // ---- weights: MFMA A-fragments straight into registers
    h8 Wh[NKS][NT], Wl[NKS][NT];
#pragma unroll
    for (int s = 0; s < NKS; ++s)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const uint4* const w = p.w + ((size_t)(s * NT + n) * 2) * 64 + lane;
            Wh[s][n] = *reinterpret_cast<const h8*>(w);
            Wl[s][n] = *reinterpret_cast<const h8*>(w + 64);
        }
    // ---- LDS byte offsets of the taps this lane's K-slots read (slots past the last tap carry zero weights: they re-read
    // tap 0, finite data)
    int toff[NKS][TPL];
#pragma unroll
    for (int s = 0; s < NKS; ++s)
#pragma unroll
        for (int j = 0; j < TPL; ++j) {
            int t = (32 * s + 8 * q) / CW + j;
            if (t >= p.ntaps) t = 0;
            const int dy = t / p.ks;
            toff[s][j] = (dy * p.hw + (t - dy * p.ks)) * PXB;
        }

    // ---- epilogue constants -> LDS (behind the halo image)
    const int halo_bytes = (p.hh * p.hw * PXB + 15) & ~15;
    float* const ecl = reinterpret_cast<float*>(smem + halo_bytes);
    for (int i = tid; i < 4 * NT * 16; i += 256) ecl[i] = p.econst[i];

    // ---- halo of the region: [hy][hx]{hi[CW] | lo[CW]}; outside the TILE zeros (SAME padding of the tile convolution)
    for (int hp = tid; hp < p.hh * p.hw; hp += 256) {
        const int hy = (int)(((float)hp + 0.5f) * p.inv_hw);   // hp < 2^16: exact after truncation
        const int hx = hp - hy * p.hw;
        const int ty = y0 + hy - pad, tx = x0 + hx - pad;
        const bool in = (unsigned)ty < (unsigned)P && (unsigned)tx < (unsigned)P;
        typedef unsigned uvec __attribute__((ext_vector_type(CW)));   // 4 * CW bytes: hi[CW] | lo[CW]
        uvec w;
#pragma unroll
        for (int c = 0; c < CW; ++c) w[c] = 0u;
        if (in) {
            const size_t px = ((size_t)img * P + ty) * P + tx;
            if (p.src_c) {   // compact source: already this kernel's pixel format
                w = *reinterpret_cast<const uvec*>(p.src_c + px * PXB);
            } else {
                _Float16 v[2 * CW];
#pragma unroll
                for (int c = 0; c < CW; ++c) { v[c] = p.src_hi[px * 8 + c]; v[CW + c] = p.src_lo[px * 8 + c]; }
                __builtin_memcpy(&w, v, sizeof w);
            }
        }
        *reinterpret_cast<uvec*>(smem + hp * PXB) = w;
    }
    __syncthreads();

    // ---- output addressing, in binary16 elements: image * dImg + pixel * dPix + octet * dOct -- NHWC: (Cds, 8); octet-planar
    // (per image [octet][pixel][8]): (8, outS * outS * 8)
    const int outS = p.outS;
    const int dPix = p.dst_planar ? 8 : p.Cds;
    const int dOct = p.dst_planar ? outS * outS * 8 : 8;
    _Float16* const ohi = p.dst_hi + (size_t)img * outS * outS * p.Cds;
    _Float16* const olo = p.dst_lo + (size_t)img * outS * outS * p.Cds;
    const float slope = p.act == ACT_RELU ? 0.f : p.act == ACT_LEAKY ? kLeakySlope : 1.f;
    const float4* const ec4 = reinterpret_cast<const float4*>(ecl);
    const bool odd = (li & 1) != 0;
    unsigned vmax = 0u;   // max |v| of this lane as a bit pattern (orders NaN and infinity above every finite value)

    const int ncb_log2 = p.rw_log2 - 4;          // 16-pixel column blocks of the region
    const int npairs = 8 << ncb_log2;            // row pairs x column blocks
    for (int pi = wave; pi < npairs; pi += kWaves) {
        const int cb = pi & ((1 << ncb_log2) - 1), rp = pi >> ncb_log2;
        f32x4 acc[2][NT];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const unsigned char* const base = smem + ((2 * rp + r) * p.hw + cb * 16 + li) * PXB;
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                u32x4 uh, ul;
                if constexpr (CW == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint2 v = *reinterpret_cast<const uint2*>(base + toff[s][j]);
                        uh[j] = v.x;
                        ul[j] = v.y;
                    }
                } else if constexpr (CW == 4) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const uint4 v = *reinterpret_cast<const uint4*>(base + toff[s][j]);
                        uh[2 * j] = v.x; uh[2 * j + 1] = v.y;
                        ul[2 * j] = v.z; ul[2 * j + 1] = v.w;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned v0 = *reinterpret_cast<const unsigned*>(base + toff[s][2 * j]);
                        const unsigned v1 = *reinterpret_cast<const unsigned*>(base + toff[s][2 * j + 1]);
                        uh[j] = __builtin_amdgcn_perm(v1, v0, 0x05040100u);   // {v0.lo16, v1.lo16}
                        ul[j] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);   // {v0.hi16, v1.hi16}
                    }
                }
                const h8 ah = __builtin_bit_cast(h8, uh), al = __builtin_bit_cast(h8, ul);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    // weights are the A operand (rows = output channels), pixels the B operand: D[channel][pixel]
                    f32x4 c = acc[r][n];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wh[s][n], al, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wl[s][n], ah, c, 0, 0, 0);
                    acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wh[s][n], ah, c, 0, 0, 0);
                }
            }
        }
        // ---- epilogue of the row pair: affine -> activation [-> affine] on both rows, max over the 2 x 2 window, (hi, lo)
        // split of this lane's two channels, one 4-byte store per plane and N-tile
        const int py = (y0 >> 1) + rp, px = ((x0 + cb * 16) >> 1) + (li >> 1);
        const int pixoff = (py * outS + px) * dPix;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const float4 ps = ec4[0 * NT * 4 + n * 4 + q], pb = ec4[1 * NT * 4 + n * 4 + q];
            const float psa[4] = {ps.x, ps.y, ps.z, ps.w}, pba[4] = {pb.x, pb.y, pb.z, pb.w};
            float qsa[4] = {1.f, 1.f, 1.f, 1.f}, qba[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.post_affine) {   // wave-uniform
                const float4 qs = ec4[2 * NT * 4 + n * 4 + q], qb = ec4[3 * NT * 4 + n * 4 + q];
                qsa[0] = qs.x; qsa[1] = qs.y; qsa[2] = qs.z; qsa[3] = qs.w;
                qba[0] = qb.x; qba[1] = qb.y; qba[2] = qb.z; qba[3] = qb.w;
            }
            float m[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v0 = acc[0][n][r] * psa[r] + pba[r];
                float v1 = acc[1][n][r] * psa[r] + pba[r];
                v0 = __builtin_amdgcn_fmed3f(v0, v0 * slope, INFINITY);   // max(v, slope * v): ReLU / LeakyReLU / none
                v1 = __builtin_amdgcn_fmed3f(v1, v1 * slope, INFINITY);
                v0 = v0 * qsa[r] + qba[r];
                v1 = v1 * qsa[r] + qba[r];
                const float a = fmaxf(v0, v1);   // rows 2rp, 2rp+1
                const float b = __builtin_bit_cast(
                    float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false));
                m[r] = fmaxf(a, b);              // columns li, li ^ 1: both lanes of the pair hold the pooled value
            }
            const float x0v = odd ? m[2] : m[0], x1v = odd ? m[3] : m[1];
            vmax = max(vmax, max(__float_as_uint(x0v) & 0x7fffffffu, __float_as_uint(x1v) & 0x7fffffffu));
            h2 hi, lo;
            hi[0] = (_Float16)x0v; hi[1] = (_Float16)x1v;
            lo[0] = (_Float16)(x0v - (float)hi[0]); lo[1] = (_Float16)(x1v - (float)hi[1]);
            const int c = n * 16 + 4 * q + (odd ? 2 : 0);
            if (c < p.Cds) {
                const int e = pixoff + (c >> 3) * dOct + (c & 7);
                *reinterpret_cast<h2*>(ohi + e) = hi;
                *reinterpret_cast<h2*>(olo + e) = lo;
            }
        }
    }
    if (vmax >= 0x476a6000u) atomicOr(p.overflow_flag, 1);   // |v| >= 60000, infinity or NaN: binary16 range exceeded
}

template <int NT, int CW, int NKS>
static hipError_t launch_first_k(const FirstParams& p, hipStream_t stream) {
    const dim3 grid((unsigned)(p.B * (p.P >> 4) * (p.P >> p.rw_log2)));
    hipLaunchKernelGGL((conv_first<NT, CW, NKS>), grid, dim3(256), (size_t)p.lds_bytes, stream, p);
    return hipGetLastError();
}

template <int CW, int NKS>
static hipError_t launch_first_nt(const FirstParams& p, hipStream_t stream) {
    switch (p.NT) {
        case 1: return launch_first_k<1, CW, NKS>(p, stream);
        case 2: return launch_first_k<2, CW, NKS>(p, stream);
        case 3: return launch_first_k<3, CW, NKS>(p, stream);
        case 4: return launch_first_k<4, CW, NKS>(p, stream);
        case 5: return launch_first_k<5, CW, NKS>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

// (channel slots per tap, k-steps) the kernel is built for: 3 x 3 taps with 1, 2 or 3-4 input channels, 5 x 5 with 1 or 2
bool conv_first_supported(int NT, int CW, int NKS) {
    return NT >= 1 && NT <= 5 && ((CW == 1 && NKS == 1) || (CW == 2 && NKS == 1) || (CW == 2 && NKS == 2) || (CW == 4 && NKS == 2));
}

hipError_t launch_conv_first(const FirstParams& p, hipStream_t stream) {
    if (p.B <= 0) return hipSuccess;
    if (p.lds_bytes > 48 * 1024) return hipErrorInvalidValue;
    if (p.CW == 1 && p.NKS == 1) return launch_first_nt<1, 1>(p, stream);
    if (p.CW == 2 && p.NKS == 1) return launch_first_nt<2, 1>(p, stream);
    if (p.CW == 2 && p.NKS == 2) return launch_first_nt<2, 2>(p, stream);
    if (p.CW == 4 && p.NKS == 2) return launch_first_nt<4, 2>(p, stream);
    return hipErrorInvalidValue;
}

}  // namespace umx
