// HIP kernels of libumx -- written for gfx950 (MI355X, CDNA4) only: 64-wide waves, v_mfma_f32_16x16x4_f32,
// 160 KiB LDS per CU.  No other target is supported.
#include <cmath>

#include "umx_kernels.h"

#include <hip/hip_fp16.h>

#include <algorithm>

namespace umx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS row pitch (floats) of the staged weight tile: NT*16 real columns, padded so that pitch = 16 (mod 32):
// lanes (j, kq) of a B-fragment read hit bank j + 16*kq -> conflict-free ds_read_b32.
__host__ __device__ constexpr int npl_of(int nt) { return (nt & 1) ? nt * 16 : nt * 16 + 16; }
__host__ __device__ constexpr int wreg_of(int nt) { return (kTapG * kCC * nt * 4 + 255) / 256; }

size_t conv_lds_bytes(int nt, int plane) {
    return sizeof(float) * ((size_t)kCC * plane + (size_t)kTapG * kCC * npl_of(nt));
}

// ------------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution on the fp32 matrix cores.
//   GEMM view: M = output pixels (16 per M-tile, 256 per workgroup), N = output channels (NT tiles of 16 per
//   workgroup), K = (tap, input channel).  No im2col buffer exists anywhere: the A operand is read straight out
//   of an LDS-staged input halo at a per-tap offset.
//   LDS: halo [kCC channel planes][plane] (channel-planar so that an A-fragment read -- 16 pixels x 2 channels per
//   32-lane half -- is bank-conflict free), weights [tap][channel][npl].
//   Pipeline: single LDS buffer; the next stage's global loads are issued into registers before the MFMA block
//   of the current stage and written to LDS after it (issue-early / write-late); two workgroups per CU cover each
//   other's barriers.
// ------------------------------------------------------------------------------------------------------------
template <int NT, int HPIX>
__global__ void __launch_bounds__(256, 2) conv_mfma_f32(const ConvParams p) {
    constexpr int NPL = npl_of(NT);
    constexpr int WREG = wreg_of(NT);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Hl = smem;                       // [kCC][plane]
    float* const Wl = smem + kCC * p.plane;       // [kTapG*kCC][NPL]   (plane is a multiple of 16 -> 64 B aligned)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4;     // k index inside one 16x16x4 MFMA step
    const int li = lane & 15;     // pixel (A) / output channel (B, C/D) index inside the tile

    // ---- workgroup -> (image group, spatial tile), N block, phase
    const int TWm = 1 << p.twm_log2, TH = 1 << p.th_log2;
    int bid = blockIdx.x;
    const int tx_i = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty_i = bid % p.tiles_y; bid /= p.tiles_y;
    const int img0 = bid * p.imgs;
    const int y0 = ty_i * TH, x0 = tx_i * TWm;
    const int ksn = p.ksplit > 1 ? p.ksplit : 1;
    const int nblk = blockIdx.y / ksn;
    const int ksi = blockIdx.y - nblk * ksn;   // K split: this workgroup takes channel chunks ksi, ksi + ksn, ...
    const ConvPhase& ph = p.ph[blockIdx.z];

    // ---- per-thread halo staging slots (constant for the whole kernel)
    const int npix = p.imgs * p.imgplane;
    int hpix[HPIX];   // source pixel index (img*H + y)*W + x, or -1 (zero fill)
    int hlds[HPIX];   // LDS offset inside a channel plane, or -1 (slot unused)
#pragma unroll
    for (int s = 0; s < HPIX; ++s) {
        const int e = s * 256 + tid;
        hpix[s] = -1;
        hlds[s] = -1;
        if (e < npix) {
            const int il = e / p.imgplane;
            const int r = e - il * p.imgplane;
            const int hy = r / p.hw;
            const int hx = r - hy * p.hw;
            const int gy = y0 + p.ymin + hy, gx = x0 + p.xmin + hx, img = img0 + il;
            hlds[s] = e;
            if (img < p.B && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) hpix[s] = (img * p.H + gy) * p.W + gx;
        }
    }

    // ---- per-lane fragment base addresses
    int abase[kMT];
#pragma unroll
    for (int m = 0; m < kMT; ++m) {
        const int t = wave * kMT + m;
        const int ig = t >> p.th_log2, ty = t & (TH - 1);
        abase[m] = kq * p.plane + (ig * p.nimg_m + (li >> p.twm_log2)) * p.imgplane + ty * p.hw + (li & (TWm - 1));
    }
    const int bbase = kq * NPL + li;

    f32x4 acc[kMT][NT];
#pragma unroll
    for (int m = 0; m < kMT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- stage iterator: (group, channel chunk, tap group)
    struct Stage { int g, c0, t0; };
    auto stage_valid = [&](const Stage& s) { return s.g < p.ngroups; };
    auto stage_skip = [&](Stage s) {   // move past groups without taps in this phase / without a chunk for this split
        while (s.g < p.ngroups && (ph.ntaps[s.g] == 0 || s.c0 >= p.Cp[s.g])) { s.g += 1; s.c0 = ksi * kCC; }
        return s;
    };
    auto stage_next = [&](Stage s) {
        s.t0 += kTapG;
        if (s.t0 >= ph.ntaps[s.g]) {
            s.t0 = 0;
            s.c0 += kCC * ksn;
        }
        return stage_skip(s);
    };

    float4 hreg[HPIX][2];
    float4 wreg[WREG];

    auto load_stage = [&](const Stage& s) {
        const int g = s.g;
        const int nk4 = (p.Cp[g] - s.c0) >= kCC ? 2 : 1;
        if (s.t0 == 0) {  // new channel chunk: (re)load the halo
            const float* __restrict__ src = p.src[g];
            const int C = p.C[g];
#pragma unroll
            for (int sl = 0; sl < HPIX; ++sl) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    const int c = s.c0 + 4 * q;
                    if (q < nk4 && hpix[sl] >= 0) {
                        const float* ptr = src + (size_t)hpix[sl] * C + c;
                        if (p.vec4[g]) {
                            if (c < C) v = *reinterpret_cast<const float4*>(ptr);
                        } else {
                            if (c + 0 < C) v.x = ptr[0];
                            if (c + 1 < C) v.y = ptr[1];
                            if (c + 2 < C) v.z = ptr[2];
                            if (c + 3 < C) v.w = ptr[3];
                        }
                    }
                    hreg[sl][q] = v;
                }
            }
        }
        // weights of taps [t0, t0+nt) x channels [c0, c0+4*nk4) x this N block
        const int nt = min(kTapG, ph.ntaps[g] - s.t0);
        const int ncs = nk4 + 1;  // log2(channels in chunk) : 4 -> 2, 8 -> 3
        const int rows = nt << ncs;
        const float* __restrict__ wsrc = ph.w[g] + ((size_t)(s.t0) * p.Cp[g] + s.c0) * p.Np + nblk * (NT * 16);
#pragma unroll
        for (int it = 0; it < WREG; ++it) {
            const int e = it * 256 + tid;
            const int row = e / (NT * 4);
            const int n4 = e - row * (NT * 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < rows) {
                const int tl = row >> ncs, c = row & ((1 << ncs) - 1);
                v = *reinterpret_cast<const float4*>(wsrc + ((size_t)tl * p.Cp[g] + c) * p.Np + n4 * 4);
            }
            wreg[it] = v;
        }
    };

    auto store_stage = [&](const Stage& s) {
        const int g = s.g;
        const int nk4 = (p.Cp[g] - s.c0) >= kCC ? 2 : 1;
        if (s.t0 == 0) {
#pragma unroll
            for (int sl = 0; sl < HPIX; ++sl) {
                if (hlds[sl] >= 0) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (q < nk4) {
                            float* d = Hl + (4 * q) * p.plane + hlds[sl];
                            d[0] = hreg[sl][q].x;
                            d[p.plane] = hreg[sl][q].y;
                            d[2 * p.plane] = hreg[sl][q].z;
                            d[3 * p.plane] = hreg[sl][q].w;
                        }
                    }
                }
            }
        }
        const int nt = min(kTapG, ph.ntaps[g] - s.t0);
        const int ncs = nk4 + 1;
        const int rows = nt << ncs;
#pragma unroll
        for (int it = 0; it < WREG; ++it) {
            const int e = it * 256 + tid;
            const int row = e / (NT * 4);
            const int n4 = e - row * (NT * 4);
            if (row < rows) {
                const int tl = row >> ncs, c = row & ((1 << ncs) - 1);
                *reinterpret_cast<float4*>(Wl + (tl * kCC + c) * NPL + n4 * 4) = wreg[it];
            }
        }
    };

    Stage cur = stage_skip(Stage{0, ksi * kCC, 0});  // (a phase may have no taps in a group)
    if (stage_valid(cur)) load_stage(cur);
    while (stage_valid(cur)) {
        __syncthreads();  // previous stage's LDS reads are done
        store_stage(cur);
        __syncthreads();
        Stage nxt = stage_next(cur);
        if (stage_valid(nxt)) load_stage(nxt);  // in flight during the MFMA block below

        const int g = cur.g;
        const int nk4 = (p.Cp[g] - cur.c0) >= kCC ? 2 : 1;
        const int nt = min(kTapG, ph.ntaps[g] - cur.t0);
        const int tapbase = ph.tap0[g] + cur.t0;
        for (int tl = 0; tl < nt; ++tl) {
            const int aoff = p.tapoff[tapbase + tl];
            for (int k4 = 0; k4 < nk4; ++k4) {
                const float* ap = Hl + aoff + k4 * 4 * p.plane;
                const float* bp = Wl + (tl * kCC + k4 * 4) * NPL + bbase;
                float a[kMT], b[NT];
#pragma unroll
                for (int m = 0; m < kMT; ++m) a[m] = ap[abase[m]];
#pragma unroll
                for (int n = 0; n < NT; ++n) b[n] = bp[n * 16];
#pragma unroll
                for (int m = 0; m < kMT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m][n], 0, 0, 0);
            }
        }
        cur = nxt;
    }

    // ---- epilogue: affine / activation / affine / (2x2 max-pool) / NHWC store
    // C/D layout of the 16x16 tile: column (output channel) = lane & 15, row (pixel) = 4*(lane>>4) + reg.
    const int ncol0 = nblk * (NT * 16) + li;
    float ps[NT], pb[NT], qs[NT], qb[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = ncol0 + n * 16;
        const bool ok = co < p.Cout;
        ps[n] = (p.pre_s && ok) ? p.pre_s[co] : 1.f;
        pb[n] = (p.pre_b && ok) ? p.pre_b[co] : 0.f;
        qs[n] = (p.post_s && ok) ? p.post_s[co] : 1.f;
        qb[n] = (p.post_b && ok) ? p.post_b[co] : 0.f;
    }
#pragma unroll
    for (int m = 0; m < kMT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[m][n][r] * ps[n] + pb[n];
                if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
                else if (p.act == ACT_LEAKY) v = v > 0.f ? v : kLeakySlope * v;
                acc[m][n][r] = v * qs[n] + qb[n];
            }

    float* const dstp = p.dst + (size_t)ksi * p.split_stride;
    if (p.pool) {
#pragma unroll
        for (int m = 0; m < kMT; m += 2) {
            const int t = wave * kMT + m;
            const int ig = t >> p.th_log2, ty = t & (TH - 1);
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const int i = 4 * kq + r;
                const int img = img0 + ig * p.nimg_m + (i >> p.twm_log2);
                const int oy = (y0 + ty) >> 1, ox = (x0 + (i & (TWm - 1))) >> 1;
                if (img < p.B) {
                    float* d = dstp + ((size_t)(img * p.outH + oy) * p.outW + ox) * p.Cout;
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int co = ncol0 + n * 16;
                        const float v = fmaxf(fmaxf(acc[m][n][r], acc[m][n][r + 1]),
                                              fmaxf(acc[m + 1][n][r], acc[m + 1][n][r + 1]));
                        if (co < p.Cout) d[co] = v;
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < kMT; ++m) {
            const int t = wave * kMT + m;
            const int ig = t >> p.th_log2, ty = t & (TH - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * kq + r;
                const int img = img0 + ig * p.nimg_m + (i >> p.twm_log2);
                const int oy = (y0 + ty) * p.o_mul + ph.oy_off, ox = (x0 + (i & (TWm - 1))) * p.o_mul + ph.ox_off;
                if (img < p.B) {
                    float* d = dstp + ((size_t)(img * p.outH + oy) * p.outW + ox) * p.Cout;
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int co = ncol0 + n * 16;
                        if (co < p.Cout) d[co] = acc[m][n][r];
                    }
                }
            }
        }
    }
}

template <int NT>
static hipError_t launch_conv_nt(const ConvParams& p, int hpix, hipStream_t stream) {
    const int img_groups = (p.B + p.imgs - 1) / p.imgs;
    const int ksn = p.ksplit > 1 ? p.ksplit : 1;
    dim3 grid((unsigned)(img_groups * p.tiles_y * p.tiles_x), (unsigned)(((p.Np / 16 + NT - 1) / NT) * ksn),
              (unsigned)p.nphase);
    const size_t lds = conv_lds_bytes(NT, p.plane);
    auto go = [&](auto kern) -> hipError_t {
        if (lds > 48 * 1024) {  // opt in to large dynamic LDS (160 KiB per CU on gfx950)
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
        return hipSuccess;
    };
    hipError_t e;
    if (hpix <= 2) e = go(conv_mfma_f32<NT, 2>);
    else if (hpix <= 4) e = go(conv_mfma_f32<NT, 4>);
    else return hipErrorInvalidValue;
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

hipError_t launch_conv(const ConvParams& p, int nt, int hpix, hipStream_t stream) {
    switch (nt) {
        case 1: return launch_conv_nt<1>(p, hpix, stream);
        case 2: return launch_conv_nt<2>(p, hpix, stream);
        case 3: return launch_conv_nt<3>(p, hpix, stream);
        case 4: return launch_conv_nt<4>(p, hpix, stream);
        case 5: return launch_conv_nt<5>(p, hpix, stream);
        case 6: return launch_conv_nt<6>(p, hpix, stream);
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------------------
// PI2D.getPatch + per-tile normalisation + batch fill (reference PartitionOfImage.py:58-63,77-82 and
// UnMicst1-5.py:700-702 / UnMicst2.py:679-681 / UnMicst.py:533): tiles[n,y,x,c] = float((padded[c][r][q] - mean)/std),
// padded = image at offset (margin, margin) in a zero canvas.  The padded float64 canvas is never materialised.
// HBM streaming: 8 B read + 4*Cn B written per tile pixel.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gather_normalise_kernel(const double* __restrict__ image, int C_img,
                                                              int band_row0, int band_rows, TileGeom g, int Cn,
                                                              double mean, double stdv, int tile0, int ntiles,
                                                              float* __restrict__ tiles) {
    const size_t total = (size_t)ntiles * g.P * g.P;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % g.P);
        const size_t r1 = e / g.P;
        const int y = (int)(r1 % g.P);
        const int tl = (int)(r1 / g.P);
        const int t = tile0 + tl;
        const int pr = t / g.npc, pc = t - pr * g.npc;
        const int iy = pr * g.sub + y - g.margin;   // image row
        const int ix = pc * g.sub + x - g.margin;
        const bool inside = iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
        float* d = tiles + e * Cn;
        for (int c = 0; c < Cn; ++c) {
            double v = 0.0;
            if (inside) {
                const int ci = C_img == 1 ? 0 : c;
                v = image[((size_t)ci * band_rows + (iy - band_row0)) * g.W + ix];
            }
            d[c] = (float)((v - mean) / stdv);
        }
    }
}

hipError_t launch_gather_normalise(const double* image, int C_img, int band_row0, int band_rows, const TileGeom& g,
                                   int Cn, double mean, double stdv, int tile0, int ntiles, float* tiles,
                                   hipStream_t stream) {
    if (ntiles <= 0) return hipSuccess;
    const size_t total = (size_t)ntiles * g.P * g.P;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(gather_normalise_kernel, dim3(blocks), dim3(256), 0, stream, image, C_img, band_row0, band_rows,
                       g, Cn, mean, stdv, tile0, ntiles, tiles);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Top layer: 1x1 conv (C -> K) + optional BN affine + softmax (reference UnMicst1-5.py:212-222,236-237;
// UnMicst.py:167-171,186).  HBM streaming: 4*C B read + 4*K B written per pixel.
// ------------------------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(256) head_softmax_kernel(const float* __restrict__ x, size_t npix, int C,
                                                          const float* __restrict__ w, const float* __restrict__ scale,
                                                          const float* __restrict__ bias, float* __restrict__ probs) {
    extern __shared__ float wl[];  // [C][K]
    for (int i = threadIdx.x; i < C * K; i += blockDim.x) wl[i] = w[i];
    __syncthreads();
    for (size_t px = (size_t)blockIdx.x * blockDim.x + threadIdx.x; px < npix; px += (size_t)gridDim.x * blockDim.x) {
        const float* xp = x + px * C;
        float acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = 0.f;
        if ((C & 3) == 0) {
            for (int c = 0; c < C; c += 4) {
                const float4 v = *reinterpret_cast<const float4*>(xp + c);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    acc[k] = fmaf(v.x, wl[(c + 0) * K + k], acc[k]);
                    acc[k] = fmaf(v.y, wl[(c + 1) * K + k], acc[k]);
                    acc[k] = fmaf(v.z, wl[(c + 2) * K + k], acc[k]);
                    acc[k] = fmaf(v.w, wl[(c + 3) * K + k], acc[k]);
                }
            }
        } else {
            for (int c = 0; c < C; ++c) {
                const float v = xp[c];
#pragma unroll
                for (int k = 0; k < K; ++k) acc[k] = fmaf(v, wl[c * K + k], acc[k]);
            }
        }
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (scale) acc[k] = acc[k] * scale[k] + bias[k];
            m = fmaxf(m, acc[k]);
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            acc[k] = expf(acc[k] - m);
            s += acc[k];
        }
        const float inv = 1.f / s;
#pragma unroll
        for (int k = 0; k < K; ++k) probs[px * K + k] = acc[k] * inv;
    }
}

hipError_t launch_head_softmax(const float* x, size_t npix, int C, int K, const float* w, const float* scale,
                               const float* bias, float* probs, hipStream_t stream) {
    if (npix == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>((npix + 255) / 256, 256 * 16);
    const size_t lds = sizeof(float) * (size_t)C * K;
    switch (K) {
        case 2: hipLaunchKernelGGL(head_softmax_kernel<2>, dim3(blocks), dim3(256), lds, stream, x, npix, C, w, scale, bias, probs); break;
        case 3: hipLaunchKernelGGL(head_softmax_kernel<3>, dim3(blocks), dim3(256), lds, stream, x, npix, C, w, scale, bias, probs); break;
        case 4: hipLaunchKernelGGL(head_softmax_kernel<4>, dim3(blocks), dim3(256), lds, stream, x, npix, C, w, scale, bias, probs); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// PI2D.patchOutput + getValidOutput as a gather (reference PartitionOfImage.py:92-122): every output pixel visits
// its <= 4 covering tiles in ascending tile index -- the order of the reference's sequential `+=` loop -- so the
// float16 accumulators see the same sequence of roundings.  fp16-compat: Count = fp16(double(Count) + W),
// Output = fp16(double(Output) + double(P)*W) per tile, result = fp16(float(Output)/float(Count)) (numpy's
// float16 divide runs in float32).  fp32 mode: blend in double, one rounding to float32.
// HBM streaming: <= 4*K*4 B read per pixel (1.78*K*4 on average), 2*K (or 4*K) B written.
// ------------------------------------------------------------------------------------------------------------
// (contraction off from here on: `o + p*w` must round the product and the sum separately, as numpy does -- with imSize a
// power of two the product is exact either way, for other tile sizes a fused multiply-add would differ in the last bit)
#pragma clang fp contract(off)
__device__ __forceinline__ double blend_weight(int r, int c, int P, int two_m) {
    const int d = min(min(r, c), min(P - 1 - r, P - 1 - c));   // ring index, reference PartitionOfImage.py:30-38
    if (d == 0) return 0.0;
    if (d >= two_m) return 1.0;
    return (double)d / (double)two_m;
}

// float64 -> binary16, round to nearest even, for the stitch (numpy's store of a float64 sum into a float16 array,
// PartitionOfImage.py:95-98) on the conversion hardware: the double is first rounded TO ODD into binary32 (truncate, then set the
// last bit if anything was lost), and v_cvt_f16_f32 rounds that to nearest even.  Rounding to odd at 24 bits keeps the sticky
// information a second rounding to <= 22 bits needs (Boldo & Melquiond: p' >= p + 2), so the result is the correctly rounded
// binary16 of the double -- subnormal results included (binary16 denormals are on) -- bit for bit double_to_half_rne's, which
// stays as the host-side statement of the conversion (umx_test_double_to_half).  ~6 instructions instead of ~40 with branches:
// the kernel converts up to 16 sums per output pixel.
__device__ __forceinline__ uint16_t d2h_rne(double d) {
    float f = __double2float_rz(d);
    if ((double)f != d) f = __uint_as_float(__float_as_uint(f) | 1u);
    return __half_as_ushort(__float2half_rn(f));
}

// test entry (umx_test_double_to_half_dev): the device routine on caller-given doubles, against the host statement of the conversion
__global__ void d2h_rne_test_kernel(const double* __restrict__ in, uint16_t* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = d2h_rne(in[i]);
}
hipError_t launch_d2h_rne_test(const double* in, uint16_t* out, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(d2h_rne_test_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, out, n);
    return hipGetLastError();
}

// the drivers' uint8 cast of one float16 probability (reference UnMicst1-5.py:848-854 at the identity grid; the same IEEE operations
// as half_to_u8_kernel below, one rounding each): np.uint8(255 * pm) in float16, resize = u8 * (1 / 255) in float64, np.uint8(255 * .)
__device__ __forceinline__ unsigned char u8_of_half(__half pm) {
    const __half p255 = __hmul(__float2half_rn(255.f), pm);
    const unsigned char first = (unsigned char)(int)__half2float(p255);
    const double f = (double)first * (1.0 / 255);
    return (unsigned char)(int)(255.0 * f);
}

// stitch: 0 = float16 planes (the reference's accumulators), 1 = float32, 2 = the float16 result cast to the drivers' uint8 on the way
// out (the raw entry points: no float16 plane is written and read back).  plane_rows: rows per class plane of the destination
// (>= y1 - y0: the sharded schedule writes straight into its padded gather buffer).
template <int K>
__global__ void __launch_bounds__(256) stitch_kernel(const float* __restrict__ probs, int tpr0, int tpr1, TileGeom g,
                                                    int mode, int stitch, int y0, int y1, int plane_rows, void* __restrict__ out) {
    const int rows = plane_rows;
    const size_t total = (size_t)(y1 - y0) * g.W;
    const int two_m = 2 * g.margin;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % g.W);
        const int y = y0 + (int)(e / g.W);
        const int R = y + g.margin, Cc = x + g.margin;  // padded coordinates
        // covering patch rows: pr*sub <= R < pr*sub + P
        int pr_hi = R / g.sub;
        if (pr_hi > g.npr - 1) pr_hi = g.npr - 1;
        int pr_lo = (R - g.P + g.sub) / g.sub;  // ceil((R-P+1)/sub) for R-P+1 possibly negative
        if (R - g.P + 1 <= 0) pr_lo = 0;
        int pc_hi = Cc / g.sub;
        if (pc_hi > g.npc - 1) pc_hi = g.npc - 1;
        int pc_lo = (Cc - g.P + g.sub) / g.sub;
        if (Cc - g.P + 1 <= 0) pc_lo = 0;

        if (mode == kModeReplace) {
            // last tile in index order wins: Output[tile] = P (float32 -> float16 store), no Count
            const int pr = pr_hi, pc = pc_hi;
            const int r = R - pr * g.sub, c = Cc - pc * g.sub;
            const float* pp = probs + ((((size_t)(pr - tpr0) * g.npc + pc) * g.P + r) * g.P + c) * K;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (stitch == 0) ((__half*)out)[((size_t)k * rows + (y - y0)) * g.W + x] = __float2half_rn(pp[k]);
                else if (stitch == 2) ((unsigned char*)out)[((size_t)k * rows + (y - y0)) * g.W + x] = u8_of_half(__float2half_rn(pp[k]));
                else ((float*)out)[((size_t)k * rows + (y - y0)) * g.W + x] = pp[k];
            }
            continue;
        }

        if (stitch != 1) {
            uint16_t cnt = 0;
            uint16_t o[K];
#pragma unroll
            for (int k = 0; k < K; ++k) o[k] = 0;
            for (int pr = pr_lo; pr <= pr_hi; ++pr)
                for (int pc = pc_lo; pc <= pc_hi; ++pc) {
                    const int r = R - pr * g.sub, c = Cc - pc * g.sub;
                    const double w = blend_weight(r, c, g.P, two_m);
                    const float* pp = probs + ((((size_t)(pr - tpr0) * g.npc + pc) * g.P + r) * g.P + c) * K;
                    cnt = d2h_rne((double)__half2float(__ushort_as_half(cnt)) + w);
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        o[k] = d2h_rne((double)__half2float(__ushort_as_half(o[k])) + (double)pp[k] * w);
                }
            const float cf = __half2float(__ushort_as_half(cnt));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float q = __half2float(__ushort_as_half(o[k])) / cf;   // IEEE float32 division
                if (stitch == 2) ((unsigned char*)out)[((size_t)k * rows + (y - y0)) * g.W + x] = u8_of_half(__float2half_rn(q));
                else ((__half*)out)[((size_t)k * rows + (y - y0)) * g.W + x] = __float2half_rn(q);
            }
        } else {
            double cnt = 0.0;
            double o[K];
#pragma unroll
            for (int k = 0; k < K; ++k) o[k] = 0.0;
            for (int pr = pr_lo; pr <= pr_hi; ++pr)
                for (int pc = pc_lo; pc <= pc_hi; ++pc) {
                    const int r = R - pr * g.sub, c = Cc - pc * g.sub;
                    const double w = blend_weight(r, c, g.P, two_m);
                    const float* pp = probs + ((((size_t)(pr - tpr0) * g.npc + pc) * g.P + r) * g.P + c) * K;
                    cnt += w;
#pragma unroll
                    for (int k = 0; k < K; ++k) o[k] += (double)pp[k] * w;
                }
#pragma unroll
            for (int k = 0; k < K; ++k) ((float*)out)[((size_t)k * rows + (y - y0)) * g.W + x] = (float)(o[k] / cnt);
        }
    }
}

hipError_t launch_stitch(const float* probs, int tpr0, int tpr1, const TileGeom& g, int K, int mode, int stitch,
                         int y0, int y1, void* out, hipStream_t stream, int plane_rows) {
    if (y1 <= y0) return hipSuccess;
    if (plane_rows <= 0) plane_rows = y1 - y0;
    if (plane_rows < y1 - y0) return hipErrorInvalidValue;
    const size_t total = (size_t)(y1 - y0) * g.W;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 16);
    switch (K) {
        case 2: hipLaunchKernelGGL(stitch_kernel<2>, dim3(blocks), dim3(256), 0, stream, probs, tpr0, tpr1, g, mode, stitch, y0, y1, plane_rows, out); break;
        case 3: hipLaunchKernelGGL(stitch_kernel<3>, dim3(blocks), dim3(256), 0, stream, probs, tpr0, tpr1, g, mode, stitch, y0, y1, plane_rows, out); break;
        case 4: hipLaunchKernelGGL(stitch_kernel<4>, dim3(blocks), dim3(256), 0, stream, probs, tpr0, tpr1, g, mode, stitch, y0, y1, plane_rows, out); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Driver-side pre/post-processing at --scalingFactor 1 (reference UnMicst1-5.py:807-821,848-854; toolbox/imtools.py:42-53):
//   raw uint8/uint16 plane -> float64 in [0,1] (skimage's resize converts with a multiply by 1/max) -> optional
//   rescale_intensity((min, max) -> (0, 0.983)); float16 probability plane -> uint8 by the reference's double cast.
// Every operation is the same IEEE operation numpy performs, one rounding each (contraction is switched off), so the
// results are bit-identical to the host path in unmicst_amd/driver.py.
// ------------------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)

template <typename T>
__global__ void __launch_bounds__(256) minmax_kernel(const T* __restrict__ x, size_t n, unsigned* __restrict__ mm /*[min,max]*/) {
    // 16-byte loads over the aligned middle of the range (a 2-byte load per lane moved 89 GB/s: 0.38 ms per 16.8 M-pixel slab),
    // scalar loads for the unaligned head and the tail
    constexpr int PER = 16 / (int)sizeof(T);
    unsigned lo = 0xFFFFFFFFu, hi = 0u;
    const size_t addr = reinterpret_cast<size_t>(x);
    size_t head = ((16 - (addr & 15)) & 15) / sizeof(T);
    if (head > n) head = n;
    const size_t nvec = (n - head) / PER;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (size_t)gridDim.x * blockDim.x;
    const uint4* const xv = reinterpret_cast<const uint4*>(x + head);
    for (size_t i = gid; i < nvec; i += gsz) {
        const uint4 w = xv[i];
        const unsigned ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (sizeof(T) == 2) {
                const unsigned a = ws[k] & 0xFFFFu, b = ws[k] >> 16;
                lo = min(lo, min(a, b));
                hi = max(hi, max(a, b));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned a = (ws[k] >> (8 * j)) & 0xFFu;
                    lo = min(lo, a);
                    hi = max(hi, a);
                }
            }
        }
    }
    for (size_t i = gid; i < head; i += gsz) { const unsigned v = x[i]; lo = min(lo, v); hi = max(hi, v); }
    for (size_t i = head + nvec * PER + gid; i < n; i += gsz) { const unsigned v = x[i]; lo = min(lo, v); hi = max(hi, v); }
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, o));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[0], lo);
        atomicMax(&mm[1], hi);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) raw_to_double_kernel(const T* __restrict__ x, size_t n, double inv_max, int rescale,
                                                           const unsigned* __restrict__ mm, double* __restrict__ out) {
    const double lo = (double)mm[0] * inv_max, hi = (double)mm[1] * inv_max;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double v = (double)x[i] * inv_max;                    // np.multiply(I, 1.0 / imax)
        if (rescale) {
            if (lo != hi) v = ((v - lo) / (hi - lo)) * 0.983;   // (I - imin) / (imax - imin) * (omax - omin) + omin, omin = 0
            else v = fmin(fmax(v, 0.0), 0.983);               // np.clip(I, omin, omax)
        }
        out[i] = v;
    }
}

__global__ void __launch_bounds__(256) half_to_u8_kernel(const __half* __restrict__ pm, size_t n, unsigned char* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const __half p255 = __hmul(__float2half_rn(255.f), pm[i]);          // 255 * float16 -> float16 (numpy)
        const unsigned char first = (unsigned char)(int)__half2float(p255);   // np.uint8: truncation
        const double f = (double)first * (1.0 / 255);                       // resize at the identity grid: u8 * (1/255)
        out[i] = (unsigned char)(int)(255.0 * f);                           // np.uint8(255 * PM)
    }
}

hipError_t launch_minmax_init(unsigned* mm, hipStream_t stream) {
    const unsigned init[2] = {0xFFFFFFFFu, 0u};
    return hipMemcpyAsync(mm, init, sizeof init, hipMemcpyHostToDevice, stream);
}

// mm = (min(mm[0], min x), max(mm[1], max x)): accumulates, so that a plane can be reduced slab by slab as it is uploaded
hipError_t launch_minmax(const void* raw, int bits, size_t n, unsigned* mm, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>((n / 8 + 255) / 256 + 1, 256 * 8);
    if (bits == 16)
        hipLaunchKernelGGL(minmax_kernel<unsigned short>, dim3(blocks), dim3(256), 0, stream, (const unsigned short*)raw, n, mm);
    else if (bits == 8)
        hipLaunchKernelGGL(minmax_kernel<unsigned char>, dim3(blocks), dim3(256), 0, stream, (const unsigned char*)raw, n, mm);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// im2double (+ rescale_intensity with the plane's (min, max) in mm) of n raw values
hipError_t launch_raw_convert(const void* raw, int bits, size_t n, int rescale, const unsigned* mm, double* out,
                              hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 16);
    if (bits == 16)
        hipLaunchKernelGGL(raw_to_double_kernel<unsigned short>, dim3(blocks), dim3(256), 0, stream, (const unsigned short*)raw, n,
                           1.0 / 65535, rescale, mm, out);
    else if (bits == 8)
        hipLaunchKernelGGL(raw_to_double_kernel<unsigned char>, dim3(blocks), dim3(256), 0, stream, (const unsigned char*)raw, n,
                           1.0 / 255, rescale, mm, out);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_raw_to_double(const void* raw, int bits, size_t n, int rescale, unsigned* mm, double* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipError_t e = launch_minmax_init(mm, stream);
    if (e != hipSuccess) return e;
    if ((e = launch_minmax(raw, bits, n, mm, stream)) != hipSuccess) return e;
    return launch_raw_convert(raw, bits, n, rescale, mm, out, stream);
}

// ------------------------------------------------------------------------------------------------------------
// skimage.transform.resize(I, (h, w)) with its defaults (order 1, mode 'reflect' -> scipy 'mirror', anti-aliasing
// Gaussian when an axis shrinks, clip to the filtered image's range), as the drivers call it around the inference at
// --scalingFactor != 1 (reference UnMicst1-5.py:813-816,850; toolbox/imtools.py:8).  Restated from scikit-image >= 0.19:
//   sigma = max(0, (in/out - 1) / 2) per axis; scipy.ndimage.gaussian_filter (truncate 4: radius int(4 sigma + 0.5),
//   weights exp(-x^2 / (2 sigma^2)) / sum, mode 'mirror': d c b | a b c d | c b a), then scipy.ndimage.zoom(order 1,
//   mode 'mirror', grid_mode True): input coordinate (o + 0.5) * in / out - 0.5, linear interpolation.
// float64 throughout, same operation order as scipy's correlate1d (centre tap, then symmetric pairs from the outside in).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int mirror_index(int i, int n) {   // scipy 'mirror' (reflect about the edge sample)
    if (n == 1) return 0;
    const int period = 2 * n - 2;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}

// one axis of the separable Gaussian: dst[y][x] = sum_j w[j] * src[.. +- j ..]; axis 0 = rows (y), 1 = columns (x)
__global__ void __launch_bounds__(256) gauss1d_kernel(const double* __restrict__ src, double* __restrict__ dst, int H, int W,
                                                     int axis, int radius, const double* __restrict__ w /* [radius+1], w[0] = centre */) {
    const size_t n = (size_t)H * W;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(e / W), x = (int)(e - (size_t)y * W);
        double acc = src[e] * w[0];
        for (int j = radius; j >= 1; --j) {
            double a, b;
            if (axis == 0) {
                a = src[(size_t)mirror_index(y - j, H) * W + x];
                b = src[(size_t)mirror_index(y + j, H) * W + x];
            } else {
                a = src[(size_t)y * W + mirror_index(x - j, W)];
                b = src[(size_t)y * W + mirror_index(x + j, W)];
            }
            acc += (a + b) * w[j];
        }
        dst[e] = acc;
    }
}

// (min, max) of non-negative doubles as ordered bit patterns (mm64[0] = min, mm64[1] = max; initialise to ~0 / 0)
__global__ void __launch_bounds__(256) minmax_f64_kernel(const double* __restrict__ x, size_t n, unsigned long long* __restrict__ mm64) {
    unsigned long long lo = ~0ull, hi = 0ull;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(x[i]);
        lo = b < lo ? b : lo;
        hi = b > hi ? b : hi;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&mm64[0], lo); atomicMax(&mm64[1], hi); }
}

// order-1 zoom (grid mode, mirror) of src [H,W] -> [h,w], clipped to [clip[0], clip[1]] (bit patterns of the source range);
// out_u8 != NULL: the drivers' final np.uint8(255 * .) instead of the float64 plane
__global__ void __launch_bounds__(256) zoom1_kernel(const double* __restrict__ src, int H, int W, int h, int w,
                                                   const unsigned long long* __restrict__ clip, double* __restrict__ dst,
                                                   unsigned char* __restrict__ out_u8) {
    const double lo = __longlong_as_double((long long)clip[0]), hi = __longlong_as_double((long long)clip[1]);
    const double zy = (double)H / (double)h, zx = (double)W / (double)w;
    const size_t n = (size_t)h * w;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int oy = (int)(e / w), ox = (int)(e - (size_t)oy * w);
        const double cy = ((double)oy + 0.5) * zy - 0.5, cx = ((double)ox + 0.5) * zx - 0.5;
        const double fy = floor(cy), fx = floor(cx);
        const double ty = cy - fy, tx = cx - fx;
        const int y0 = mirror_index((int)fy, H), y1 = mirror_index((int)fy + 1, H);
        const int x0 = mirror_index((int)fx, W), x1 = mirror_index((int)fx + 1, W);
        // scipy's separable evaluation: rows weighted (1-ty, ty), columns (1-tx, tx), accumulated in this order
        double v = 0.0;
        v += src[(size_t)y0 * W + x0] * (1.0 - ty) * (1.0 - tx);   // value * w_axis0 * w_axis1, C order of the 2x2 support
        v += src[(size_t)y0 * W + x1] * (1.0 - ty) * tx;
        v += src[(size_t)y1 * W + x0] * ty * (1.0 - tx);
        v += src[(size_t)y1 * W + x1] * ty * tx;
        v = fmin(fmax(v, lo), hi);
        if (out_u8) out_u8[e] = (unsigned char)(int)(255.0 * v);
        else dst[e] = v;
    }
}

// rescale_intensity(I, in_range = (min, max) from mm64, out_range = (0, 0.983)) in place (reference UnMicst1-5.py:817-821)
__global__ void __launch_bounds__(256) rescale_f64_kernel(double* __restrict__ x, size_t n, const unsigned long long* __restrict__ mm64) {
    const double lo = __longlong_as_double((long long)mm64[0]), hi = __longlong_as_double((long long)mm64[1]);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double v = fmin(fmax(x[i], lo), hi);
        if (lo != hi) v = ((v - lo) / (hi - lo)) * 0.983;
        else v = fmin(fmax(v, 0.0), 0.983);
        x[i] = v;
    }
}

// uint8 plane -> float64 u8 * (1/255) (resize's img_as_float), or float16 plane -> np.uint8(255 * pm) -> same
__global__ void __launch_bounds__(256) half_to_u8_f64_kernel(const __half* __restrict__ pm, size_t n, double* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const __half p255 = __hmul(__float2half_rn(255.f), pm[i]);
        out[i] = (double)(unsigned char)(int)__half2float(p255) * (1.0 / 255);
    }
}

// ------------------------------------------------------------------------------------------------------------
// np.percentile(I, q) of a float64 plane of non-negative values (reference UnMicst1-5.py:820, the --outlier limit), exact:
// the two order statistics a[k], a[k+1] by radix selection on the bit patterns (non-negative doubles order like their
// bits), 8 passes of 8 bits, both targets in one read of the plane per pass; then numpy's linear interpolation.
// st[t] = {prefix bits found so far, rank left inside the bucket}; hist = 2 x 256 counters, zeroed by the pick kernel.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rsel_hist_kernel(const double* __restrict__ x, size_t n, const unsigned long long* __restrict__ st,
                                                       int pass, unsigned* __restrict__ hist) {
    __shared__ unsigned h[2][256];
    h[0][threadIdx.x] = 0; h[1][threadIdx.x] = 0;
    __syncthreads();
    const int shift = 56 - 8 * pass;
    const unsigned long long p0 = st[0], p1 = st[2];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(x[i]);
        const unsigned d = (unsigned)(b >> shift) & 255u;
        const bool m0 = pass == 0 || (b >> (shift + 8)) == (p0 >> (shift + 8));
        const bool m1 = pass == 0 || (b >> (shift + 8)) == (p1 >> (shift + 8));
        if (m0) atomicAdd(&h[0][d], 1u);
        if (m1) atomicAdd(&h[1][d], 1u);
    }
    __syncthreads();
    if (h[0][threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[0][threadIdx.x]);
    if (h[1][threadIdx.x]) atomicAdd(&hist[256 + threadIdx.x], h[1][threadIdx.x]);
}

__global__ void rsel_init_kernel(unsigned long long* __restrict__ st, unsigned long long k0, unsigned long long k1,
                                 unsigned* __restrict__ hist) {
    if (threadIdx.x == 0) { st[0] = 0; st[1] = k0; st[2] = 0; st[3] = k1; }
    for (int i = threadIdx.x; i < 512; i += blockDim.x) hist[i] = 0;
}

__global__ void rsel_pick_kernel(unsigned long long* __restrict__ st, int pass, unsigned* __restrict__ hist) {
    const int t = threadIdx.x;   // one thread per target
    if (t < 2) {
        const int shift = 56 - 8 * pass;
        unsigned long long k = st[2 * t + 1], cum = 0;
        int b = 0;
        for (; b < 255; ++b) {
            const unsigned long long c = hist[256 * t + b];
            if (k < cum + c) break;
            cum += c;
        }
        st[2 * t] |= (unsigned long long)b << shift;
        st[2 * t + 1] = k - cum;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += blockDim.x) hist[i] = 0;
}

// numpy's _lerp(a, b, t): a + (b - a) * t, and b - (b - a) * (1 - t) where t >= 0.5; the result replaces the plane's max
__global__ void percentile_lerp_kernel(const unsigned long long* __restrict__ st, double gamma, unsigned long long* __restrict__ mm64) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double a = __longlong_as_double((long long)st[0]), b = __longlong_as_double((long long)st[2]);
        const double d = b - a;
        double r = a + d * gamma;
        if (gamma >= 0.5) r = b - d * (1.0 - gamma);
        mm64[1] = (unsigned long long)__double_as_longlong(r);
    }
}

static unsigned blocks_for(size_t n) { return (unsigned)std::min<size_t>((n + 255) / 256, 256 * 16); }

hipError_t launch_gauss1d(const double* src, double* dst, int H, int W, int axis, int radius, const double* w_dev,
                          hipStream_t stream) {
    hipLaunchKernelGGL(gauss1d_kernel, dim3(blocks_for((size_t)H * W)), dim3(256), 0, stream, src, dst, H, W, axis, radius, w_dev);
    return hipGetLastError();
}
hipError_t launch_minmax_f64(const double* x, size_t n, unsigned long long* mm64, hipStream_t stream) {
    const unsigned long long init[2] = {~0ull, 0ull};
    hipError_t e = hipMemcpyAsync(mm64, init, sizeof init, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(minmax_f64_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, x, n, mm64);
    return hipGetLastError();
}
hipError_t launch_zoom1(const double* src, int H, int W, int h, int w, const unsigned long long* clip, double* dst,
                        unsigned char* out_u8, hipStream_t stream) {
    hipLaunchKernelGGL(zoom1_kernel, dim3(blocks_for((size_t)h * w)), dim3(256), 0, stream, src, H, W, h, w, clip, dst, out_u8);
    return hipGetLastError();
}
// mm64[1] <- np.percentile(x, q) for the n non-negative values of x; st: 4 x 8 bytes, hist: 512 x 4 bytes of device scratch
hipError_t launch_percentile_f64(const double* x, size_t n, double q, unsigned long long* st, unsigned* hist,
                                 unsigned long long* mm64, hipStream_t stream) {
    // numpy: virtual index = (n - 1) * (q / 100), previous = floor, next = previous + 1 (clipped), gamma = virtual - previous
    const double virt = (double)(n - 1) * (q / 100.0);
    double prev = std::floor(virt);
    if (prev < 0) prev = 0;
    if (prev > (double)(n - 1)) prev = (double)(n - 1);
    const unsigned long long k0 = (unsigned long long)prev, k1 = std::min<unsigned long long>(k0 + 1, n - 1);
    const double gamma = virt - prev;
    hipLaunchKernelGGL(rsel_init_kernel, dim3(1), dim3(256), 0, stream, st, k0, k1, hist);
    for (int pass = 0; pass < 8; ++pass) {
        hipLaunchKernelGGL(rsel_hist_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, x, n, st, pass, hist);
        hipLaunchKernelGGL(rsel_pick_kernel, dim3(1), dim3(256), 0, stream, st, pass, hist);
    }
    hipLaunchKernelGGL(percentile_lerp_kernel, dim3(1), dim3(64), 0, stream, st, gamma, mm64);
    return hipGetLastError();
}
hipError_t launch_rescale_f64(double* x, size_t n, const unsigned long long* mm64, hipStream_t stream) {
    hipLaunchKernelGGL(rescale_f64_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, x, n, mm64);
    return hipGetLastError();
}
hipError_t launch_half_to_u8_f64(const void* pm_half, size_t n, double* out, hipStream_t stream) {
    hipLaunchKernelGGL(half_to_u8_f64_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, (const __half*)pm_half, n, out);
    return hipGetLastError();
}

hipError_t launch_half_to_u8(const void* pm_half, size_t n, unsigned char* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(half_to_u8_kernel, dim3(blocks), dim3(256), 0, stream, (const __half*)pm_half, n, out);
    return hipGetLastError();
}

}  // namespace umx
