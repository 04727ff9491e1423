// Split-precision implicit-GEMM convolution for gfx950 (MI355X, CDNA4): every fp32 product x*w is evaluated as
//     x_hi*w_hi + x_hi*w_lo + x_lo*w_hi          (x = x_hi + x_lo, w = w_hi + w_lo, all four IEEE binary16)
// on v_mfma_f32_16x16x32_f16 with fp32 accumulation.  A two-term binary16 split carries 22 significand bits, the dropped
// x_lo*w_lo term is ~2^-22 relative, so the result is fp32-equivalent (measured ~1e-6 on the softmax output, the same
// as the exact-fp32 MFMA path) at 16/3 = 5.3x the fp32 matrix rate.  gfx950 has no TF32/xf32 path; this is the
// fast route that still holds the 1e-4 parity tolerance.
//
// Data layout
//   activations  two binary16 planes (hi, lo) per tensor, NHWC with the channel count padded to a multiple of 8
//                ("octets": 8 channels = 16 bytes = one lane's A-fragment of the 16x16x32 MFMA); pad channels are 0.
//   weights      pre-packed on the host in the exact LDS image order, per (phase, N-block): for every stage, for every
//                k-step, for every 16-wide N-tile: a hi image then a lo image of [64 lanes][8 halves] -- so staging is a
//                linear copy and a lane's B-fragment read is base + lane*16 (conflict-free ds_read_b128).
//   LDS          input halo, octet-planar: plane[octet][halo pixel] of 16-byte slots, hi planes then lo planes (an
//                A-fragment read = 16 consecutive pixels of one plane per 16-lane group: conflict-free when the four
//                groups read four planes at the same tap); then the stage's weight images.
// Staging is LDS-DMA (global_load_lds_dwordx4: no staging VGPRs), single-buffered; two 256-thread workgroups per CU
// cover each other's load phases.  A k-step is any 4 (tap, octet) pairs (table built on the host), so channel counts
// only need to be multiples of 8, not 32.
#include "umx_kernels.h"

namespace umx {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define UMX_GLDS16(gptr, lptr)                                                                        \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),           \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

template <int NT>
__global__ void __launch_bounds__(256, 2) conv_f16x3(const HConvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4;    // which 8-wide k group of the 16x16x32 MFMA this lane feeds
    const int li = lane & 15;   // pixel (A) / output channel (B, C/D) inside the tile

    // ---- workgroup -> (image group, spatial tile), N block, phase
    const int TWm = 1 << p.twm_log2, TH = 1 << p.th_log2;
    int bid = blockIdx.x;
    const int tx_i = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty_i = bid % p.tiles_y; bid /= p.tiles_y;
    const int img0 = bid * p.imgs;
    const int y0 = ty_i * TH, x0 = tx_i * TWm;
    const int nblk = blockIdx.y;
    const HPhase& ph = p.ph[blockIdx.z];

    // ---- halo slot -> source pixel, fixed for the whole kernel: slot e = c*64 + lane
    int hsrc[kHaloChunks];   // >= 0: pixel index (img*H + y)*W + x;  -1: outside the image (zeros);  -2: no such slot
#pragma unroll
    for (int c = 0; c < kHaloChunks; ++c) {
        const int e = c * 64 + lane;
        int v = -2;
        if (e < p.nhalo) {
            const int il = e / p.imgplane;
            const int r = e - il * p.imgplane;
            const int hy = r / p.hw;
            const int hx = r - hy * p.hw;
            const int gy = y0 + p.ymin + hy, gx = x0 + p.xmin + hx, img = img0 + il;
            v = (img < p.B && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? (img * p.H + gy) * p.W + gx : -1;
        }
        hsrc[c] = v;
    }

    // ---- per-lane A-fragment pixel offsets (bytes) of this wave's M-tiles
    int abase[kMT];
#pragma unroll
    for (int m = 0; m < kMT; ++m) {
        const int t = wave * kMT + m;
        const int ig = t >> p.th_log2, ty = t & (TH - 1);
        abase[m] = ((ig * p.nimg_m + (li >> p.twm_log2)) * p.imgplane + ty * p.hw + (li & (TWm - 1))) << 4;
    }

    f32x4 acc[kMT][NT];
#pragma unroll
    for (int m = 0; m < kMT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int plane_bytes = p.plane_slots << 4;
    const int lo_off = p.lo_off;   // byte offset of the lo planes
    unsigned char* const Bl = smem + p.b_off;
    const uint4* const wbase = ph.w + (size_t)nblk * ph.wblk_stride;
    const int nch = (p.nhalo + 63) >> 6;

    for (int s = 0; s < ph.nstages; ++s) {
        const HStage st = p.stages[ph.stage0 + s];
        __syncthreads();   // every wave is done reading the previous stage's LDS images

        unsigned kb[kStageK];
#pragma unroll
        for (int j = 0; j < kStageK; ++j) kb[j] = j < st.nk ? (unsigned)p.kmap[(st.k0 + j) * 4 + q] << 4 : 0u;

        if (st.group >= 0) {   // (re)load the halo: octets [oct0, oct0+noct) of operand group `group`
            const int g = st.group;
            const _Float16* const shi = p.src_hi[g];
            const _Float16* const slo = p.src_lo[g];
            const int Cs = p.Cs[g];
            for (int t = wave; t < st.noct * 2; t += kWaves) {   // one (plane, hi|lo) per iteration, wave-uniform
                const int pl = t >> 1;
                const _Float16* const sb = ((t & 1) ? slo : shi) + (st.oct0 + pl) * 8;
                unsigned char* const dst = smem + ((t & 1) ? lo_off : 0) + pl * plane_bytes;
#pragma unroll
                for (int c = 0; c < kHaloChunks; ++c) {
                    if (c < nch && hsrc[c] != -2) {
                        const void* src = hsrc[c] >= 0 ? (const void*)(sb + (size_t)hsrc[c] * Cs) : (const void*)p.zeros;
                        UMX_GLDS16(src, dst + c * 1024);
                    }
                }
            }
        }
        {   // weight images of this stage: a linear copy, 1 KiB per wave-instruction
            const uint4* const wsrc = wbase + st.woff + lane;
            const int npieces = st.nk * NT * 2;
            for (int pc = wave; pc < npieces; pc += kWaves) UMX_GLDS16(wsrc + pc * 64, Bl + pc * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

#pragma unroll
        for (int j = 0; j < kStageK; ++j) {
            if (j < st.nk) {
                const unsigned char* const ap = smem + kb[j];
                const unsigned char* const bp = Bl + j * (NT * 2048) + lane * 16;
                h8 ah[kMT], al[kMT];
#pragma unroll
                for (int m = 0; m < kMT; ++m) {
                    ah[m] = *reinterpret_cast<const h8*>(ap + abase[m]);
                    al[m] = *reinterpret_cast<const h8*>(ap + abase[m] + lo_off);
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) {   // B fragments are streamed: two live at a time
                    const h8 bh = *reinterpret_cast<const h8*>(bp + n * 2048);
                    const h8 bl = *reinterpret_cast<const h8*>(bp + n * 2048 + 1024);
#pragma unroll
                    for (int m = 0; m < kMT; ++m) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh, acc[m][n], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue: (acc * pre_s + pre_b) -> activation -> (* post_s + post_b) -> [2x2 max-pool] -> * out_scale
    //      -> split to (hi, lo) binary16 NHWC, or fp32 NHWC for the layer feeding the softmax head.
    // C/D layout of the 16x16 tile: column (output channel) = lane & 15, row (pixel) = 4*(lane>>4) + reg.
    const int ncol0 = nblk * (NT * 16) + li;
    float ps[NT], pb[NT], qs[NT], qb[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = ncol0 + n * 16;
        const bool ok = co < p.Cout;
        ps[n] = ok ? p.pre_s[co] : 0.f;
        pb[n] = (p.pre_b && ok) ? p.pre_b[co] : 0.f;
        qs[n] = ((p.post_s && ok) ? p.post_s[co] : 1.f) * p.out_scale;
        qb[n] = ((p.post_b && ok) ? p.post_b[co] : 0.f) * p.out_scale;
    }
    bool big = false;
#pragma unroll
    for (int m = 0; m < kMT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[m][n][r] * ps[n] + pb[n];
                if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
                else if (p.act == ACT_LEAKY) v = v > 0.f ? v : 0.2f * v;
                v = v * qs[n] + qb[n];
                big |= !(fabsf(v) < 60000.f);
                acc[m][n][r] = v;
            }
    if (big && p.dst_f32 == nullptr) atomicOr(p.overflow_flag, 1);   // binary16 range exceeded: the host reports it

    if (p.dst_f32) {
        // fp32 NHWC output (the tensor the softmax head reads): direct stores, 64 B per 16-lane group
        auto store = [&](int img, int oy, int ox, int n, float v) {
            const int co = ncol0 + n * 16;
            if (co < p.Cout) p.dst_f32[((size_t)(img * p.outH + oy) * p.outW + ox) * p.Cout + co] = v;
        };
#pragma unroll
        for (int m = 0; m < kMT; ++m) {
            const int t = wave * kMT + m;
            const int ig = t >> p.th_log2, ty = t & (TH - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * q + r;
                const int img = img0 + ig * p.nimg_m + (i >> p.twm_log2);
                const int oy = (y0 + ty) * p.o_mul + ph.oy_off, ox = (x0 + (i & (TWm - 1))) * p.o_mul + ph.ox_off;
                if (img < p.B) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) store(img, oy, ox, n, acc[m][n][r]);
                }
            }
        }
        return;
    }

    // (hi, lo) binary16 output through a per-wave LDS transpose: the C/D layout has one channel per lane (2-byte
    // stores, 32 B per 16-lane group); staged as [pixel][channel] rows, every lane stores 16 contiguous bytes and a
    // wave-instruction covers whole pixels' channel vectors (consecutive pixels are contiguous in NHWC).
    __syncthreads();   // every wave is done with the halo / weight images: LDS is free
    constexpr int PITCH = NT * 32 + 16;   // bytes per staged pixel row of one plane (16-byte aligned)
    constexpr int PLANE = 16 * PITCH;
    constexpr int UR = NT * 2;            // 16-byte units per staged row
    unsigned char* const stg = smem + wave * (2 * PLANE);
    const bool odd = li & 1;

    // two values of this lane (rows ra and ra+1, same channel) -> packed channel pairs: even lanes write row ra, odd
    // lanes row ra+1, after swapping one (hi, lo) pair with the neighbouring lane (one DPP move per two values)
    auto put2 = [&](int ra, int n, float va, float vb) {
        const _Float16 ha = (_Float16)va, hb = (_Float16)vb;
        const _Float16 la = (_Float16)(va - (float)ha), lb = (_Float16)(vb - (float)hb);
        union { _Float16 h[2]; int i; } pa, pb, rc, wh, wl;
        pa.h[0] = ha; pa.h[1] = la;
        pb.h[0] = hb; pb.h[1] = lb;
        const int give = odd ? pa.i : pb.i;
        rc.i = __builtin_amdgcn_update_dpp(0, give, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false);
        if (odd) { wh.h[0] = rc.h[0]; wh.h[1] = hb; wl.h[0] = rc.h[1]; wl.h[1] = lb; }
        else { wh.h[0] = ha; wh.h[1] = rc.h[0]; wl.h[0] = la; wl.h[1] = rc.h[1]; }
        unsigned char* const d = stg + (ra + (odd ? 1 : 0)) * PITCH + (n * 16 + (li & ~1)) * 2;
        *reinterpret_cast<int*>(d) = wh.i;
        *reinterpret_cast<int*>(d + PLANE) = wl.i;
    };

    // staged rows [0, R) -> global; pixel_of(row) gives the NHWC pixel index or -1
    auto flush = [&](int R, auto pixel_of) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < (16 * UR + 63) / 64; ++k) {
            const int u = lane + 64 * k;
            const int row = u / UR, cu = u - row * UR;
            const int c0 = nblk * (NT * 16) + cu * 8;
            if (row < R && c0 < p.Cds) {
                const long pix = pixel_of(row);
                if (pix >= 0) {
                    const uint4 vh = *reinterpret_cast<const uint4*>(stg + row * PITCH + cu * 16);
                    const uint4 vl = *reinterpret_cast<const uint4*>(stg + row * PITCH + cu * 16 + PLANE);
                    *reinterpret_cast<uint4*>(p.dst_hi + pix * p.Cds + c0) = vh;
                    *reinterpret_cast<uint4*>(p.dst_lo + pix * p.Cds + c0) = vl;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staged rows are in registers before the next tile overwrites
    };

    if (p.pool) {
#pragma unroll
        for (int m = 0; m < kMT; m += 2) {
            const int t = wave * kMT + m;
            const int ig = t >> p.th_log2, ty = t & (TH - 1);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                // pooled pixel j = 2q + r/2 of the 8 this M-tile pair produces
                const float v0 = fmaxf(fmaxf(acc[m][n][0], acc[m][n][1]), fmaxf(acc[m + 1][n][0], acc[m + 1][n][1]));
                const float v1 = fmaxf(fmaxf(acc[m][n][2], acc[m][n][3]), fmaxf(acc[m + 1][n][2], acc[m + 1][n][3]));
                put2(2 * q, n, v0, v1);
            }
            flush(8, [&](int j) -> long {
                const int i = 2 * j;
                const int img = img0 + ig * p.nimg_m + (i >> p.twm_log2);
                if (img >= p.B) return -1;
                return (long)(img * p.outH + ((y0 + ty) >> 1)) * p.outW + ((x0 + (i & (TWm - 1))) >> 1);
            });
        }
    } else {
#pragma unroll
        for (int m = 0; m < kMT; ++m) {
            const int t = wave * kMT + m;
            const int ig = t >> p.th_log2, ty = t & (TH - 1);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                put2(4 * q, n, acc[m][n][0], acc[m][n][1]);
                put2(4 * q + 2, n, acc[m][n][2], acc[m][n][3]);
            }
            flush(16, [&](int i) -> long {
                const int img = img0 + ig * p.nimg_m + (i >> p.twm_log2);
                if (img >= p.B) return -1;
                return (long)(img * p.outH + (y0 + ty) * p.o_mul + ph.oy_off) * p.outW +
                       (x0 + (i & (TWm - 1))) * p.o_mul + ph.ox_off;
            });
        }
    }
}

template <int NT>
static hipError_t launch_h_nt(const HConvParams& p, hipStream_t stream) {
    const int img_groups = (p.B + p.imgs - 1) / p.imgs;
    dim3 grid((unsigned)(img_groups * p.tiles_y * p.tiles_x), (unsigned)p.nblocks, (unsigned)p.nphase);
    const size_t lds = (size_t)p.lds_bytes;
    const void* kern = reinterpret_cast<const void*>(conv_f16x3<NT>);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(conv_f16x3<NT>, grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_conv_f16(const HConvParams& p, hipStream_t stream) {
    switch (p.NT) {
        case 1: return launch_h_nt<1>(p, stream);
        case 2: return launch_h_nt<2>(p, stream);
        case 3: return launch_h_nt<3>(p, stream);
        case 4: return launch_h_nt<4>(p, stream);
        case 5: return launch_h_nt<5>(p, stream);
        case 6: return launch_h_nt<6>(p, stream);
        case 7: return launch_h_nt<7>(p, stream);
        case 8: return launch_h_nt<8>(p, stream);
        case 9: return launch_h_nt<9>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------------------
// fp32 NHWC [npix, C] -> (hi, lo) binary16 NHWC [npix, Cs], Cs = C rounded up to 8, pad channels zero, values scaled
// by `scale` (a power of two) first.  Feeds the first conv from umx_forward_tiles / the gather kernel.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) split_f32_kernel(const float* __restrict__ x, size_t npix, int C, int Cs, float scale,
                                                       _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
    const size_t total = npix * (size_t)Cs;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t px = e / Cs;
        const int c = (int)(e - px * Cs);
        const float v = c < C ? x[px * C + c] * scale : 0.f;
        const _Float16 h = (_Float16)v;
        hi[e] = h;
        lo[e] = (_Float16)(v - (float)h);
    }
}

hipError_t launch_split_f32(const float* x, size_t npix, int C, int Cs, float scale, _Float16* hi, _Float16* lo,
                            hipStream_t stream) {
    if (npix == 0) return hipSuccess;
    const size_t total = npix * (size_t)Cs;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 256 * 16 ? (total + 255) / 256 : 256 * 16);
    hipLaunchKernelGGL(split_f32_kernel, dim3(blocks), dim3(256), 0, stream, x, npix, C, Cs, scale, hi, lo);
    return hipGetLastError();
}

}  // namespace umx
