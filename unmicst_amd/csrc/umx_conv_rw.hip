// Register-resident-weight convolution for gfx950 (MI355X): the narrow full-resolution layers of the split-precision plan.
//
// conv_f16x3 (umx_conv_f16.hip) re-streams a layer's weights through LDS for every 256-pixel workgroup; for a layer such as
// the v2 top up-convolution (38 -> 36 channels, 3x3: reference UnMicst1-5.py:197-203 followed by the 1x1 head + BN + softmax
// of :212-222,236-237) that is 84 KB of weights per 62 KB of pixels, a barrier per weight stage and a serial
// load -> wait -> compute -> epilogue timeline per workgroup.  Here the roles are turned around:
//   * the WHOLE packed weight set of the layer (NK k-steps x NT N-tiles x (hi, lo) MFMA A-fragments, <= 336 VGPRs) is
//     loaded ONCE per wave into registers -- one wave per SIMD owns the 512-entry register file (launch bounds 256, 1);
//   * workgroups are persistent (one per CU) and walk the 16x16-pixel tiles of the layer; the only thing that moves is the
//     input halo of the NEXT tile, by LDS-DMA (buffer_load .. lds, out-of-image lanes come back as zeros = the padding),
//     into the other of two LDS tile buffers while the current tile computes: one barrier per tile, no weight traffic, no
//     LDS reads of weights, no per-stage waits;
//   * per 16-pixel M-tile a wave issues 2*NK ds_read_b128 (its pixel fragments) for 3*NK*NT MFMAs.
// Arithmetic, data layout of the activations, packed weight fragments, k-map and epilogue constants are those of
// conv_f16x3 (x*w = x_hi*w_hi + x_hi*w_lo + x_lo*w_hi on v_mfma_f32_16x16x32_f16, fp32 accumulate), so results agree
// with it to fp32 rounding of a different summation order inside a k-step only -- there is none: the k-step order and the
// order of the three products are identical, the outputs are bit-identical to conv_f16x3's.
#include <cstdlib>

#include "umx_kernels.h"

namespace umx {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {   // compile-time loop: the body sees its index as a constant expression
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}

__device__ __forceinline__ unsigned lds_offset(const void* p) {   // byte offset of a __shared__ address inside the LDS
    return (unsigned)(unsigned long)((__attribute__((address_space(3))) const void*)p);
}

#define UMX_BLDS16(rsrc, lptr, voff, soff) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lptr), 16, voff, soff, 0, 0)

template <int NT, int NK, int DIAG = 0>   // DIAG (timing-only builds): 1 no halo loads, 2 no epilogue
__global__ void __launch_bounds__(256, 1) conv_rw(const RwParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int MAXP = 12;   // halo pieces per wave and tile (the planner keeps ceil(ninst/4) <= MAXP)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4;    // which 8-wide k group of the 16x16x32 MFMA this lane feeds
    const int li = lane & 15;   // pixel (B operand) / output channel (A operand, C/D) inside the tile

    // ---- the layer's weights: MFMA A-fragments, k-step major, straight from the packed image into registers
    h8 Wh[NK][NT], Wl[NK][NT];
#pragma unroll
    for (int j = 0; j < NK; ++j)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const uint4* const w = p.w + ((size_t)(j * NT + n) * 2) * 64 + lane;
            Wh[j][n] = *reinterpret_cast<const h8*>(w);
            Wl[j][n] = *reinterpret_cast<const h8*>(w + 64);
        }
    // k-map: LDS byte offset of the (tap, octet) pair this lane group reads at k-step j, pixel column li folded in
    int kb[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) kb[j] = (int)p.kmap[j * 4 + q] * 16 + li * p.pix_bytes;

    // epilogue constants -> LDS, once: [pre_s | pre_b | post_s | post_b] x NT*16, then FOUR head-weight rows (rows >= head_K
    // zero, so that the head's dot products need no class-count branches), then the head's BN [scale x 8 | bias x 8]
    float* const ecl = reinterpret_cast<float*>(smem + p.ec_off);
    const int K = p.head_K;
    {
        const float* const src = reinterpret_cast<const float*>(p.econst);
        for (int i = tid; i < 8 * NT * 16 + 16; i += 256) {
            float v = 0.f;
            if (i < 4 * NT * 16) v = src[i];
            else if (i < 8 * NT * 16) { if (i < (4 + K) * NT * 16) v = src[i]; }
            else {
                v = src[(4 + K) * NT * 16 + (i - 8 * NT * 16)];
                if (i - 8 * NT * 16 < 8) v *= p.head_unscale;   // the head's BN scale absorbs the 2^-hs of its packed weights
            }
            ecl[i] = v;
        }
    }
    const float slope = p.act == ACT_RELU ? 0.f : p.act == ACT_LEAKY ? 0.2f : 1.f;
    h8 Hh[(NT + 1) / 2], Hl[(NT + 1) / 2];   // the 1x1 head as MFMA A-fragments (rows = classes)
#pragma unroll
    for (int s2 = 0; s2 < (NT + 1) / 2; ++s2) {
        Hh[s2] = *reinterpret_cast<const h8*>(p.head_frag + (s2 * 2 + 0) * 64 + lane);
        Hl[s2] = *reinterpret_cast<const h8*>(p.head_frag + (s2 * 2 + 1) * 64 + lane);
    }
    // ---- halo pieces: piece i = PP consecutive halo pixels x OCT octet columns; wave w owns pieces w, w+4, ..
    const int pl = (lane * p.inv_oct_q16) >> 16;   // lane / OCT
    const int kq = lane - pl * p.OCT;              // octet column of the LDS image
    const bool g1 = p.ngroups > 1 && kq >= p.goct[1];
    const int okq = kq - (g1 ? p.goct[1] : 0);     // octet inside its operand group
    const bool lane_ok = lane < p.nact && okq < (g1 ? p.noct[1] : p.noct[0]);
    const unsigned Cs2 = (unsigned)(g1 ? p.Cs[1] : p.Cs[0]) * 2u;   // bytes per pixel of one plane of this lane's source
    const size_t plane_elems = (size_t)p.B * p.H * p.W;
    auto act_rsrc = [&](const _Float16* base, int Cs) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(plane_elems * Cs * 2), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t r0h = act_rsrc(p.src_hi[0], p.Cs[0]), r0l = act_rsrc(p.src_lo[0], p.Cs[0]);
    const __amdgpu_buffer_rsrc_t r1h = act_rsrc(p.ngroups > 1 ? p.src_hi[1] : p.src_hi[0], p.ngroups > 1 ? p.Cs[1] : p.Cs[0]),
                                 r1l = act_rsrc(p.ngroups > 1 ? p.src_lo[1] : p.src_lo[0], p.ngroups > 1 ? p.Cs[1] : p.Cs[0]);

    auto issue_halo = [&](int tile, int buf) {
        const int tx_i = tile & ((1 << p.tx_log2) - 1);
        const int ty_i = (tile >> p.tx_log2) & ((1 << p.ty_log2) - 1);
        const int img = tile >> (p.tx_log2 + p.ty_log2);
        const int yb = ty_i * 16 + p.ymin, xb = tx_i * 16 + p.xmin;
        unsigned char* const dst = smem + buf * (2 * p.plane_bytes);
        // one pass per operand group: a piece's lanes of group 0 and of group 1 use different (wave-uniform) descriptors, and
        // an if / else over the lane's group would be if-converted into a per-lane descriptor select (a waterfall loop)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (g == 1 && p.ngroups < 2) break;
            const __amdgpu_buffer_rsrc_t rh = g ? r1h : r0h, rl = g ? r1l : r0l;
#pragma unroll
            for (int j = 0; j < MAXP; ++j) {
                const int i = wave + 4 * j;
                if (i < p.ninst) {                          // wave-uniform
                    const int px = i * p.PP + pl;           // halo pixel this lane fetches for piece i
                    if (lane_ok && px < p.nhalo && g1 == (g == 1)) {   // lanes of this piece that belong to group g
                        const int hy = (int)(((float)px + 0.5f) * p.inv_hw);   // px < 1024: exact after truncation
                        const int gy = yb + hy, gx = xb + (px - hy * p.hw);
                        const bool inside = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                        const int voff = inside ? (int)(((unsigned)(img * p.H + gy) * (unsigned)p.W + (unsigned)gx) * Cs2) + okq * 16
                                                : 0x7fffffff;   // outside the descriptor: the DMA writes zeros
                        unsigned char* const d = dst + i * p.piece_bytes;
                        UMX_BLDS16(rh, d, voff, 0);
                        UMX_BLDS16(rl, d + p.plane_bytes, voff, 0);
                    }
                }
            }
        }
    };

    const float4* const ec4 = reinterpret_cast<const float4*>(ecl);
    const float* const hsb = ecl + 8 * (NT * 16);         // [scale x 8 | bias x 8] of the head's BN
    const int rowpitch = p.hw * p.pix_bytes;

    int tile = blockIdx.x;
    if (tile < p.ntiles) issue_halo(tile, 0);
    for (int it = 0; tile < p.ntiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        // this tile's halo (issued one tile ago) has landed, for every wave; every wave is done reading the other buffer
        // (vector-memory operations retire in issue order: the one store of the previous tile's probabilities is the youngest
        // and may stay in flight -- waiting for it would put a store round trip in front of every tile)
        if (K == 3 && it > 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tile + (int)gridDim.x < p.ntiles && !(DIAG & 1)) issue_halo(tile + gridDim.x, buf ^ 1);
        const unsigned char* const hb = smem + buf * (2 * p.plane_bytes);
        const int tx_i = tile & ((1 << p.tx_log2) - 1);
        const int ty_i = (tile >> p.tx_log2) & ((1 << p.ty_log2) - 1);
        const int img = tile >> (p.tx_log2 + p.ty_log2);

#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const unsigned char* const rb = hb + (wave * 4 + m) * rowpitch;   // halo row of this M-tile's taps (dy = ymin)
            f32x4 acc[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // Pixel fragments are read kAPre k-steps ahead of their MFMAs.  hipcc sinks plain LDS loads down to their first use
            // (one wave per SIMD: every exposed ds_read latency idles the matrix pipe), so the reads are issued by hand
            // and retired with counted waits: LDS returns in order, "at most 2*d reads outstanding" means the fragments
            // of k-step j have arrived while those of the next d k-steps are still in flight.
            constexpr int kAPre = 3;
            h8 ahq[kAPre + 1], alq[kAPre + 1];
            const unsigned rbo = lds_offset(rb);
#define UMX_RW_READ(slot, jj)                                                                                        \
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3"                                                        \
                 : "=v"(ahq[slot]), "=v"(alq[slot]) : "v"(rbo + (unsigned)kb[jj]), "v"(rbo + (unsigned)kb[jj] + (unsigned)p.plane_bytes))
#pragma unroll
            for (int j = 0; j < kAPre && j < NK; ++j) UMX_RW_READ(j, j);
            static_for<0, NK>([&](auto J) {
                constexpr int j = decltype(J)::value;
                constexpr int R = kAPre + 1;
                if constexpr (j + kAPre < NK) UMX_RW_READ((j + kAPre) % R, j + kAPre);
                constexpr int ahead = (NK - 1 - j) < kAPre ? (NK - 1 - j) : kAPre;   // k-steps whose reads may stay in flight
                // the wait names the registers it releases, so that nothing touches them before it
                if constexpr (ahead == 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(ahq[j % R]), "+v"(alq[j % R]));
                else if constexpr (ahead == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ahq[j % R]), "+v"(alq[j % R]));
                else if constexpr (ahead == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ahq[j % R]), "+v"(alq[j % R]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ahq[j % R]), "+v"(alq[j % R]));
                const h8 ah = ahq[j % R], al = alq[j % R];
                static_for<0, NT>([&](auto N) {
                    constexpr int n = decltype(N)::value;
                    // weights are the A operand (rows = output channels), pixels the B operand: D[channel][pixel]
                    f32x4 c = acc[n];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wh[j][n], al, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wl[j][n], ah, c, 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wh[j][n], ah, c, 0, 0, 0);
                });
            });
#undef UMX_RW_READ

            // ---- epilogue of this M-tile: BN affine -> activation -> affine (reference UnMicst1-5.py:199-203 / UnMicst.py:157-161),
            // then the fused 1x1 head (UnMicst1-5.py:212-222 / UnMicst.py:167-171) ON THE MATRIX CORES: the lane's 4 channels of
            // N-tiles (2s, 2s+1) are exactly the 8 k-elements lane group q feeds at head k-step s, so the activations go
            // from the accumulators into B-fragments without leaving the lane; D[class][pixel] lands with all classes of
            // pixel li in lanes 0..15, where the softmax runs without any cross-lane traffic.
            if constexpr (DIAG & 2) { asm volatile("" :: "v"(acc[0]), "v"(acc[NT - 1])); continue; }
            asm volatile("" ::: "memory");   // the constants are re-read from LDS here: hoisted, they cost registers and speed
            float v[NT][4];
            unsigned vmax = 0u;   // max |v| as a bit pattern: NaN and infinity order above every finite value
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float4 ps = ec4[0 * NT * 4 + n * 4 + q], pb = ec4[1 * NT * 4 + n * 4 + q];
                const float psa[4] = {ps.x, ps.y, ps.z, ps.w}, pba[4] = {pb.x, pb.y, pb.z, pb.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = acc[n][r] * psa[r] + pba[r];
                    t = fmaxf(t, t * slope);   // slope 0: ReLU, 0.2: LeakyReLU, 1: none
                    if (p.post_affine) t = t * ecl[2 * NT * 16 + n * 16 + 4 * q + r] + ecl[3 * NT * 16 + n * 16 + 4 * q + r];
                    vmax = max(vmax, __float_as_uint(t) & 0x7fffffffu);
                    v[n][r] = t;
                }
            }
            const bool big = vmax >= 0x476a6000u;   // |v| >= 60000, infinity or NaN
            f32x4 lg = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < (NT + 1) / 2; ++s2) {
                h8 bh, bl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int nt = 2 * s2 + (j >> 2);
                    const float t = nt < NT ? v[nt < NT ? nt : 0][j & 3] : 0.f;
                    bh[j] = (_Float16)t;
                    bl[j] = (_Float16)(t - (float)bh[j]);
                }
                lg = __builtin_amdgcn_mfma_f32_16x16x32_f16(Hh[s2], bl, lg, 0, 0, 0);
                lg = __builtin_amdgcn_mfma_f32_16x16x32_f16(Hl[s2], bh, lg, 0, 0, 0);
                lg = __builtin_amdgcn_mfma_f32_16x16x32_f16(Hh[s2], bh, lg, 0, 0, 0);
            }
            if (big) atomicOr(p.overflow_flag, 1);   // binary16 range exceeded in front of the head: the host reports it
            if (q == 0) {   // lanes 0..15: classes 0..3 of pixel li in lg[0..3]
                float e[4], mx = -INFINITY;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    e[k] = k < K ? lg[k] * hsb[k] + hsb[8 + k] : -INFINITY;
                    mx = fmaxf(mx, e[k]);
                }
                float sum = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    e[k] = __expf(e[k] - mx);
                    sum += e[k];
                }
                const float inv = __builtin_amdgcn_rcpf(sum);
                float* const d = p.probs + ((size_t)(img * p.H + ty_i * 16 + wave * 4 + m) * p.W + tx_i * 16 + li) * K;
                if (K == 3) {   // one 12-byte store per pixel instead of three 4-byte ones
                    struct __attribute__((packed, aligned(4))) F3 { float a, b, c; };
                    *reinterpret_cast<F3*>(d) = F3{e[0] * inv, e[1] * inv, e[2] * inv};
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < K) d[k] = e[k] * inv;
                }
            }
        }
    }
}

template <int NT, int NK, int DIAG = 0>
static hipError_t launch_rw_nt(const RwParams& p, int ncu, hipStream_t stream) {
    const void* kern = reinterpret_cast<const void*>(conv_rw<NT, NK, DIAG>);
    if (p.lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, p.lds_bytes);
        if (e != hipSuccess) return e;
    }
    const int grid = p.ntiles < ncu ? p.ntiles : ncu;   // persistent: one workgroup per CU walks the tiles
    hipLaunchKernelGGL((conv_rw<NT, NK, DIAG>), dim3((unsigned)grid), dim3(256), (size_t)p.lds_bytes, stream, p);
    return hipGetLastError();
}

bool conv_rw_supported(int NT, int NK) { return NT == 3 && (NK == 12 || NK == 14); }

hipError_t launch_conv_rw(const RwParams& p, int NT, int ncu, hipStream_t stream) {
    if (p.ntiles <= 0) return hipSuccess;
    if (NT == 3 && p.nk == 12) {
        static const int diag = getenv("UMX_RW_DIAG") ? atoi(getenv("UMX_RW_DIAG")) : 0;
        switch (diag) {
            case 1: return launch_rw_nt<3, 12, 1>(p, ncu, stream);
            case 2: return launch_rw_nt<3, 12, 2>(p, ncu, stream);
            case 3: return launch_rw_nt<3, 12, 3>(p, ncu, stream);
            default: return launch_rw_nt<3, 12>(p, ncu, stream);
        }
    }
    if (NT == 3 && p.nk == 14) return launch_rw_nt<3, 14>(p, ncu, stream);
    return hipErrorInvalidValue;
}

}  // namespace umx
