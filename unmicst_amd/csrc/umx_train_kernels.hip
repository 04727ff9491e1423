// Kernels of the training step of libumx (gfx950 only): weight repacking, batch-statistics BN forward/backward,
// activation / dropout / max-pool forward and backward, the top layer with the weighted cross-entropy, the weight
// gradient of a convolution on the fp32 matrix cores, and the optimiser.  The convolutions themselves (forward,
// input gradient, transposed forward, transposed input gradient on a space-to-depth tensor) run on conv_mfma_f32
// (umx_kernels.hip) with operands packed by pack_weights_kernel.
//
// Every reduction has a fixed summation order (per-thread serial sums, in-block sums in thread order, partials summed
// in block order, fp64): a step is bit-reproducible from run to run.  No floating-point atomics anywhere.
#include "umx_kernels.h"

#include <algorithm>
#include <cstdlib>

#pragma clang fp contract(off)

namespace umx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ weight packing
__global__ void __launch_bounds__(256) pack_weights_kernel(const PackDesc* __restrict__ descs) {
    const PackDesc& d = descs[blockIdx.y];
    const size_t n = (size_t)d.ntaps * d.Cp * d.Np;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int nn = (int)(e % d.Np);
        const size_t r = e / d.Np;
        const int c = (int)(r % d.Cp);
        const int t = (int)(r / d.Cp);
        float v = 0.f;
        if (c < d.C && nn < d.N) {
            const int par = c / d.Cblk, cc = c - par * d.Cblk;
            const int m = d.mtap[t * d.npar + par];
            if (m >= 0) {
                const size_t idx = d.transpose ? ((size_t)m * d.d2 + d.c_off + nn) * d.d3 + cc
                                               : ((size_t)m * d.d2 + d.c_off + cc) * d.d3 + nn;
                v = d.w[idx];
                if (d.w2) v += d.w2[idx];
            }
        }
        d.dst[e] = v;
    }
}

hipError_t launch_pack_weights(const PackDesc* descs_dev, int ndesc, size_t max_elems, hipStream_t stream) {
    if (ndesc <= 0) return hipSuccess;
    const unsigned gx = (unsigned)std::min<size_t>(512, (max_elems + 255) / 256);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(gx ? gx : 1, (unsigned)ndesc), dim3(256), 0, stream, descs_dev);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ channel layout
// An NHWC tensor is [N rows][C channels].  A block of Cb*k threads (Cb = min(C, 256), k = 256 / Cb) walks k rows at
// a time: thread (r, c) always sees channel c, so per-channel sums live in registers and the loads are contiguous.
struct ChanLayout { int Cb, k, threads, cblocks; };
static inline ChanLayout chan_layout(int C) {
    ChanLayout l;
    l.Cb = std::min(C, 256);
    l.k = std::max(1, 256 / l.Cb);
    l.threads = l.Cb * l.k;
    l.cblocks = (C + 255) / 256;
    return l;
}

int chan_blocks(size_t N, int C) {
    const ChanLayout l = chan_layout(C);
    // (a handful of rows per block: the deep layers have 512 - 2048 rows, and 17 blocks walking 30 rows each left their
    //  activation-backward kernel latency-bound at 158 us for 1.2 MB -- round 4 time line)
    const size_t want = N / ((size_t)l.k * 4) + 1;
    const size_t cap = std::max(64, 1024 / l.cblocks);
    return (int)std::min<size_t>(cap, std::max<size_t>(1, want));
}

__global__ void __launch_bounds__(256) chan_stats_kernel(const float* __restrict__ x, size_t N, int C, int Cb, int k,
                                                         double* __restrict__ part) {
    __shared__ double sm[2][256];
    const int tid = threadIdx.x;
    const int r = tid / Cb, cl = tid - r * Cb;
    const int c = cl + blockIdx.y * 256;
    const size_t rpb = (N + gridDim.x - 1) / gridDim.x;
    const size_t r0 = (size_t)blockIdx.x * rpb, r1 = std::min(N, r0 + rpb);
    double s = 0.0, q = 0.0;
    if (c < C)
        for (size_t row = r0 + r; row < r1; row += k) {
            const double v = (double)x[row * C + c];
            s += v;
            q += v * v;
        }
    sm[0][tid] = s;
    sm[1][tid] = q;
    __syncthreads();
    if (r == 0 && c < C) {
        for (int j = 1; j < k; ++j) { s += sm[0][j * Cb + cl]; q += sm[1][j * Cb + cl]; }
        part[((size_t)blockIdx.x * 2 + 0) * C + c] = s;
        part[((size_t)blockIdx.x * 2 + 1) * C + c] = q;
    }
}

// (C % 4 == 0: four channels per thread, 16-byte loads; same partial-sum layout)
__global__ void __launch_bounds__(256) chan_stats_v4_kernel(const float* __restrict__ x, unsigned N, int C, int Qb, int R,
                                                            double* __restrict__ part) {
    __shared__ double sm[2][256][4];
    const int tid = threadIdx.x;
    const int r = tid / Qb, ql = tid - r * Qb;
    const int Q = C >> 2, q = ql + blockIdx.y * 256;
    const unsigned rpb = (N + gridDim.x - 1) / gridDim.x;
    const unsigned r0 = blockIdx.x * rpb, r1 = min(N, r0 + rpb);
    double s[4] = {0.0, 0.0, 0.0, 0.0}, w[4] = {0.0, 0.0, 0.0, 0.0};
    if (r < R && q < Q)
        for (unsigned row = r0 + r; row < r1; row += R) {
            const float4 v4 = reinterpret_cast<const float4*>(x)[(size_t)row * Q + q];
            const double v[4] = {(double)v4.x, (double)v4.y, (double)v4.z, (double)v4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { s[k] += v[k]; w[k] += v[k] * v[k]; }
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) { sm[0][tid][k] = s[k]; sm[1][tid][k] = w[k]; }
    __syncthreads();
    if (r == 0 && q < Q) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double t1 = s[k], t2 = w[k];
            for (int j = 1; j < R; ++j) { t1 += sm[0][j * Qb + ql][k]; t2 += sm[1][j * Qb + ql][k]; }
            part[((size_t)blockIdx.x * 2 + 0) * C + 4 * q + k] = t1;
            part[((size_t)blockIdx.x * 2 + 1) * C + 4 * q + k] = t2;
        }
    }
}

hipError_t launch_chan_stats(const float* x, size_t N, int C, double* part, int nblk, hipStream_t stream) {
    if (C % 4 == 0 && N < 0x7fffffffull) {
        const int Q = C / 4, Qb = std::min(Q, 256), R = std::max(1, 256 / Qb), cbl = (Q + 255) / 256;
        hipLaunchKernelGGL(chan_stats_v4_kernel, dim3((unsigned)nblk, (unsigned)cbl), dim3(256), 0, stream, x, (unsigned)N, C, Qb, R, part);
        return hipGetLastError();
    }
    const ChanLayout l = chan_layout(C);
    hipLaunchKernelGGL(chan_stats_kernel, dim3((unsigned)nblk, (unsigned)l.cblocks), dim3((unsigned)l.threads), 0, stream,
                       x, N, C, l.Cb, l.k, part);
    return hipGetLastError();
}

// fixed-order sum of the per-block partials of one channel by a 64-lane block: lane l adds partials l, l+64, ... in
// order, then a binary tree over the lanes
__device__ __forceinline__ void block64_sum2(double& a, double& b, double (*sm)[64]) {
    const int l = threadIdx.x;
    sm[0][l] = a;
    sm[1][l] = b;
    __syncthreads();
    for (int st = 32; st > 0; st >>= 1) {
        if (l < st) { sm[0][l] += sm[0][l + st]; sm[1][l] += sm[1][l + st]; }
        __syncthreads();
    }
    a = sm[0][0];
    b = sm[1][0];
}

__global__ void __launch_bounds__(64) bn_finalize_kernel(const double* __restrict__ part, int nblk, double N, int C,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* mov_mean, float* mov_var, float momentum,
                                                         float* __restrict__ stat, unsigned* smax) {
    __shared__ double sm[2][64];
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) {
        s += part[((size_t)b * 2 + 0) * C + c];
        q += part[((size_t)b * 2 + 1) * C + c];
    }
    block64_sum2(s, q, sm);
    if (threadIdx.x != 0) return;
    const double mean = s / N;
    double var = q / N - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + 1e-3);   // eps of tf.layers.batch_normalization in the reference graphs
    const double scale = (double)gamma[c] * rstd;
    stat[c] = (float)mean;
    stat[C + c] = (float)rstd;
    stat[2 * C + c] = (float)scale;
    stat[3 * C + c] = (float)((double)beta[c] - mean * scale);
    if (smax) atomicMax(smax, __float_as_uint(fabsf((float)scale)));   // max_c |gamma * rstd|: bounds the BN input gradient (bn_bwd_apply)
    const double mom = (double)momentum;
    const double unbiased = N > 1.0 ? var * (N / (N - 1.0)) : var;
    mov_mean[c] = (float)((double)mov_mean[c] * mom + mean * (1.0 - mom));
    mov_var[c] = (float)((double)mov_var[c] * mom + unbiased * (1.0 - mom));
}

hipError_t launch_bn_finalize(const double* part, int nblk, size_t N, int C, const float* gamma, const float* beta,
                              float* mov_mean, float* mov_var, float momentum, float* stat, unsigned* smax, hipStream_t stream) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)C), dim3(64), 0, stream, part, nblk, (double)N, C, gamma, beta,
                       mov_mean, mov_var, momentum, stat, smax);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) bn_stat_from_moving_kernel(int C, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta,
                                                                  const float* __restrict__ mov_mean,
                                                                  const float* __restrict__ mov_var,
                                                                  float* __restrict__ stat) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double rstd = 1.0 / sqrt((double)mov_var[c] + 1e-3);
    const double scale = (double)gamma[c] * rstd;
    stat[c] = mov_mean[c];
    stat[C + c] = (float)rstd;
    stat[2 * C + c] = (float)scale;
    stat[3 * C + c] = (float)((double)beta[c] - (double)mov_mean[c] * scale);
}

hipError_t launch_bn_stat_from_moving(int C, const float* gamma, const float* beta, const float* mov_mean,
                                      const float* mov_var, float* stat, hipStream_t stream) {
    hipLaunchKernelGGL(bn_stat_from_moving_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream, C, gamma, beta,
                       mov_mean, mov_var, stat);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ dropout stream
// == oracle/train_oracle.py dropout_mask: u = top 24 bits of mix(key ^ flat NHWC index); keep iff u >= rate
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float drop_mul(unsigned long long key, size_t idx, float rate, float keep_scale) {
    if (rate <= 0.f) return 1.f;
    const unsigned long long h = mix64((unsigned long long)idx ^ key);
    const float u = (float)(unsigned)(h >> 40) * (1.0f / 16777216.0f);
    return u >= rate ? keep_scale : 0.f;
}
__device__ __forceinline__ float act_of(float v, int act) {
    if (act == ACT_LEAKY) return v > 0.f ? v : 0.2f * v;
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    return v;
}
__device__ __forceinline__ float dact_of(float v, int act) {
    if (act == ACT_LEAKY) return v > 0.f ? 1.f : 0.2f;
    if (act == ACT_RELU) return v > 0.f ? 1.f : 0.f;
    return 1.f;
}

// max |v| over a block -> one atomicMax on the tensor's word (uint order == float order for non-negative floats;
// integer max is associative: the result does not depend on the order of the blocks)
__device__ __forceinline__ void block_absmax_to(unsigned* dst, float v) {
    __shared__ unsigned smax[4];
    unsigned u = __float_as_uint(fabsf(v));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, off));
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = u;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned m = max(max(smax[0], smax[1]), max(smax[2], smax[3]));
        // most blocks cannot raise the maximum once a few have reported: a plain (possibly stale, never too large) read
        // keeps them off the atomic unit; the atomic itself stays the arbiter
        if (m > *reinterpret_cast<volatile unsigned*>(dst)) atomicMax(dst, m);
    }
}

// (grid-stride; the block's max |output| goes to the tensor's max word with at most one atomic)
__global__ void __launch_bounds__(256) act_fwd_kernel(const ActParams a, float* __restrict__ out, size_t nout,
                                                      unsigned* omax, _Float16* __restrict__ hi, _Float16* __restrict__ lo, int Cs,
                                                      int* __restrict__ overflow) {
    const int C = a.C;
    bool bad = false;   // (hi != NULL: the output also as the (hi, lo) planes the next convolution reads, unscaled)
    const float ks = 1.0f / (1.0f - a.drop_rate);
    const int OW = a.W >> 1, OH = a.H >> 1;
    float mx = 0.f;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nout; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const float sc = a.stat[2 * C + c], sh = a.stat[3 * C + c];
        float y;
        if (!a.pool) {
            const float v = a.z[e] * sc + sh;
            y = act_of(v, a.act) * drop_mul(a.drop_key, e, a.drop_rate, ks);
        } else {
            size_t r = e / C;
            const int ox = (int)(r % OW); r /= OW;
            const int oy = (int)(r % OH);
            const int b = (int)(r / OH);
            y = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t idx = (((size_t)b * a.H + 2 * oy + (j >> 1)) * a.W + 2 * ox + (j & 1)) * C + c;
                const float v = a.z[idx] * sc + sh;
                const float yj = act_of(v, a.act) * drop_mul(a.drop_key, idx, a.drop_rate, ks);
                if (j == 0 || yj > y) y = yj;
            }
        }
        out[e] = y;
        mx = fmaxf(mx, fabsf(y));
        if (hi) {
            const size_t at = (e / C) * Cs + c;
            bad = bad || !(fabsf(y) < 60000.f);
            const _Float16 h = (_Float16)y;
            hi[at] = h;
            lo[at] = (_Float16)(y - (float)h);
        }
    }
    if (bad) atomicOr(overflow, 1);
    if (omax) block_absmax_to(omax, mx);
}

// (C % 4 == 0: four channels of one output pixel per thread -- 16-byte accesses, 32-bit index arithmetic)
__global__ void __launch_bounds__(256) act_fwd_v4_kernel(const ActParams a, float* __restrict__ out, unsigned nq, unsigned* omax,
                                                         _Float16* __restrict__ hi, _Float16* __restrict__ lo, int Cs,
                                                         int* __restrict__ overflow) {
    const int C = a.C;
    const unsigned Q = (unsigned)C >> 2;
    const float ks = 1.0f / (1.0f - a.drop_rate);
    const unsigned OW = a.W >> 1, OH = a.H >> 1;
    float mx = 0.f;
    bool bad = false;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < nq; i += gridDim.x * 256) {
        const unsigned row = i / Q, q = i - row * Q;
        const int c = 4 * (int)q;
        const float4 sc4 = *reinterpret_cast<const float4*>(a.stat + 2 * C + c), sh4 = *reinterpret_cast<const float4*>(a.stat + 3 * C + c);
        const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
        float y[4];
        if (!a.pool) {
            const float4 z4 = reinterpret_cast<const float4*>(a.z)[i];
            const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
            const size_t e = (size_t)row * C + c;
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = act_of(zz[k] * sc[k] + sh[k], a.act) * drop_mul(a.drop_key, e + k, a.drop_rate, ks);
        } else {
            const unsigned ox = row % OW, t = row / OW, oy = t % OH, b = t / OH;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t px = ((size_t)b * a.H + 2 * oy + (j >> 1)) * a.W + 2 * ox + (j & 1);
                const float4 z4 = reinterpret_cast<const float4*>(a.z)[px * Q + q];
                const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float yj = act_of(zz[k] * sc[k] + sh[k], a.act) * drop_mul(a.drop_key, px * C + c + k, a.drop_rate, ks);
                    if (j == 0 || yj > y[k]) y[k] = yj;
                }
            }
        }
        reinterpret_cast<float4*>(out)[i] = make_float4(y[0], y[1], y[2], y[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) mx = fmaxf(mx, fabsf(y[k]));
        if (hi) {
            union { _Float16 h[4]; uint2 u; } va, vb;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bad = bad || !(fabsf(y[k]) < 60000.f);
                va.h[k] = (_Float16)y[k];
                vb.h[k] = (_Float16)(y[k] - (float)va.h[k]);
            }
            const size_t at = (size_t)row * Cs + c;
            *reinterpret_cast<uint2*>(hi + at) = va.u;
            *reinterpret_cast<uint2*>(lo + at) = vb.u;
        }
    }
    if (bad) atomicOr(overflow, 1);
    if (omax) block_absmax_to(omax, mx);
}

hipError_t launch_act_fwd(const ActParams& a, float* out, unsigned* omax, _Float16* hi, _Float16* lo, int Cs, int* overflow,
                          hipStream_t stream) {
    const size_t nout = (size_t)a.B * (a.pool ? a.H / 2 : a.H) * (a.pool ? a.W / 2 : a.W) * a.C;
    if (a.C % 4 == 0 && (!hi || Cs % 4 == 0) && nout < 0xfffffff0ull) {
        const unsigned nq = (unsigned)(nout / 4);
        const unsigned blocks = std::min<unsigned>(4096, (nq + 255) / 256);
        hipLaunchKernelGGL(act_fwd_v4_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, a, out, nq, omax, hi, lo, Cs, overflow);
        return hipGetLastError();
    }
    const unsigned blocks = (unsigned)std::min<size_t>(2048, (nout + 255) / 256);
    hipLaunchKernelGGL(act_fwd_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, a, out, nout, omax, hi, lo, Cs, overflow);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, size_t n, unsigned* omax) {
    float mx = 0.f;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) mx = fmaxf(mx, fabsf(x[e]));
    block_absmax_to(omax, mx);
}

hipError_t launch_absmax(const float* x, size_t n, unsigned* omax, hipStream_t stream) {
    const unsigned blocks = (unsigned)std::min<size_t>(4096, (n + 255) / 256);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, x, n, omax);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Forward / input-gradient convolutions on conv_f16x3 (round 4).  The filters change every step: this kernel gathers them into
// the LDS-image order the planner laid out once (HWRef, umx_internal.h) -- straight from the master tensors where the trainer
// could rewrite the planner's references into master coordinates (estride != 0), else from the fp32 operands [tap][Cp][Np]
// pack_weights_kernel rebuilds -- scaled by the power of two that puts the layer's largest |w| of the START of training in
// [2^10, 2^11) -- 32 x headroom; a weight that outgrows binary16 raises the range flag -- as (hi, lo) binary16 pairs.
// One thread per 16-byte unit; consecutive lanes read consecutive output channels of the fp32 operand (coalesced per element).
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) repack_f16x3_kernel(const RepackDesc* __restrict__ descs, int* __restrict__ overflow) {
    const RepackDesc d = descs[blockIdx.y];
    bool bad = false;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < d.n; i += gridDim.x * 256) {
        const HWRefDev r = d.refs[i];
        const float* const src = d.arr[r.arr] + r.base;
        const float* const src2 = d.arr2[r.arr] ? d.arr2[r.arr] + r.base : nullptr;
        const int es = d.estride[r.arr] ? d.estride[r.arr] : d.stride;
        union { _Float16 h[8]; uint4 u; } vh, vl;
        vh.u = make_uint4(0, 0, 0, 0);
        vl.u = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (e < (int)r.nvalid) {
                float w = src[(size_t)e * es];
                if (src2) w += src2[(size_t)e * es];   // (the sum pack_weights_kernel makes, same order)
                const float v = w * d.scale;
                bad = bad || !(fabsf(v) < 60000.f);
                vh.h[e] = (_Float16)v;
                vl.h[e] = (_Float16)(v - (float)vh.h[e]);
            }
        }
        d.slab[r.dst] = vh.u;
        d.slab[r.dst + 64] = vl.u;
    }
    if (bad) atomicOr(overflow, 1);
}

hipError_t launch_repack_f16x3(const RepackDesc* descs_dev, int ndesc, int max_n, int* overflow, hipStream_t stream) {
    if (ndesc <= 0) return hipSuccess;
    const unsigned bx = (unsigned)std::min(512, (max_n + 255) / 256);
    hipLaunchKernelGGL(repack_f16x3_kernel, dim3(bx ? bx : 1, (unsigned)ndesc), dim3(256), 0, stream, descs_dev, overflow);
    return hipGetLastError();
}

// fp32 NHWC [npix, C] -> (hi, lo) binary16 NHWC [npix, Cs] (Cs = C rounded up to 8, pad channels zero) for conv_f16x3.  `maxw`
// (nullable): the tensor's max |v| as float bits, written by its producer -- a gradient tensor is scaled by the power of two that
// puts that maximum in [2^11, 2^12) (gradients would flush to zero otherwise) and the inverse factor goes to *inv_scale for the
// consuming convolution's epilogue (HConvParams::dyn); without it the tensor is stored as it is.  Non-finite or out-of-range
// values raise the flag.
__global__ void __launch_bounds__(256) split_dyn_kernel(const float* __restrict__ x, size_t npix, int C, int Cs,
                                                       const unsigned* __restrict__ maxw, float* __restrict__ inv_scale,
                                                       _Float16* __restrict__ hi, _Float16* __restrict__ lo, int* __restrict__ overflow,
                                                       unsigned* __restrict__ omax) {
    float scale = 1.f, mx = 0.f;   // (omax: the tensor's max |v| for the weight-gradient kernel, one integer atomicMax per block)
    if (maxw) {
        const unsigned mb = *maxw;
        const int e = (int)((mb >> 23) & 0xFFu);
        if (mb != 0u && e > 0 && e < 255) {
            const int sh = max(-100, min(100, 11 - (e - 127)));
            scale = __uint_as_float((unsigned)(127 + sh) << 23);
        }
        if (inv_scale && blockIdx.x == 0 && threadIdx.x == 0) *inv_scale = 1.f / scale;
    }
    const size_t total = npix * (size_t)Cs;
    bool bad = false;
    if ((C & 3) == 0 && Cs == C) {   // whole float4 / 8-byte units
        const size_t n4 = total / 4;
        const float4* const x4 = reinterpret_cast<const float4*>(x);
        uint2* const h2 = reinterpret_cast<uint2*>(hi);
        uint2* const l2 = reinterpret_cast<uint2*>(lo);
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
            const float4 v = x4[i];
            const float f[4] = {v.x * scale, v.y * scale, v.z * scale, v.w * scale};
            union { _Float16 h[4]; uint2 u; } a, b;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mx = fmaxf(mx, fabsf(f[k]));
                bad = bad || !(fabsf(f[k]) < 60000.f);
                a.h[k] = (_Float16)f[k];
                b.h[k] = (_Float16)(f[k] - (float)a.h[k]);
            }
            h2[i] = a.u;
            l2[i] = b.u;
        }
    } else {
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
            const size_t px = e / Cs;
            const int c = (int)(e - px * Cs);
            const float v = c < C ? x[px * C + c] * scale : 0.f;
            mx = fmaxf(mx, fabsf(v));
            bad = bad || !(fabsf(v) < 60000.f);
            const _Float16 h = (_Float16)v;
            hi[e] = h;
            lo[e] = (_Float16)(v - (float)h);
        }
    }
    if (bad) atomicOr(overflow, 1);
    if (omax) block_absmax_to(omax, mx);   // (only asked for unscaled tensors: scale == 1)
}

hipError_t launch_split_dyn(const float* x, size_t npix, int C, int Cs, const unsigned* maxw, float* inv_scale, _Float16* hi,
                            _Float16* lo, int* overflow, unsigned* omax, hipStream_t stream) {
    if (npix == 0) return hipSuccess;
    const size_t total = npix * (size_t)Cs;
    const unsigned blocks = (unsigned)std::min<size_t>(4096, (total / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(split_dyn_kernel, dim3(blocks), dim3(256), 0, stream, x, npix, C, Cs, maxw, inv_scale, hi, lo, overflow, omax);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) act_bwd_kernel(const ActParams a, const float* __restrict__ dy0,
                                                      const float* __restrict__ dy1, float* __restrict__ g, size_t Nrows,
                                                      int Cb, int k, double* __restrict__ part, unsigned* gx) {
    __shared__ double sm[2][256];
    __shared__ unsigned smx[2][256];
    float mg = 0.f, mxh = 0.f;   // max |g| and max |xhat| seen by this thread: with max |gamma rstd| they bound |dz| (bn_bwd_apply)
    const int tid = threadIdx.x;
    const int r = tid / Cb, cl = tid - r * Cb;
    const int C = a.C;
    const int c = cl + blockIdx.y * 256;
    const size_t rpb = (Nrows + gridDim.x - 1) / gridDim.x;
    const size_t r0 = (size_t)blockIdx.x * rpb, r1 = std::min(Nrows, r0 + rpb);
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        const float mean = a.stat[c], rstd = a.stat[C + c], sc = a.stat[2 * C + c], sh = a.stat[3 * C + c];
        const float ks = 1.0f / (1.0f - a.drop_rate);
        const int OW = a.W >> 1, OH = a.H >> 1;
        for (size_t row = r0 + r; row < r1; row += k) {
            const size_t e = row * C + c;
            float d = dy0[e];
            if (dy1) d += dy1[e];
            if (!a.pool) {
                const float zz = a.z[e];
                const float v = zz * sc + sh;
                const float gg = d * drop_mul(a.drop_key, e, a.drop_rate, ks) * dact_of(v, a.act);
                g[e] = gg;
                const float xh = (zz - mean) * rstd;
                s1 += (double)gg;
                s2 += (double)gg * (double)xh;
                mg = fmaxf(mg, fabsf(gg));
                mxh = fmaxf(mxh, fabsf(xh));
            } else {
                size_t q = row;
                const int ox = (int)(q % OW); q /= OW;
                const int oy = (int)(q % OH);
                const int b = (int)(q / OH);
                size_t idx[4];
                float zz[4], dm[4], vv[4];
                int arg = 0;
                float best = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    idx[j] = (((size_t)b * a.H + 2 * oy + (j >> 1)) * a.W + 2 * ox + (j & 1)) * C + c;
                    zz[j] = a.z[idx[j]];
                    vv[j] = zz[j] * sc + sh;
                    dm[j] = drop_mul(a.drop_key, idx[j], a.drop_rate, ks);
                    const float y = act_of(vv[j], a.act) * dm[j];
                    if (j == 0 || y > best) { best = y; arg = j; }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gg = j == arg ? d * dm[j] * dact_of(vv[j], a.act) : 0.f;
                    g[idx[j]] = gg;
                    const float xh = (zz[j] - mean) * rstd;
                    mxh = fmaxf(mxh, fabsf(xh));
                    if (j == arg) {
                        s1 += (double)gg;
                        s2 += (double)gg * (double)xh;
                        mg = fmaxf(mg, fabsf(gg));
                    }
                }
            }
        }
    }
    sm[0][tid] = s1;
    sm[1][tid] = s2;
    smx[0][tid] = __float_as_uint(mg);
    smx[1][tid] = __float_as_uint(mxh);
    __syncthreads();
    if (r == 0 && c < C) {
        for (int j = 1; j < k; ++j) { s1 += sm[0][j * Cb + cl]; s2 += sm[1][j * Cb + cl]; }
        part[((size_t)blockIdx.x * 2 + 0) * C + c] = s1;
        part[((size_t)blockIdx.x * 2 + 1) * C + c] = s2;
    }
    if (gx && tid < 2) {   // (integer max of non-negative float bits: order-independent)
        unsigned m = 0u;
        for (int j = 0; j < (int)blockDim.x; ++j) m = max(m, smx[tid][j]);
        if (m > *reinterpret_cast<volatile unsigned*>(gx + tid)) atomicMax(gx + tid, m);
    }
}

// The same for C % 4 == 0 (every layer of the shipped widths): a thread owns FOUR consecutive channels of the rows it walks -- 16-byte
// loads and stores, the per-channel constants in registers, no division in the row loop (the scalar kernel above spends most of
// its time on 4-byte accesses and 64-bit index arithmetic: lu0, 19 M elements, 149 us against 50 us of memory time).  Same partial-sum
// layout (part[block][2][C], fp64, fixed order), so the finalize kernel does not change.
__global__ void __launch_bounds__(256) act_bwd_v4_kernel(const ActParams a, const float* __restrict__ dy0,
                                                         const float* __restrict__ dy1, float* __restrict__ g, unsigned Nrows,
                                                         int Qb, int R, double* __restrict__ part, unsigned* gx) {
    __shared__ double sm[2][256][4];
    __shared__ unsigned smx[2][256];
    const int tid = threadIdx.x;
    const int r = tid / Qb, ql = tid - r * Qb;
    const int C = a.C, Q = C >> 2;
    const int q = ql + blockIdx.y * 256;            // channel quad
    const unsigned rpb = (Nrows + gridDim.x - 1) / gridDim.x;
    const unsigned r0 = blockIdx.x * rpb, r1 = min(Nrows, r0 + rpb);
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    float mg = 0.f, mxh = 0.f;
    const bool on = r < R && q < Q;
    if (on) {
        const int c = 4 * q;
        const float4 mean4 = *reinterpret_cast<const float4*>(a.stat + c), rstd4 = *reinterpret_cast<const float4*>(a.stat + C + c);
        const float4 sc4 = *reinterpret_cast<const float4*>(a.stat + 2 * C + c), sh4 = *reinterpret_cast<const float4*>(a.stat + 3 * C + c);
        const float mean[4] = {mean4.x, mean4.y, mean4.z, mean4.w}, rstd[4] = {rstd4.x, rstd4.y, rstd4.z, rstd4.w};
        const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
        const float ks = 1.0f / (1.0f - a.drop_rate);
        const unsigned OW = a.W >> 1, OH = a.H >> 1;
        for (unsigned row = r0 + r; row < r1; row += R) {
            float4 d4 = reinterpret_cast<const float4*>(dy0)[(size_t)row * Q + q];
            if (dy1) {
                const float4 e4 = reinterpret_cast<const float4*>(dy1)[(size_t)row * Q + q];
                d4.x += e4.x; d4.y += e4.y; d4.z += e4.z; d4.w += e4.w;
            }
            const float d[4] = {d4.x, d4.y, d4.z, d4.w};
            if (!a.pool) {
                const size_t e = (size_t)row * C + c;
                const float4 z4 = reinterpret_cast<const float4*>(a.z)[(size_t)row * Q + q];
                const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
                float gg[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = zz[k] * sc[k] + sh[k];
                    gg[k] = d[k] * drop_mul(a.drop_key, e + k, a.drop_rate, ks) * dact_of(v, a.act);
                    const float xh = (zz[k] - mean[k]) * rstd[k];
                    s1[k] += (double)gg[k];
                    s2[k] += (double)gg[k] * (double)xh;
                    mg = fmaxf(mg, fabsf(gg[k]));
                    mxh = fmaxf(mxh, fabsf(xh));
                }
                reinterpret_cast<float4*>(g)[(size_t)row * Q + q] = make_float4(gg[0], gg[1], gg[2], gg[3]);
            } else {
                const unsigned ox = row % OW, t = row / OW, oy = t % OH, b = t / OH;
                size_t px[4];
                float zz[4][4], vv[4][4], dm[4][4];
                int arg[4] = {0, 0, 0, 0};
                float best[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    px[j] = ((size_t)b * a.H + 2 * oy + (j >> 1)) * a.W + 2 * ox + (j & 1);
                    const float4 z4 = reinterpret_cast<const float4*>(a.z)[px[j] * Q + q];
                    const float zj[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        zz[j][k] = zj[k];
                        vv[j][k] = zj[k] * sc[k] + sh[k];
                        dm[j][k] = drop_mul(a.drop_key, px[j] * C + c + k, a.drop_rate, ks);
                        const float y = act_of(vv[j][k], a.act) * dm[j][k];
                        if (j == 0 || y > best[k]) { best[k] = y; arg[k] = j; }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float gg[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        gg[k] = j == arg[k] ? d[k] * dm[j][k] * dact_of(vv[j][k], a.act) : 0.f;
                        const float xh = (zz[j][k] - mean[k]) * rstd[k];
                        mxh = fmaxf(mxh, fabsf(xh));
                        if (j == arg[k]) {
                            s1[k] += (double)gg[k];
                            s2[k] += (double)gg[k] * (double)xh;
                            mg = fmaxf(mg, fabsf(gg[k]));
                        }
                    }
                    reinterpret_cast<float4*>(g)[px[j] * Q + q] = make_float4(gg[0], gg[1], gg[2], gg[3]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { sm[0][tid][k] = s1[k]; sm[1][tid][k] = s2[k]; }
    smx[0][tid] = __float_as_uint(mg);
    smx[1][tid] = __float_as_uint(mxh);
    __syncthreads();
    if (r == 0 && q < Q) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double t1 = s1[k], t2 = s2[k];
            for (int j = 1; j < R; ++j) { t1 += sm[0][j * Qb + ql][k]; t2 += sm[1][j * Qb + ql][k]; }
            part[((size_t)blockIdx.x * 2 + 0) * C + 4 * q + k] = t1;
            part[((size_t)blockIdx.x * 2 + 1) * C + 4 * q + k] = t2;
        }
    }
    if (gx && tid < 2) {
        unsigned m = 0u;
        for (int j = 0; j < 256; ++j) m = max(m, smx[tid][j]);
        if (m > *reinterpret_cast<volatile unsigned*>(gx + tid)) atomicMax(gx + tid, m);
    }
}

hipError_t launch_act_bwd(const ActParams& a, const float* dy0, const float* dy1, float* g, double* part, int nblk,
                          unsigned* gx, hipStream_t stream) {
    const ChanLayout l = chan_layout(a.C);
    const size_t rows = (size_t)a.B * (a.pool ? a.H / 2 : a.H) * (a.pool ? a.W / 2 : a.W);
    if (a.C % 4 == 0 && rows < 0x7fffffffull) {
        const int Q = a.C / 4, Qb = std::min(Q, 256), R = std::max(1, 256 / Qb), cbl = (Q + 255) / 256;
        hipLaunchKernelGGL(act_bwd_v4_kernel, dim3((unsigned)nblk, (unsigned)cbl), dim3(256), 0, stream, a, dy0, dy1, g, (unsigned)rows,
                           Qb, R, part, gx);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)nblk, (unsigned)l.cblocks), dim3((unsigned)l.threads), 0, stream, a,
                       dy0, dy1, g, rows, l.Cb, l.k, part, gx);
    return hipGetLastError();
}

__global__ void __launch_bounds__(64) bn_bwd_finalize_kernel(const double* __restrict__ part, int nblk, double N, int C,
                                                             float* dgamma, float* dbeta, float* __restrict__ m12) {
    __shared__ double sm[2][64];
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) {
        s1 += part[((size_t)b * 2 + 0) * C + c];
        s2 += part[((size_t)b * 2 + 1) * C + c];
    }
    block64_sum2(s1, s2, sm);
    if (threadIdx.x != 0) return;
    m12[c] = (float)(s1 / N);
    m12[C + c] = (float)(s2 / N);
    dgamma[c] = (float)s2;
    dbeta[c] = (float)s1;
}

hipError_t launch_bn_bwd_finalize(const double* part, int nblk, size_t N, int C, float* dgamma, float* dbeta, float* m12,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)C), dim3(64), 0, stream, part, nblk, (double)N, C, dgamma,
                       dbeta, m12);
    return hipGetLastError();
}

// (grid-stride; at most one atomic per block on the tensor's max word)
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(float* __restrict__ g, const float* __restrict__ z,
                                                           const float* __restrict__ stat, const float* __restrict__ m12,
                                                           size_t n, int C, unsigned* gmax) {
    float mx = 0.f;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const float xh = (z[e] - stat[c]) * stat[C + c];
        const float v = stat[2 * C + c] * (g[e] - m12[c] - xh * m12[C + c]);
        g[e] = v;
        mx = fmaxf(mx, fabsf(v));
    }
    if (gmax) block_absmax_to(gmax, mx);
}

// The same with the (hi, lo) planes conv_f16x3's input-gradient launches read written in the same pass (round 4; no second trip
// through the tensor).  Their power-of-two scale has to be known before the first element: |dz| <= max_c|gamma rstd| * (|g| +
// |mean g| + |xhat| |mean g xhat|) <= smax * gmax * (2 + xhmax)  (|mean g| <= gmax, |mean g xhat| <= gmax * E|xhat| <= gmax), the
// three maxima tracked by bn_finalize / act_bwd -- a true bound, 2^2 .. 2^3 above the real maximum, placed at 2^14.
template <bool VEC>
__global__ void __launch_bounds__(256) bn_bwd_apply_planes_kernel(float* __restrict__ g, const float* __restrict__ z,
                                                                  const float* __restrict__ stat, const float* __restrict__ m12,
                                                                  unsigned n, int C, int Cs, unsigned* gmax, const unsigned* __restrict__ bw,
                                                                  float* __restrict__ inv_scale, _Float16* __restrict__ hi,
                                                                  _Float16* __restrict__ lo, int* __restrict__ overflow) {
    float scale = 1.f;
    {
        const float bound = __uint_as_float(bw[0]) * __uint_as_float(bw[1]) * (2.f + __uint_as_float(bw[2]));
        const unsigned bb = __float_as_uint(bound);
        const int e = (int)((bb >> 23) & 0xFFu);
        if (bb != 0u && e > 0 && e < 255) {
            const int sh = max(-100, min(100, 14 - (e - 127)));   // bound -> [2^14, 2^15)
            scale = __uint_as_float((unsigned)(127 + sh) << 23);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) *inv_scale = 1.f / scale;
    }
    float mx = 0.f;
    bool bad = false;
    if constexpr (VEC) {   // C % 4 == 0: four channels of one pixel per thread
        const unsigned n4 = n / 4;
        for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
            const unsigned e = i * 4, px = e / (unsigned)C, c = e - px * (unsigned)C;
            const float4 gv = reinterpret_cast<const float4*>(g)[i], zv = reinterpret_cast<const float4*>(z)[i];
            const float4 mean = *reinterpret_cast<const float4*>(stat + c), rstd = *reinterpret_cast<const float4*>(stat + C + c);
            const float4 sc = *reinterpret_cast<const float4*>(stat + 2 * C + c);
            const float4 m1 = *reinterpret_cast<const float4*>(m12 + c), m2 = *reinterpret_cast<const float4*>(m12 + C + c);
            float v[4];
            v[0] = sc.x * (gv.x - m1.x - (zv.x - mean.x) * rstd.x * m2.x);
            v[1] = sc.y * (gv.y - m1.y - (zv.y - mean.y) * rstd.y * m2.y);
            v[2] = sc.z * (gv.z - m1.z - (zv.z - mean.z) * rstd.z * m2.z);
            v[3] = sc.w * (gv.w - m1.w - (zv.w - mean.w) * rstd.w * m2.w);
            reinterpret_cast<float4*>(g)[i] = make_float4(v[0], v[1], v[2], v[3]);
            union { _Float16 h[4]; uint2 u; } a, b;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mx = fmaxf(mx, fabsf(v[k]));
                const float f = v[k] * scale;
                bad = bad || !(fabsf(f) < 60000.f);
                a.h[k] = (_Float16)f;
                b.h[k] = (_Float16)(f - (float)a.h[k]);
            }
            const size_t at = (size_t)px * Cs + c;
            *reinterpret_cast<uint2*>(hi + at) = a.u;
            *reinterpret_cast<uint2*>(lo + at) = b.u;
            if (c + 4 == (unsigned)C && Cs > C) {   // the pad channels of the pixel's last octet (the slot is shared between layers)
                *reinterpret_cast<uint2*>(hi + at + 4) = make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(lo + at + 4) = make_uint2(0u, 0u);
            }
        }
    } else {
        for (unsigned e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
            const unsigned px = e / (unsigned)C, c = e - px * (unsigned)C;
            const float xh = (z[e] - stat[c]) * stat[C + c];
            const float v = stat[2 * C + c] * (g[e] - m12[c] - xh * m12[C + c]);
            g[e] = v;
            mx = fmaxf(mx, fabsf(v));
            const float f = v * scale;
            bad = bad || !(fabsf(f) < 60000.f);
            const _Float16 h = (_Float16)f;
            hi[(size_t)px * Cs + c] = h;
            lo[(size_t)px * Cs + c] = (_Float16)(f - (float)h);
            if (c + 1 == (unsigned)C)
                for (int k = C; k < Cs; ++k) { hi[(size_t)px * Cs + k] = (_Float16)0.f; lo[(size_t)px * Cs + k] = (_Float16)0.f; }
        }
    }
    if (bad) atomicOr(overflow, 1);
    if (gmax) block_absmax_to(gmax, mx);
}

// (hi == NULL: the fp32 tensor only -- the top layer, the first down layer, the exact-fp32 route)
hipError_t launch_bn_bwd_apply_max(float* g, const float* z, const float* stat, const float* m12, size_t N, int C,
                                   unsigned* gmax, const unsigned* bw, float* inv_scale, _Float16* hi, _Float16* lo, int Cs,
                                   int* overflow, hipStream_t stream) {
    const size_t n = N * C;
    const unsigned blocks = (unsigned)std::min<size_t>(1024, (n + 255) / 256);
    // (planes asked for on a tensor the 32-bit-index kernel cannot address: refuse -- the caller's convolutions and weight
    // gradients would read planes and an inverse scale nobody wrote.  Batch x resolution x channels >= 2^32: not a shape of this graph)
    if (hi && n >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    if (hi) {
        const unsigned bv = (unsigned)std::min<size_t>(2048, (n / 4 + 255) / 256 + 1);
        if (C % 4 == 0 && Cs % 4 == 0)
            hipLaunchKernelGGL(bn_bwd_apply_planes_kernel<true>, dim3(bv), dim3(256), 0, stream, g, z, stat, m12, (unsigned)n, C, Cs, gmax,
                               bw, inv_scale, hi, lo, overflow);
        else
            hipLaunchKernelGGL(bn_bwd_apply_planes_kernel<false>, dim3(blocks ? blocks : 1), dim3(256), 0, stream, g, z, stat, m12,
                               (unsigned)n, C, Cs, gmax, bw, inv_scale, hi, lo, overflow);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, g, z, stat, m12, n, C, gmax);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) leaky_bwd_s2d_kernel(const float* __restrict__ d_us, const float* __restrict__ us,
                                                            int S, int C, float* __restrict__ gS, size_t n, unsigned* gmax) {
    float mx = 0.f;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        size_t r = e / C;
        const int q = (int)(r & 3); r >>= 2;
        const int j = (int)(r % S); r /= S;
        const int i = (int)(r % S);
        const size_t b = r / S;
        const size_t src = ((b * (2 * S) + 2 * i + (q >> 1)) * (size_t)(2 * S) + 2 * j + (q & 1)) * C + c;
        const float v = d_us[src] * (us[src] > 0.f ? 1.f : 0.2f);
        gS[e] = v;
        mx = fmaxf(mx, fabsf(v));
    }
    if (gmax) block_absmax_to(gmax, mx);
}

// C % 4 == 0 and < 2^32 elements (every layer of the shipped widths): four channels per thread, 16-byte accesses, 32-bit index
// arithmetic (the scalar kernel: 146 us for 3 x 75 MB at 8 x 256 x 256 x 36, this one the memory time)
__global__ void __launch_bounds__(256) leaky_bwd_s2d_v4_kernel(const float4* __restrict__ d_us, const float4* __restrict__ us,
                                                               unsigned S, unsigned Q, float4* __restrict__ gS, unsigned n4, unsigned* gmax) {
    float mx = 0.f;
    for (unsigned e = blockIdx.x * 256 + threadIdx.x; e < n4; e += gridDim.x * 256) {
        const unsigned q4 = e % Q;
        unsigned r = e / Q;
        const unsigned q = r & 3u; r >>= 2;
        const unsigned j = r % S; r /= S;
        const unsigned i = r % S;
        const unsigned b = r / S;
        const unsigned src = ((b * (2 * S) + 2 * i + (q >> 1)) * (2 * S) + 2 * j + (q & 1u)) * Q + q4;
        const float4 d = d_us[src], u = us[src];
        float4 v;
        v.x = d.x * (u.x > 0.f ? 1.f : 0.2f);
        v.y = d.y * (u.y > 0.f ? 1.f : 0.2f);
        v.z = d.z * (u.z > 0.f ? 1.f : 0.2f);
        v.w = d.w * (u.w > 0.f ? 1.f : 0.2f);
        gS[e] = v;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (gmax) block_absmax_to(gmax, mx);
}

hipError_t launch_leaky_bwd_s2d_max(const float* d_us, const float* us, int B, int S, int C, float* gS, unsigned* gmax,
                                    hipStream_t stream) {
    const size_t n = (size_t)B * S * S * 4 * C;
    if (C % 4 == 0 && n < 0xffffffffull) {
        const unsigned n4 = (unsigned)(n / 4);
        const unsigned blocks = std::min(2048u, (n4 + 255u) / 256u);
        hipLaunchKernelGGL(leaky_bwd_s2d_v4_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, reinterpret_cast<const float4*>(d_us),
                           reinterpret_cast<const float4*>(us), (unsigned)S, (unsigned)(C / 4), reinterpret_cast<float4*>(gS), n4, gmax);
        return hipGetLastError();
    }
    const unsigned blocks = (unsigned)std::min<size_t>(1024, (n + 255) / 256);
    hipLaunchKernelGGL(leaky_bwd_s2d_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, d_us, us, S, C, gS, n, gmax);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ top layer
constexpr int kMaxK = 8;

// 256 pixels per block: the [256][C] slab is loaded with consecutive threads on consecutive addresses and parked in LDS
// (pixel pitch C+1: a thread walking its own pixel then touches a new bank every step), each thread reduces one pixel
__global__ void __launch_bounds__(256) head_fwd_kernel(const float* __restrict__ x, size_t N, int C, int K,
                                                       const float* __restrict__ w, float* __restrict__ t0) {
    extern __shared__ float hl[];    // [C][K] weights, then [256][C+1] pixels
    float* const wl = hl;
    float* const xl = hl + C * K;
    const size_t p0 = (size_t)blockIdx.x * 256;
    const int npx = (int)min((size_t)256, N - p0);
    for (int i = threadIdx.x; i < C * K; i += 256) wl[i] = w[i];
    for (int i = threadIdx.x; i < npx * C; i += 256) {
        const int px = i / C, c = i - px * C;
        xl[px * (C + 1) + c] = x[p0 * C + i];
    }
    __syncthreads();
    if ((int)threadIdx.x >= npx) return;
    float acc[kMaxK];
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) acc[k] = 0.f;
    const float* xp = xl + threadIdx.x * (C + 1);
    for (int c = 0; c < C; ++c) {
        const float v = xp[c];
#pragma unroll
        for (int k = 0; k < kMaxK; ++k)
            if (k < K) acc[k] = fmaf(v, wl[c * K + k], acc[k]);
    }
    const size_t p = p0 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < kMaxK; ++k)
        if (k < K) t0[p * K + k] = acc[k];
}

hipError_t launch_head_fwd(const float* x, size_t N, int C, int K, const float* w, float* t0, hipStream_t stream) {
    if (K > kMaxK) return hipErrorInvalidValue;
    const size_t lds = sizeof(float) * ((size_t)C * K + 256 * (size_t)(C + 1));
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), lds, stream, x, N, C, K, w, t0);
    return hipGetLastError();
}

int loss_blocks(size_t N) { return (int)std::min<size_t>(1024, (N + 255) / 256); }

// reference UnMicst1-5.py:362-367: mean over pixels of -sum_k weights*labels*log(clip(p, eps, 1-eps)); duo: log(p)
__global__ void __launch_bounds__(256) softmax_loss_kernel(const float* __restrict__ t0, const float* __restrict__ stat,
                                                           const float* __restrict__ labels,
                                                           const float* __restrict__ weights, size_t N, int K,
                                                           float clip_eps, float* __restrict__ probs,
                                                           float* __restrict__ dt, double* __restrict__ part) {
    __shared__ double sm[256];
    double lsum = 0.0;
    const float invN = (float)(1.0 / (double)N);
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < N; p += (size_t)gridDim.x * 256) {
        float t[kMaxK], pr[kMaxK], dp[kMaxK];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < kMaxK; ++k)
            if (k < K) { t[k] = t0[p * K + k] * stat[2 * K + k] + stat[3 * K + k]; mx = fmaxf(mx, t[k]); }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxK; ++k)
            if (k < K) { pr[k] = expf(t[k] - mx); den += pr[k]; }
        float dot = 0.f, L = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxK; ++k)
            if (k < K) {
                pr[k] = pr[k] / den;
                const float wy = weights[p * K + k] * labels[p * K + k];
                float pc = pr[k];
                bool pass = true;
                if (clip_eps > 0.f) {
                    pass = pc >= clip_eps && pc <= 1.0f - clip_eps;
                    pc = fminf(fmaxf(pc, clip_eps), 1.0f - clip_eps);
                }
                L -= wy * logf(pc);
                dp[k] = (pass && wy != 0.f) ? -wy / pc : 0.f;
                dot += dp[k] * pr[k];
                if (probs) probs[p * K + k] = pr[k];
            }
#pragma unroll
        for (int k = 0; k < kMaxK; ++k)
            if (k < K) dt[p * K + k] = pr[k] * (dp[k] - dot) * invN;
        lsum += (double)L;
    }
    sm[threadIdx.x] = lsum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}

hipError_t launch_softmax_loss(const float* t0, const float* stat, const float* labels, const float* weights, size_t N,
                               int K, float clip_eps, float* probs, float* dt, double* part, int nblk, hipStream_t stream) {
    if (K > kMaxK) return hipErrorInvalidValue;
    hipLaunchKernelGGL(softmax_loss_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, t0, stat, labels, weights, N, K,
                       clip_eps, probs, dt, part);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) softmax_only_kernel(const float* __restrict__ t0, const float* __restrict__ stat,
                                                           size_t N, int K, float* __restrict__ probs) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= N) return;
    float t[kMaxK];
    float mx = -3.0e38f, den = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxK; ++k)
        if (k < K) { t[k] = t0[p * K + k] * stat[2 * K + k] + stat[3 * K + k]; mx = fmaxf(mx, t[k]); }
#pragma unroll
    for (int k = 0; k < kMaxK; ++k)
        if (k < K) { t[k] = expf(t[k] - mx); den += t[k]; }
#pragma unroll
    for (int k = 0; k < kMaxK; ++k)
        if (k < K) probs[p * K + k] = t[k] / den;
}

hipError_t launch_softmax_only(const float* t0, const float* stat, size_t N, int K, float* probs, hipStream_t stream) {
    if (K > kMaxK) return hipErrorInvalidValue;
    hipLaunchKernelGGL(softmax_only_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, t0, stat, N, K, probs);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) head_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dt0,
                                                       const float* __restrict__ w, size_t N, int C, int K, int Cb, int k,
                                                       float* __restrict__ dx, double* __restrict__ part) {
    __shared__ double sm[kMaxK][256];
    const int tid = threadIdx.x;
    const int r = tid / Cb, cl = tid - r * Cb;
    const int c = cl + blockIdx.y * 256;
    const size_t rpb = (N + gridDim.x - 1) / gridDim.x;
    const size_t r0 = (size_t)blockIdx.x * rpb, r1 = std::min(N, r0 + rpb);
    double acc[kMaxK];
    float wk[kMaxK];
#pragma unroll
    for (int j = 0; j < kMaxK; ++j) { acc[j] = 0.0; wk[j] = (c < C && j < K) ? w[c * K + j] : 0.f; }
    if (c < C)
        for (size_t row = r0 + r; row < r1; row += k) {
            const float xv = x[row * C + c];
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < kMaxK; ++j)
                if (j < K) {
                    const float g = dt0[row * K + j];
                    acc[j] += (double)xv * (double)g;
                    d = fmaf(g, wk[j], d);
                }
            dx[row * C + c] = d;
        }
#pragma unroll
    for (int j = 0; j < kMaxK; ++j) sm[j][tid] = acc[j];
    __syncthreads();
    if (r == 0 && c < C) {
#pragma unroll
        for (int j = 0; j < kMaxK; ++j)
            if (j < K) {
                double s = acc[j];
                for (int i = 1; i < k; ++i) s += sm[j][i * Cb + cl];
                part[((size_t)blockIdx.x * C + c) * K + j] = s;
            }
    }
}

hipError_t launch_head_bwd(const float* x, const float* dt0, const float* w, size_t N, int C, int K, float* dx,
                           double* part, int nblk, hipStream_t stream) {
    if (K > kMaxK) return hipErrorInvalidValue;
    const ChanLayout l = chan_layout(C);
    hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)nblk, (unsigned)l.cblocks), dim3((unsigned)l.threads), 0, stream, x,
                       dt0, w, N, C, K, l.Cb, l.k, dx, part);
    return hipGetLastError();
}

__device__ __forceinline__ float reg_grad(float w, int kind, float c) {
    if (kind == 1) return w > 0.f ? c : (w < 0.f ? -c : 0.f);
    if (kind == 2) return 2.0f * c * w;
    return 0.f;
}

__global__ void __launch_bounds__(64) reduce_partials_kernel(const double* __restrict__ part, int nblk, int n,
                                                             double scale, float* __restrict__ dst,
                                                             const float* __restrict__ w, int reg_kind, float reg_c) {
    __shared__ double sm[2][64];
    const int i = blockIdx.x;
    double s = 0.0, unused = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += part[(size_t)b * n + i];
    block64_sum2(s, unused, sm);
    if (threadIdx.x != 0) return;
    float v = (float)(s * scale);
    if (w && reg_kind) v += reg_grad(w[i], reg_kind, reg_c);
    dst[i] = v;
}

hipError_t launch_reduce_partials(const double* part, int nblk, int n, double scale, float* dst, const float* w,
                                  int reg_kind, float reg_c, hipStream_t stream) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)n), dim3(64), 0, stream, part, nblk, n, scale, dst, w,
                       reg_kind, reg_c);
    return hipGetLastError();
}

// (one block of 256: thread t adds the contiguous run [t * per, (t + 1) * per) in order, thread 0 adds the 256 run sums in order --
//  a fixed order; the single-thread loop it replaces took 61 us for 1024 partials at the head of the backward pass)
__global__ void __launch_bounds__(256) sum_to_scalar_kernel(const double* __restrict__ part, int n, double scale, double* out, int slot,
                                                            int accumulate) {
    __shared__ double sm[256];
    const int per = (n + 255) / 256;
    const int i0 = threadIdx.x * per, i1 = min(n, i0 + per);
    double s = 0.0;
    for (int i = i0; i < i1; ++i) s += part[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x != 0) return;
    double t = 0.0;
    for (int j = 0; j < 256; ++j) t += sm[j];
    t *= scale;
    out[slot] = accumulate ? out[slot] + t : t;
}

hipError_t launch_sum_to_scalar(const double* part, int n, double scale, double* out, int slot, int accumulate,
                                hipStream_t stream) {
    hipLaunchKernelGGL(sum_to_scalar_kernel, dim3(1), dim3(256), 0, stream, part, n, scale, out, slot, accumulate);
    return hipGetLastError();
}

// regularisation loss of every regularised tensor in two launches: part[seg][64] = sum |w| or w^2 over a strided share,
// then out[slot] += sum_seg coef[seg] * sum_b part[seg][b]
__global__ void __launch_bounds__(256) reg_partials_kernel(const RegSeg* __restrict__ segs, int kind,
                                                           double* __restrict__ part) {
    __shared__ double sm[256];
    const RegSeg sg = segs[blockIdx.y];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < sg.n; i += (size_t)gridDim.x * 256) {
        const double v = (double)sg.w[i];
        s += kind == 1 ? fabs(v) : v * v;
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = sm[0];
}

__global__ void reg_finalize_kernel(const RegSeg* __restrict__ segs, int nseg, const double* __restrict__ part, int nb,
                                    double* out, int slot) {
    double total = 0.0;
    for (int g = 0; g < nseg; ++g) {
        double s = 0.0;
        for (int b = 0; b < nb; ++b) s += part[(size_t)g * nb + b];
        total += (double)segs[g].coef * s;
    }
    out[slot] += total;
}

hipError_t launch_reg_loss(const RegSeg* segs_dev, int nseg, int kind, double* part, double* out, int slot,
                           hipStream_t stream) {
    if (nseg <= 0) return hipSuccess;
    hipLaunchKernelGGL(reg_partials_kernel, dim3(64, (unsigned)nseg), dim3(256), 0, stream, segs_dev, kind, part);
    hipLaunchKernelGGL(reg_finalize_kernel, dim3(1), dim3(1), 0, stream, segs_dev, nseg, part, 64, out, slot);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ weight gradient
// GEMM view: M = input channels of a slab (16*MI per workgroup, MI = 1..3 chosen per layer), N = output channels
// (64 per workgroup, 16 per wave), K = pixels.  A workgroup walks pixel tiles of 128 pixels (imgs x TH x TW) of its
// slice: the X halo tile and the G tile are staged in LDS once and serve every slab (filter tap) of the group --
// 9 taps x MI channel tiles accumulator tiles per wave, one ds_read_b32 per MFMA.  LDS pitches (16 or 48 / 80 floats
// per pixel) put the four pixels of a 16x16x4 fragment on disjoint bank quarters.  Partial sums go to ws[slice];
// wgrad_reduce_kernel adds the slices in order.
constexpr int kWgCO = 64, kWgPG = 80, kWgNS = 9, kWgPix = 128;
__host__ __device__ constexpr int wg_px(int mi) { return mi == 1 ? 16 : 48; }   // X pitch: = 16 or 48 (mod 64)
// halo pixels of one tile: 10 x 18 = 180 (8 x 16 tile, 3 x 3 taps), 2 x 10 x 10 = 200; the 4 x 4 layers (8 images
// x 6 x 6 = 288) use the BIG variants, which exist for MI <= 2 only (register budget of the staging slots)
constexpr int kWgHalo = 208, kWgHaloBig = 288;
__host__ __device__ constexpr int wg_hx(int mi, bool big) { return ((big ? kWgHaloBig : kWgHalo) * 4 * mi + 255) / 256; }

template <int MI, bool BIG>
__global__ void __launch_bounds__(256, 2) wgrad_mfma_f32(const WgradParams p) {
    constexpr int CI = 16 * MI, PX = wg_px(MI);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Xl = smem;                  // [nhalo][PX]
    float* const Gl = smem + p.nhalo * PX;   // [128][kWgPG]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, li = lane & 15;
    const int TW = 1 << p.tw_log2, TH = 1 << p.th_log2;

    const int slice = blockIdx.x;
    const int nco = (p.Cg + kWgCO - 1) / kWgCO;
    const int ci0 = (blockIdx.y / nco) * CI, co0 = (blockIdx.y % nco) * kWgCO;
    const int slab0 = p.gstart[blockIdx.z], ns = p.gcount[blockIdx.z];
    const int coff = p.coff[slab0];
    const bool wave_live = co0 + wave * 16 < p.Cg;

    f32x4 acc[kWgNS][MI];
#pragma unroll
    for (int s = 0; s < kWgNS; ++s)
#pragma unroll
        for (int t2 = 0; t2 < MI; ++t2) acc[s][t2] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // register-staged pipeline: the next tile's global loads are issued before the MFMA block of the current tile and
    // written to LDS after it
    constexpr int HX = wg_hx(MI, BIG);
    float4 xr[HX], gr[kWgPix * (kWgCO / 4) / 256];
    const int nx4 = p.nhalo * (CI / 4);
    auto load_tile = [&](int t) {
        const int tx = t % p.tiles_x;
        const int ty = (t / p.tiles_x) % p.tiles_y;
        const int img0 = (t / (p.tiles_x * p.tiles_y)) * p.imgs;
        const int y0 = ty * TH, x0 = tx * TW;
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));   // opaque per call: keeps the slot decode out of the tile loop's live registers
#pragma unroll
        for (int i = 0; i < HX; ++i) {
            const int e = tid_o + i * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < nx4) {
                const int hp = e / (CI / 4), q = e - hp * (CI / 4);
                const int il = hp / p.imgplane;
                const int rr = hp - il * p.imgplane;
                const int hy = rr / p.hw, hx = rr - hy * p.hw;
                const int gy = y0 + p.ymin + hy, gx = x0 + p.xmin + hx, img = img0 + il;
                const int c = ci0 + 4 * q;
                if (img < p.B && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W && c < p.Cx) {
                    const float* src = p.X + ((size_t)(img * p.H + gy) * p.W + gx) * p.Cxt + coff + c;
                    if (p.vecx && c + 3 < p.Cx) v = *reinterpret_cast<const float4*>(src);
                    else {
                        v.x = src[0];
                        if (c + 1 < p.Cx) v.y = src[1];
                        if (c + 2 < p.Cx) v.z = src[2];
                        if (c + 3 < p.Cx) v.w = src[3];
                    }
                }
            }
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < kWgPix * (kWgCO / 4) / 256; ++i) {
            const int e = tid_o + i * 256;
            const int px = e >> 4, q = e & 15;
            const int il = px >> (p.th_log2 + p.tw_log2);
            const int y = (px >> p.tw_log2) & (TH - 1), x = px & (TW - 1);
            const int img = img0 + il;
            const int co = co0 + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (il < p.imgs && img < p.B && co < p.Cg) {   // (tiny layers: fewer images than pixel slots -> zero rows of G)
                const float* src = p.G + ((size_t)(img * p.H + y0 + y) * p.W + x0 + x) * p.Cg + co;
                if (p.vecg && co + 3 < p.Cg) v = *reinterpret_cast<const float4*>(src);
                else {
                    v.x = src[0];
                    if (co + 1 < p.Cg) v.y = src[1];
                    if (co + 2 < p.Cg) v.z = src[2];
                    if (co + 3 < p.Cg) v.w = src[3];
                }
            }
            gr[i] = v;
        }
    };
    auto store_tile = [&]() {
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
#pragma unroll
        for (int i = 0; i < HX; ++i) {
            const int e = tid_o + i * 256;
            if (e < nx4) {
                const int hp = e / (CI / 4), q = e - hp * (CI / 4);
                *reinterpret_cast<float4*>(Xl + hp * PX + 4 * q) = xr[i];
            }
        }
#pragma unroll
        for (int i = 0; i < kWgPix * (kWgCO / 4) / 256; ++i) {
            const int e = tid_o + i * 256;
            *reinterpret_cast<float4*>(Gl + (e >> 4) * kWgPG + 4 * (e & 15)) = gr[i];
        }
    };

    int so[kWgNS];   // LDS offset of each slab of the group (wave-uniform)
#pragma unroll
    for (int s = 0; s < kWgNS; ++s) {
        const int sb = slab0 + (s < ns ? s : 0);
        so[s] = __builtin_amdgcn_readfirstlane(((p.dy[sb] - p.ymin) * p.hw + (p.dx[sb] - p.xmin)) * PX);
    }
    const int t_begin = slice * p.tiles_per_slice;
    const int t_end = min(p.ntiles, (slice + 1) * p.tiles_per_slice);
    if (t_begin < t_end) load_tile(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();   // the previous tile's fragment reads are done
        store_tile();
        __syncthreads();
        if (t + 1 < t_end) load_tile(t + 1);
        if (wave_live) {
            for (int ks = 0; ks < kWgPix / 4; ++ks) {
                const int px = 4 * ks + kq;
                const int il = min(px >> (p.th_log2 + p.tw_log2), p.imgs - 1);   // unused slots: any finite X (their G is 0)
                const int y = (px >> p.tw_log2) & (TH - 1), x = px & (TW - 1);
                const float* ap = Xl + (il * p.imgplane + y * p.hw + x) * PX + li;
                const float b = Gl[px * kWgPG + wave * 16 + li];
                float a[kWgNS][MI];
#pragma unroll
                for (int s = 0; s < kWgNS; ++s)
                    if (s < ns) {
#pragma unroll
                        for (int t2 = 0; t2 < MI; ++t2) a[s][t2] = ap[so[s] + 16 * t2];
                    }
#pragma unroll
                for (int s = 0; s < kWgNS; ++s)
                    if (s < ns) {
#pragma unroll
                        for (int t2 = 0; t2 < MI; ++t2)
                            acc[s][t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][t2], b, acc[s][t2], 0, 0, 0);
                    }
            }
        }
    }

    // D tile: row (input channel) = 4*kq + r, column (output channel) = li
    const size_t slab_sz = (size_t)p.Cx * p.Cg;
    const int co = co0 + wave * 16 + li;
#pragma unroll
    for (int s = 0; s < kWgNS; ++s) {
        if (s < ns && co < p.Cg) {
            float* dst = p.ws + ((size_t)slice * p.nslab + slab0 + s) * slab_sz;
#pragma unroll
            for (int t2 = 0; t2 < MI; ++t2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = ci0 + t2 * 16 + 4 * kq + r;
                    if (ci < p.Cx) dst[(size_t)ci * p.Cg + co] = acc[s][t2][r];
                }
        }
    }
}


// ------------------------------------------------------------------------------------------------ weight gradient, thin X
// The first layer (and the v2 graph's top skip connection) has 1 - 4 input channels: a 16-channel MFMA tile is 6 - 25 % full and
// the 128-pixel tiles of the kernel above spend their time staging (165 - 277 us for 2 x 36 channels at 8 x 256 x 256, the last
// kernel of the backward pass with nothing left to hide behind).  Here the GEMM runs on the vector ALUs: a workgroup owns `thin`
// whole image rows, stages their X halo (a few KB) in LDS, and thread (output channel co, pixel group pg) walks the strip's
// pixels pg, pg + PG, ... with one coalesced load of G (four output channels) per pixel and nslab * CX broadcast LDS reads.  The PG
// partial sums are added in a fixed order; the slices by wgrad_reduce_kernel as before (same ws layout).
template <int CX, int V>   // V output channels per thread: 4 (Cg % 4 == 0: 16-byte loads of G) or 1
__global__ void __launch_bounds__(256) wgrad_thin_kernel(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int R = p.thin, hw = p.hw, hh = p.hh;
    const int slice = blockIdx.x;
    const int row0 = slice * R;                       // global row = img * H + y; R divides H: a strip stays inside one image
    const int img = row0 / p.H, y0 = row0 - img * p.H;
    const int ns = p.gcount[0];
    const int coff = p.coff[0];
    float* const Xl = smem;                            // [hh][hw][CX]
    float* const red = smem + ((hh * hw * CX + 3) & ~3);   // [ceil(PG / 4)][ns * CX][Cg]
    for (int e = tid; e < hh * hw; e += 256) {
        const int hy = e / hw, hx = e - hy * hw;
        const int gy = y0 + p.ymin + hy, gx = p.xmin + hx;
        const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const float* src = p.X + ((size_t)(img * p.H + (in ? gy : 0)) * p.W + (in ? gx : 0)) * p.Cxt + coff;
#pragma unroll
        for (int c = 0; c < CX; ++c) Xl[e * CX + c] = in ? src[c] : 0.f;
    }
    int so[kWgNS];
#pragma unroll
    for (int s = 0; s < kWgNS; ++s) {
        const int sb = s < ns ? s : 0;
        so[s] = __builtin_amdgcn_readfirstlane(((p.dy[sb] - p.ymin) * hw + (p.dx[sb] - p.xmin)) * CX);
    }
    __syncthreads();
    const int Cg = p.Cg, Q = Cg / V, PG = 256 / Q;
    const int pg = tid / Q, co = (tid - pg * Q) * V;
    float acc[kWgNS][CX][V];
#pragma unroll
    for (int s = 0; s < kWgNS; ++s)
#pragma unroll
        for (int c = 0; c < CX; ++c)
#pragma unroll
            for (int v = 0; v < V; ++v) acc[s][c][v] = 0.f;
    const int nout = ns * CX * Cg;                     // (s, ci, co) in the order of the ws slab: [slab][Cx][Cg]
    {   // branch-free: every thread walks ceil(npx / PG) pixels, a pixel past the strip is clamped and its G taken as zero (the four
        // threads past the last pixel group do the same and drop their sums) -- the compiler keeps two loads of G in flight
        const int npx = R * p.W;
        const int wl = 31 - __builtin_clz(p.W);      // W is a power of two (wgrad_setup)
        const float* const G = p.G + (size_t)row0 * p.W * Cg + co;
        auto load_g = [&](int px, float* g) {
            const int pc = min(px, npx - 1);
            if constexpr (V == 4) {
                const float4 g4 = *reinterpret_cast<const float4*>(G + (size_t)pc * Cg);
                g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w;
            } else {
                g[0] = G[(size_t)pc * Cg];
            }
        };
        auto step = [&](int px, const float* gin) {
            const bool live = px < npx;
            const int pc = min(px, npx - 1);
            float g[V];
#pragma unroll
            for (int v = 0; v < V; ++v) g[v] = live ? gin[v] : 0.f;
            const int y = pc >> wl, x = pc & (p.W - 1);
            const float* const xp = Xl + (y * hw + x) * CX;
            float xv[kWgNS][CX];   // (all kWgNS slabs: a slab past ns re-reads slab 0 and its sums are dropped)
#pragma unroll
            for (int s = 0; s < kWgNS; ++s)
#pragma unroll
                for (int c = 0; c < CX; ++c) xv[s][c] = xp[so[s] + c];
#pragma unroll
            for (int s = 0; s < kWgNS; ++s)
#pragma unroll
                for (int c = 0; c < CX; ++c)
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[s][c][v] = __builtin_fmaf(xv[s][c], g[v], acc[s][c][v]);
        };
        const int niter = (npx + PG - 1) / PG;
        int ahead = 2 * PG;   // (opaque: hipcc otherwise proves the prefetch is the next iteration's own load and sinks it back there)
        asm volatile("" : "+s"(ahead));
        float ga[V], gb[V];
        load_g(pg, ga);
        load_g(pg + PG, gb);
        for (int i = 0; i < niter; i += 2) {
            const int px = pg + i * PG;
            float g0[V], g1[V];
#pragma unroll
            for (int v = 0; v < V; ++v) { g0[v] = ga[v]; g1[v] = gb[v]; }
            load_g(px + ahead, ga);
            load_g(px + ahead + PG, gb);
            step(px, g0);
            step(px + PG, g1);
        }
    }
    // the PG partial sums, in a fixed order and through S = ceil(PG / 4) LDS slots (a slot per group would cost 36 KB for 36 output
    // channels and leave a CU three workgroups, one short of the 1024-workgroup launch fitting in one round): the top m = min(S, n - S)
    // groups are folded into the bottom m until S are left, and those are added in group order
    const int S = (PG + 3) / 4;
    auto slot = [&](int g, int s, int c, int v) { return (size_t)g * nout + (size_t)(s * CX + c) * Cg + co + v; };
    int n = PG;
    while (n > S) {
        const int m = min(S, n - S);
        if (pg >= n - m && pg < n) {
#pragma unroll
            for (int s = 0; s < kWgNS; ++s)
                if (s < ns) {
#pragma unroll
                    for (int c = 0; c < CX; ++c)
#pragma unroll
                        for (int v = 0; v < V; ++v) red[slot(pg - (n - m), s, c, v)] = acc[s][c][v];
                }
        }
        __syncthreads();
        if (pg < m) {
#pragma unroll
            for (int s = 0; s < kWgNS; ++s)
                if (s < ns) {
#pragma unroll
                    for (int c = 0; c < CX; ++c)
#pragma unroll
                        for (int v = 0; v < V; ++v) acc[s][c][v] += red[slot(pg, s, c, v)];
                }
        }
        __syncthreads();
        n -= m;
    }
    if (pg < n) {
#pragma unroll
        for (int s = 0; s < kWgNS; ++s)
            if (s < ns) {
#pragma unroll
                for (int c = 0; c < CX; ++c)
#pragma unroll
                    for (int v = 0; v < V; ++v) red[slot(pg, s, c, v)] = acc[s][c][v];
            }
    }
    __syncthreads();
    float* const dst = p.ws + (size_t)slice * p.nslab * CX * Cg;
    for (int e = tid; e < nout; e += 256) {
        float t = red[e];
        for (int j = 1; j < n; ++j) t += red[(size_t)j * nout + e];
        dst[e] = t;
    }
}

static size_t wgrad_thin_lds(const WgradParams& p) {
    const int V = p.Cg % 4 == 0 ? 4 : 1, PG = 256 / (p.Cg / V);
    return sizeof(float) * ((size_t)((p.hh * p.hw * p.Cx + 3) & ~3) + (size_t)((PG + 3) / 4) * p.gcount[0] * p.Cx * p.Cg);
}

// ------------------------------------------------------------------------------------------------ weight gradient, f16x3
// The same GEMM on the binary16 matrix cores with fp32-equivalent products: x*g = xh*gh + xh*gl + xl*gh, (hi, lo) exact
// binary16 pairs, fp32 accumulation (the scheme of conv_f16x3).  One v_mfma_f32_16x16x32_f16 covers
// 32 pixels of K where the fp32 MFMA covers 4: 3 x 16 cycles instead of 8 x 32.
//   * Operands are converted while they are staged: fp32 NHWC from HBM -> scaled -> (hi, lo) -> channel-planar LDS
//     [plane][channel][pixel] (a fragment = 8 consecutive pixels of one channel = one ds_read_b128).  Every operand is
//     scaled by the power of two that brings its tracked max |v| (written by its producer) to [2^11, 2^12): gradients
//     would flush to zero in binary16 otherwise, and activations keep their lo parts normal; the reduce kernel divides
//     the product of the two scales out.
//   * A tap shifts the X window by dx pixels = dx halves: the fragment is read as an aligned b128 + b32 and funnel-
//     shifted (v_alignbit) -- no per-tap copies of the halo.
//   * 48 x 48 channel tiles (the 36*2^k and 80*2^k widths of the shipped models pad to multiples of 48 with <= 1.33x):
//     the 9 taps x 3 channel tiles = 27 (tap, tile) pairs are dealt 7/7/7/6 to the four waves, each wave against all
//     three output-channel tiles, so every X fragment is read once per workgroup and every wave is busy.
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
constexpr int kHwC = 48, kHwPW = 7, kHwXT = 5, kHwGT = 3;

__device__ __forceinline__ float wg_scale(const unsigned* mx) {
    const int e = (int)((*mx >> 23) & 0xff);          // biased exponent of the tensor's tracked max |v|
    if (e == 0 || e == 255) return 1.f;
    const int se = min(max(127 + 11 - (e - 127), 1), 254);
    return __uint_as_float((unsigned)se << 23);        // max * scale in [2^11, 2^12)
}

__device__ __forceinline__ unsigned pack_h2(_Float16 a, _Float16 b) {
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

// PL = true (round 4): the operands are staged from the (hi, lo) binary16 NHWC planes their producers already wrote for conv_f16x3
// (WgradParams::Xhi ..): half the bytes, no conversion, and a staging task is 16 bit operations instead of ~56 conversions and
// subtractions.  X planes are unscaled, G planes carry the power of two of their tensor (undone by the reduce through *ginv).
// OC (with PL): a staging task is (pixel pair, channel OCTET) with octets fastest over the lanes -- 16-byte loads, six lanes on one
// pixel's 96 contiguous bytes -- instead of (pixel pair, channel quad) with pixel pairs fastest, whose 8-byte loads put every lane of
// an instruction on a cache line of its own (timing ablation, profiles/r04/train_wgrad_ablation.txt: this kernel's global loads are
// 30 % of its time and slow the main stream's memory-bound kernels by 7 - 25 %).  Needs 16-byte aligned octets: channel offsets
// and stored channel counts that are multiples of 8.
template <bool PL, bool OC = false>
__global__ void __launch_bounds__(256, 2) wgrad_f16x3(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    _Float16* const Xh = reinterpret_cast<_Float16*>(smem_b);   // [2][48][xs]
    _Float16* const Gh = Xh + 2 * kHwC * p.xs;                  // [2][48][gs]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, li = lane & 15;
    const int TW = 1 << p.tw_log2, TH = 1 << p.th_log2;
    const int imgplaneP = p.hh * p.hp;

    const int slice = blockIdx.x;
    const int nco = (p.Cg + kHwC - 1) / kHwC;
    const int ci0 = (blockIdx.y / nco) * kHwC, co0 = (blockIdx.y % nco) * kHwC;
    const int slab0 = p.gstart[blockIdx.z], ns = p.gcount[blockIdx.z];
    const int coff = p.coff[slab0];
    const float sx = PL ? 1.f : wg_scale(p.xmax), sg = PL ? 1.f : wg_scale(p.gmax);

    // this wave's (slab, channel tile) pairs
    // (pairs dealt evenly: a transposed convolution's parity groups have 1 - 4 slabs = 3 - 12 pairs, and seven per wave in wave order
    //  left two or three of the four waves without work)
    const int npairs = ns * 3;
    const int ppw = __builtin_amdgcn_readfirstlane((npairs + 3) >> 2), pair0 = wave * ppw, pair1 = min(npairs, pair0 + ppw);
    int aoff[kHwPW], dxs[kHwPW], sbi[kHwPW], t2i[kHwPW];
#pragma unroll
    for (int i = 0; i < kHwPW; ++i) {
        const int id = min(pair0 + i, npairs - 1);
        const int sb = id / 3, t2 = id - sb * 3;
        sbi[i] = __builtin_amdgcn_readfirstlane(sb);
        t2i[i] = __builtin_amdgcn_readfirstlane(t2);
        aoff[i] = __builtin_amdgcn_readfirstlane(t2 * 16 * p.xs + (p.dy[slab0 + sb] - p.ymin) * p.hp);
        dxs[i] = __builtin_amdgcn_readfirstlane(p.dx[slab0 + sb] - p.xmin);
    }
    f32x4 acc[kHwPW][3];
#pragma unroll
    for (int i = 0; i < kHwPW; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int prow = (p.hw + 1) >> 1;                       // pixel pairs per halo row
    const int nxt = p.imgs * p.hh * prow * (kHwC / 4);      // X staging tasks: (halo row, pixel pair, channel quad)
    // e / prow and r / hh by reciprocal multiplication (e < kHwXT * 256 = 1280, divisors < 32: exact) -- three runtime integer
    // divisions per staging task, twice per tile, were ~35 instructions each
    const unsigned inv_prow = 65536u / (unsigned)prow + 1u, inv_hh = 65536u / (unsigned)p.hh + 1u;
    float4 xr[OC ? 1 : kHwXT][2], gr[OC ? 1 : kHwGT][2];
    constexpr int kOX = 3, kOG = 2;                          // octet tasks per thread: 256 * 3 >= imgs * hh * prow * 6, 256 * 2 >= 64 * 6
    uint4 xo[OC ? kOX : 1][4], go[OC ? kOG : 1][4];          // [task][px0 hi, px1 hi, px0 lo, px1 lo]
    const int nxo = p.imgs * p.hh * prow * 6;
    auto load_tile = [&](int t) {
        const int tx = t % p.tiles_x;
        const int ty = (t / p.tiles_x) % p.tiles_y;
        const int img0 = (t / (p.tiles_x * p.tiles_y)) * p.imgs;
        const int y0 = ty * TH, x0 = tx * TW;
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        if constexpr (OC) {
            const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int i = 0; i < kOX; ++i) {
                const int e = tid_o + i * 256;
                xo[i][0] = xo[i][1] = xo[i][2] = xo[i][3] = z4;
                if (e < nxo) {
                    const int e2 = (int)(((unsigned)e * 10923u) >> 16);   // e / 6 (e < 768: exact)
                    const int o = e - e2 * 6;
                    int r = (int)(((unsigned)e2 * inv_prow) >> 16);
                    const int pr = e2 - r * prow;
                    const int il = (int)(((unsigned)r * inv_hh) >> 16), hy = r - il * p.hh;
                    const int gy = y0 + p.ymin + hy, gx = x0 + p.xmin + 2 * pr, img = img0 + il;
                    const int c = ci0 + 8 * o;
                    if (img < p.B && gy >= 0 && gy < p.H && c < p.Cx) {
                        const size_t row = ((size_t)(img * p.H + gy) * p.W) * p.XCs + coff + c;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int gxx = gx + h;
                            if (gxx >= 0 && gxx < p.W && 2 * pr + h < p.hw) {
                                xo[i][h] = *reinterpret_cast<const uint4*>(p.Xhi + row + (size_t)gxx * p.XCs);
                                xo[i][2 + h] = *reinterpret_cast<const uint4*>(p.Xlo + row + (size_t)gxx * p.XCs);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < kOG; ++i) {
                const int e = tid_o + i * 256;              // 64 pixel pairs x 6 channel octets = 384 tasks
                go[i][0] = go[i][1] = go[i][2] = go[i][3] = z4;
                const int pp = (int)(((unsigned)e * 10923u) >> 16), o = e - pp * 6;
                const int px = 2 * pp;
                const int il = px >> (p.th_log2 + p.tw_log2);
                const int y = (px >> p.tw_log2) & (TH - 1), x = px & (TW - 1);
                const int img = img0 + il;
                const int co = co0 + 8 * o;
                if (pp < 64 && img < p.B && co < p.Cg) {
                    const size_t at = ((size_t)(img * p.H + y0 + y) * p.W + x0 + x) * p.GCs + co;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        go[i][h] = *reinterpret_cast<const uint4*>(p.Ghi + at + (size_t)h * p.GCs);
                        go[i][2 + h] = *reinterpret_cast<const uint4*>(p.Glo + at + (size_t)h * p.GCs);
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < kHwXT; ++i) {
            const int e = tid_o + i * 256;
            float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
            if (e < nxt) {
                int r = (int)(((unsigned)e * inv_prow) >> 16);
                const int pr = e - r * prow;                  // pixel pair fastest: conflict-free LDS stores
                const int q = r % (kHwC / 4); r /= (kHwC / 4);
                const int il = (int)(((unsigned)r * inv_hh) >> 16), hy = r - il * p.hh;
                const int gy = y0 + p.ymin + hy, gx = x0 + p.xmin + 2 * pr, img = img0 + il;
                const int c = ci0 + 4 * q;
                if (PL && img < p.B && gy >= 0 && gy < p.H && c < p.Cx) {
                    // (planes: 4 halves of hi and of lo per pixel -- the pad channels of a tensor's last octet are zeros)
                    const size_t row = ((size_t)(img * p.H + gy) * p.W) * p.XCs + coff + c;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int gxx = gx + h;
                        if (gxx >= 0 && gxx < p.W && 2 * pr + h < p.hw) {
                            const uint2 vh = *reinterpret_cast<const uint2*>(p.Xhi + row + (size_t)gxx * p.XCs);
                            const uint2 vl = *reinterpret_cast<const uint2*>(p.Xlo + row + (size_t)gxx * p.XCs);
                            const float4 v = make_float4(__uint_as_float(vh.x), __uint_as_float(vh.y), __uint_as_float(vl.x),
                                                         __uint_as_float(vl.y));
                            if (h == 0) v0 = v; else v1 = v;
                        }
                    }
                } else if (!PL && img < p.B && gy >= 0 && gy < p.H && c < p.Cx) {
                    const float* row = p.X + ((size_t)(img * p.H + gy) * p.W) * p.Cxt + coff + c;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int gxx = gx + h;
                        if (gxx >= 0 && gxx < p.W && 2 * pr + h < p.hw) {
                            const float* src = row + (size_t)gxx * p.Cxt;
                            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (p.vecx && c + 3 < p.Cx) v = *reinterpret_cast<const float4*>(src);
                            else {
                                v.x = src[0];
                                if (c + 1 < p.Cx) v.y = src[1];
                                if (c + 2 < p.Cx) v.z = src[2];
                                if (c + 3 < p.Cx) v.w = src[3];
                            }
                            if (h == 0) v0 = v; else v1 = v;
                        }
                    }
                }
            }
            xr[i][0] = v0;
            xr[i][1] = v1;
        }
#pragma unroll
        for (int i = 0; i < kHwGT; ++i) {
            const int e = tid_o + i * 256;              // 64 pixel pairs x 12 channel quads = 768 tasks
            const int pp = e & 63, q = e >> 6;
            const int px = 2 * pp;
            const int il = px >> (p.th_log2 + p.tw_log2);
            const int y = (px >> p.tw_log2) & (TH - 1), x = px & (TW - 1);
            const int img = img0 + il;
            const int co = co0 + 4 * q;
            float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
            if (PL && img < p.B && co < p.Cg) {
                const size_t at = ((size_t)(img * p.H + y0 + y) * p.W + x0 + x) * p.GCs + co;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint2 vh = *reinterpret_cast<const uint2*>(p.Ghi + at + (size_t)h * p.GCs);
                    const uint2 vl = *reinterpret_cast<const uint2*>(p.Glo + at + (size_t)h * p.GCs);
                    const float4 v = make_float4(__uint_as_float(vh.x), __uint_as_float(vh.y), __uint_as_float(vl.x), __uint_as_float(vl.y));
                    if (h == 0) v0 = v; else v1 = v;
                }
            } else if (!PL && img < p.B && co < p.Cg) {
                const float* src = p.G + ((size_t)(img * p.H + y0 + y) * p.W + x0 + x) * p.Cg + co;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float* s2 = src + (size_t)h * p.Cg;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.vecg && co + 3 < p.Cg) v = *reinterpret_cast<const float4*>(s2);
                    else {
                        v.x = s2[0];
                        if (co + 1 < p.Cg) v.y = s2[1];
                        if (co + 2 < p.Cg) v.z = s2[2];
                        if (co + 3 < p.Cg) v.w = s2[3];
                    }
                    if (h == 0) v0 = v; else v1 = v;
                }
            }
            gr[i][0] = v0;
            gr[i][1] = v1;
        }
    };
    auto split_store = [&](_Float16* base, int plane_stride, int chan_stride, int pos, float4 a, float4 b, float sc) {
        if constexpr (PL) {
            // a, b = pixels 2 pr and 2 pr + 1: {hi(c0, c1), hi(c2, c3), lo(c0, c1), lo(c2, c3)} as bit patterns; the LDS image wants, per
            // channel, the pixel pair in one word
            const unsigned ah0 = __float_as_uint(a.x), ah1 = __float_as_uint(a.y), al0 = __float_as_uint(a.z), al1 = __float_as_uint(a.w);
            const unsigned bh0 = __float_as_uint(b.x), bh1 = __float_as_uint(b.y), bl0 = __float_as_uint(b.z), bl1 = __float_as_uint(b.w);
            const unsigned hw[4] = {(ah0 & 0xffffu) | (bh0 << 16), (ah0 >> 16) | (bh0 & 0xffff0000u), (ah1 & 0xffffu) | (bh1 << 16),
                                    (ah1 >> 16) | (bh1 & 0xffff0000u)};
            const unsigned lw[4] = {(al0 & 0xffffu) | (bl0 << 16), (al0 >> 16) | (bl0 & 0xffff0000u), (al1 & 0xffffu) | (bl1 << 16),
                                    (al1 >> 16) | (bl1 & 0xffff0000u)};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                *reinterpret_cast<unsigned*>(base + k * chan_stride + pos) = hw[k];
                *reinterpret_cast<unsigned*>(base + plane_stride + k * chan_stride + pos) = lw[k];
            }
            return;
        }
        const float va[4] = {a.x * sc, a.y * sc, a.z * sc, a.w * sc};
        const float vb[4] = {b.x * sc, b.y * sc, b.z * sc, b.w * sc};
        float big = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) big = fmaxf(big, fmaxf(fabsf(va[k]), fabsf(vb[k])));
        if (!(big < 6.0e4f)) atomicOr(p.overflow, 1);   // also catches NaN
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const _Float16 ha = (_Float16)va[k], hb = (_Float16)vb[k];
            const _Float16 la = (_Float16)(va[k] - (float)ha), lb = (_Float16)(vb[k] - (float)hb);
            *reinterpret_cast<unsigned*>(base + k * chan_stride + pos) = pack_h2(ha, hb);
            *reinterpret_cast<unsigned*>(base + plane_stride + k * chan_stride + pos) = pack_h2(la, lb);
        }
    };
    // an octet task's two pixels -> per channel one word (the pixel pair), hi and lo planes
    auto store_octet = [&](_Float16* base, int plane_stride, int chan_stride, int pos, const uint4 (&v)[4]) {
        const unsigned a[4] = {v[0].x, v[0].y, v[0].z, v[0].w}, b[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
        const unsigned c[4] = {v[2].x, v[2].y, v[2].z, v[2].w}, d[4] = {v[3].x, v[3].y, v[3].z, v[3].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<unsigned*>(base + (2 * j) * chan_stride + pos) = (a[j] & 0xffffu) | (b[j] << 16);
            *reinterpret_cast<unsigned*>(base + (2 * j + 1) * chan_stride + pos) = (a[j] >> 16) | (b[j] & 0xffff0000u);
            *reinterpret_cast<unsigned*>(base + plane_stride + (2 * j) * chan_stride + pos) = (c[j] & 0xffffu) | (d[j] << 16);
            *reinterpret_cast<unsigned*>(base + plane_stride + (2 * j + 1) * chan_stride + pos) = (c[j] >> 16) | (d[j] & 0xffff0000u);
        }
    };
    auto store_tile = [&]() {
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        if constexpr (OC) {
#pragma unroll
            for (int i = 0; i < kOX; ++i) {
                const int e = tid_o + i * 256;
                if (e < nxo) {
                    const int e2 = (int)(((unsigned)e * 10923u) >> 16);
                    const int o = e - e2 * 6;
                    const int r = (int)(((unsigned)e2 * inv_prow) >> 16);   // = il * hh + hy
                    const int pr = e2 - r * prow;
                    store_octet(Xh + (8 * o) * p.xs, kHwC * p.xs, p.xs, r * p.hp + 2 * pr, xo[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < kOG; ++i) {
                const int e = tid_o + i * 256;
                const int pp = (int)(((unsigned)e * 10923u) >> 16), o = e - pp * 6;
                if (pp < 64) store_octet(Gh + (8 * o) * p.gs, kHwC * p.gs, p.gs, 2 * pp, go[i]);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < kHwXT; ++i) {
            const int e = tid_o + i * 256;
            if (e < nxt) {
                int r = (int)(((unsigned)e * inv_prow) >> 16);
                const int pr = e - r * prow;
                const int q = r % (kHwC / 4); r /= (kHwC / 4);   // r = il * hh + hy
                split_store(Xh + (4 * q) * p.xs, kHwC * p.xs, p.xs, r * p.hp + 2 * pr, xr[i][0], xr[i][1], sx);
            }
        }
#pragma unroll
        for (int i = 0; i < kHwGT; ++i) {
            const int e = tid_o + i * 256;
            const int pp = e & 63, q = e >> 6;
            split_store(Gh + (4 * q) * p.gs, kHwC * p.gs, p.gs, 2 * pp, gr[i][0], gr[i][1], sg);
        }
    };

    const bool live1 = co0 + 16 < p.Cg, live2 = co0 + 32 < p.Cg;
    const int t_begin = slice * p.tiles_per_slice;
    const int t_end = min(p.ntiles, (slice + 1) * p.tiles_per_slice);
    if (t_begin < t_end) load_tile(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (t + 1 < t_end) load_tile(t + 1);
        // (not unrolled: the four k-steps of a tile unrolled cost 54 more registers and 10 scratch spills inside this loop -- found by a
        // timing ablation whose run-time loop bound kept hipcc from unrolling: backward pass 4.35 -> 4.08 ms)
#pragma unroll 1
        for (int ks = 0; ks < kWgPix / 32; ++ks) {
            const int p0 = 32 * ks + 8 * kq;
            const int il = p0 >> (p.th_log2 + p.tw_log2);
            const int y = (p0 >> p.tw_log2) & (TH - 1), x = p0 & (TW - 1);
            const _Float16* const gp = Gh + li * p.gs + p0;
            h8v gh[3], gl[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                gh[j] = *reinterpret_cast<const h8v*>(gp + j * 16 * p.gs);
                gl[j] = *reinterpret_cast<const h8v*>(gp + (kHwC + j * 16) * p.gs);
            }
            const _Float16* const xp = Xh + li * p.xs + il * imgplaneP + y * p.hp + x;
#pragma unroll
            for (int i = 0; i < kHwPW; ++i) {
                if (pair0 + i < pair1) {
                    h8v xf[2];
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const _Float16* a = xp + aoff[i] + pl * kHwC * p.xs;
                        const uint4 d = *reinterpret_cast<const uint4*>(a);
                        const unsigned d4 = *reinterpret_cast<const unsigned*>(a + 8);
                        uint4 f;
                        if (dxs[i] == 0) f = d;
                        else if (dxs[i] == 1) {
                            f.x = __builtin_amdgcn_alignbit(d.y, d.x, 16);
                            f.y = __builtin_amdgcn_alignbit(d.z, d.y, 16);
                            f.z = __builtin_amdgcn_alignbit(d.w, d.z, 16);
                            f.w = __builtin_amdgcn_alignbit(d4, d.w, 16);
                        } else f = make_uint4(d.y, d.z, d.w, d4);
                        xf[pl] = __builtin_bit_cast(h8v, f);
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        if (j == 0 || (j == 1 && live1) || (j == 2 && live2)) {
                            f32x4 c = acc[i][j];
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[0], gl[j], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[1], gh[j], c, 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[0], gh[j], c, 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    const size_t slab_sz = (size_t)p.Cx * p.Cg;
#pragma unroll
    for (int i = 0; i < kHwPW; ++i) {
        if (pair0 + i < pair1) {
            float* dst = p.ws + ((size_t)slice * p.nslab + slab0 + sbi[i]) * slab_sz;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int co = co0 + j * 16 + li;
                if (co < p.Cg) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ci = ci0 + t2i[i] * 16 + 4 * kq + r;
                        if (ci < p.Cx) dst[(size_t)ci * p.Cg + co] = acc[i][j][r];
                    }
                }
            }
        }
    }
}

static size_t wgrad_f16_lds(const WgradParams& p) { return sizeof(_Float16) * 2 * kHwC * ((size_t)p.xs + p.gs); }

bool wgrad_setup(WgradParams* p, std::string* why) {
    auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    if (p->nslab < 1 || p->nslab > kWgMaxSlabs) { *why = "wgrad: too many filter taps"; return false; }
    p->vecx = (p->Cxt % 4 == 0);
    for (int s = 0; s < p->nslab; ++s)
        if (p->coff[s] % 4) p->vecx = 0;
    p->vecg = (p->Cg % 4 == 0);
    const int TW = std::min(16, p->W), TH = std::min(8, p->H);
    if ((TW & (TW - 1)) || (TH & (TH - 1)) || p->H % TH || p->W % TW || kWgPix % (TH * TW)) {
        *why = "wgrad: layer size must be a power of two";
        return false;
    }
    p->tw_log2 = lg2(TW);
    p->th_log2 = lg2(TH);
    p->imgs = kWgPix / (TH * TW);
    int ymin = 0, ymax = 0, xmin = 0, xmax = 0;
    for (int s = 0; s < p->nslab; ++s) {
        ymin = std::min<int>(ymin, p->dy[s]); ymax = std::max<int>(ymax, p->dy[s]);
        xmin = std::min<int>(xmin, p->dx[s]); xmax = std::max<int>(xmax, p->dx[s]);
    }
    p->ymin = ymin;
    p->xmin = xmin;
    p->hh = TH + ymax - ymin;
    p->hw = TW + xmax - xmin;
    p->imgplane = p->hh * p->hw;
    // tiny layers (2x2, 4x4): 128 pixel slots would span dozens of images and their halos; use as many images as the
    // staging budget holds and leave the other slots empty (their rows of G are staged as zeros)
    p->imgs = std::max(1, std::min(p->imgs, kWgHaloBig / p->imgplane));
    p->nhalo = p->imgs * p->imgplane;
    p->tiles_y = p->H / TH;
    p->tiles_x = p->W / TW;
    p->ntiles = ((p->B + p->imgs - 1) / p->imgs) * p->tiles_y * p->tiles_x;
    // slab groups: runs of <= 9 consecutive slabs with one channel offset
    p->ngroups = 0;
    for (int s = 0; s < p->nslab;) {
        int n = 1;
        while (s + n < p->nslab && n < kWgNS && p->coff[s + n] == p->coff[s]) ++n;
        p->gstart[p->ngroups] = (short)s;
        p->gcount[p->ngroups] = (short)n;
        ++p->ngroups;
        s += n;
    }
    p->f16 = 0;
    p->thin = 0;
    if (p->Cx <= 4 && p->ngroups == 1 && p->Cg <= 128 && (p->W & (p->W - 1)) == 0) {
        // whole rows per workgroup, ~1024 workgroups, the strip's halo + the partial sums inside 64 KB of LDS
        int R = 1;
        while (R * 2 <= p->H && p->H % (R * 2) == 0 && (long)p->B * p->H / (R * 2) >= 1024) R *= 2;
        WgradParams q = *p;
        for (;; R /= 2) {
            q.thin = R;
            q.hh = R + ymax - ymin;
            q.hw = p->W + xmax - xmin;
            if (wgrad_thin_lds(q) <= 64 * 1024 || R == 1) break;
        }
        if (wgrad_thin_lds(q) <= 64 * 1024) {
            p->thin = q.thin; p->hh = q.hh; p->hw = q.hw;
            p->imgplane = p->hh * p->hw;
            p->nslices = p->B * p->H / p->thin;
            p->ntiles = p->nslices;
            p->tiles_per_slice = 1;
            p->mi = 1;
            return true;
        }
    }
    if (TW >= 8 && p->Cx > 4 && !getenv("UMX_TRAIN_WGRAD_F32")) {
        auto stride_halves = [](int halves) { int dw = (halves + 1) / 2; while (dw % 16 != 8) ++dw; return 2 * dw; };
        p->hp = (p->hw + 7) / 8 * 8;
        p->xs = stride_halves(p->imgs * p->hh * p->hp);
        p->gs = stride_halves(kWgPix);
        const int tasks = p->imgs * p->hh * ((p->hw + 1) / 2) * (kHwC / 4);
        if (tasks <= kHwXT * 256 && wgrad_f16_lds(*p) <= 160 * 1024) {
            p->f16 = 1;
            p->mi = 3;
            const int chunks = ((p->Cx + kHwC - 1) / kHwC) * ((p->Cg + kHwC - 1) / kHwC) * p->ngroups;
            // ~384 workgroups a launch (A/B on the box: 128 -> 1162 images/s at batch 8, 256 -> 1293, 384 -> 1294, 512 -> 1279,
            // 1024 -> 1247, 2048 -> 1172): fewer slices mean fewer prologues / accumulator flushes and a shorter reduce; below
            // one workgroup per CU the launch no longer fills the chip beside the main stream's kernels
            const int target = 384;
            int nslices = std::max(1, std::min(p->ntiles, target / std::max(1, chunks)));
            p->tiles_per_slice = (p->ntiles + nslices - 1) / nslices;
            p->nslices = (p->ntiles + p->tiles_per_slice - 1) / p->tiles_per_slice;
            return true;
        }
    }
    // input-channel tiles per workgroup: fewest padded tiles, each chunk charged one tile for its G traffic
    if (p->nhalo > kWgHaloBig) { *why = "wgrad: halo too large"; return false; }
    const int mi_max = p->nhalo > kWgHalo ? 2 : 3;
    int best = 1;
    double best_cost = 1e300;
    for (int mi = 1; mi <= mi_max; ++mi) {
        const double cost = (double)((p->Cx + 16 * mi - 1) / (16 * mi)) * (mi + 1.0);
        if (cost <= best_cost) { best = mi; best_cost = cost; }
    }
    p->mi = best;
    const int chunks = ((p->Cx + 16 * p->mi - 1) / (16 * p->mi)) * ((p->Cg + kWgCO - 1) / kWgCO) * p->ngroups;
    int nslices = std::max(1, std::min(p->ntiles, 1024 / std::max(1, chunks)));
    p->tiles_per_slice = (p->ntiles + nslices - 1) / nslices;
    p->nslices = (p->ntiles + p->tiles_per_slice - 1) / p->tiles_per_slice;
    const size_t lds = sizeof(float) * ((size_t)p->nhalo * wg_px(p->mi) + (size_t)kWgPix * kWgPG);
    if (lds > 160 * 1024) { *why = "wgrad: halo too large for the LDS"; return false; }
    return true;
}

size_t wgrad_ws_floats(const WgradParams& p) { return (size_t)p.nslices * p.nslab * p.Cx * p.Cg; }

template <int MI, bool BIG>
static hipError_t launch_wgrad_mi(const WgradParams& p, hipStream_t stream) {
    const size_t lds = sizeof(float) * ((size_t)p.nhalo * wg_px(MI) + (size_t)kWgPix * kWgPG);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_mfma_f32<MI, BIG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const unsigned chunks = (unsigned)(((p.Cx + 16 * MI - 1) / (16 * MI)) * ((p.Cg + kWgCO - 1) / kWgCO));
    hipLaunchKernelGGL((wgrad_mfma_f32<MI, BIG>), dim3((unsigned)p.nslices, chunks, (unsigned)p.ngroups), dim3(256), lds,
                       stream, p);
    return hipGetLastError();
}

hipError_t launch_wgrad(const WgradParams& p, hipStream_t stream) {
    if (p.thin > 0) {
        const size_t lds = wgrad_thin_lds(p);
        const dim3 grid((unsigned)p.nslices);
        const bool v4 = p.Cg % 4 == 0;
#define UMX_THIN(CX) \
    if (v4) hipLaunchKernelGGL((wgrad_thin_kernel<CX, 4>), grid, dim3(256), lds, stream, p); \
    else hipLaunchKernelGGL((wgrad_thin_kernel<CX, 1>), grid, dim3(256), lds, stream, p); \
    break;
        switch (p.Cx) {
            case 1: UMX_THIN(1)
            case 2: UMX_THIN(2)
            case 3: UMX_THIN(3)
            case 4: UMX_THIN(4)
            default: return hipErrorInvalidValue;
        }
#undef UMX_THIN
        return hipGetLastError();
    }
    if (p.f16) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_f16x3<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_f16x3<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        160 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_f16x3<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        160 * 1024);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
        const unsigned chunks = (unsigned)(((p.Cx + kHwC - 1) / kHwC) * ((p.Cg + kHwC - 1) / kHwC));
        bool oc = p.planes && p.XCs % 8 == 0 && p.GCs % 8 == 0 && p.imgs * p.hh * ((p.hw + 1) / 2) * 6 <= 3 * 256;
        for (int sb = 0; sb < p.nslab && oc; ++sb) oc = p.coff[sb] % 8 == 0;
        if (oc)
            hipLaunchKernelGGL((wgrad_f16x3<true, true>), dim3((unsigned)p.nslices, chunks, (unsigned)p.ngroups), dim3(256), wgrad_f16_lds(p),
                               stream, p);
        else if (p.planes)
            hipLaunchKernelGGL(wgrad_f16x3<true>, dim3((unsigned)p.nslices, chunks, (unsigned)p.ngroups), dim3(256), wgrad_f16_lds(p),
                               stream, p);
        else
            hipLaunchKernelGGL(wgrad_f16x3<false>, dim3((unsigned)p.nslices, chunks, (unsigned)p.ngroups), dim3(256), wgrad_f16_lds(p),
                               stream, p);
        return hipGetLastError();
    }
    const bool big = p.nhalo > kWgHalo;
    if (p.nhalo > kWgHaloBig || (big && p.mi > 2)) return hipErrorInvalidValue;
    switch (p.mi) {
        case 1: return big ? launch_wgrad_mi<1, true>(p, stream) : launch_wgrad_mi<1, false>(p, stream);
        case 2: return big ? launch_wgrad_mi<2, true>(p, stream) : launch_wgrad_mi<2, false>(p, stream);
        case 3: return launch_wgrad_mi<3, false>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

struct WgReduce {
    const float* ws;
    int nslices, nslab, Cx, Cg, Ctot, c_off;
    float* g;
    const float* w;
    int reg_kind;
    float reg_c;
    float* g2;
    const unsigned* xmax;     // split-precision launches: the operand scales are divided out here
    const unsigned* gmax;
    const float* xinv;        // ... plane-staged launches: the inverse scales of the planes (NULL: 1)
    const float* ginv;
    int planes;
    int f16;
    short mslab[kWgMaxSlabs];
};

// 256 threads = OUT outputs x SP slice partitions (SP a power of two chosen by the host): thread (o, sp) adds slices
// sp, sp+SP, ... in order; the SP partial sums are then added in order by the sp == 0 thread.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const WgReduce q, int SP) {
    __shared__ double sm[256];
    const int OUT = 256 / SP;
    const int o = threadIdx.x % OUT, sp = threadIdx.x / OUT;
    const size_t n = (size_t)q.nslab * q.Cx * q.Cg;
    const size_t e = (size_t)blockIdx.x * OUT + o;
    double s = 0.0;
    if (e < n)
        for (int sl = sp; sl < q.nslices; sl += SP) s += (double)q.ws[(size_t)sl * n + e];
    sm[threadIdx.x] = s;
    __syncthreads();
    if (sp != 0 || e >= n) return;
    for (int j = 1; j < SP; ++j) s += sm[j * OUT + o];
    const int co = (int)(e % q.Cg);
    const size_t r = e / q.Cg;
    const int ci = (int)(r % q.Cx);
    const int sb = (int)(r / q.Cx);
    const size_t dsti = ((size_t)q.mslab[sb] * q.Ctot + q.c_off + ci) * q.Cg + co;
    if (q.f16 && q.planes) s *= (double)(q.xinv ? *q.xinv : 1.f) * (double)(q.ginv ? *q.ginv : 1.f);
    else if (q.f16) s /= (double)wg_scale(q.xmax) * (double)wg_scale(q.gmax);
    float v = (float)s;
    if (q.g2) q.g2[dsti] = v;
    if (q.w && q.reg_kind) v += reg_grad(q.w[dsti], q.reg_kind, q.reg_c);
    q.g[dsti] = v;
}

hipError_t launch_wgrad_reduce(const WgradParams& p, int Ctot, int c_off, float* g, const float* w, int reg_kind,
                               float reg_c, float* g2, hipStream_t stream) {
    WgReduce q;
    q.ws = p.ws; q.nslices = p.nslices; q.nslab = p.nslab; q.Cx = p.Cx; q.Cg = p.Cg; q.Ctot = Ctot; q.c_off = c_off;
    q.g = g; q.w = w; q.reg_kind = reg_kind; q.reg_c = reg_c; q.g2 = g2;
    q.xmax = p.xmax; q.gmax = p.gmax; q.f16 = p.f16;
    q.xinv = p.xinv; q.ginv = p.ginv; q.planes = p.f16 && p.planes;
    for (int s = 0; s < kWgMaxSlabs; ++s) q.mslab[s] = s < p.nslab ? p.mslab[s] : 0;
    const size_t n = (size_t)p.nslab * p.Cx * p.Cg;
    int SP = 1;
    while (SP < 64 && SP * 8 < p.nslices) SP *= 2;
    const int OUT = 256 / SP;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + OUT - 1) / OUT)), dim3(256), 0, stream, q, SP);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) split_reduce_kernel(const float* __restrict__ part, int nsplit, size_t stride,
                                                           size_t n, int act, float* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = part[i];
    for (int k = 1; k < nsplit; ++k) s += part[(size_t)k * stride + i];
    dst[i] = act_of(s, act);
}

hipError_t launch_split_reduce(const float* part, int nsplit, size_t stride, size_t n, int act, float* dst,
                               hipStream_t stream) {
    hipLaunchKernelGGL(split_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, part, nsplit, stride, n,
                       act, dst);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ optimiser
// Adam as tf.train.AdamOptimizer applies it (reference UnMicst1-5.py:371): m, v slots, lr_t = lr*sqrt(1-b2^t)/(1-b1^t)
// formed on the host, w -= lr_t * m / (sqrt(v) + eps).  Momentum as tf.train.MomentumOptimizer (UnMicst.py:279).
// Positions whose gradient is always zero (BN moving statistics) are left untouched by both rules.
__global__ void __launch_bounds__(256) optimizer_kernel(const OptParams o, float* __restrict__ w,
                                                        const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    if (o.kind == 0) {
        const float mi = o.beta1 * m[i] + (1.0f - o.beta1) * gi;
        const float vi = o.beta2 * v[i] + (1.0f - o.beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        w[i] = w[i] - o.lr_t * mi / (sqrtf(vi) + o.eps);
    } else {
        const float mi = o.momentum * m[i] + gi;
        m[i] = mi;
        w[i] = w[i] - o.lr * mi;
    }
}

hipError_t launch_optimizer(const OptParams& o, float* w, const float* g, float* m, float* v, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(optimizer_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, o, w, g, m, v, n);
    return hipGetLastError();
}

}  // namespace umx
