// Internal kernel interface of libumx (gfx950 only). See DESIGN.md for the data layout and roofline of each kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace umx {

constexpr int kWaves = 4;        // waves per workgroup (256 threads)
constexpr int kMT = 4;           // 16-pixel M-tiles per wave
constexpr int kMW = kWaves * kMT;  // 16 M-tiles (256 output pixels) per workgroup
constexpr int kCC = 8;           // input channels staged per K-chunk
constexpr int kTapG = 9;         // filter taps staged per weight stage
constexpr int kMaxTaps = 100;    // flat tap table: two groups of <= 7x7, or 4 transposed-conv phases
constexpr int kMaxNT = 6;        // 16-wide N-tiles per workgroup

struct ConvPhase {
    const float* w[2];   // packed weights per operand group: [ntaps][Cp][Np], zero padded
    int tap0[2];         // first entry of this (phase, group) in ConvParams::tapoff
    int ntaps[2];
    int oy_off, ox_off;  // output offset (transposed-conv sub-pixel phase)
};

// One launch = one convolution-like layer: up to two operand groups (concat-free skip connection, or the
// legacy 1x1 shortcut as an extra K slab), 1 or 4 output phases, fused epilogue.
struct ConvParams {
    const float* src[2];  // NHWC float32 [B,H,W,C]
    int C[2], Cp[2];      // real channels, channels padded to a multiple of 4 (packed-weight K extent)
    int vec4[2];          // 1: C % 4 == 0 -> 16-byte loads
    int ngroups;
    int B, H, W;          // compute grid == source grid
    int Cout, Np;         // real output channels; padded width of the packed weights (multiple of 16)
    int twm_log2, th_log2, nimg_m, imgs;  // M-tile = nimg_m images x 1 row x 2^twm_log2 cols; tile = imgs x TH x TWm
    int hh, hw, imgplane, plane;         // LDS halo geometry (floats); plane = 16 (mod 32)
    int ymin, xmin;                      // halo origin relative to the tile origin
    int tiles_y, tiles_x;                // spatial tiles per image group
    int nphase, o_mul;                   // 1 phase / o_mul 1 (conv) or 4 phases / o_mul 2 (stride-2 transposed conv)
    ConvPhase ph[4];
    short tapoff[kMaxTaps];              // LDS halo offset of each tap: (dy-ymin)*hw + (dx-xmin)
    float* dst;                          // NHWC [B,outH,outW,Cout]
    int outH, outW, pool;                // pool: fused 2x2/2 max-pool (outH = H/2)
    const float* pre_s;                  // epilogue: v = acc*pre_s+pre_b ; act ; v = v*post_s+post_b (NULL = skip)
    const float* pre_b;
    const float* post_s;
    const float* post_b;
    int act;                             // 0 none, 1 ReLU, 2 LeakyReLU(0.2)
    int ksplit;                          // > 1 (training, small batches): the K loop is split over ksplit workgroups by
    size_t split_stride;                 // input-channel chunk; workgroup s writes raw partial sums to dst + s*split_stride
                                         // (no epilogue, no pool) and launch_split_reduce adds them in order
};

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2 };
constexpr float kLeakySlope = 0.2f;   // tf.nn.leaky_relu's default alpha (reference UnMicst1-5.py:114,134,195,198)
constexpr double kBnEpsilon = 0.001;  // tf.layers.batch_normalization's default epsilon
constexpr int kModeReplace = 1;   // == UMX_MODE_REPLACE

size_t conv_lds_bytes(int nt, int plane);
hipError_t launch_d2h_rne_test(const double* in, uint16_t* out, size_t n, hipStream_t stream);   // (test entry: the stitch kernel's double -> binary16)
hipError_t launch_conv(const ConvParams& p, int nt, int hpix, hipStream_t stream);

// ---- split-precision (3 x fp16 MFMA) convolution, umx_conv_f16.hip --------------------------------------------
constexpr int kHaloChunks = 16;  // 64-slot pieces of the LDS halo per plane (halo <= 1024 pixels, the fp32 kernel's limit too)
constexpr int kMaxNT16 = 9;      // N-tiles per workgroup of the split-precision kernel (144 output channels)
constexpr int kStageK = 4;       // max k-steps (of 32 K-slots = 4 (tap, octet) pairs) per weight stage
constexpr int kMaxLdsPerWG = 80 * 1024;   // two workgroups per CU share the 160 KiB
constexpr int kMaxKSplit = 24;   // K-split rows of a trainer launch (HConvParams::ks0)

struct HStage {      // one pipeline stage: optional halo chunk load + one weight block of nk k-steps
    int woff;        // offset (uint4 units) of this stage's weight block inside one (phase, N-block) slab; a block is
                     // a 64-byte header (k-map [nk][4] of LDS slots, unsigned short) + nk * NT * (hi, lo) images
    short group;     // operand group whose halo chunk is loaded with this stage, or -1 (chunk already resident)
    short oct0;      // first octet (8 channels) of that group to load ...
    short noct;      // ... and how many ...
    short nk;        // k-steps in this stage (<= kStageK); F6 form: bit 8 set = "B" stage over the (nk & 0xff) k-steps of the stage before it
    short plane0;    // ... into halo slot plane0 (0 or 1: consecutive chunks alternate between the two slots)
    short phase;     // fused transposed convolution: the sub-pixel phase (accumulator set) this stage feeds
};

struct HPhase {
    const uint4* w;     // packed weights of this phase: [N-block][stage][k-step][N-tile][hi|lo][64 lanes] x 16 B
    int wblk_stride;    // uint4 per N-block
    int stage0, nstages;
    int oy_off, ox_off;
};

struct HConvParams {
    const _Float16* src_hi[2];   // NHWC binary16 [B,H,W,Cs], hi and lo planes
    const _Float16* src_lo[2];
    int Cs[2];                   // stored channels (multiple of 8)
    int srcA[2], srcB[2];        // address form of an operand plane: byte offset of (pixel, octet) inside an image =
                                 // pixel * srcA + octet * srcB -- NHWC: (2 * Cs, 16); octet-planar [octet][pixel][8]: (16, H*W*16)
    int dst_planar;              // the output planes are octet-planar (per image [octet][pixel][8]) instead of NHWC
    int B, H, W;
    int Cout, Cds, NT, nblocks;  // real / stored output channels; N-tiles per workgroup; N blocks
    int twm_log2, th_log2, nimg_m, imgs;
    int hh, hw, imgplane, nhalo; // halo geometry in pixels; nhalo = imgs * imgplane
    int plane_slots;             // halo pixels per LDS slot image, rounded up to a multiple of 16
    int OC, pix_bytes, slot_bytes;   // octets per staged pixel (odd), OC*16, bytes of one halo slot's hi image
    float inv_OC;
    int PP, nact, ninst;           // halo chunk = ninst LDS-DMA pieces of PP = 64/OC pixels x OC octets (nact = PP*OC lanes)
    int piece_bytes;               // PP*OC*16: LDS bytes one piece fills (the image stays dense: slot = pixel*OC + octet)
    int inv_oc_q16;                // 65536/OC + 1: lane / OC == (lane * inv_oc_q16) >> 16 for lane < 64
    int kmt;                       // M-tiles per wave: 4 (plain kernel) or 2 (fused transposed convolution)
    int xcd_order;                 // 1: workgroup ids are re-ordered so that each XCD (ids = x mod 8) walks a contiguous range of tiles;
    int ntiles_grid, tiles_per_xcd;   // 2: the same ranges on a one-dimensional grid with (N-block, phase) fastest inside an XCD
    int maxp;                      // halo pieces per wave and chunk the launched instantiation indexes: 4 or 12
    int lo_off, b_off, lds_bytes;  // LDS byte offsets: lo planes, weight buffers; total dynamic LDS
    int wbuf_bytes;                // one weight buffer (there are two): 64 + S * NT * 2048
    int ymin, xmin, tiles_y, tiles_x;
    int nphase, o_mul;
    int fused_phases;            // 1: stride-2 transposed convolution, all 4 phases per workgroup (KMT = 2, ph[0] only)
    HPhase ph[4];
    const HStage* stages;
    const uint4* zeros;          // >= 16 bytes of zeros in global memory (source for out-of-image halo slots)
    _Float16* dst_hi;
    _Float16* dst_lo;
    const unsigned char* app_c;  // non-NULL: word `app_word` (2 channels) of the stored octet `app_oct` is replaced by the channels of
    int app_cw;                  // this compact tensor ([pixel]{hi[app_cw] | lo[app_cw]}, app_cw = 1 or 2; the output's pixel grid):
    int app_oct, app_word;       // the raw-input skip rides in the spare channels of the up-sampled tensor
    float* dst_f32;              // non-NULL: write fp32 NHWC [..,Cout] instead of the (hi, lo) pair
    float* probs;                // head_K > 0: fused 1x1 conv + BN affine + softmax head, probabilities NHWC [..,head_K]
    int head_K;                  // (needs nblocks == 1: every channel of a pixel in one workgroup)
    const uint4* head_frag;      // the 1x1 head as MFMA A-fragments: [ceil(NT/2) k-steps][hi | lo][64 lanes] x 16 B, rows = classes
    int outH, outW, pool;
    const uint4* econst;         // epilogue constants per N-block: [pre_s | pre_b | post_s | post_b] x NT*16 floats
                                 // (+ with a fused head: head_K rows of NT*16 head weights, then [scale | bias] x 8);
                                 // pre_s = BN scale (or 1) * 2^-(weight shift + activation shift), post_* carry the
                                 // 2^(activation shift) of the output; everything 0 for padded channels
    float inv_imgplane, inv_hw;  // 1 / imgplane, 1 / hw
    int act;
    int post_affine;             // 0: post_s == 1 and post_b == 0 on every real channel (the epilogue skips the second affine)
    int d2s;                     // 1: depth-to-space transposed convolution (conv_f16x3's D2S form; blockIdx.z = N-block of phases):
    int d2s_mix;                 //    the last N-tile is the remainder tile (lane group q = phase slot q).  Behind the epilogue
                                 //    constants, 64 ints: per block z [2 NT + 8]: element offset of (sub-pixel phase, destination
                                 //    octet) relative to output pixel (2y, 2x) per stored octet of the N axis (< 0: padding), then the
                                 //    same for the remainder tile's 4 lane groups, then their sub-pixel codes oy * 2 + ox
    int w2;                      // 1: conv_f16x3's W2 form (two tiles per eight-wave workgroup, one workgroup per CU); set with f6 as well
    int f6;                      // 1: conv_f16x3's F6 form -- stages alternate between "A" (x_hi * w_hi, hi weight images only) and "B"
                                 //    (HStage::nk bit 8: one block-scaled fp6 MFMA per tile pair for both cross terms of the A stage's k-steps)
    int pk;                      // 1: the last N-tile's weight image is [w_hi | w_lo] of its <= 8 real channels (conv_f16x3's PK form)
    int ksplit;                  // > 1 (trainer, fp32 output): the stage list is cut into `ksplit` runs of whole halo chunks, run s by workgroup
    short ks0[4][kMaxKSplit + 1];   //   rows [s * nblocks, (s + 1) * nblocks): stages [ks0[z][s], ks0[z][s + 1]) of phase z, relative to its
    size_t split_stride;         //      stage0; partial sums of run s at dst_f32 + s * split_stride
    const float* dyn[2];         // fp32 output only (the trainer's launches): non-NULL device scalars multiplied into pre_s when the
                                 // epilogue runs -- the power-of-two scales of this step's repacked weights and of a gradient tensor's
                                 // (hi, lo) planes, both chosen on the device
    int* overflow_flag;
    long long* dbg;              // diagnostic builds only (UMX_DEBUG_STAMPS): per-workgroup s_memtime segments, or NULL
};

hipError_t launch_conv_f16(const HConvParams& p, hipStream_t stream);

// ---- first down-sampling layer, umx_conv_first.hip: the K extent of the layer (taps x input channels: 9 x 2 = 18 for the duo
// models, 9 for solo, 25 for the legacy 5 x 5 graph) packed DENSELY into 1-2 MFMA k-steps instead of one octet (8 channel
// slots, 1-2 of them real) per tap; weights live in registers, a workgroup owns a 16 x (16..64)-pixel region of one tile
struct TileGeom {
    int H, W;            // full image size
    int P, margin, sub;  // patch size, margin, sub-patch
    int npr, npc;        // patch rows / cols
};

struct FirstParams {
    const _Float16* src_hi;   // the input tiles as (hi, lo) binary16 NHWC planes with 8 stored channels [B,P,P,8] ...
    const _Float16* src_lo;
    const unsigned char* src_c;   // ... or (non-NULL) in the compact form [B,P,P]{hi[CW] | lo[CW]}: 4 * CW bytes per pixel, exactly
                                  // the kernel's LDS pixel format (written by the gather / split kernels for this consumer)
    int B, P, Ci;
    int ks, ntaps;            // ks x ks taps, SAME padding
    int NT, CW, NKS;          // N-tiles of 16 output channels; channel slots per tap (1, 2, 4); k-steps (K = ntaps * CW <= 32 * NKS)
    int rw_log2;              // region = 16 rows x 2^rw_log2 columns of the tile per workgroup
    int hh, hw;               // halo of a region (pixels)
    float inv_hw;
    int lds_bytes;
    const uint4* w;           // MFMA A-fragments [NKS][NT][hi | lo][64 lanes] x 16 B, rows = output channels
    const float* econst;      // [pre_s | pre_b | post_s | post_b] x NT*16 (HConvParams::econst of the same layer)
    int act, post_affine;
    _Float16* dst_hi;         // pooled output, (hi, lo) planes, NHWC or octet-planar
    _Float16* dst_lo;
    int Cds, dst_planar, outS;
    int* overflow_flag;
};
bool conv_first_supported(int NT, int CW, int NKS);
hipError_t launch_conv_first(const FirstParams& p, hipStream_t stream);

// (cw > 0: the compact form [pixel]{hi[cw] | lo[cw]} into `hi` instead of the two 8-channel planes)
hipError_t launch_split_f32(const float* x, size_t npix, int C, int Cs, float scale, _Float16* hi, _Float16* lo, int cw,
                            hipStream_t stream);

hipError_t launch_gather_split(const void* image, int raw_bits, int C_img, int band_row0, int band_rows, const TileGeom& g, int Cn,
                               double mean, double stdv, int tile0, int ntiles, float scale, _Float16* hi, _Float16* lo, int cw,
                               hipStream_t stream, const unsigned* mm = nullptr /* raw sources: rescale to the planes' (min, max) words */);

hipError_t launch_gather_normalise(const double* image, int C_img, int band_row0, int band_rows, const TileGeom& g,
                                   int Cn, double mean, double stdv, int tile0, int ntiles, float* tiles,
                                   hipStream_t stream);

hipError_t launch_head_softmax(const float* x, size_t npix, int C, int K, const float* w /*[C][K]*/,
                               const float* scale, const float* bias, float* probs, hipStream_t stream);

constexpr int kStitchU8 = 2;   // (internal; beside UMX_STITCH_FP16_COMPAT = 0 / UMX_STITCH_FP32 = 1) the fp16-compat result as the drivers' uint8
hipError_t launch_stitch(const float* probs, int tpr0, int tpr1, const TileGeom& g, int K, int mode, int stitch,
                         int y0, int y1, void* out, hipStream_t stream, int plane_rows = 0 /* rows per class plane of `out`; 0: y1 - y0 */);

// driver-side pre/post-processing at scalingFactor 1 (raw integer planes in, uint8 probability planes out)
hipError_t launch_raw_to_double(const void* raw, int bits, size_t n, int rescale, unsigned* mm /*2 words*/, double* out,
                                hipStream_t stream);
hipError_t launch_minmax_init(unsigned* mm, hipStream_t stream);
hipError_t launch_minmax(const void* raw, int bits, size_t n, unsigned* mm, hipStream_t stream);
hipError_t launch_raw_convert(const void* raw, int bits, size_t n, int rescale, const unsigned* mm, double* out,
                              hipStream_t stream);
hipError_t launch_half_to_u8(const void* pm_half, size_t n, unsigned char* out, hipStream_t stream);
// skimage.transform.resize on the device (float64; umx_kernels.hip): separable Gaussian, range reduce, order-1 zoom + clip
hipError_t launch_gauss1d(const double* src, double* dst, int H, int W, int axis, int radius, const double* w_dev,
                          hipStream_t stream);
hipError_t launch_minmax_f64(const double* x, size_t n, unsigned long long* mm64, hipStream_t stream);
hipError_t launch_zoom1(const double* src, int H, int W, int h, int w, const unsigned long long* clip, double* dst,
                        unsigned char* out_u8, hipStream_t stream);
hipError_t launch_percentile_f64(const double* x, size_t n, double q, unsigned long long* st, unsigned* hist,
                                 unsigned long long* mm64, hipStream_t stream);
hipError_t launch_rescale_f64(double* x, size_t n, const unsigned long long* mm64, hipStream_t stream);
hipError_t launch_half_to_u8_f64(const void* pm_half, size_t n, double* out, hipStream_t stream);


// ---- training step (umx_train_kernels.hip): fp32 everywhere, reductions in fp64 with a fixed summation order ----------
constexpr int kMaxPackTaps = 25;

// One packed operand of the fp32 conv kernel, rebuilt on the device from the master tensors every step:
//   packed[t][c][n] ([ntaps][Cp][Np], zero outside C x N);  (par, c') = (c / Cblk, c % Cblk);  m = mtap[t*npar + par]
//   transpose 0:  w[m][c_off + c'][n]        forward conv group / space-to-depth input-gradient of a transposed conv
//   transpose 1:  w[m][c_off + n][c']        input-gradient of a conv (channel slice c_off..) / transposed conv phase
//   (+ w2 at the same index when non-NULL: main filter + shortcut filter of a v2 down layer)
struct PackDesc {
    float* dst;
    const float* w;
    const float* w2;
    int ntaps, Cp, Np, C, N;
    int d2, d3;            // master tensor [taps][d2][d3]
    int transpose, c_off;
    int npar, Cblk;
    int bwd;               // (host bookkeeping) an operand only the backward pass reads: packed on the side stream under the forward pass
    short mtap[4 * kMaxPackTaps];
};
hipError_t launch_pack_weights(const PackDesc* descs_dev, int ndesc, size_t max_elems, hipStream_t stream);

// per-channel sums over the rows of an NHWC tensor [N][C]: part[blk][2][C] = (sum, sum of squares), fp64
int chan_blocks(size_t N, int C);
hipError_t launch_chan_stats(const float* x, size_t N, int C, double* part, int nblk, hipStream_t stream);
// batch statistics -> stat[4][C] = mean | rstd | scale = gamma*rstd | shift = beta - mean*scale; moving statistics
// updated in place (momentum; the unbiased variance feeds the moving variance, like TF's fused kernel)
// (smax, nullable: max_c |gamma * rstd| as float bits, atomicMax)
hipError_t launch_bn_finalize(const double* part, int nblk, size_t N, int C, const float* gamma, const float* beta,
                              float* mov_mean, float* mov_var, float momentum, float* stat, unsigned* smax, hipStream_t stream);

// inference-mode statistics: stat from the moving averages (tf.layers.batch_normalization(training=False))
hipError_t launch_bn_stat_from_moving(int C, const float* gamma, const float* beta, const float* mov_mean,
                                      const float* mov_var, float* stat, hipStream_t stream);
// probs = softmax(t0*scale + shift)
hipError_t launch_softmax_only(const float* t0, const float* stat, size_t N, int K, float* probs, hipStream_t stream);

struct ActParams {          // y = dropout(act(z*scale + shift)) [-> 2x2 max-pool]
    const float* z;         // [B,H,W,C] pre-BN conv output
    const float* stat;      // [4][C]
    int B, H, W, C;
    int pool, act;
    float drop_rate;
    unsigned long long drop_key;
};
// (hi / lo / Cs, nullable: the output also as (hi, lo) binary16 NHWC planes with Cs stored channels; overflow: range flag)
hipError_t launch_act_fwd(const ActParams& a, float* out, unsigned* omax /* max |out| word or NULL */, _Float16* hi, _Float16* lo,
                          int Cs, int* overflow, hipStream_t stream);
// omax = max(omax, max |x|)  (float bits; integer atomicMax)
hipError_t launch_absmax(const float* x, size_t n, unsigned* omax, hipStream_t stream);
// ---- forward / input-gradient convolutions of the trainer on conv_f16x3: weight images rebuilt on the device every step, and
// fp32 tensors turned into the (hi, lo) planes the kernel reads (umx_train_kernels.hip)
struct HWRefDev { int dst; int base; unsigned short nvalid, arr; };   // == umx::HWRef (umx_internal.h)
struct RepackDesc {
    const HWRefDev* refs;
    int n;                  // units to fill
    const float* arr[2];    // fp32 operands of the launch's groups for this phase: [tap][Cp][Np]; or (direct) the groups' master tensors
    const float* arr2[2];   // (direct, nullable) a second master added element for element (a block's 3x3 filter + its shortcut's)
    int stride;             // Np: elements between consecutive input channels
    int estride[2];         // (direct) the same per group inside the master tensor; 0 = the packed operand (stride above)
    float scale;            // 2^s applied to the weights (the launch's HConvParams::dyn[0] points at 2^-s)
    uint4* slab;            // the phase list's weight slab (headers stay as the planner wrote them)
    int bwd;                // (host bookkeeping) as PackDesc::bwd
    int owner;              // (host bookkeeping) the trainer's convolution this slab belongs to: whose scale `scale` follows
};
hipError_t launch_repack_f16x3(const RepackDesc* descs_dev, int ndesc, int max_n, int* overflow, hipStream_t stream);
hipError_t launch_split_dyn(const float* x, size_t npix, int C, int Cs, const unsigned* maxw, float* inv_scale, _Float16* hi,
                            _Float16* lo, int* overflow, unsigned* omax /* max |v| word or NULL */, hipStream_t stream);
// backward of the same: g = d(loss)/d(BN output) written full-res [B,H,W,C]; part[blk][2][C] = (sum g, sum g*xhat)
// (gx, nullable: two words, max |g| and max |xhat| as float bits)
hipError_t launch_act_bwd(const ActParams& a, const float* dy0, const float* dy1, float* g, double* part, int nblk,
                          unsigned* gx, hipStream_t stream);
// sums -> m12[2][C] = (mean g, mean g*xhat); dgamma = sum g*xhat, dbeta = sum g
hipError_t launch_bn_bwd_finalize(const double* part, int nblk, size_t N, int C, float* dgamma, float* dbeta, float* m12,
                                  hipStream_t stream);
// (bn_bwd_apply: g <- scale * (g - m1 - xhat*m2) in place, and leaky_bwd_s2d: gS[b,i,j,(pa*2+pb)*C + c] =
//  d_us[b,2i+pa,2j+pb,c] * (us > 0 ? 1 : 0.2), are declared with the weight-gradient kernels below: they track max |v|)

// top layer: t0 = x W (1x1 conv, K <= 8 classes)
hipError_t launch_head_fwd(const float* x, size_t N, int C, int K, const float* w, float* t0, hipStream_t stream);
// p = softmax(t0*scale + shift); loss partials; dt = d(mean loss)/dt;  part[blk] fp64
int loss_blocks(size_t N);
hipError_t launch_softmax_loss(const float* t0, const float* stat, const float* labels, const float* weights, size_t N,
                               int K, float clip_eps, float* probs, float* dt, double* part, int nblk, hipStream_t stream);
// dx = dt0 W^T ; part[blk][C][K] = sum_p x[p][c] dt0[p][k]
hipError_t launch_head_bwd(const float* x, const float* dt0, const float* w, size_t N, int C, int K, float* dx,
                           double* part, int nblk, hipStream_t stream);
// dst[i] = scale * sum_b part[b][i] (+ reg'(w[i])) ; fixed order
hipError_t launch_reduce_partials(const double* part, int nblk, int n, double scale, float* dst, const float* w,
                                  int reg_kind, float reg_c, hipStream_t stream);
// out[slot] (+)= scale * sum part[0..n)
hipError_t launch_sum_to_scalar(const double* part, int n, double scale, double* out, int slot, int accumulate,
                                hipStream_t stream);
// out[slot] += sum over the segments of coef * (sum |w| (kind 1) or sum w^2 (kind 2)); part: 64 doubles per segment
struct RegSeg { const float* w; size_t n; float coef; };
hipError_t launch_reg_loss(const RegSeg* segs_dev, int nseg, int kind, double* part, double* out, int slot,
                           hipStream_t stream);

// weight gradient of a convolution on the fp32 matrix cores:  dW[s][ci][co] = sum_p X[p + (dy_s,dx_s)][coff_s + ci] * G[p][co]
constexpr int kWgMaxSlabs = 32;
struct WgradParams {
    const float* X; int Cxt;      // NHWC, Cxt channels per pixel
    const float* G; int Cg;       // NHWC [B,H,W,Cg]
    int B, H, W;
    int nslab, Cx;
    short dy[kWgMaxSlabs], dx[kWgMaxSlabs];
    short mslab[kWgMaxSlabs];     // slab -> tap index of the master tensor (row block of dW this slab produces)
    int coff[kWgMaxSlabs];
    int ngroups;                  // slabs are processed in groups of <= 9 that share one channel offset
    short gstart[kWgMaxSlabs], gcount[kWgMaxSlabs];
    int tw_log2, th_log2, imgs;   // pixel tile = imgs x TH x TW = 128 pixels
    int hh, hw, imgplane, nhalo, ymin, xmin, tiles_y, tiles_x;
    int ntiles, tiles_per_slice, nslices;
    int mi;                       // input-channel tiles (of 16) per workgroup: 1..3
    int vecx, vecg;               // 16-byte loads allowed
    float* ws;                    // [slice][slab][Cx][Cg]
    // split-precision variant (wgrad_f16x3): 48 x 48 channel tiles, operands staged as (hi, lo) binary16 planes
    int f16;                      // 1: use it (W >= 8, Cx > 4, <= 9 slabs per group); 0: fp32 MFMA kernel
    int hp, xs, gs;               // halo row pitch, X / G channel strides in LDS (halves)
    const unsigned* xmax;         // device word holding max|X| as float bits, written by the tensor's producer
    const unsigned* gmax;         // same for G
    int* overflow;                // set when a scaled operand leaves the binary16 range
    // plane-staged variant (planes = 1, f16 only): the operands as (hi, lo) binary16 NHWC planes with XCs / GCs stored channels
    // (multiples of 8, pad channels zero) and the inverse of the power-of-two scale each was stored with (NULL: unscaled)
    int thin;                     // > 0: wgrad_thin_kernel (Cx <= 4, one slab group): rows of the image per workgroup
    int planes;
    const _Float16 *Xhi, *Xlo, *Ghi, *Glo;
    int XCs, GCs;
    const float *xinv, *ginv;
};
// BN input gradient g <- scale * (g - m1 - xhat*m2) (in place) and LeakyReLU backward + space-to-depth of the transposed
// conv's output gradient; both track max |v| of what they write (bit pattern of a non-negative float, atomicMax on uint;
// gmax may be NULL)
// (hi != NULL: dz also as (hi, lo) planes scaled by the power of two the bound bw[0] * bw[1] * (2 + bw[2]) -- max |gamma rstd|,
//  max |g|, max |xhat| -- asks for; *inv_scale receives the inverse factor)
hipError_t launch_bn_bwd_apply_max(float* g, const float* z, const float* stat, const float* m12, size_t N, int C,
                                   unsigned* gmax, const unsigned* bw, float* inv_scale, _Float16* hi, _Float16* lo, int Cs,
                                   int* overflow, hipStream_t stream);
hipError_t launch_leaky_bwd_s2d_max(const float* d_us, const float* us, int B, int S, int C, float* gS, unsigned* gmax,
                                    hipStream_t stream);
bool wgrad_setup(WgradParams* p, std::string* why);   // fills geometry, slab groups and slices from B,H,W,Cx,Cg,nslab,dy,dx,coff
size_t wgrad_ws_floats(const WgradParams& p);
hipError_t launch_wgrad(const WgradParams& p, hipStream_t stream);
// g[(s*Ctot + c_off + ci)*Cg + co] = sum_slices ws (+ reg'(w)) ; optional second destination g2 (no reg) for the pair
hipError_t launch_wgrad_reduce(const WgradParams& p, int Ctot, int c_off, float* g, const float* w, int reg_kind,
                               float reg_c, float* g2, hipStream_t stream);

struct OptParams {
    int kind;                 // 0 Adam, 1 Momentum
    float lr, lr_t, beta1, beta2, eps, momentum;
};
// dst[i] = act(sum_s part[s*stride + i]) : the second half of a K-split convolution
hipError_t launch_split_reduce(const float* part, int nsplit, size_t stride, size_t n, int act, float* dst,
                               hipStream_t stream);
hipError_t launch_optimizer(const OptParams& o, float* w, const float* g, float* m, float* v, size_t n, hipStream_t stream);

__host__ __device__ inline uint16_t double_to_half_rne(double d);

}  // namespace umx

// ---- correctly rounded (round-to-nearest-even) double -> IEEE binary16, the conversion numpy performs when a
// float64 result is stored into a float16 array (reference PartitionOfImage.py:95-98: `Output[...] += P*W`).
__host__ __device__ inline uint16_t umx::double_to_half_rne(double d) {
    union { double f; uint64_t u; } v;
    v.f = d;
    const uint64_t u = v.u;
    const uint16_t sign = (uint16_t)((u >> 48) & 0x8000u);
    const int e = (int)((u >> 52) & 0x7FF);
    const uint64_t mant = u & 0xFFFFFFFFFFFFFull;
    if (e == 0x7FF) return (uint16_t)(sign | 0x7C00u | (mant ? 0x200u : 0u));  // inf / nan
    const int eh = e - 1023 + 15;  // biased half exponent
    if (eh >= 31) return (uint16_t)(sign | 0x7C00u);  // overflow -> inf (values >= 65520 after rounding handled below)
    if (eh <= 0) {
        // subnormal half or zero: value = 1.mant * 2^(e-1023); half subnormal unit = 2^-24
        if (eh < -10) return sign;  // < 2^-25: rounds to zero (2^-25 exactly is a tie -> even = 0; slightly above handled below)
        const uint64_t full = mant | (1ull << 52);  // 53-bit significand
        const int shift = 52 - 10 + (1 - eh);       // bits to drop so that the unit becomes 2^-24
        const uint64_t kept = full >> shift;
        const uint64_t rem = full & ((1ull << shift) - 1);
        const uint64_t half = 1ull << (shift - 1);
        uint64_t r = kept;
        if (rem > half || (rem == half && (kept & 1))) r += 1;
        return (uint16_t)(sign | (uint16_t)r);  // may carry into the exponent field: that is the correct normal
    }
    uint64_t kept = mant >> 42;  // top 10 mantissa bits
    const uint64_t rem = mant & ((1ull << 42) - 1);
    const uint64_t half = 1ull << 41;
    uint32_t h = ((uint32_t)eh << 10) | (uint32_t)kept;
    if (rem > half || (rem == half && (kept & 1))) h += 1;  // carry propagates into the exponent correctly
    if (h >= 0x7C00u) h = 0x7C00u;
    return (uint16_t)(sign | h);
}
