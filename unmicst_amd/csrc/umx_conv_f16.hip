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
//   LDS          input halo, pixel-major: [halo pixel][OC octets] of 16-byte slots (OC odd: an activation-fragment read of
//                16 consecutive pixels at one octet is conflict-free), a hi image then a lo image per halo slot, two
//                slots; then the two weight buffers.
// Staging is LDS-DMA (buffer_load_dwordx4 .. lds: no staging VGPRs, out-of-range lanes write the zero padding) in a two-deep pipeline: two weight buffers and two halo
// slots, the loads of stage s+1 in flight under the MFMAs of stage s, one barrier per stage; 2-3 workgroups per CU.  A k-step is any 4 (tap, octet) pairs (table built on the host), so channel counts
// only need to be multiples of 8, not 32.
#include <type_traits>

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

#define UMX_GLDS16(gptr, lptr)                                                                        \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),           \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h32 __attribute__((ext_vector_type(32)));

// ---- F6 form: the two CROSS terms of the split product on the block-scaled matrix instruction ------------------------------------
// x*w = x_hi*w_hi + [x_hi*w_lo + x_lo*w_hi]: the bracket is a 2^-11 correction, so it needs ~11 fewer significant bits than the leading
// term -- OCP MX fp6 (e2m3: 4 significant bits, one power-of-two e8m0 scale per 32 consecutive K elements) holds the tolerance on the
// layers at <= 1/4 of the input resolution (tests/fp8_cross_term_report.py: 2.2e-5 on the reference's trained weights, 3e-6 on random
// graphs, against 1e-4).  v_mfma_scale_f32_16x16x128_f8f6f4 with e2m3 operands runs 4 x the K of a binary16 MFMA in the same 16
// cycles: ONE instruction whose K axis is [32 K of k-step j | 32 K of k-step j+1] x [x_hi*w_lo | x_lo*w_hi] replaces the four
// binary16 cross-term MFMAs of two k-steps -- 1.5 units of matrix time per k-step instead of 3 (tools/probes/mfma_loops.hip:
// the bare mixed trip runs 1.71 x the 3-product trip).  Weights are quantised once by the planner; the pixel operand is quantised
// on the fly from the (hi, lo) halo already in LDS: lane (pixel, q) reads the four (tap, octet) units of k-step j + (q & 1) from the hi
// (q < 2) or lo plane, takes the block's exponent from its largest magnitude and converts all 32 values with one
// v_cvt_scalef32_pk32_fp6_f16 (semantics probed: tools/probes/mx_fp6_semantics.hip).
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

// NPH = 1: one output phase per workgroup (plain convolutions; transposed convolutions one sub-pixel phase at a time,
//          phase = blockIdx.z), KMT = 4 M-tiles per wave (256 pixels per workgroup: a 16 x 16 tile).
// NPH = 4: stride-2 transposed convolution with all four sub-pixel phases in one workgroup: the input halo is loaded
//          once instead of four times, every stage carries the phase whose accumulators it feeds, KMT = 2 (128 input
//          pixels -> 512 output pixels per workgroup), and the epilogue interleaves the phases so that whole output
//          rows (32 consecutive pixels) are stored contiguously.
// DBG = true: the same kernel with in-kernel s_memtime stamps (UMX_DEBUG_STAMPS); the product build carries none of it.
// MAXP: halo pieces per wave and chunk the kernel keeps a pixel index for (4 or 12; the planner picks the smallest that holds
//       its chunking -- 12 costs 8 vector registers, i.e. a wave per SIMD on the narrow kernels).
constexpr int conv_f16x3_waves(int NT, int KMT, int NPH, int MAXP, bool D2S = false) {   // resident waves per SIMD the register budget is set for
    return NPH != 1 || D2S ? 2 : (NT <= 3 && MAXP == 4) ? 4 : (NT <= 3 || (NT <= 5 && MAXP == 4)) ? 3 : 2;
}
// PK = true: the LAST N-tile holds <= 8 real output channels and is packed [w_hi of channels 0..7 | w_lo of channels 0..7] in its
//       16 rows (one weight image): x_hi and x_lo against it are 2 MFMAs instead of 3 (rows 0..7 collect w_hi.x_hi + w_hi.x_lo,
//       rows 8..15 w_lo.x_hi + the 2^-22 term w_lo.x_lo the 3-product form drops), and rows j, j + 8 -- lanes l, l + 32 -- are
//       added once, after the K loop (v_permlane32_swap).  36 channels: 8 MFMAs per (M-tile, k-step) instead of 9, 72: 14 of 15.
// D2S = true: a stride-2 transposed convolution in its depth-to-space form (umx_plan.hip, make_d2s): ONE plain convolution over the
//       union of the phases' taps whose N axis is [phase][channel] -- 4 x 36 channels are 9 full N-tiles instead of 4 x 3 --
//       with KMT = 4 and no phase loop; only the epilogue differs: a stored octet's phase (sub-pixel offset) and destination
//       octet come from a per-octet table, and an optional last "remainder" tile carries <= 4 channels of each phase in lane
//       group q = phase slot, stored without the set exchange together with the appended raw-skip channels.
// F6 = true: the cross terms run on the block-scaled fp6 matrix instruction (above).  A stage = up to two k-steps of x_hi * w_hi alone
//       (hi weight images only, 1 KiB per k-step and N-tile) followed by ONE scaled MFMA per (M-tile, N-tile) that adds both cross terms
//       of those k-steps (per N-tile a 2-KiB image of [64 lanes][24 bytes of e2m3 | scale byte] in two 16-byte planes): a 36-KiB weight
//       block per stage.  Halving the matrix time of a stage exposes what it was hiding -- a stage cannot be shorter than the L2 -> LDS
//       latency of the weight block issued one stage ahead (timing-only ablations, docs/experiments.md: with 2 / 3 of the matrix work
//       removed the 4-wave kernel was 12 % faster) -- so the F6 form keeps the matrix time per stage where it was by doubling the K per
//       stage, and pays for the 36-KiB blocks with the whole LDS of a CU: ONE workgroup of EIGHT waves per CU, waves 0..3 on tile 2 b,
//       waves 4..7 on tile 2 b + 1 (each half with its own halo slots), both halves fed by the SAME weight stream -- half the L2 -> LDS
//       weight traffic per tile as well.
// W2 = true: ONE workgroup of EIGHT waves per CU over TWO x-adjacent tiles (waves 0..3 on tile 2 b, waves 4..7 on tile 2 b + 1, each half
//       with its own halo slots), both fed by the same weight stream, with the CU's whole LDS: half the L2 -> LDS weight bytes per tile,
//       and room for two k-steps per stage (half the barriers) where the four-wave form holds one.  The F6 form implies it.
template <int NT, int KMT, int NPH, bool DBG = false, int MAXP = 4, bool PK = false, bool D2S = false, bool F6 = false, bool W2 = F6>
__global__ void __launch_bounds__(W2 ? 512 : 256, conv_f16x3_waves(NT, KMT, NPH, MAXP, D2S)) conv_f16x3(const HConvParams p) {
    static_assert(!F6 || (NPH == 1 && !PK && !D2S && !DBG && W2), "the fp6 cross-term form runs on the plain / per-phase kernel (no stamped twin: it spills 426 registers)");
    static_assert(!W2 || (NPH == 1 && !D2S && !DBG), "two tiles per workgroup: the plain / per-phase kernel");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    constexpr int kNW = W2 ? 2 * kWaves : kWaves;                       // waves that share a weight stream
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = W2 ? wave_all >> 2 : 0;                            // (W2) which of the workgroup's two tiles this wave works on
    const int wave = W2 ? wave_all & 3 : wave_all;                      // the wave's place inside its tile: every geometry below
    const int q = lane >> 4;    // which 8-wide k group of the 16x16x32 MFMA this lane feeds
    const int li = lane & 15;   // pixel (A) / output channel (B, C/D) inside the tile

    // ---- workgroup -> (image group, spatial tile), N block, phase
    const int TWm = 1 << p.twm_log2, TH = 1 << p.th_log2;
    int bid = blockIdx.x, nblk_i = blockIdx.y, z_i = NPH == 1 ? blockIdx.z : 0;
    if (p.xcd_order == 2) {
        // One-dimensional grid, (N-block, phase) fastest inside an XCD: the workgroups that read the SAME input tile run on one
        // XCD at the same time and share its halo through that L2.  For the layers of many N-blocks x phases over small images
        // (the solo model's 4 x 4 and 8 x 8-pixel transposed convolutions: 32 and 16 workgroups per tile) the x-fastest order
        // keeps 16 - 64 different tiles resident per XCD, nothing is shared, and the layer runs at the fabric's 6 - 6.7 TB/s on
        // 30 - 70 x its compulsory bytes.  Tiles are dealt to the XCDs in the same contiguous ranges as in order 1; the grid holds
        // `tiles_per_xcd` = ceil(tiles / 8) slots per XCD, the spare ones leave at once.
        const int YZ = p.nblocks * (NPH == 1 ? p.nphase : 1);
        const int xcd = bid & 7, idx = bid >> 3;
        const int tl = idx / YZ, yz = idx - tl * YZ;
        const int q8 = p.ntiles_grid >> 3, r8 = p.ntiles_grid & 7;   // the first r8 XCDs own q8 + 1 tiles, the others q8
        bid = xcd < r8 ? xcd * (q8 + 1) + tl : r8 * (q8 + 1) + (xcd - r8) * q8 + tl;
        nblk_i = yz % p.nblocks;
        z_i = yz / p.nblocks;
        if (tl >= (xcd < r8 ? q8 + 1 : q8)) return;   // (wave-uniform: the whole workgroup; the grid holds tiles_per_xcd slots per XCD)
    } else if (p.xcd_order) {   // consecutive ids go round-robin over the 8 XCDs: give each XCD a contiguous run of tiles, so that
                                // tiles that share halo lines meet in ONE L2
        const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7, idx = bid >> 3;
        bid = xcd < r8 ? xcd * (q8 + 1) + idx : r8 * (q8 + 1) + (xcd - r8) * q8 + idx;
    }
    if constexpr (W2) bid = 2 * bid + half;   // (a tile index past the launch's last one lands on image >= B: zero halo, nothing stored)
    const int tx_i = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty_i = bid % p.tiles_y; bid /= p.tiles_y;
    const int img0 = min(bid * p.imgs, p.B);
    const int y0 = ty_i * TH, x0 = tx_i * TWm;
    const HPhase& ph = p.ph[z_i];
    // K split (the trainer's launches on small batches: a deep layer of 8 images is 16 - 64 workgroups walking 80 - 240 k-steps at
    // memory latency): workgroup row y = split * nblocks + N-block runs the stages [ks0[split], ks0[split + 1]) of the list -- whole
    // halo chunks -- and stores raw partial sums `split_stride` floats apart; launch_split_reduce adds them in order
    int ph_stage0 = ph.stage0, ph_nstages = ph.nstages;
    size_t split_off = 0;
    if (p.ksplit > 1) {
        const int ks = nblk_i / p.nblocks;
        nblk_i -= ks * p.nblocks;
        ph_stage0 += p.ks0[z_i][ks];
        ph_nstages = p.ks0[z_i][ks + 1] - p.ks0[z_i][ks];
        split_off = (size_t)ks * p.split_stride;
    }
    const int nblk = nblk_i;
    // diagnostic stamps (shader-clock cycles), wave 0 only: [0] prologue, [1] waiting for loads + barriers, [2] MFMA
    // blocks, [3] epilogue, [4] whole kernel
    long long t_in = 0, t_wait = 0, t_comp = 0, t_a = 0, t_b = 0, t_vm = 0, t_iss = 0;
    if (DBG && p.dbg) t_in = __builtin_amdgcn_s_memtime();
    // ---- per-lane A-fragment pixel offsets (bytes) of this wave's M-tiles
    int abase[KMT];
#pragma unroll
    for (int m = 0; m < KMT; ++m) {
        const int t = wave * KMT + m;
        const int ig = t >> p.th_log2, ty = t & (TH - 1);
        abase[m] = ((ig * p.nimg_m + (li >> p.twm_log2)) * p.imgplane + ty * p.hw + (li & (TWm - 1))) * p.pix_bytes;
    }

    f32x4 accs[NPH][KMT][NT];
#pragma unroll
    for (int h = 0; h < NPH; ++h)
#pragma unroll
        for (int m = 0; m < KMT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) accs[h][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int lo_off = p.lo_off;   // byte offset of the lo planes
    unsigned char* const smem_h = smem + half * p.b_off;          // this tile's halo slots
    unsigned char* const Bl = smem + (W2 ? 2 : 1) * p.b_off;
    const uint4* const wbase = ph.w + (size_t)nblk * ph.wblk_stride;

    // ---- load path.  Everything is LDS-DMA (no staging VGPRs) through buffer descriptors: `buffer_load_dwordx4 .. offen lds`
    // takes a wave-uniform descriptor + scalar offset and a 32-bit per-lane offset, and lanes whose offset is outside the
    // descriptor write ZEROS to LDS (probed on MI355X, tools/probes/lds_dma_oob.hip) -- which is exactly the zero padding
    // of the convolution, so no address arithmetic and no zero source survive in the stage loop:
    //   halo    piece i of a chunk = PP = 64/OC consecutive halo pixels x OC octets (lanes >= PP*OC masked); wave w owns the
    //           pieces w, w+4, ..; the NHWC pixel index each lane fetches for its j-th piece depends on the tile only, so
    //           it is computed ONCE per workgroup (pix[j]); per chunk a piece costs one multiply-add + one select.
    //   weights a linear copy: per-lane offset lane*16, the piece's position in the scalar offset: no VALU at all.
    const int pl = (lane * p.inv_oc_q16) >> 16;   // lane / OC   (exact for lane < 64, OC <= 9)
    const int kq = lane - pl * p.OC;              // lane % OC: octet inside the chunk
    int pix[MAXP];   // >= 0: pixel index relative to image img0; -2: zero padding; -1: this lane writes nothing
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
        const int px = (wave + kWaves * j) * p.PP + pl;
        int v = -1;
        if (lane < p.nact && px < p.nhalo) {
            const int il = (int)(((float)px + 0.5f) * p.inv_imgplane);   // px < 1024: exact after truncation
            const int r = px - il * p.imgplane;
            const int hy = (int)(((float)r + 0.5f) * p.inv_hw);
            const int hx = r - hy * p.hw;
            const int gy = y0 + p.ymin + hy, gx = x0 + p.xmin + hx;
            const bool inside = img0 + il < p.B && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            v = inside ? (il * p.H + gy) * p.W + gx : -2;
        }
        pix[j] = v;
    }
    // descriptors: activation planes from this workgroup's first image on (offsets stay far below 2^31), the weight slab of
    // this (phase, N-block), the epilogue constants of this N-block
    const size_t img_elems = (size_t)p.H * p.W;
    auto act_rsrc = [&](const _Float16* base, int Cs) {
        const size_t left = (size_t)(p.B - img0) * img_elems * Cs * 2;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)img0 * img_elems * Cs), 0,
                                                 (int)(left < 0x7fffffffu ? left : 0x7fffffffu), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t r0h = act_rsrc(p.src_hi[0], p.Cs[0]), r0l = act_rsrc(p.src_lo[0], p.Cs[0]);
    const __amdgpu_buffer_rsrc_t r1h = act_rsrc(p.src_hi[1] ? p.src_hi[1] : p.src_hi[0], p.Cs[1]),
                                 r1l = act_rsrc(p.src_lo[1] ? p.src_lo[1] : p.src_lo[0], p.Cs[1]);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, ph.wblk_stride * 16, 0x00020000);
    const int lane16 = lane * 16;
#define UMX_BLDS16(rsrc, lptr, voff, soff) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lptr), 16, voff, soff, 0, 0)

    // all pieces of one halo chunk: octets [oct0, oct0+noct) of operand group `group` into halo slot `plane0`
    // (pixel, octet) -> byte offset inside the image: pixel * A + octet * B, (A, B) = (2 Cs, 16) for NHWC planes and
    // (16, H*W*16) for octet-planar ones (every 128-byte line is then read by exactly one octet chunk)
    const int kqB0 = kq * p.srcB[0], kqB1 = kq * p.srcB[1];
    auto issue_halo = [&](const HStage& st) {
        const bool g1 = st.group > 0;
        const int pixA = p.srcA[g1 ? 1 : 0];
        const int kqB = g1 ? kqB1 : kqB0;
        const __amdgpu_buffer_rsrc_t rh = g1 ? r1h : r0h, rl = g1 ? r1l : r0l;
        unsigned char* const slot = smem_h + st.plane0 * p.slot_bytes;
        const int soff = st.oct0 * p.srcB[g1 ? 1 : 0];
        const bool kok = kq < st.noct;
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const int i = wave + kWaves * j;
            if (i >= p.ninst) break;             // wave-uniform: the wave's pieces are the first ones
            int pj = pix[j];
            asm volatile("" : "+v"(pj));         // compare here: hoisted, the 12 lane masks live in (spilled) scalar pairs
            if (pj != -1 && kok) {               // lanes of this piece
                const int voff = pj >= 0 ? (int)__umul24(pj, pixA) + kqB : 0x7fffffff;
                UMX_BLDS16(rh, slot + i * p.piece_bytes, voff, soff);
                UMX_BLDS16(rl, slot + lo_off + i * p.piece_bytes, voff, soff);
            }
        }
    };
    // weight block of a stage: [64 B header: k-map of its k-steps][nk x NT x (hi, lo) images].  The header goes out at once
    // (one 64-byte piece by the last wave); the 1-KiB pieces are handed out one per N-tile iteration of the MFMA loop below
    // (wq_* state), so that a wave never queues a burst of vector-memory instructions in front of its matrix work.
    // (no queue state: the piece a wave issues at its c-th call is piece wave + 4c, and c is a compile-time constant of the
    // unrolled loops -- a loop-carried position ended up in a vector register with a waterfall loop around every piece:
    // per k-step of the 9-tile kernel 68 VALU + 90 SALU + 27 branches)
    int wq_np = 0, wq_soff = 0;
    unsigned char* wq_dst = Bl;
    const int wq_w = wave_all * 1024;
    auto wq_begin = [&](const HStage& st, int buf) {
        unsigned char* const wl = Bl + buf * p.wbuf_bytes;
        if (wave_all == kNW - 1 && lane < 4) UMX_BLDS16(rw, wl, lane16, st.woff * 16);
        wq_np = F6 ? NT * 4096 : st.nk * (NT * 2048);   // bytes of the block's 1-KiB pieces (F6: two k-steps of hi images + the fp6 images, always)
        wq_soff = st.woff * 16 + 64;
        wq_dst = wl + 64;
    };
    auto wq_one = [&](int c) {                 // the wave's c-th piece of the block
        const int pc = wq_w + c * (kNW * 1024);
        if (pc < wq_np) UMX_BLDS16(rw, wq_dst + pc, lane16, wq_soff + pc);
    };
    auto wq_drain = [&](int c0) {              // pieces c0, c0+1, .. (those the stage had no N-tile iteration for)
        for (int pc = wq_w + c0 * (kNW * 1024); pc < wq_np; pc += kNW * 1024)
            UMX_BLDS16(rw, wq_dst + pc, lane16, wq_soff + pc);
    };

    // epilogue constants of this N-block ([pre_s | pre_b | post_s | post_b] x NT*16 floats, defaults and 2^shift factors
    // folded on the host): loaded by LDS-DMA under the MFMAs of the last stage, into the weight buffer that stage frees
    const int ec_units = (p.head_K > 0 ? (4 + p.head_K) * (NT * 4) + 4 : NT * 16) + (D2S ? 16 : 0);   // uint4 per N-block
    auto issue_econst = [&](int buf) {
        if (wave_all == kNW - 1) {
            float* const dst = reinterpret_cast<float*>(Bl + buf * p.wbuf_bytes);
#pragma unroll
            for (int i = 0; i < (8 * NT * 4 + 4 + 63) / 64; ++i)
                if (i * 64 + lane < ec_units)
                    UMX_GLDS16(p.econst + (size_t)nblk * ec_units + i * 64 + lane, dst + i * 256);
        }
    };
    // depth-to-space form with appended channels: the workgroup's 32 x 32 output pixels of the compact raw-skip tensor go to LDS
    // behind the constants, under the MFMAs of the last stage as well (read by global loads in the epilogue their latency was
    // exposed twice per wave: 14 k of a 71 k-cycle workgroup).  One piece = the two output rows of two M-tiles: lane = (M-tile
    // parity, output row parity, 16-byte segment of the row's 32 pixels).
    auto issue_app = [&](int buf) {
        if constexpr (D2S) {
            if (p.app_c != nullptr && p.d2s_mix && wave == kWaves - 2) {
                const int apb = 4 * p.app_cw;   // bytes per compact pixel
                const size_t oHW = (size_t)p.outH * p.outW, left = (size_t)(p.B - img0) * oHW * apb;
                const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(p.app_c + (size_t)img0 * oHW * apb), 0, (int)(left < 0x7fffffffu ? left : 0x7fffffffu), 0x00020000);
                unsigned char* const dst = Bl + buf * p.wbuf_bytes + ec_units * 16;
                const int oy = (lane >> 4) & 1, seg = lane & 15;
#pragma unroll
                for (int i = 0; i < kMW / 2; ++i) {
                    const int t = 2 * i + (lane >> 5);
                    const int irel = (t >> p.th_log2) * p.nimg_m, yt = y0 + (t & (TH - 1));
                    const bool ok = img0 + irel < p.B && seg * 16 < 32 * apb;
                    const int voff = ok ? (int)((irel * (int)oHW + (2 * yt + oy) * p.outW + 2 * x0) * apb + seg * 16) : 0x7fffffff;
                    UMX_BLDS16(ra, dst + i * 1024, voff, 0);
                }
            }
        }
    };
    // stage s uses weight buffer (s + par0) & 1, with par0 such that the buffer the LAST stage frees -- where the epilogue
    // constants go -- is always buffer 1: the epilogue's transpose staging may then use everything below it
    const int par0 = (ph_nstages + 1) & 1;
    const float* const ec = reinterpret_cast<const float*>(Bl + p.wbuf_bytes);
    if (ph_nstages == 0) { issue_econst(1); issue_app(1); }
    // the stage table is read through the constant address space: a plain global pointer gets a VECTOR load and a full
    // s_waitcnt vmcnt(0) round trip at the top of every stage (the kernel stores and fences, so the compiler will not
    // prove the table unclobbered); the host writes it before the launch and nothing writes it afterwards
    // (tried: 32-byte records with every per-stage quantity resolved on the host, loaded two stages ahead -- hipcc waits for
    // the scalar load where it is issued, because it reuses a destination register at once, and the 8 extra scalar
    // registers cost more than the arithmetic they save: 1-2 % slower)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    static_assert(sizeof(HStage) == 16, "one stage = one 16-byte scalar load");
    auto load_stage = [&](int idx) {
        const u32x4 raw = ((const __attribute__((address_space(4))) u32x4*)(unsigned long long)p.stages)[idx];
        HStage st;
        __builtin_memcpy(&st, &raw, sizeof(st));
        return st;
    };
    HStage cur = load_stage(ph_stage0);
    if (ph_nstages > 0) {
        if (cur.group >= 0) issue_halo(cur);
        wq_begin(cur, par0);
        wq_drain(0);
    }
    long long t_pro = 0;
    if (DBG && p.dbg) { t_pro = __builtin_amdgcn_s_memtime(); t_a = t_pro; }
    // Two-deep software pipeline: the loads of stage s+1 (next weight block into the other weight buffer, and, at a chunk
    // boundary, the next halo chunk into the other halo slot) are in flight while stage s runs its MFMAs.  One barrier
    // per stage: it orders "everyone's loads of stage s have landed" (each wave waits for its own first) and "everyone is
    // done computing stage s-1" (so the buffers stage s+1 loads into are free).
    auto stage_body = [&](const int s) {
        const HStage nxt = load_stage(ph_stage0 + (s + 1 < ph_nstages ? s + 1 : s));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (DBG && p.dbg) { const long long t_v = __builtin_amdgcn_s_memtime(); t_vm += t_v - t_a; }
        __syncthreads();
        if (DBG && p.dbg) { t_b = __builtin_amdgcn_s_memtime(); t_wait += t_b - t_a; }
        if (s + 1 < ph_nstages) {
            if (nxt.group >= 0) issue_halo(nxt);
            wq_begin(nxt, (s + 1 + par0) & 1);
        } else {
            wq_np = 0;   // no block after the last stage
            issue_econst(1);   // == (s + 1 + par0) & 1
            issue_app(1);
        }
        if (DBG && p.dbg) t_iss += (long long)__builtin_amdgcn_s_memtime() - t_b;

        const unsigned char* const wl = Bl + ((s + par0) & 1) * p.wbuf_bytes;
        // k-map of the stage up front: one LDS round trip per stage instead of one on every k-step's critical path
        const int cur_nk = F6 ? (cur.nk & 0xff) : cur.nk;
        unsigned kbs[kStageK];
#pragma unroll
        for (int j = 0; j < (F6 ? 2 : kStageK); ++j)
            kbs[j] = (unsigned)*reinterpret_cast<const unsigned short*>(wl + (j * 4 + q) * 2) << 4;
        // the k-steps of this stage on one accumulator set (the phase loop is unrolled: static register indexing)
        // one k-step on accumulator set h.  SK (depth-to-space form, one block of four phases): the k-step's four (tap, octet) pairs
        // are taps the odd output rows do not have, so the N-tiles of that row pair -- [NT / 2, 2 (NT / 2)) -- hold zero weights:
        // no fragment reads and no MFMAs for them (lu0.convT: 5 of 9 tiles in every second k-step)
        auto kstep = [&](int h, int j, bool sk) {   // (sk: wave-uniform; the skippable tiles sit under one scalar branch each)
            constexpr int kLo = NT / 2, kHi = 2 * (NT / 2);
            constexpr int kImg = F6 ? 1024 : 2048;    // bytes of one (k-step, N-tile) weight image: hi alone, or hi then lo
            auto live = [&](int n) { return !(D2S && sk && n >= kLo && n < kHi); };
            const unsigned char* const bp = wl + 64 + j * (NT * kImg) + lane * 16;
            const unsigned char* const ap = smem_h + kbs[j];
            h8 ah[KMT], al[KMT];
#pragma unroll
            for (int m = 0; m < KMT; ++m) {
                ah[m] = *reinterpret_cast<const h8*>(ap + abase[m]);
                if constexpr (!F6) al[m] = *reinterpret_cast<const h8*>(ap + abase[m] + lo_off);
            }
            // B (weight) fragments are streamed with a prefetch distance of two N-tiles (three (hi, lo) pairs live),
            // and the scheduler is asked for its DS-read / MFMA interleaving pipeline (iglp_opt 0): left to itself it
            // puts a full lgkmcnt(0) wait between a fragment read and the MFMA block of the previous tile
            // (+4 % on the whole bench; explicit sched_group_barrier patterns were slower).
            constexpr int kBPre = 2;
            h8 bhq[kBPre + 1], blq[kBPre + 1];
#pragma unroll
            for (int n = 0; n < kBPre && n < NT; ++n) {
                if (!live(n)) continue;
                bhq[n] = *reinterpret_cast<const h8*>(bp + n * kImg);
                if constexpr (!F6) blq[n] = *reinterpret_cast<const h8*>(bp + n * kImg + 1024);
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                wq_one(j * NT + n);   // next stage's weight pieces, one per N-tile: spread under the MFMAs
                __builtin_amdgcn_iglp_opt(0);
                if (n + kBPre < NT && live(n + kBPre)) {
                    bhq[(n + kBPre) % (kBPre + 1)] = *reinterpret_cast<const h8*>(bp + (n + kBPre) * kImg);
                    if constexpr (!F6) blq[(n + kBPre) % (kBPre + 1)] = *reinterpret_cast<const h8*>(bp + (n + kBPre) * kImg + 1024);
                }
                if (!live(n)) continue;
                const h8 bh = bhq[n % (kBPre + 1)];
#pragma unroll
                for (int m = 0; m < KMT; ++m) {
                    // weights are the A operand (rows = output channels), activations the B operand (columns =
                    // pixels): D[channel][pixel], so a lane ends up with 4 consecutive channels of one pixel
                    f32x4 c = accs[h][m][n];
                    if constexpr (!F6) {
                        const h8 bl = blq[n % (kBPre + 1)];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al[m], c, 0, 0, 0);
                        if (!(PK && n == NT - 1)) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah[m], c, 0, 0, 0);
                    }
                    accs[h][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah[m], c, 0, 0, 0);
                }
            }
        };
        // the second half of an F6 stage: both cross terms of the stage's k-steps (0, 1) in one scaled MFMA per (M-tile, N-tile).
        // This lane's K block: k-step q & 1, plane q >> 1 (hi: the x_hi * w_lo term; lo: x_lo * w_hi)
        auto f6step = [&]() {
            if constexpr (F6) {
                const uint2 km = *reinterpret_cast<const uint2*>(wl + (q & 1) * 8);   // the k-step's four LDS slots
                const unsigned pl = (q >> 1) ? (unsigned)lo_off : 0u;
                const unsigned ko[4] = {((km.x & 0xffffu) << 4) + pl, ((km.x >> 16) << 4) + pl, ((km.y & 0xffffu) << 4) + pl, ((km.y >> 16) << 4) + pl};
                i32x6 xb[KMT];
                int sx[KMT];
                // the first weight images on their way before the pixel blocks are converted
                const unsigned char* const bpq = wl + 64 + 2 * NT * 1024 + lane * 16;   // (behind the room of two k-steps of hi images)
                uint4 w0q[3];
                uint4 w1q[3];
#pragma unroll
                for (int n = 0; n < 2 && n < NT; ++n) {
                    w0q[n] = *reinterpret_cast<const uint4*>(bpq + n * 2048);
                    w1q[n] = *reinterpret_cast<const uint4*>(bpq + n * 2048 + 1024);
                }
                // (timing-only ablations of the B stage, UMX_F6_ABLATE: bit 1 no pixel reads / conversion, bit 2 no scaled MFMAs)
                if (p.f6 & 2) {
#pragma unroll
                    for (int m = 0; m < KMT; ++m) { xb[m] = (i32x6){0, 0, 0, 0, 0, 0}; sx[m] = 127; }
                } else
#pragma unroll
                for (int m = 0; m < KMT; m += 2) {   // (two M-tiles at a time: all four blocks' fragments in flight at once would spill)
                    h8 v[2][4];
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[b][t] = *reinterpret_cast<const h8*>(smem_h + ko[t] + abase[m + b]);
                    i32x6 o[2];
                    int sc2[2];
                    quant_blocks_e2m3<2>(v, o, sc2);
                    xb[m] = o[0]; xb[m + 1] = o[1];
                    sx[m] = sc2[0]; sx[m + 1] = sc2[1];
                    __builtin_amdgcn_sched_barrier(0);
                }
                const unsigned char* const bp = bpq;
                constexpr int kBPre = 2;   // (prefetch distance of the weight images, like the binary16 k-steps; everything at once spills)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    wq_one(2 * NT + n);
                    if (n + kBPre < NT) {
                        w0q[(n + kBPre) % (kBPre + 1)] = *reinterpret_cast<const uint4*>(bp + (n + kBPre) * 2048);
                        w1q[(n + kBPre) % (kBPre + 1)] = *reinterpret_cast<const uint4*>(bp + (n + kBPre) * 2048 + 1024);
                    }
                    const uint4 w0 = w0q[n % (kBPre + 1)], w1 = w1q[n % (kBPre + 1)];
                    const i32x6 wa = (i32x6){(int)w0.x, (int)w0.y, (int)w0.z, (int)w0.w, (int)w1.x, (int)w1.y};
                    const int sa = (int)w1.z;
                    // (inline assembly with the accumulator as ONE in/out operand: the builtin's destination is not tied to its C
                    // operand, every result lands in a fresh register tuple and the allocator spills the accumulators moving them back
                    // -- 109 .. 240 registers.  cbsz / blgp 2 = e2m3; op_sel 0: scale = byte 0 of the scale registers.  Operands come
                    // from LDS reads and from conversions tens of instructions earlier; the results are next read behind a barrier.)
                    if (p.f6 & 4) { asm volatile("" :: "v"(wa), "v"(sa)); continue; }
#pragma unroll
                    for (int m = 0; m < KMT; ++m)
                        asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2"
                                     : "+v"(accs[0][m][n]) : "v"(wa), "v"(xb[m]), "v"(sa), "v"(sx[m]));
                    __builtin_amdgcn_sched_barrier(0);   // (no iglp_opt here: with 4 MFMAs per image it hoists all nine images' reads to the top)
                }
            }
        };
        // (depth-to-space form: one flag byte per k-step behind the k-map)
        unsigned skips = 0u;
        if constexpr (D2S) skips = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const unsigned*>(wl + 32));
#pragma unroll
        for (int h = 0; h < NPH; ++h) {
            if (NPH > 1 && cur.phase != h) continue;   // wave-uniform
#pragma unroll
            for (int j = 0; j < (F6 ? 2 : kStageK); ++j) {
                if (j >= cur_nk) break;
                kstep(h, j, D2S && ((skips >> (8 * j)) & 1u) != 0u);
            }
        }
        f6step();
        wq_drain(F6 ? 3 * NT : cur_nk * NT);
        cur = nxt;
        if (DBG && p.dbg) { t_a = __builtin_amdgcn_s_memtime(); t_comp += t_a - t_b; }
    };
    for (int s = 0; s < ph_nstages; ++s) stage_body(s);
    // packed last N-tile: rows j (w_hi products) += rows j + 8 (w_lo products), i.e. lane l += lane l + 32.  The plain kernels do it
    // here, once; the fused-phase kernels where an accumulator is consumed (emit_t): their four accumulator sets moved to vector
    // registers at this point cost 19 registers, i.e. the third workgroup per CU of the 3-tile kernel
    auto pk_sum = [](float v) {
        const unsigned u = __float_as_uint(v);
        const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // {lanes 0..31 twice, lanes 32..63 twice}
        return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    };
    if constexpr (PK && NPH == 1) {
#pragma unroll
        for (int m = 0; m < KMT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) accs[0][m][NT - 1][r] = pk_sum(accs[0][m][NT - 1][r]);
    }
    struct DbgOut {   // written when the kernel returns (both epilogue paths)
        const HConvParams& p; long long t_in, t_pro, t_wait, t_comp, t_epi; int tid; const long long& t_vm; const long long& t_iss;
        __device__ ~DbgOut() {
            if (DBG && p.dbg && tid == 0) {
                const long long t_end = __builtin_amdgcn_s_memtime();
                const size_t w = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
                long long* d = p.dbg + w * 7;
                d[0] = t_pro - t_in; d[1] = t_wait; d[2] = t_comp; d[3] = t_end - t_epi; d[4] = t_end - t_in; d[5] = t_vm; d[6] = t_iss;
            }
        }
    } dbg_out{p, t_in, t_pro, t_wait, t_comp, DBG && p.dbg ? (long long)__builtin_amdgcn_s_memtime() : 0, tid, t_vm, t_iss};

    // ---- epilogue: (acc * pre_s + pre_b) -> activation -> (* post_s + post_b, output shift folded in) -> [2x2 max-pool]
    //      -> (hi, lo) binary16 NHWC, or fp32 NHWC for the tensor the softmax head reads.
    // C/D layout (weights as the A operand): row = output channel 4*(lane>>4) + reg of the N-tile, column = pixel
    // lane & 15 of the M-tile.  A lane therefore stores 4 consecutive channels of its pixel in one 8-byte (binary16)
    // or 16-byte (fp32) store; the 4 lane groups complete the 16 channels (32 / 64 contiguous bytes per pixel).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the constants have landed ...
    __syncthreads();                                    // ... for every wave
    const float4* const ec4 = reinterpret_cast<const float4*>(ec);

    // fp32 output: direct 16-byte stores (64 contiguous bytes per pixel and N-tile).
    // (hi, lo) output: straight from registers in 16-byte units (below); no LDS staging, every wave runs arithmetic ->
    // exchange -> stores on its own and accumulators die as they are consumed.

    // One M-tile of one phase: (acc * pre_s + pre_b) -> activation -> [* post_s + post_b].  Nothing in the per-value code may
    // branch: hipcc does not unswitch loops on p.act / p.post_affine, it tests them per value (5 VALU + 3 SALU + 4 branches
    // per value measured).  The activation is max(v, slope*v) with slope 0 (ReLU: a negative v gives -0, numerically the
    // reference's 0), 0.2 (LeakyReLU, == v > 0 ? v : 0.2 v) or 1 (none), written as the median of (v, slope*v, +inf) because
    // fmaxf() costs an extra canonicalising v_max per value; the second affine exists only in the legacy graph (BN after
    // ReLU) and with a non-zero activation shift: two copies of the loop behind one uniform branch.  Results are scalars,
    // not written back into the accumulator tuples (a 4-vector rebuilt by inserts gets a fresh register tuple).
    unsigned vmax = 0u;   // max |v| of this lane as a bit pattern (orders NaN and infinity above every finite value): one
                          // compare against binary16's range at the end
    const float slope = p.act == ACT_RELU ? 0.f : p.act == ACT_LEAKY ? kLeakySlope : 1.f;
    auto arith_v = [&](const f32x4 (&A)[NT], float (&out)[NT][4], auto POST, auto TRACK) {
        constexpr bool post = decltype(POST)::value, track = decltype(TRACK)::value;
        // (different first instructions: otherwise the common head of the two copies is hoisted above the branch in one batch)
        if constexpr (post) asm volatile("; epilogue arithmetic, second affine");
        else asm volatile("; epilogue arithmetic");
        // (trainer launches: the dynamic operand scales, two device scalars; 1 otherwise -- four multiplies per N-tile, no branch)
        // (compiled into the fp32-output instance of this function only: the fused head is the other, range-tracked one)
        float dyn = 1.f;
        if constexpr (!track) dyn = (p.dyn[0] ? *p.dyn[0] : 1.f) * (p.dyn[1] ? *p.dyn[1] : 1.f);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const float4 ps = ec4[0 * NT * 4 + n * 4 + q], pb = ec4[1 * NT * 4 + n * 4 + q];
            float psa[4] = {ps.x, ps.y, ps.z, ps.w};
            const float pba[4] = {pb.x, pb.y, pb.z, pb.w};
            if constexpr (!track) {
#pragma unroll
                for (int r = 0; r < 4; ++r) psa[r] *= dyn;
            }
            float qsa[4] = {1.f, 1.f, 1.f, 1.f}, qba[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (post) {
                const float4 qs = ec4[2 * NT * 4 + n * 4 + q], qb = ec4[3 * NT * 4 + n * 4 + q];
                qsa[0] = qs.x; qsa[1] = qs.y; qsa[2] = qs.z; qsa[3] = qs.w;
                qba[0] = qb.x; qba[1] = qb.y; qba[2] = qb.z; qba[3] = qb.w;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = A[n][r] * psa[r] + pba[r];
                v = __builtin_amdgcn_fmed3f(v, v * slope, INFINITY);
                if constexpr (post) v = v * qsa[r] + qba[r];
                if constexpr (track) vmax = max(vmax, __float_as_uint(v) & 0x7fffffffu);
                out[n][r] = v;
            }
            // four values at a time: left alone, the scheduler runs all fmas, then all multiplies, then all medians, and
            // spills the temporaries in between
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto arith = [&](const f32x4 (&A)[NT], float (&out)[NT][4], auto TRACK) {
        if (p.post_affine) arith_v(A, out, std::true_type{}, TRACK);
        else arith_v(A, out, std::false_type{}, TRACK);
    };
    constexpr std::true_type kTrack{};
    constexpr std::false_type kNoTrack{};

    if constexpr (NPH == 1) {
        if (p.head_K > 0) {
            // fused top layer (reference UnMicst1-5.py:212-222,236-237 / UnMicst.py:167-171,186): 1x1 conv over this pixel's
            // channels, BN affine, softmax -- the 1x1 conv ON THE MATRIX CORES: the lane's 4 channels of N-tiles (2s, 2s+1)
            // are exactly the 8 k-elements lane group q feeds at head k-step s, so the activations go from the accumulators
            // into B-fragments without leaving the lane (as (hi, lo) binary16 pairs, the same 3-product arithmetic as the
            // convolutions).  The head's A-fragment has the classes in rows 0..3; shifted down by 4 m rows (DPP row_shr, zeros
            // shifted in) it puts M-tile m's logits into lane group m, so ONE accumulator collects the four M-tiles of a wave
            // and every lane ends with all classes of one pixel: no cross-lane reduction, one softmax per lane.
            const int K = p.head_K;
            constexpr int NS2 = (NT + 1) / 2;
            const float* const hsb = ec + (4 + K) * (NT * 16);   // [scale x 8 | bias x 8] (the scale carries the 2^-hs of the packed head weights)
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 Hh[NS2], Hl[NS2];
#pragma unroll
            for (int s2 = 0; s2 < NS2; ++s2) {
                Hh[s2] = *reinterpret_cast<const u32x4*>(p.head_frag + (s2 * 2 + 0) * 64 + lane);
                Hl[s2] = *reinterpret_cast<const u32x4*>(p.head_frag + (s2 * 2 + 1) * 64 + lane);
            }
            f32x4 lg = (f32x4){0.f, 0.f, 0.f, 0.f};
            static_for<0, KMT>([&](auto MC) {
                constexpr int m = decltype(MC)::value;
                float res[NT][4];
                arith(accs[0][m], res, kTrack);   // (range tracked: the head's operands are binary16 pairs)
#pragma unroll
                for (int s2 = 0; s2 < NS2; ++s2) {
                    h8 bh, bl;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int nt = 2 * s2 + (j >> 2);
                        const float t = nt < NT ? res[nt < NT ? nt : 0][j & 3] : 0.f;
                        bh[j] = (_Float16)t;
                        bl[j] = (_Float16)(t - (float)bh[j]);
                    }
                    u32x4 ah = Hh[s2], al = Hl[s2];
                    if constexpr ((m & 3) != 0) {
                        constexpr int ctrl = 0x110 + 4 * (m & 3);   // row_shr:4m, bound_ctrl: rows shifted in are zero
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            ah[d] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ah[d], ctrl, 0xF, 0xF, true);
                            al[d] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)al[d], ctrl, 0xF, 0xF, true);
                        }
                    }
                    const h8 wh = __builtin_bit_cast(h8, ah), wl = __builtin_bit_cast(h8, al);
                    lg = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl, lg, 0, 0, 0);
                    lg = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh, lg, 0, 0, 0);
                    lg = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh, lg, 0, 0, 0);
                }
                if ((m & 3) == 3 || m == KMT - 1) {   // lane group q holds the logits of pixel li of M-tile (m & ~3) + q
                    const int mq = (m & ~3) + q;
                    const int t = wave * KMT + mq;
                    const int ig = t >> p.th_log2, ty = t & (TH - 1);
                    const int img = img0 + ig * p.nimg_m + (li >> p.twm_log2);
                    if (mq < KMT && img < p.B) {
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
                        float* const d = p.probs + ((long)(img * p.outH + y0 + ty) * p.outW + x0 + (li & (TWm - 1))) * K;
                        if (K == 3) {   // one 12-byte store per pixel instead of three 4-byte ones
                            struct __attribute__((packed, aligned(4))) F3 { float a, b, c; };
                            *reinterpret_cast<F3*>(d) = F3{e[0] * inv, e[1] * inv, e[2] * inv};
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (k < K) d[k] = e[k] * inv;
                        }
                    }
                    lg = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            });
            if (vmax >= 0x476a6000u) atomicOr(p.overflow_flag, 1);   // binary16 range exceeded in front of the head
            return;
        }
        if (p.dst_f32) {   // (the planner never pairs fp32 output with the fused transposed convolution)
#pragma unroll
            for (int m = 0; m < KMT; ++m) {
                const int t = wave * KMT + m;
                const int ig = t >> p.th_log2, ty = t & (TH - 1);
                const int img = img0 + ig * p.nimg_m + (li >> p.twm_log2);
                float res[NT][4];
                arith(accs[0][m], res, kNoTrack);
                if (img >= p.B) continue;
                const long pix = (long)(img * p.outH + (y0 + ty) * p.o_mul + ph.oy_off) * p.outW +
                                 (x0 + (li & (TWm - 1))) * p.o_mul + ph.ox_off;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int c0 = nblk * (NT * 16) + n * 16 + 4 * q;
                    const float* const v = res[n];
                    float* const d = p.dst_f32 + split_off + pix * p.Cout + c0;
                    if ((p.Cout & 3) == 0) {
                        if (c0 < p.Cout) *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (c0 + r < p.Cout) d[r] = v[r];
                    }
                }
            }
            return;
        }
    }

    // ---- (hi, lo) output.  A lane holds 4 consecutive channels (8 bytes per plane) of one pixel; the unit of both
    // destination layouts is (pixel, octet) = 16 bytes: element offset image * dImg + pixel * dPix + octet * dOct with
    // (dPix, dOct) = (Cds, 8) for NHWC and (8, outH*outW*8) for octet-planar ([octet][pixel][8] per image).
    // v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of another.  Fed with two result
    // sets A and B of the SAME N-tile (two sets of pixels), it leaves lane group q with all 8 channels of octet
    // 2n + (q >> 1) -- of A's pixel for q even, of B's pixel for q odd: every lane stores one 16-byte unit per plane and
    // N-tile, 16 consecutive pixels of an octet per lane row (256 contiguous bytes in the planar form), with no LDS
    // transpose, no staging area and no waits in between.
    const bool dpl = p.dst_planar != 0;
    const int dImg = p.Cds * p.outH * p.outW;
    const int dPix = dpl ? 8 : p.Cds;
    const int dOct = dpl ? p.outH * p.outW * 8 : 8;
    const int noct = p.Cds >> 3;
    const int oct_q = q >> 1;            // octet (of the N-tile's two) this lane stores
    const bool setB = (q & 1) != 0;      // ... and whose pixel: set A (q even) or set B (q odd)
    // wave-uniform bases: this workgroup's first image; a lane adds a 32-bit byte offset (the planner keeps a tile's images
    // and every (pixel, octet) offset inside one image far below 2^31)
    unsigned char* const bhi = reinterpret_cast<unsigned char*>(p.dst_hi + (size_t)img0 * dImg);
    unsigned char* const blo = reinterpret_cast<unsigned char*>(p.dst_lo + (size_t)img0 * dImg);
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    auto pack_split = [](float v0, float v1, unsigned& hw, unsigned& lw) {
        h2 h, l;
        h[0] = (_Float16)v0; h[1] = (_Float16)v1;
        l[0] = (_Float16)(v0 - (float)h[0]); l[1] = (_Float16)(v1 - (float)h[1]);
        hw = __builtin_bit_cast(unsigned, h);
        lw = __builtin_bit_cast(unsigned, l);
    };
    // One value: (acc * pre_s + pre_b) -> activation -> [* post_s + post_b], range-tracked (as arith_v above, one N-tile at a
    // time so that the results of an N-tile leave for memory before the next one is touched: a whole M-tile pair of
    // results live at once spilled 69 registers in the 9-tile kernel)
    struct EC { float ps[4], pb[4], qs[4], qb[4]; };
    auto load_ec = [&](int n, auto POST) {
        EC e;
        const float4 ps = ec4[0 * NT * 4 + n * 4 + q], pb = ec4[1 * NT * 4 + n * 4 + q];
        e.ps[0] = ps.x; e.ps[1] = ps.y; e.ps[2] = ps.z; e.ps[3] = ps.w;
        e.pb[0] = pb.x; e.pb[1] = pb.y; e.pb[2] = pb.z; e.pb[3] = pb.w;
        if constexpr (decltype(POST)::value) {
            const float4 qs = ec4[2 * NT * 4 + n * 4 + q], qb = ec4[3 * NT * 4 + n * 4 + q];
            e.qs[0] = qs.x; e.qs[1] = qs.y; e.qs[2] = qs.z; e.qs[3] = qs.w;
            e.qb[0] = qb.x; e.qb[1] = qb.y; e.qb[2] = qb.z; e.qb[3] = qb.w;
        }
        return e;
    };
    auto act1 = [&](float a, const EC& e, int r, auto POST) {
        float v = a * e.ps[r] + e.pb[r];
        v = __builtin_amdgcn_fmed3f(v, v * slope, INFINITY);
        if constexpr (decltype(POST)::value) v = v * e.qs[r] + e.qb[r];
        return v;
    };
    // this lane's 16-byte unit of N-tile n after the exchange of sets (a, b) -> both planes
    constexpr bool kApp = NT <= 5;   // (appended channels exist for narrow tensors only: not in the 6..9-tile kernels)
    // appended channels: the lanes that will hold octet app_oct read their pixel's two binary16 pairs AHEAD of the arithmetic
    // (issued at the top of a tile's epilogue, consumed N-tiles later: the loads' latency hides behind the other N-tiles)
    const bool lane_app = kApp && p.app_c != nullptr && oct_q == (p.app_oct & 1) && (p.app_oct >> 1) >= nblk * NT &&
                          (p.app_oct >> 1) < (nblk + 1) * NT;
    auto app_load = [&](int apix, unsigned& ah, unsigned& al) {
        ah = al = 0u;
        if (lane_app && apix >= 0) {   // the compact tensor: [pixel]{hi[cw] | lo[cw]}
            const size_t px = (size_t)img0 * p.outH * p.outW + apix;
            if (p.app_cw == 2) {
                const uint2 w = *reinterpret_cast<const uint2*>(p.app_c + px * 8);
                ah = w.x; al = w.y;
            } else {
                const unsigned w = *reinterpret_cast<const unsigned*>(p.app_c + px * 4);
                ah = w & 0xffffu; al = w >> 16;
            }
        }
    };
    auto store_unit = [&](int n, const float (&a)[4], const float (&b)[4], int off, bool ok, int apix, unsigned app_h, unsigned app_l) {
        // appended channels (the raw-input skip riding in a spare word of this unit), loaded by app_load below
        const int oct_u = (nblk * NT + n) * 2 + oct_q;
        const bool app = kApp && lane_app && apix >= 0 && oct_u == p.app_oct;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            vmax = max(vmax, max(__float_as_uint(a[r]) & 0x7fffffffu, __float_as_uint(b[r]) & 0x7fffffffu));
        unsigned ah0, ah1, al0, al1, bh0, bh1, bl0, bl1;
        pack_split(a[0], a[1], ah0, al0);
        pack_split(a[2], a[3], ah1, al1);
        pack_split(b[0], b[1], bh0, bl0);
        pack_split(b[2], b[3], bh1, bl1);
        // (every lane takes part in the exchange: never under a divergent branch)
        const auto h0 = __builtin_amdgcn_permlane16_swap(ah0, bh0, false, false);
        const auto h1 = __builtin_amdgcn_permlane16_swap(ah1, bh1, false, false);
        const auto l0 = __builtin_amdgcn_permlane16_swap(al0, bl0, false, false);
        const auto l1 = __builtin_amdgcn_permlane16_swap(al1, bl1, false, false);
        const int oct = oct_u;
        if (ok && oct < noct) {
            const unsigned o = (unsigned)(off + oct * dOct) * 2u;
            uint4 uh = make_uint4(h0[0], h1[0], h0[1], h1[1]), ul = make_uint4(l0[0], l1[0], l0[1], l1[1]);
            if (app) {
                if (p.app_word == 1) { uh.y = app_h; ul.y = app_l; }
                else if (p.app_word == 2) { uh.z = app_h; ul.z = app_l; }
                else { uh.w = app_h; ul.w = app_l; }
            }
            *reinterpret_cast<uint4*>(bhi + o) = uh;
            *reinterpret_cast<uint4*>(blo + o) = ul;
        }
    };
    // accumulator sets A, B (two sets of pixels) -> stores; `off`: element offset of THIS lane's pixel (set A's for q even,
    // set B's for q odd) relative to image img0, octet 0; `ok`: the lane's pixel exists
    auto emit_t = [&](const f32x4 (&A)[NT], const f32x4 (&B)[NT], int off, bool ok, int apix, unsigned app_h, unsigned app_l, auto POST) {
        if constexpr (decltype(POST)::value) asm volatile("; epilogue stores, second affine");
        else asm volatile("; epilogue stores");
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const EC e = load_ec(n, POST);
            float a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float av = A[n][r], bv = B[n][r];
                if constexpr (PK && NPH == 4) {
                    if (n == NT - 1) { av = pk_sum(av); bv = pk_sum(bv); }
                }
                a[r] = act1(av, e, r, POST); b[r] = act1(bv, e, r, POST);
            }
            store_unit(n, a, b, off, ok, apix, app_h, app_l);
            __builtin_amdgcn_sched_barrier(0);   // one N-tile at a time
        }
    };
    auto emit = [&](const f32x4 (&A)[NT], const f32x4 (&B)[NT], int off, bool ok, int apix, unsigned app_h, unsigned app_l) {
        if (p.post_affine) emit_t(A, B, off, ok, apix, app_h, app_l, std::true_type{});
        else emit_t(A, B, off, ok, apix, app_h, app_l, std::false_type{});
    };
    // the same with a fused 2 x 2 max-pool: set A = the pooled row of M-tiles (R0, R1), set B of (R2, R3); rows are vertical
    // neighbours, pixels li, li^1 horizontal ones; both lanes of a pixel pair hold the pooled value, the even one stores it
    auto emit_pool_t = [&](const f32x4 (&R0)[NT], const f32x4 (&R1)[NT], const f32x4 (&R2)[NT], const f32x4 (&R3)[NT], int off,
                           bool ok, auto POST) {
        if constexpr (decltype(POST)::value) asm volatile("; pooled epilogue stores, second affine");
        else asm volatile("; pooled epilogue stores");
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const EC e = load_ec(n, POST);
            float a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float va = fmaxf(act1(R0[n][r], e, r, POST), act1(R1[n][r], e, r, POST));
                const float vb = fmaxf(act1(R2[n][r], e, r, POST), act1(R3[n][r], e, r, POST));
                const float sa = __builtin_bit_cast(
                    float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, va), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false));
                const float sb = __builtin_bit_cast(
                    float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, vb), 0xB1, 0xF, 0xF, false));
                a[r] = fmaxf(va, sa);
                b[r] = fmaxf(vb, sb);
            }
            store_unit(n, a, b, off, ok, -1, 0u, 0u);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // element offset (relative to image img0) of the image M-tile t's column i belongs to, and the tile-relative (y, x) of
    // that pixel; -1: no such image
    const int dImgPix = p.outH * p.outW;   // pixels per output image: the appended tensor is indexed by pixel
    int app_img = 0;                       // (set by pix_off: pixel index of the lane's image inside the appended tensor)
    auto pix_off = [&](int t, int i, int& y_tile, int& x_tile) {
        const int ig = t >> p.th_log2;
        y_tile = y0 + (t & (TH - 1));
        x_tile = x0 + (i & (TWm - 1));
        const int irel = ig * p.nimg_m + (i >> p.twm_log2);
        if constexpr (kApp) app_img = irel * dImgPix;
        return img0 + irel < p.B ? irel * dImg : -1;
    };
    if constexpr (D2S) {
        static_assert(NPH == 1 && (KMT & 1) == 0 && !PK, "the depth-to-space form runs on the plain kernel");
        const int z = z_i;
        // (the table sits behind the epilogue constants in LDS; an entry is read where it is used: 9 of them held in registers
        // spilled 34)
        const int* const dtab = reinterpret_cast<const int*>(ec) + 4 * NT * 16 + z * (2 * NT + 8);
        const bool mix = p.d2s_mix != 0;                  // (wave-uniform) the last N-tile is the remainder tile
        constexpr int NTR = NT - 1;                       // N-tiles that are regular whatever `mix` says
        // Two passes, so that neither carries the other's registers (one pass spilled 37, and every scratch reload waits for the
        // stores in front of it).  Pass 1, the regular N-tiles: sets A, B = M-tiles m, m + 1 (rows y, y + 1 of one image: the planner
        // asks for >= 2 rows), exchanged into 16-byte units; the octet's phase and destination come from the table.
        const int rowB = 2 * p.outW * dPix;               // element offset of set B's pixel relative to set A's
#pragma unroll
        for (int m = 0; m < KMT; m += 2) {
            int yt, xt;
            const int ib = pix_off(wave * KMT + m, li, yt, xt);
            const int off = ib + ((yt * 2) * p.outW + xt * 2) * dPix + (setB ? rowB : 0);
            const bool ok = ib >= 0;
            auto body = [&](auto POST) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if (n == NTR && mix) break;   // (wave-uniform)
                    const EC e = load_ec(n, POST);
                    float a[4], b[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { a[r] = act1(accs[0][m][n][r], e, r, POST); b[r] = act1(accs[0][m + 1][n][r], e, r, POST); }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        vmax = max(vmax, max(__float_as_uint(a[r]) & 0x7fffffffu, __float_as_uint(b[r]) & 0x7fffffffu));
                    unsigned ah0, ah1, al0, al1, bh0, bh1, bl0, bl1;
                    pack_split(a[0], a[1], ah0, al0);
                    pack_split(a[2], a[3], ah1, al1);
                    pack_split(b[0], b[1], bh0, bl0);
                    pack_split(b[2], b[3], bh1, bl1);
                    // (every lane takes part in the exchange: never under a divergent branch)
                    const auto h0 = __builtin_amdgcn_permlane16_swap(ah0, bh0, false, false);
                    const auto h1 = __builtin_amdgcn_permlane16_swap(ah1, bh1, false, false);
                    const auto l0 = __builtin_amdgcn_permlane16_swap(al0, bl0, false, false);
                    const auto l1 = __builtin_amdgcn_permlane16_swap(al1, bl1, false, false);
                    const int dsel = dtab[2 * n + oct_q];
                    if (ok && dsel >= 0) {
                        const unsigned o = (unsigned)(off + dsel) * 2u;
                        *reinterpret_cast<uint4*>(bhi + o) = make_uint4(h0[0], h1[0], h0[1], h1[1]);
                        *reinterpret_cast<uint4*>(blo + o) = make_uint4(l0[0], l1[0], l0[1], l1[1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // one N-tile at a time
                }
            };
            if (p.post_affine) body(std::true_type{});
            else body(std::false_type{});
        }
        // Pass 2, the remainder tile: lane group q holds 4 channels of phase slot q for its pixel; with the appended raw-skip
        // channels (from the LDS copy, issue_app) and zeros they are that phase's last stored octet -- no exchange, one M-tile at a time
        if (mix) {
            const int dmix = dtab[2 * NT + q];            // this lane group's (sub-pixel phase, last octet) offset, < 0: no phase
            const int cmix = dtab[2 * NT + 4 + q];        // ... and the phase's sub-pixel code oy * 2 + ox
            const bool app_on = p.app_c != nullptr;
            const int apb = 4 * p.app_cw;
            const unsigned char* const apl = reinterpret_cast<const unsigned char*>(ec) + ec_units * 16 + (cmix >> 1) * 256 +
                                             (2 * li + (cmix & 1)) * apb;
            auto tail = [&](auto POST) {
                const EC e = load_ec(NTR, POST);
#pragma unroll
                for (int m = 0; m < KMT; ++m) {
                    int yt, xt;
                    const int ib = pix_off(wave * KMT + m, li, yt, xt);
                    float a[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        a[r] = act1(accs[0][m][NTR][r], e, r, POST);
                        vmax = max(vmax, __float_as_uint(a[r]) & 0x7fffffffu);
                    }
                    uint4 uh = make_uint4(0u, 0u, 0u, 0u), ul = make_uint4(0u, 0u, 0u, 0u);
                    pack_split(a[0], a[1], uh.x, ul.x);
                    pack_split(a[2], a[3], uh.y, ul.y);
                    if (app_on && dmix >= 0) {
                        unsigned aph, apl_w;
                        const unsigned char* const src = apl + (wave * KMT + m) * 512;
                        if (p.app_cw == 2) {
                            const uint2 w = *reinterpret_cast<const uint2*>(src);
                            aph = w.x; apl_w = w.y;
                        } else {
                            const unsigned w = *reinterpret_cast<const unsigned*>(src);
                            aph = w & 0xffffu; apl_w = w >> 16;
                        }
                        if (p.app_word == 1) { uh.y = aph; ul.y = apl_w; }
                        else if (p.app_word == 2) { uh.z = aph; ul.z = apl_w; }
                        else { uh.w = aph; ul.w = apl_w; }
                    }
                    if (ib >= 0 && dmix >= 0) {
                        const unsigned o = (unsigned)(ib + ((yt * 2) * p.outW + xt * 2) * dPix + dmix) * 2u;
                        *reinterpret_cast<uint4*>(bhi + o) = uh;
                        *reinterpret_cast<uint4*>(blo + o) = ul;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (p.post_affine) tail(std::true_type{});
            else tail(std::false_type{});
        } else {   // (no remainder tile: the last N-tile is a regular one)
#pragma unroll
            for (int m = 0; m < KMT; m += 2) {
                int yt, xt;
                const int ib = pix_off(wave * KMT + m, li, yt, xt);
                const int off = ib + ((yt * 2) * p.outW + xt * 2) * dPix + (setB ? rowB : 0);
                auto last = [&](auto POST) {
                    const EC e = load_ec(NTR, POST);
                    float a[4], b[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { a[r] = act1(accs[0][m][NTR][r], e, r, POST); b[r] = act1(accs[0][m + 1][NTR][r], e, r, POST); }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        vmax = max(vmax, max(__float_as_uint(a[r]) & 0x7fffffffu, __float_as_uint(b[r]) & 0x7fffffffu));
                    unsigned ah0, ah1, al0, al1, bh0, bh1, bl0, bl1;
                    pack_split(a[0], a[1], ah0, al0);
                    pack_split(a[2], a[3], ah1, al1);
                    pack_split(b[0], b[1], bh0, bl0);
                    pack_split(b[2], b[3], bh1, bl1);
                    const auto h0 = __builtin_amdgcn_permlane16_swap(ah0, bh0, false, false);
                    const auto h1 = __builtin_amdgcn_permlane16_swap(ah1, bh1, false, false);
                    const auto l0 = __builtin_amdgcn_permlane16_swap(al0, bl0, false, false);
                    const auto l1 = __builtin_amdgcn_permlane16_swap(al1, bl1, false, false);
                    const int dsel = dtab[2 * NTR + oct_q];
                    if (ib >= 0 && dsel >= 0) {
                        const unsigned o = (unsigned)(off + dsel) * 2u;
                        *reinterpret_cast<uint4*>(bhi + o) = make_uint4(h0[0], h1[0], h0[1], h1[1]);
                        *reinterpret_cast<uint4*>(blo + o) = make_uint4(l0[0], l1[0], l0[1], l1[1]);
                    }
                };
                if (p.post_affine) last(std::true_type{});
                else last(std::false_type{});
            }
        }
    } else if constexpr (NPH == 4) {
        // output row 2y+pu of M-tile row y: its 32 pixels 2x+pv come from phases (pu, 0) = set A and (pu, 1) = set B
        int ibs[KMT], pixs[KMT][2], apx[KMT][2];
        unsigned aph[KMT][2], apl[KMT][2];
#pragma unroll
        for (int m = 0; m < KMT; ++m) {
            int yt, xt;
            ibs[m] = pix_off(wave * KMT + m, li, yt, xt);
#pragma unroll
            for (int pu = 0; pu < 2; ++pu) {
                pixs[m][pu] = (yt * 2 + pu) * p.outW + xt * 2 + (setB ? 1 : 0);
                apx[m][pu] = ibs[m] >= 0 ? app_img + pixs[m][pu] : -1;
                app_load(apx[m][pu], aph[m][pu], apl[m][pu]);
            }
        }
#pragma unroll
        for (int m = 0; m < KMT; ++m)
#pragma unroll
            for (int pu = 0; pu < 2; ++pu)
                emit(accs[pu * 2 + 0][m], accs[pu * 2 + 1][m], ibs[m] + pixs[m][pu] * dPix, ibs[m] >= 0, apx[m][pu], aph[m][pu], apl[m][pu]);
    } else if (p.pool) {
        static_assert(NPH == 4 || KMT == 4, "the pooled epilogue pairs the two pooled rows of a wave");
        int yt, xt;
        const int ib = pix_off(wave * KMT + (setB ? 2 : 0), li, yt, xt);
        const int off = ib + ((yt >> 1) * p.outW + (xt >> 1)) * dPix;
        const bool ok = ib >= 0 && (li & 1) == 0;
        if (p.post_affine) emit_pool_t(accs[0][0], accs[0][1], accs[0][2], accs[0][3], off, ok, std::true_type{});
        else emit_pool_t(accs[0][0], accs[0][1], accs[0][2], accs[0][3], off, ok, std::false_type{});
    } else {
        // sets A, B = M-tiles m, m+1 (every o_mul-th pixel for a per-phase transposed convolution)
        static_assert((KMT & 1) == 0, "M-tiles are stored in pairs");
        const int om = p.o_mul;
#pragma unroll
        for (int m = 0; m < KMT; m += 2) {
            int yt, xt;
            const int ib = pix_off(wave * KMT + m + (setB ? 1 : 0), li, yt, xt);
            const int pix = (yt * om + ph.oy_off) * p.outW + xt * om + ph.ox_off;
            const int apix = ib >= 0 ? app_img + pix : -1;
            unsigned ah, al;
            app_load(apix, ah, al);
            emit(accs[0][m], accs[0][m + 1], ib + pix * dPix, ib >= 0, apix, ah, al);
        }
    }
    if (vmax >= 0x476a6000u) atomicOr(p.overflow_flag, 1);   // |v| >= 60000, infinity or NaN: binary16 range exceeded, the host reports it
}

template <int NT, int KMT, int NPH, bool DBG, int MAXP, bool PK = false, bool D2S = false, bool F6 = false, bool W2 = F6>
static hipError_t launch_h_k(const HConvParams& p, hipStream_t stream) {
    const int img_groups = (p.B + p.imgs - 1) / p.imgs;
    dim3 grid((unsigned)(img_groups * p.tiles_y * p.tiles_x), (unsigned)p.nblocks, (unsigned)(NPH == 1 ? p.nphase : 1));
    if (W2) {   // two tiles per workgroup of eight waves, one workgroup per CU (the whole LDS)
        if (p.xcd_order == 2 || p.ksplit > 1) return hipErrorInvalidValue;
        grid.x = (grid.x + 1) / 2;
    }
    if (p.ksplit > 1) {   // (trainer: fp32 partial sums, one row of N-blocks per split)
        if (p.xcd_order == 2 || !p.dst_f32 || p.head_K > 0 || p.ksplit > kMaxKSplit) return hipErrorInvalidValue;
        grid.y *= (unsigned)p.ksplit;
    }
    if (p.xcd_order == 2) {   // (the caller filled ntiles_grid / tiles_per_xcd)
        if (p.ntiles_grid != (int)grid.x || p.tiles_per_xcd * 8 < p.ntiles_grid) return hipErrorInvalidValue;
        grid = dim3((unsigned)(8 * p.tiles_per_xcd) * grid.y * grid.z, 1, 1);
    }
    const size_t lds = (size_t)p.lds_bytes;
    const void* kern = reinterpret_cast<const void*>(conv_f16x3<NT, KMT, NPH, DBG, MAXP, PK, D2S, F6, W2>);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((conv_f16x3<NT, KMT, NPH, DBG, MAXP, PK, D2S, F6, W2>), grid, dim3(W2 ? 512 : 256), lds, stream, p);
    return hipGetLastError();
}

// instantiations: MAXP = 4 everywhere; MAXP = 12 in addition for <= 5 N-tiles (12 only for the fused kernels of <= 3)
template <int NT, int KMT, int NPH>
static hipError_t launch_h_nt(const HConvParams& p, hipStream_t stream) {
    constexpr bool has4 = true, has12 = NT <= 5 && !(NPH == 4 && NT > 3);
    if (p.maxp != 4 && p.maxp != 12) return hipErrorInvalidValue;
    if (p.f6) {   // fp6 cross terms: the 9-tile plain / per-phase kernel (the planner asks for nothing else)
        if constexpr (NT == 9 && NPH == 1) {
            if (p.maxp == 4 && !p.pk && !p.d2s && !p.dbg) return launch_h_k<NT, KMT, NPH, false, 4, false, false, true>(p, stream);
        }
        return hipErrorInvalidValue;
    }
    if (p.w2) {   // two tiles per eight-wave workgroup: the 9-tile plain / per-phase kernel (the planner asks for nothing else)
        if constexpr (NT == 9 && NPH == 1) {
            if (p.maxp == 4 && !p.pk && !p.d2s && !p.dbg) return launch_h_k<NT, KMT, NPH, false, 4, false, false, false, true>(p, stream);
        }
        return hipErrorInvalidValue;
    }
    if (p.d2s) {   // depth-to-space transposed convolution: the wide plain kernels, 4 pixel indices per wave
        if constexpr (NT >= 5 && NPH == 1) {
            if constexpr (NT == 9) {   // (stamped twin for the tile count the bench graph uses)
                if (p.maxp == 4 && !p.pk && p.dbg) return launch_h_k<NT, KMT, NPH, true, 4, false, true>(p, stream);
            }
            if (p.maxp == 4 && !p.pk) return launch_h_k<NT, KMT, NPH, false, 4, false, true>(p, stream);
        }
        return hipErrorInvalidValue;
    }
    if (p.pk) {   // packed last N-tile: the narrow kernels only (the planner asks for it at <= 5 N-tiles), no stamped twins
        if constexpr (NT >= 2 && NT <= 5) {
            if constexpr (has12) {
                if (p.maxp == 12) return launch_h_k<NT, KMT, NPH, false, 12, true>(p, stream);
            }
            if (p.maxp == 4) return launch_h_k<NT, KMT, NPH, false, 4, true>(p, stream);
        }
        return hipErrorInvalidValue;
    }
    if constexpr (has12) {
        if (p.maxp == 12) {
            if constexpr (NT == 3 && NPH == 4) {   // (stamped twin of the top transposed convolution)
                if (p.dbg) return launch_h_k<NT, KMT, NPH, true, 12>(p, stream);
            }
            return launch_h_k<NT, KMT, NPH, false, 12>(p, stream);
        }
    }
    if constexpr (has4) {
        if (p.maxp == 4) {
            if constexpr ((NT == 3 || NT == 5 || NT == 9) && !(NPH == 4 && NT == 3)) {   // the stamped twins exist for the tile counts the bench graphs use
                if (p.dbg) return launch_h_k<NT, KMT, NPH, true, 4>(p, stream);
            }
            return launch_h_k<NT, KMT, NPH, false, 4>(p, stream);
        }
    }
    return hipErrorInvalidValue;
}

hipError_t launch_conv_f16(const HConvParams& p, hipStream_t stream) {
    if (p.fused_phases) {
        switch (p.NT) {
            case 1: return launch_h_nt<1, 2, 4>(p, stream);
            case 2: return launch_h_nt<2, 2, 4>(p, stream);
            case 3: return launch_h_nt<3, 2, 4>(p, stream);
            case 4: return launch_h_nt<4, 2, 4>(p, stream);
            case 5: return launch_h_nt<5, 2, 4>(p, stream);
            default: return hipErrorInvalidValue;
        }
    }
    switch (p.NT) {
        case 1: return launch_h_nt<1, kMT, 1>(p, stream);
        case 2: return launch_h_nt<2, kMT, 1>(p, stream);
        case 3: return launch_h_nt<3, kMT, 1>(p, stream);
        case 4: return launch_h_nt<4, kMT, 1>(p, stream);
        case 5: return launch_h_nt<5, kMT, 1>(p, stream);
        case 6: return launch_h_nt<6, kMT, 1>(p, stream);
        case 7: return launch_h_nt<7, kMT, 1>(p, stream);
        case 8: return launch_h_nt<8, kMT, 1>(p, stream);
        case 9: return launch_h_nt<9, kMT, 1>(p, stream);
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
// the compact form for the dense-K first layer and the raw-skip append: [pixel]{hi[CW] | lo[CW]}
template <int CW>
__global__ void __launch_bounds__(256) split_f32_compact_kernel(const float* __restrict__ x, size_t npix, int C, float scale,
                                                               _Float16* __restrict__ out) {
    for (size_t px = (size_t)blockIdx.x * blockDim.x + threadIdx.x; px < npix; px += (size_t)gridDim.x * blockDim.x) {
        _Float16 w[2 * CW];
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const float v = c < C ? x[px * C + c] * scale : 0.f;
            w[c] = (_Float16)v;
            w[CW + c] = (_Float16)(v - (float)w[c]);
        }
#pragma unroll
        for (int c = 0; c < 2 * CW; ++c) out[px * (2 * CW) + c] = w[c];
    }
}

hipError_t launch_split_f32(const float* x, size_t npix, int C, int Cs, float scale, _Float16* hi, _Float16* lo, int cw,
                            hipStream_t stream) {
    if (npix == 0) return hipSuccess;
    if (cw > 0) {
        const unsigned blocks = (unsigned)((npix + 255) / 256 < 256 * 16 ? (npix + 255) / 256 : 256 * 16);
        if (cw == 1) hipLaunchKernelGGL(split_f32_compact_kernel<1>, dim3(blocks), dim3(256), 0, stream, x, npix, C, scale, hi);
        else if (cw == 2) hipLaunchKernelGGL(split_f32_compact_kernel<2>, dim3(blocks), dim3(256), 0, stream, x, npix, C, scale, hi);
        else if (cw == 4) hipLaunchKernelGGL(split_f32_compact_kernel<4>, dim3(blocks), dim3(256), 0, stream, x, npix, C, scale, hi);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    const size_t total = npix * (size_t)Cs;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 256 * 16 ? (total + 255) / 256 : 256 * 16);
    hipLaunchKernelGGL(split_f32_kernel, dim3(blocks), dim3(256), 0, stream, x, npix, C, Cs, scale, hi, lo);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// PI2D.getPatch + per-tile normalisation + batch fill (reference PartitionOfImage.py:58-63,77-82, UnMicst1-5.py:700-702)
// written straight into the (hi, lo) binary16 input planes of the first convolution (channels padded to Cs = 8) -- or, CW > 0,
// into the compact form [pixel]{hi[CW] | lo[CW]} the dense-K first layer and the raw-skip append read (8 bytes per pixel
// instead of 32 for a two-channel input).  One thread per tile pixel.  Same float64 arithmetic as gather_normalise_kernel.
// ------------------------------------------------------------------------------------------------------------
template <int CW, int SRC>
__global__ void __launch_bounds__(256) gather_split_kernel(const void* __restrict__ image_v, double inv_max, int C_img, int band_row0,
                                                          int band_rows, TileGeom g, int Cn, double mean, double stdv,
                                                          int tile0, int ntiles, float scale, uint4* __restrict__ hi,
                                                          uint4* __restrict__ lo, const unsigned* __restrict__ mm) {
    constexpr int NC = CW > 0 ? CW : 8;
    const size_t total = (size_t)ntiles * g.P * g.P;
    // raw planes with the drivers' intensity rescale (mm != NULL: words [16 * plane] = the plane's min, max raw value): the float64
    // recipe of raw_to_double_kernel, operation for operation, in front of the normalisation
    double rlo[NC], rhi[NC];
    if constexpr (SRC != 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int pl = C_img == 1 ? 0 : (c < Cn ? c : 0);
            rlo[c] = mm ? __dmul_rn((double)mm[16 * pl], inv_max) : 0.0;
            rhi[c] = mm ? __dmul_rn((double)mm[16 * pl + 1], inv_max) : 0.0;
        }
    }
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % g.P);
        const size_t r1 = e / g.P;
        const int y = (int)(r1 % g.P);
        const int t = tile0 + (int)(r1 / g.P);
        const int pr = t / g.npc, pc = t - pr * g.npc;
        const int iy = pr * g.sub + y - g.margin, ix = pc * g.sub + x - g.margin;
        const bool inside = iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
        union { _Float16 h[8]; uint4 u; uint2 u2; unsigned u1; } vh, vl;
        vh.u = make_uint4(0, 0, 0, 0);
        vl.u = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float f = 0.f;
            if (c < Cn) {
                double v = 0.0;
                if (inside) {
                    const size_t at = ((size_t)(C_img == 1 ? 0 : c) * band_rows + (iy - band_row0)) * g.W + ix;
                    // SRC 1 / 2: the raw uint16 / uint8 plane; im2double is the host path's own multiply, one rounding, never fused
                    // with the subtraction below (umx_kernels.hip raw_to_double_kernel: np.multiply(I, 1.0 / imax))
                    if constexpr (SRC == 0) v = static_cast<const double*>(image_v)[at];
                    else if constexpr (SRC == 1) v = __dmul_rn((double)static_cast<const unsigned short*>(image_v)[at], inv_max);
                    else v = __dmul_rn((double)static_cast<const unsigned char*>(image_v)[at], inv_max);
                    if constexpr (SRC != 0) {
                        if (mm) {
                            if (rlo[c] != rhi[c]) v = __dmul_rn(__ddiv_rn(__dsub_rn(v, rlo[c]), __dsub_rn(rhi[c], rlo[c])), 0.983);
                            else v = fmin(fmax(v, 0.0), 0.983);
                        }
                    }
                }
                f = (float)((v - mean) / stdv) * scale;
            }
            vh.h[c] = (_Float16)f;
            vl.h[c] = (_Float16)(f - (float)vh.h[c]);
        }
        if constexpr (CW == 0) {
            hi[e] = vh.u;
            lo[e] = vl.u;
        } else if constexpr (CW == 1) {
            reinterpret_cast<unsigned*>(hi)[e] = (vh.u1 & 0xffffu) | (vl.u1 << 16);
        } else if constexpr (CW == 2) {
            reinterpret_cast<uint2*>(hi)[e] = make_uint2(vh.u1, vl.u1);
        } else {
            hi[e] = make_uint4(vh.u2.x, vh.u2.y, vl.u2.x, vl.u2.y);
        }
    }
}

// `raw_bits` 0: `image` holds float64 planes; 16 / 8: the raw integer planes (im2double happens in the gather)
hipError_t launch_gather_split(const void* image, int raw_bits, int C_img, int band_row0, int band_rows, const TileGeom& g, int Cn,
                               double mean, double stdv, int tile0, int ntiles, float scale, _Float16* hi, _Float16* lo, int cw,
                               hipStream_t stream, const unsigned* mm) {
    if (ntiles <= 0) return hipSuccess;
    if (mm && raw_bits == 0) return hipErrorInvalidValue;   // (the rescale rides on the raw sources only)
    if (Cn > 8 || (cw > 0 && Cn > cw) || (raw_bits != 0 && raw_bits != 8 && raw_bits != 16)) return hipErrorInvalidValue;
    const size_t total = (size_t)ntiles * g.P * g.P;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 256 * 16 ? (total + 255) / 256 : 256 * 16);
    uint4* const h4 = reinterpret_cast<uint4*>(hi);
    uint4* const l4 = reinterpret_cast<uint4*>(lo);
    const double inv_max = raw_bits == 16 ? 1.0 / 65535 : 1.0 / 255;
#define UMX_GS2(CWV, SRCV) hipLaunchKernelGGL((gather_split_kernel<CWV, SRCV>), dim3(blocks), dim3(256), 0, stream, image, inv_max, C_img, \
                                              band_row0, band_rows, g, Cn, mean, stdv, tile0, ntiles, scale, h4, l4, mm)
#define UMX_GS(CWV) do { if (raw_bits == 0) UMX_GS2(CWV, 0); else if (raw_bits == 16) UMX_GS2(CWV, 1); else UMX_GS2(CWV, 2); } while (0)
    if (cw == 0) UMX_GS(0);
    else if (cw == 1) UMX_GS(1);
    else if (cw == 2) UMX_GS(2);
    else if (cw == 4) UMX_GS(4);
    else return hipErrorInvalidValue;
#undef UMX_GS
#undef UMX_GS2
    return hipGetLastError();
}

}  // namespace umx
