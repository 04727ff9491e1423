// libumx training step: host side (plan, buffers, launch sequence) and the C ABI of include/umx_train.h.
//
// The step is written out layer by layer rather than as a generic autograd tape: the v2 graph is fixed
// (reference UnMicst1-5.py:83-237), so every tensor the backward pass needs is known at create time and lives in a
// preallocated HBM buffer (pre-BN conv outputs z, layer inputs, up-sampled tensors); nothing is recomputed except the
// cheap element-wise BN/activation/dropout chain, and nothing is allocated during a step.
//
//   forward   conv (fp32 MFMA, raw) -> per-channel batch statistics -> BN + LeakyReLU + dropout (+ 2x2 max-pool)
//   backward  activation/pool/dropout backward + BN reductions -> BN input gradient -> weight gradient (fp32 MFMA,
//             umx_train_kernels.hip) and input gradient (the forward conv kernel on flipped/transposed filters; for a
//             stride-2 transposed conv: a 2x2-tap conv over the space-to-depth form of the output gradient)
//   update    Adam / Momentum over the flat parameter vector
#include "../../include/umx_train.h"
#include "umx_internal.h"
#include "umx_kernels.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace umx;

namespace {

thread_local std::string g_terr;

struct TConv {                       // one convolution of the step: device-packed fp32 operands [tap][Cp][Np], rebuilt every step,
    ConvParams cp;                   // run by conv_mfma_f32 -- or (hidx >= 0, the default since round 4) repacked into conv_f16x3's
    int nt = 1, hpix = 2;            // weight images and run in split precision
    float* packed[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    double mac = 0.0;                // algorithmic multiply-accumulates per image
    int hidx = -1;                   // index into umx_trainer::hls (split-precision plan), -1: fp32 kernel
    bool direct = false;             // its weight images are gathered straight from the master tensors (no fp32 operand)
    float* winv = nullptr;           // device scalar 2^-s: undoes the scale of the repacked weights in the epilogue
    int wsh = 0;                     // s
    struct WSrc { size_t off, off2, cnt; };
    std::vector<WSrc> wsrc;          // master tensor(s) each operand group reads: where the scale is re-derived from (refresh_wscales)
};

struct H16 {                         // (hi, lo) binary16 NHWC planes of one tensor, channels padded to Cs (what conv_f16x3 reads)
    _Float16* hi = nullptr;
    _Float16* lo = nullptr;
    int Cs = 0;
};

constexpr int kSlots = 4;   // dz / gS buffers the main stream may run ahead of the weight gradients by

struct TapSet {
    std::vector<std::pair<int, int>> off;   // (dy, dx) input offsets
    std::vector<int> m;                     // master tap index per (tap, parity) : size off.size() * npar
};

struct BnSite {            // one batch-normalised tensor
    int C = 0, H = 0, W = 0;
    size_t gamma = 0, beta = 0, mean = 0, var = 0;   // offsets in the parameter vector
    float* z = nullptr;    // [B,H,W,C] pre-BN
    float* stat = nullptr; // [4][C]
    float* m12 = nullptr;  // [2][C]
    unsigned* gmax = nullptr;   // max |dz| of the current step (float bits), for the split-precision weight gradient
    unsigned* bw = nullptr;     // three words: max_c |gamma rstd| (bn_finalize), max |g| and max |xhat| (act_bwd) -- the bound on |dz|
                                // that scales its (hi, lo) planes before the tensor exists (bn_bwd_apply)
};

struct Seg { std::string name; size_t off, n; float reg; };

}  // namespace

struct umx_trainer {
    umx_hparams hp;
    umx_train_options o;
    int device = 0, B = 0, L = 0, K = 0, P = 0;
    std::vector<int> n;                 // channel widths
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<void*> allocs;
    int64_t step = 0;
    // parameters
    size_t nparams = 0;
    std::vector<Seg> segs;
    float *d_w = nullptr, *d_g = nullptr, *d_m = nullptr, *d_v = nullptr;
    // per-layer offsets into the parameter vector
    std::vector<size_t> o_w1, o_ws, o_wt, o_w2;
    size_t o_lb = 0, o_lt = 0;
    // activations
    std::vector<float*> ds;             // ds[0] = data, ds[i+1] = pooled output of down layer i
    std::vector<BnSite> bn_d, bn_u;     // down layers / up layers (index = idx)
    BnSite bn_b, bn_t;
    float* act_b = nullptr;             // bottom output
    std::vector<float*> us, cv;         // per up layer idx
    float *d_labels = nullptr, *d_weights = nullptr, *d_probs = nullptr, *d_dt = nullptr;
    std::vector<float*> dskip;          // gradient w.r.t. ds[idx] from the up path (idx >= 1)
    float *DA = nullptr, *DB = nullptr, *DZ = nullptr, *GS = nullptr;
    float* DZ2[kSlots] = {};   // gradient w.r.t. a conv output, one per slot: the weight gradients of layer l run on a side stream
    float* GS2[kSlots] = {};   // while the main stream moves on to layers l+1 .. l+nslots-1
    int nslots = kSlots;
    hipStream_t side = nullptr;
    hipStream_t side2 = nullptr;          // a second side stream: the dz / gS slots alternate between the two
    hipEvent_t ev_join2 = nullptr;
    hipEvent_t ev_dz[kSlots] = {}, ev_gs[kSlots] = {}, ev_side[kSlots] = {}, ev_join = nullptr;
    hipStream_t aux = nullptr;                    // the skip connections' input gradients: needed only on the way back down the U, so
    hipEvent_t ev_aux[kSlots] = {};               // they leave the main stream's dependent chain (slot free again / gradient ready)
    std::vector<hipEvent_t> ev_skip;
    bool overlap = true;
    double* d_part = nullptr;  size_t part_doubles = 0;
    double* d_part2 = nullptr;          // the side stream's own partial sums (regularisation loss under the forward pass)
    double* d_loss = nullptr;           // [0] data term, [1] regularisation
    unsigned* d_maxw = nullptr;  int n_maxw = 0;   // per-tensor max |gradient| words, then the binary16 range flag
    std::vector<unsigned*> smax;        // max |gS| per up layer
    std::vector<unsigned*> dsmax, usmax, cvmax;   // max |activation| of ds[i], us[idx], cv[idx]
    unsigned* bmax = nullptr;           // ... of the bottom layer's output
    float* d_ws = nullptr;  size_t ws_floats = 0;
    float* d_ws2 = nullptr;             // ... of the second side stream
    float* d_split = nullptr;  size_t split_floats = 0;   // partial outputs of K-split convolutions
    float* d_split2 = nullptr;                            // ... of those enqueued on the side stream
    float* d_split3 = nullptr;                            // ... on the aux stream
    // forward / input-gradient convolutions on conv_f16x3 (UMX_TRAIN_CONV_F32=1: the exact-fp32 kernels of rounds 1-3)
    bool range_pending = false;         // a training step raised the range flag and an eval pass cleared it before umx_trainer_loss saw it
    bool hconv = true;
    bool wg_planes = true;              // the split-precision weight gradient stages from the planes
    umx_ctx* pctx = nullptr;            // owner of the planner's device allocations (stage tables, weight slabs, constants)
    const float* h_blob = nullptr;      // (during build) the initial parameters on the host: weight scales
    std::vector<umx::Launch> hls;
    std::vector<TConv*> hconvs;         // the convolutions that took the split-precision route, index = RepackDesc::owner
    int wscale_every = 256;             // steps between two refreshes of the weight scales (UMX_TRAIN_WSCALE_EVERY)
    std::vector<float> h_params;        // (refresh) host copy of the parameters
    std::vector<RepackDesc> rdescs;
    RepackDesc* d_rdescs = nullptr;
    int max_refs = 0;
    bool cur_bwd = false;               // (during build) the convolution being set up belongs to the backward pass
    int n_fwd_packs = 0, n_fwd_rdescs = 0;   // descriptors [0, n_fwd) serve the forward pass, the rest the backward pass only
    hipEvent_t ev_begin = nullptr, ev_packed = nullptr;
    std::vector<H16> h_ds, h_us, h_cv;  // planes of ds[i], us[idx], cv[idx]
    H16 h_b, h_dz[kSlots], h_gs[kSlots];
    float* d_xinv = nullptr;            // [2 * kSlots] inverse scales of the dz / gS slots' planes, written by split_dyn_kernel
    // launches
    std::vector<TConv> c_fwd_d, c_dg_d, c_T, c_fwd_u, c_dg_us, c_dg_skip, c_dg_T;
    TConv c_fwd_b, c_dg_b;
    std::vector<WgradParams> wg_d, wg_u0, wg_u1, wg_T;
    WgradParams wg_b;
    std::vector<PackDesc> packs;
    PackDesc* d_packs = nullptr;
    RegSeg* d_regsegs = nullptr;  int n_regsegs = 0;
    size_t max_pack = 0;
    double flops_per_image = 0.0;
    // profiling
    bool prof = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    double t_fwd = 0, t_bwd = 0, t_opt = 0;
    int t_steps = 0;
    bool pending = false;
};

namespace {

int tfail(umx_trainer* tr, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (tr) tr->err = buf; else g_terr = buf;
    return code;
}

#define T_HIP(tr, call)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return tfail(tr, e_ == hipErrorOutOfMemory ? UMX_ERR_OOM : UMX_ERR_HIP, "%s failed: %s", #call, \
                         hipGetErrorString(e_));                                                          \
    } while (0)
#define T_TRY(call)                  \
    do {                             \
        int rc_ = (call);            \
        if (rc_ != UMX_OK) return rc_; \
    } while (0)

template <typename T>
int talloc(umx_trainer* tr, T** out, size_t count) {
    void* d = nullptr;
    T_HIP(tr, hipMalloc(&d, std::max<size_t>(16, count * sizeof(T))));
    tr->allocs.push_back(d);
    *out = reinterpret_cast<T*>(d);
    return UMX_OK;
}

template <typename T>
int tzero(umx_trainer* tr, T** out, size_t count) {
    T_TRY(talloc(tr, out, count));
    T_HIP(tr, hipMemset(*out, 0, std::max<size_t>(16, count * sizeof(T))));
    return UMX_OK;
}

// N-tiles per workgroup.  A training batch is small (8 images): the deep layers have a handful of 256-pixel M-tiles, so
// wide N blocks would leave most of the 256 CUs idle.  Cost model: rounds of 512 resident workgroups (2 per CU) times
// the work of one workgroup (NT MFMAs per fragment pair + a fixed staging share); padded N counts as work.
void choose_nt(int Cout, int mtiles, int* nt, int* Np) {
    const int t16 = (Cout + 15) / 16;
    int best = 1;
    double best_cost = 1e300;
    for (int c = 1; c <= kMaxNT; ++c) {
        const long wgs = (long)mtiles * ((t16 + c - 1) / c);
        const double cost = (double)((wgs + 511) / 512) * (c + 0.5);
        if (cost < best_cost - 1e-9 || (std::fabs(cost - best_cost) <= 1e-9 && c > best)) { best = c; best_cost = cost; }
    }
    *nt = best;
    *Np = round_up(t16, best) * 16;
}

// Geometry + packed-operand buffers + pack descriptors of one conv launch.
//   groups: ngroups sources with C[g] channels; per phase and group a TapSet (master taps of tensor w_off);
//   pack_mode per group: transpose flag, channel offset on the master's d2 axis, (npar, Cblk), optional second tensor
struct GroupSpec {
    int C;                      // source channels seen by the kernel
    size_t w_off, w2_off;       // master tensor(s) in the parameter vector (w2_off = SIZE_MAX: none)
    int d2, d3;                 // master dims [taps][d2][d3]
    int transpose, c_off, npar, Cblk;
    TapSet taps[4];
};

// largest |w| a convolution's packed operand can hold: the master tensor(s) each group reads (a summed pair: the sum of the maxima)
double operand_wmax(const TConv& tc, const float* params) {
    double wmax = 0.0;
    for (const TConv::WSrc& w : tc.wsrc) {
        double m1 = 0.0, m2 = 0.0;
        for (size_t i = 0; i < w.cnt; ++i) m1 = std::max(m1, (double)std::fabs(params[w.off + i]));
        if (w.off2 != SIZE_MAX)
            for (size_t i = 0; i < w.cnt; ++i) m2 = std::max(m2, (double)std::fabs(params[w.off2 + i]));
        wmax = std::max(wmax, m1 + m2);
    }
    return wmax;
}
// s such that wmax * 2^s is in [2^10, 2^11): 32 x headroom below binary16's largest finite value for the weights to grow into
int wscale_shift(double wmax) {
    if (!(wmax > 0.0) || !std::isfinite(wmax)) return 0;
    int e;
    std::frexp(wmax, &e);              // wmax = m * 2^e, m in [0.5, 1)
    return std::max(-24, std::min(40, 11 - e));
}

// The same convolution as a split-precision plan (conv_f16x3, fp32 output): stage tables and the LDS-image layout of the weights
// come from plan_f16 once; the values are filled every step by repack_f16x3_kernel from the fp32 operands this TConv already
// rebuilds on the device.  A shape the planner refuses stays on the fp32 kernel.
int setup_hconv(umx_trainer* tr, TConv& tc, const char* what, int H, int W, int Cout, int act, int nphase, int o_mul, const int* oy,
                const int* ox, int ngroups, GroupSpec* gs) {
    if (!tr->hconv) return UMX_OK;
    umx::Launch L;
    L.name = what;
    L.train = true;
    L.ngroups = ngroups; L.nphase = nphase; L.o_mul = o_mul;
    for (int ph = 0; ph < nphase; ++ph) { L.oy_off[ph] = oy ? oy[ph] : 0; L.ox_off[ph] = ox ? ox[ph] : 0; }
    L.H = H; L.W = W; L.Cout = Cout; L.outH = H * o_mul; L.outW = W * o_mul; L.pool = 0; L.act = act; L.dst = -1;
    L.Np = tc.cp.Np;
    double wmax = 0.0;
    for (int g = 0; g < ngroups; ++g) {
        L.g[g].src = g;
        L.g[g].C = gs[g].C;
        for (int ph = 0; ph < nphase; ++ph) L.g[g].taps[ph] = gs[g].taps[ph].off;
        if (gs[g].npar > 1 && gs[g].Cblk % 8 == 0) {
            const int noct = (gs[g].C + 7) / 8;
            for (int ph = 0; ph < nphase; ++ph) {
                const TapSet& ts = gs[g].taps[ph];
                L.g[g].dead[ph].assign(ts.off.size() * (size_t)noct, 0);
                for (size_t t = 0; t < ts.off.size(); ++t)
                    for (int o = 0; o < noct; ++o)
                        if (ts.m[t * gs[g].npar + (size_t)(8 * o / gs[g].Cblk)] < 0) L.g[g].dead[ph][t * noct + o] = 1;
            }
        }
        // largest |w| the packed operand can hold now: the master tensor(s) this group reads (summed pair: the sum of the maxima)
        int ntap_master = 0;
        for (int ph = 0; ph < nphase; ++ph)
            for (int m : gs[g].taps[ph].m) ntap_master = std::max(ntap_master, m + 1);
        const size_t cnt = (size_t)ntap_master * gs[g].d2 * gs[g].d3;
        tc.wsrc.push_back({gs[g].w_off, gs[g].w2_off, cnt});
    }
    wmax = operand_wmax(tc, tr->h_blob);
    std::string why;
    if (!umx::conv_geometry(L, &why) || umx::plan_f16(tr->pctx, L, 0, true, nullptr, &why) != UMX_OK) {
        if (getenv("UMX_DEBUG_PLAN")) fprintf(stderr, "[umx train] %s stays on the fp32 kernel: %s\n", what, why.c_str());
        return UMX_OK;
    }
    const int sh = wscale_shift(wmax);
    tc.wsh = sh;
    const float inv = std::ldexp(1.f, -sh);
    T_TRY(talloc(tr, &tc.winv, 1));
    T_HIP(tr, hipMemcpy(tc.winv, &inv, sizeof inv, hipMemcpyHostToDevice));
    static_assert(sizeof(umx::HWRef) == sizeof(HWRefDev) && sizeof(HWRefDev) == 12, "reference record layout");
    // The planner's references address the packed fp32 operand [tap][Cp][Np]; rewritten here into the master tensor's own
    // coordinates (the index arithmetic of pack_weights_kernel, done once on the host), the per-step repack reads the parameters
    // directly and this convolution needs no fp32 operand at all.  An octet that straddles two parity blocks keeps the packed route.
    bool direct = true;
    for (int ph = 0; ph < nphase && direct; ++ph)
        for (const umx::HWRef& r : L.wrefs[ph]) {
            const GroupSpec& G = gs[r.arr];
            const int c0 = (int)(((size_t)r.base / L.Np) % round_up(G.C, 4));
            if (c0 / G.Cblk != (c0 + r.nvalid - 1) / G.Cblk) { direct = false; break; }
        }
    if (direct)
        for (int ph = 0; ph < nphase; ++ph) {
            std::vector<umx::HWRef> out;
            out.reserve(L.wrefs[ph].size());
            for (const umx::HWRef& r : L.wrefs[ph]) {
                const GroupSpec& G = gs[r.arr];
                const int Cp = round_up(G.C, 4);
                const int co = (int)((size_t)r.base % L.Np);
                const size_t row = (size_t)r.base / L.Np;
                const int c0 = (int)(row % Cp), t = (int)(row / Cp);
                const int par = c0 / G.Cblk, cc = c0 - par * G.Cblk;
                const int m = G.taps[ph].m[(size_t)t * G.npar + par];
                if (m < 0) continue;   // a tap this parity block does not have: the unit stays zero
                const size_t idx = G.transpose ? ((size_t)m * G.d2 + G.c_off + co) * G.d3 + cc : ((size_t)m * G.d2 + G.c_off + cc) * G.d3 + co;
                if (idx > (size_t)INT_MAX) return tfail(tr, UMX_ERR_INVALID, "%s: master tensor too large for a weight reference", what);
                out.push_back(umx::HWRef{r.dst, (int)idx, r.nvalid, r.arr});
            }
            L.wrefs[ph].swap(out);
        }
    tc.direct = direct;
    for (int ph = 0; ph < nphase; ++ph) {
        const std::vector<umx::HWRef>& refs = L.wrefs[ph];
        if (refs.empty()) continue;
        HWRefDev* d = nullptr;
        T_TRY(talloc(tr, &d, refs.size()));
        T_HIP(tr, hipMemcpy(d, refs.data(), refs.size() * sizeof(HWRefDev), hipMemcpyHostToDevice));
        RepackDesc rd;
        memset(&rd, 0, sizeof rd);
        rd.refs = d;
        rd.n = (int)refs.size();
        for (int g = 0; g < ngroups; ++g) {
            rd.arr[g] = direct ? tr->d_w + gs[g].w_off : tc.packed[ph][g];
            rd.arr2[g] = direct && gs[g].w2_off != SIZE_MAX ? tr->d_w + gs[g].w2_off : nullptr;
            rd.estride[g] = direct ? (gs[g].transpose ? 1 : gs[g].d3) : 0;
        }
        rd.stride = tc.cp.Np;
        rd.scale = std::ldexp(1.f, sh);
        rd.slab = const_cast<uint4*>(L.hcp.ph[ph].w);
        rd.bwd = tr->cur_bwd ? 1 : 0;
        rd.owner = (int)tr->hconvs.size();
        tr->rdescs.push_back(rd);
        tr->max_refs = std::max(tr->max_refs, rd.n);
        std::vector<umx::HWRef>().swap(L.wrefs[ph]);
    }
    {   // K split: enough workgroups to occupy the chip (two per CU) where a batch of 8 leaves a deep layer a few dozen
        HConvParams& h = L.hcp;
        const long wgs = (long)((tr->B + h.imgs - 1) / h.imgs) * h.tiles_y * h.tiles_x * h.nblocks * nphase;
        // (one workgroup per CU: 512 / 768 / 1024 lose 3 / 7 / 10 % of the step to the ordered reduce over more partial sums, 128 / 192
        // lose 1 % to idle CUs -- profiles/r04/train_ksplit_sweep.txt)
        const long target = 256;
        int S = (int)std::min<long>(kMaxKSplit, wgs > 0 ? (target + wgs - 1) / wgs : 1);
        if (getenv("UMX_TRAIN_NO_KSPLIT") || wgs * 2 > target) S = 1;
        std::vector<std::vector<int>> starts(nphase);   // per phase: stage indices (relative) where a halo chunk begins
        for (int ph = 0; ph < nphase && S > 1; ++ph) {
            for (int si = 0; si < h.ph[ph].nstages; ++si)
                if (L.stages_host[h.ph[ph].stage0 + si].group >= 0) starts[ph].push_back(si);
            S = std::min(S, (int)starts[ph].size());
        }
        if (S > 1) {
            for (int ph = 0; ph < nphase; ++ph) {
                const int nch = (int)starts[ph].size();
                for (int k = 0; k < S; ++k) h.ks0[ph][k] = (short)starts[ph][(size_t)((long)nch * k / S)];
                h.ks0[ph][S] = (short)h.ph[ph].nstages;
            }
            h.ksplit = S;
            h.split_stride = (size_t)tr->B * L.outH * L.outW * Cout;
            if (h.xcd_order == 2) h.xcd_order = 1;
            tr->split_floats = std::max(tr->split_floats, (size_t)S * h.split_stride);
        }
        if (getenv("UMX_DEBUG_PLAN")) {
            int nk = 0;
            for (int ph = 0; ph < nphase; ++ph)
                for (int si = 0; si < h.ph[ph].nstages; ++si) nk += L.stages_host[h.ph[ph].stage0 + si].nk;
            fprintf(stderr, "[umx train] %s%s: %d x %d -> %d ch, %d phase(s), NT %d x %d blocks, %d k-steps, %ld workgroups, K split %d\n",
                    what, tr->cur_bwd ? " (backward)" : "", H, W, Cout, nphase, h.NT, h.nblocks, nk, wgs, S);
        }
    }
    std::vector<HStage>().swap(L.stages_host);
    tc.hidx = (int)tr->hls.size();
    tr->hls.push_back(std::move(L));
    tr->hconvs.push_back(&tc);
    return UMX_OK;
}

int setup_conv(umx_trainer* tr, TConv& tc, const char* what, int H, int W, int Cout, int act, int nphase, int o_mul,
               const int* oy, const int* ox, int ngroups, GroupSpec* gs) {
    ConvParams& p = tc.cp;
    memset(&p, 0, sizeof p);
    {
        const int TWm0 = std::min(16, W), TH0 = std::min(16, H);
        const int imgs0 = (16 / TWm0) * (16 / TH0);
        choose_nt(Cout, ((tr->B + imgs0 - 1) / imgs0) * (H / TH0) * (W / TWm0) * nphase, &tc.nt, &p.Np);
    }
    int ymin = 0, ymax = 0, xmin = 0, xmax = 0, ntaps_total = 0;
    for (int g = 0; g < ngroups; ++g)
        for (int ph = 0; ph < nphase; ++ph)
            for (auto& t : gs[g].taps[ph].off) {
                ymin = std::min(ymin, t.first); ymax = std::max(ymax, t.first);
                xmin = std::min(xmin, t.second); xmax = std::max(xmax, t.second);
                ++ntaps_total;
            }
    if (ntaps_total > kMaxTaps) return tfail(tr, UMX_ERR_INVALID, "%s: too many filter taps", what);
    auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    const int TWm = std::min(16, W), TH = std::min(16, H);
    if ((TWm & (TWm - 1)) || (TH & (TH - 1)))
        return tfail(tr, UMX_ERR_INVALID, "%s: layer size %d must be a power of two", what, H);
    p.twm_log2 = lg2(TWm);
    p.th_log2 = lg2(TH);
    p.nimg_m = 16 / TWm;
    p.imgs = p.nimg_m * (16 / TH);
    p.hh = TH + ymax - ymin;
    p.hw = TWm + xmax - xmin;
    p.imgplane = p.hh * p.hw;
    int plane = round_up(p.imgs * p.imgplane, 32) + 16;
    if (plane - 32 >= p.imgs * p.imgplane) plane -= 32;
    p.plane = plane;
    p.ymin = ymin;
    p.xmin = xmin;
    p.tiles_y = H / TH;
    p.tiles_x = W / TWm;
    tc.hpix = (p.imgs * p.imgplane + 255) / 256;
    if (tc.hpix > 4) return tfail(tr, UMX_ERR_INVALID, "%s: halo too large", what);
    tc.hpix = tc.hpix <= 2 ? 2 : 4;
    p.ngroups = ngroups;
    p.H = H; p.W = W; p.Cout = Cout;
    p.nphase = nphase; p.o_mul = o_mul;
    p.outH = H * o_mul; p.outW = W * o_mul; p.pool = 0; p.act = act;
    {   // K split: a deep layer of a small batch has a few dozen workgroups, each walking thousands of K steps with
        // exposed staging latency -- spread the channel chunks over more workgroups (partials + ordered reduce)
        const int img_groups = (tr->B + p.imgs - 1) / p.imgs;
        const long wgs = (long)img_groups * p.tiles_y * p.tiles_x * nphase * (p.Np / 16 / tc.nt);
        int min_chunks = 1 << 30;
        for (int g = 0; g < ngroups; ++g) min_chunks = std::min(min_chunks, (round_up(gs[g].C, 4) + kCC - 1) / kCC);
        int S = 1;
        const long target = 1536, smax = 8;
        if (wgs < target / 2 && !getenv("UMX_TRAIN_NO_KSPLIT"))
            S = (int)std::max<long>(1, std::min<long>(std::min<long>(smax, target / wgs), min_chunks / 4));
        p.ksplit = S;
        p.split_stride = (size_t)tr->B * p.outH * p.outW * Cout;
        if (S > 1) tr->split_floats = std::max(tr->split_floats, (size_t)S * p.split_stride);
    }
    if (conv_lds_bytes(tc.nt, p.plane) > 160 * 1024) return tfail(tr, UMX_ERR_INVALID, "%s: LDS footprint too large", what);
    int tpos = 0;
    for (int ph = 0; ph < nphase; ++ph) {
        p.ph[ph].oy_off = oy ? oy[ph] : 0;
        p.ph[ph].ox_off = ox ? ox[ph] : 0;
        for (int g = 0; g < ngroups; ++g) {
            const TapSet& ts = gs[g].taps[ph];
            p.ph[ph].tap0[g] = tpos;
            p.ph[ph].ntaps[g] = (int)ts.off.size();
            for (auto& t : ts.off) p.tapoff[tpos++] = (short)((t.first - ymin) * p.hw + (t.second - xmin));
        }
    }
    for (int g = 0; g < ngroups; ++g) {
        p.C[g] = gs[g].C;
        p.Cp[g] = round_up(gs[g].C, 4);
        p.vec4[g] = (gs[g].C % 4) == 0;
    }
    // packed operands + descriptors
    size_t npacks = 0;
    for (int ph = 0; ph < nphase; ++ph)
        for (int g = 0; g < ngroups; ++g) {
            const TapSet& ts = gs[g].taps[ph];
            const int nt = (int)ts.off.size();
            if (nt == 0) continue;
            if (nt * gs[g].npar > 4 * kMaxPackTaps) return tfail(tr, UMX_ERR_INVALID, "%s: too many taps to pack", what);
            const size_t elems = (size_t)nt * p.Cp[g] * p.Np;
            T_TRY(talloc(tr, &tc.packed[ph][g], elems));
            p.ph[ph].w[g] = tc.packed[ph][g];
            PackDesc d;
            memset(&d, 0, sizeof d);
            d.dst = tc.packed[ph][g];
            d.w = tr->d_w + gs[g].w_off;
            d.w2 = gs[g].w2_off == SIZE_MAX ? nullptr : tr->d_w + gs[g].w2_off;
            d.ntaps = nt; d.Cp = p.Cp[g]; d.Np = p.Np; d.C = gs[g].C; d.N = Cout;
            d.d2 = gs[g].d2; d.d3 = gs[g].d3;
            d.transpose = gs[g].transpose; d.c_off = gs[g].c_off;
            d.npar = gs[g].npar; d.Cblk = gs[g].Cblk;
            for (size_t i = 0; i < ts.m.size(); ++i) d.mtap[i] = (short)ts.m[i];
            d.bwd = tr->cur_bwd ? 1 : 0;
            tr->packs.push_back(d);
            ++npacks;
            tr->max_pack = std::max(tr->max_pack, elems);
            tc.mac += (double)H * W * nt * gs[g].C * Cout;
        }
    const size_t packs_before = tr->packs.size() - npacks;
    T_TRY(setup_hconv(tr, tc, what, H, W, Cout, act, nphase, o_mul, oy, ox, ngroups, gs));
    if (tc.hidx >= 0 && tc.direct) tr->packs.resize(packs_before);   // nothing reads this convolution's fp32 operands
    return UMX_OK;
}

int alloc_h16(umx_trainer* tr, H16& h, size_t npix, int C) {
    h.Cs = round_up(std::max(C, 1), 8);
    T_TRY(tzero(tr, &h.hi, npix * h.Cs));
    T_TRY(tzero(tr, &h.lo, npix * h.Cs));
    return UMX_OK;
}

// fp32 NHWC tensor -> the (hi, lo) planes conv_f16x3 reads (maxw / inv: a gradient tensor's dynamic scale, else NULL)
int to_h16(umx_trainer* tr, const float* x, size_t npix, int C, const H16& h, const unsigned* maxw, float* inv, hipStream_t st,
           unsigned* omax = nullptr) {
    if (!tr->hconv) return UMX_OK;
    T_HIP(tr, launch_split_dyn(x, npix, C, h.Cs, maxw, inv, h.hi, h.lo, reinterpret_cast<int*>(tr->d_maxw + tr->n_maxw), omax, st));
    return UMX_OK;
}

int run_hconv(umx_trainer* tr, TConv& tc, const H16& s0, const H16* s1, float* dst, const float* xinv, hipStream_t st) {
    HConvParams p = tr->hls[tc.hidx].hcp;
    p.B = tr->B;
    p.overflow_flag = reinterpret_cast<int*>(tr->d_maxw + tr->n_maxw);
    const H16* src[2] = {&s0, s1 ? s1 : &s0};
    for (int gi = 0; gi < 2; ++gi) {
        p.src_hi[gi] = gi == 0 || s1 ? src[gi]->hi : nullptr;
        p.src_lo[gi] = gi == 0 || s1 ? src[gi]->lo : nullptr;
        p.Cs[gi] = src[gi]->Cs;
        p.srcA[gi] = src[gi]->Cs * 2;
        p.srcB[gi] = 16;
    }
    p.dst_f32 = dst;
    p.dyn[0] = tc.winv;
    p.dyn[1] = xinv;
    if (p.ksplit > 1) {   // raw partial sums, then the ordered reduce applies the activation
        const int act = p.act;
        p.act = ACT_NONE;
        float* const scratch = st == tr->stream ? tr->d_split : st == tr->aux ? tr->d_split3 : tr->d_split2;   // (one scratch per stream)
        p.dst_f32 = scratch;
        T_HIP(tr, launch_conv_f16(p, st));
        T_HIP(tr, launch_split_reduce(scratch, p.ksplit, p.split_stride, p.split_stride, act, dst, st));
        return UMX_OK;
    }
    T_HIP(tr, launch_conv_f16(p, st));
    return UMX_OK;
}

int pack_all(umx_trainer* tr, hipStream_t st, int part);
int refresh_wscales(umx_trainer* tr);

int run_conv(umx_trainer* tr, TConv& tc, const float* src0, const float* src1, float* dst, bool on_side = false) {
    ConvParams p = tc.cp;
    p.src[0] = src0;
    p.src[1] = src1;
    p.dst = dst;
    p.B = tr->B;
    hipStream_t st = on_side ? tr->side : tr->stream;
    float* split = on_side ? tr->d_split2 : tr->d_split;     // each stream has its own K-split scratch
    if (p.ksplit > 1) {
        p.dst = split;
        p.act = ACT_NONE;
        T_HIP(tr, launch_conv(p, tc.nt, tc.hpix, st));
        T_HIP(tr, launch_split_reduce(split, p.ksplit, p.split_stride, p.split_stride, tc.cp.act, dst, st));
        return UMX_OK;
    }
    T_HIP(tr, launch_conv(p, tc.nt, tc.hpix, st));
    return UMX_OK;
}

TapSet same_taps(int ks, bool flipped) {   // stride-1 SAME conv (or its input gradient: offsets negated)
    TapSet t;
    const int ph = (ks - 1) / 2;
    for (int a = 0; a < ks; ++a)
        for (int b = 0; b < ks; ++b) {
            t.off.push_back(flipped ? std::make_pair(ph - a, ph - b) : std::make_pair(a - ph, b - ph));
            t.m.push_back(a * ks + b);
        }
    return t;
}

int setup_wgrad(umx_trainer* tr, WgradParams& w, const char* what, int H, int Cxt, int Cx, int Cg, const TapSet& slabs,
                const std::vector<int>& coff) {
    memset(&w, 0, sizeof w);
    w.B = tr->B; w.H = H; w.W = H;
    w.Cxt = Cxt; w.Cx = Cx; w.Cg = Cg;
    w.nslab = (int)slabs.off.size();
    if (w.nslab > kWgMaxSlabs) return tfail(tr, UMX_ERR_INVALID, "%s: too many filter taps", what);
    for (int s = 0; s < w.nslab; ++s) {
        w.dy[s] = (short)slabs.off[s].first;
        w.dx[s] = (short)slabs.off[s].second;
        w.mslab[s] = (short)slabs.m[s];
        w.coff[s] = coff.empty() ? 0 : coff[s];
    }
    std::string why;
    if (!wgrad_setup(&w, &why)) return tfail(tr, UMX_ERR_INVALID, "%s: %s", what, why.c_str());
    tr->ws_floats = std::max(tr->ws_floats, wgrad_ws_floats(w));
    if (getenv("UMX_DEBUG_PLAN")) {
        std::string groups;
        for (int g = 0; g < w.ngroups; ++g) groups += (g ? "/" : "") + std::to_string(w.gcount[g]);
        fprintf(stderr, "[umx train] wgrad %s: %d x %d, X %d (of %d) ch x G %d ch, %d slabs in groups %s, %s, %d img/tile, %d tiles, %d slices x %d tiles, "
                        "grid %d x %d x %d\n",
                what, H, H, Cx, Cxt, Cg, w.nslab, groups.c_str(), w.thin ? "thin" : w.f16 ? "f16x3" : "fp32 mfma", w.imgs, w.ntiles, w.nslices,
                w.tiles_per_slice, w.nslices,
                w.thin ? 1 : w.f16 ? ((Cx + 47) / 48) * ((Cg + 47) / 48) : ((Cx + 16 * w.mi - 1) / (16 * w.mi)) * ((Cg + 63) / 64), w.thin ? 1 : w.ngroups);
    }
    return UMX_OK;
}

unsigned long long mix64_host(unsigned long long x) {
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
unsigned long long drop_key(const umx_trainer* tr, int layer_id) {
    return mix64_host(tr->o.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(64 * tr->step + layer_id + 1));
}

int bn_alloc(umx_trainer* tr, BnSite& s, int C, int H, size_t off_gamma) {
    s.C = C; s.H = s.W = H;
    s.gamma = off_gamma; s.beta = off_gamma + C; s.mean = off_gamma + 2 * (size_t)C; s.var = off_gamma + 3 * (size_t)C;
    T_TRY(talloc(tr, &s.z, (size_t)tr->B * H * H * C));
    T_TRY(talloc(tr, &s.stat, 4 * (size_t)C));
    T_TRY(talloc(tr, &s.m12, 2 * (size_t)C));
    return UMX_OK;
}

// z -> statistics -> stat (and the moving averages when updating)
int bn_forward_stats(umx_trainer* tr, BnSite& s, bool update, bool training = true) {
    if (!training) {
        T_HIP(tr, launch_bn_stat_from_moving(s.C, tr->d_w + s.gamma, tr->d_w + s.beta, tr->d_w + s.mean, tr->d_w + s.var,
                                             s.stat, tr->stream));
        return UMX_OK;
    }
    const size_t N = (size_t)tr->B * s.H * s.W;
    const int nblk = chan_blocks(N, s.C);
    T_HIP(tr, launch_chan_stats(s.z, N, s.C, tr->d_part, nblk, tr->stream));
    T_HIP(tr, launch_bn_finalize(tr->d_part, nblk, N, s.C, tr->d_w + s.gamma, tr->d_w + s.beta, tr->d_w + s.mean,
                                 tr->d_w + s.var, update ? tr->o.bn_momentum : 1.0f, s.stat, s.bw, tr->stream));
    return UMX_OK;
}

ActParams act_params(const umx_trainer* tr, const BnSite& s, int pool, int act, float rate, int layer_id) {
    ActParams a;
    a.z = s.z; a.stat = s.stat;
    a.B = tr->B; a.H = s.H; a.W = s.W; a.C = s.C;
    a.pool = pool; a.act = act;
    a.drop_rate = rate > 0.f ? rate : 0.f;
    a.drop_key = drop_key(tr, layer_id);
    return a;
}

// dy (+dy1) -> gradient w.r.t. z in tr->DZ ; BN parameter gradients into the gradient vector
// (planes / inv: where an input-gradient convolution on conv_f16x3 reads dz, its (hi, lo) planes are written in the same pass)
int bn_backward(umx_trainer* tr, BnSite& s, const ActParams& a, const float* dy0, const float* dy1, float* dz,
                const H16* planes = nullptr, float* inv = nullptr) {
    const size_t N = (size_t)tr->B * s.H * s.W;
    const size_t rows = a.pool ? N / 4 : N;
    const int nblk = chan_blocks(rows, s.C);
    T_HIP(tr, launch_act_bwd(a, dy0, dy1, dz, tr->d_part, nblk, planes ? s.bw + 1 : nullptr, tr->stream));
    T_HIP(tr, launch_bn_bwd_finalize(tr->d_part, nblk, N, s.C, tr->d_g + s.gamma, tr->d_g + s.beta, s.m12, tr->stream));
    T_HIP(tr, launch_bn_bwd_apply_max(dz, s.z, s.stat, s.m12, N, s.C, s.gmax, s.bw, inv, planes ? planes->hi : nullptr,
                                      planes ? planes->lo : nullptr, planes ? planes->Cs : 0,
                                      reinterpret_cast<int*>(tr->d_maxw + tr->n_maxw), tr->stream));
    return UMX_OK;
}

// (xp / gp: the operands' (hi, lo) planes and xi / gi the inverse scales they were stored with, where they exist: the split-precision
// kernel then stages from them -- half the bytes, no conversion)
int run_wgrad(umx_trainer* tr, WgradParams& w, const float* X, const float* G, int Ctot, int c_off, size_t w_off,
              float reg, size_t pair_off /* SIZE_MAX: none */, const unsigned* xmax, const unsigned* gmax,
              hipStream_t stream, const H16* xp = nullptr, const float* xi = nullptr, const H16* gp = nullptr,
              const float* gi = nullptr) {
    w.X = X;
    w.G = G;
    w.planes = 0;
    if (w.f16 && tr->wg_planes && xp && gp && xp->hi && gp->hi && w.Cxt <= xp->Cs && w.Cg <= gp->Cs) {
        w.planes = 1;
        w.Xhi = xp->hi; w.Xlo = xp->lo; w.XCs = xp->Cs;
        w.Ghi = gp->hi; w.Glo = gp->lo; w.GCs = gp->Cs;
        w.xinv = xi; w.ginv = gi;
    }
    w.ws = stream == tr->side2 ? tr->d_ws2 : tr->d_ws;   // (each side stream has its own slice workspace)
    w.xmax = xmax;
    w.gmax = gmax;
    w.overflow = reinterpret_cast<int*>(tr->d_maxw + tr->n_maxw);
    T_HIP(tr, launch_wgrad(w, stream));
    T_HIP(tr, launch_wgrad_reduce(w, Ctot, c_off, tr->d_g + w_off, tr->d_w + w_off, reg > 0.f ? tr->o.reg_kind : 0, reg,
                                  pair_off == SIZE_MAX ? nullptr : tr->d_g + pair_off, stream));
    return UMX_OK;
}

float up_rate(const umx_trainer* tr, int idx) { return tr->o.drop_up0 - tr->o.drop_up_step * idx; }
float down_rate(const umx_trainer* tr, int i) { return tr->o.drop_down_step * i; }

enum { LAYER_DOWN = 0, LAYER_BOTTOM = 16, LAYER_UP = 32 };   // dropout stream ids (oracle/train_oracle.py)

// forward of the graph up to the top layer's pre-BN output (UnMicst1-5.py:83-222).  training: batch statistics + dropout;
// otherwise the moving statistics, no dropout (tfTraining: 0)
int forward_pass(umx_trainer* tr, const float* data, bool training, bool update) {
    const int L = tr->L;
    hipStream_t st = tr->stream;
    const umx_train_options& o = tr->o;
    auto rate = [&](float r) { return training ? r : 0.f; };
    // (training: every activation that the backward pass feeds to the split-precision weight gradient gets its max |v|
    // tracked by the kernel that writes it -- the input batch and the transposed-conv outputs by a pass of their own)
    // a convolution on whichever route its plan took: conv_f16x3 reads the (hi, lo) planes, conv_mfma_f32 the fp32 tensors
    auto conv = [&](TConv& tc, const float* x0, const float* x1, const H16* h0, const H16* h1, float* dst) -> int {
        if (tc.hidx >= 0) return run_hconv(tr, tc, *h0, h1, dst, nullptr, st);
        return run_conv(tr, tc, x0, x1, dst);
    };
    const size_t Bz = (size_t)tr->B;
    int* const flagp = reinterpret_cast<int*>(tr->d_maxw + tr->n_maxw);
    // (the (hi, lo) planes of an activation are written by the kernel that produces it)
    static const H16 kNoPlanes;   // (the exact-fp32 route has no plane buffers: the vectors below are empty then)
#define PL(h) (tr->hconv ? (h) : kNoPlanes).hi, (tr->hconv ? (h) : kNoPlanes).lo, (tr->hconv ? (h) : kNoPlanes).Cs, flagp
    tr->ds[0] = const_cast<float*>(data);
    if (training && !tr->hconv) T_HIP(tr, launch_absmax(data, Bz * tr->P * tr->P * tr->n[0], tr->dsmax[0], st));
    if (tr->hconv) T_TRY(to_h16(tr, data, Bz * tr->P * tr->P, tr->n[0], tr->h_ds[0], nullptr, nullptr, st, training ? tr->dsmax[0] : nullptr));
    int S = tr->P;
    for (int i = 0; i < L; ++i) {
        BnSite& s = tr->bn_d[i];
        T_TRY(conv(tr->c_fwd_d[i], tr->ds[i], nullptr, tr->hconv ? &tr->h_ds[i] : nullptr, nullptr, s.z));
        T_TRY(bn_forward_stats(tr, s, update, training));
        T_HIP(tr, launch_act_fwd(act_params(tr, s, 1, ACT_LEAKY, rate(down_rate(tr, i)), LAYER_DOWN + i), tr->ds[i + 1],
                                 training ? tr->dsmax[i + 1] : nullptr, PL(tr->h_ds[i + 1]), st));
        S /= 2;
    }
    T_TRY(conv(tr->c_fwd_b, tr->ds[L], nullptr, tr->hconv ? &tr->h_ds[L] : nullptr, nullptr, tr->bn_b.z));
    T_TRY(bn_forward_stats(tr, tr->bn_b, update, training));
    T_HIP(tr, launch_act_fwd(act_params(tr, tr->bn_b, 0, ACT_LEAKY, rate(o.drop_bottom), LAYER_BOTTOM), tr->act_b,
                             training ? tr->bmax : nullptr, PL(tr->h_b), st));
    const float* cur = tr->act_b;
    const H16* hcur = tr->hconv ? &tr->h_b : nullptr;
    for (int idx = L - 1; idx >= 0; --idx) {
        BnSite& s = tr->bn_u[idx];
        S *= 2;
        T_TRY(conv(tr->c_T[idx], cur, nullptr, hcur, nullptr, tr->us[idx]));
        // (max |us| for the split-precision weight gradient: tracked by the plane split where there is one)
        if (training && !tr->hconv) T_HIP(tr, launch_absmax(tr->us[idx], Bz * S * S * tr->n[idx + 1], tr->usmax[idx], st));
        if (tr->hconv) T_TRY(to_h16(tr, tr->us[idx], Bz * S * S, tr->n[idx + 1], tr->h_us[idx], nullptr, nullptr, st,
                                    training ? tr->usmax[idx] : nullptr));
        T_TRY(conv(tr->c_fwd_u[idx], tr->ds[idx], tr->us[idx], tr->hconv ? &tr->h_ds[idx] : nullptr, tr->hconv ? &tr->h_us[idx] : nullptr, s.z));
        T_TRY(bn_forward_stats(tr, s, update, training));
        if (tr->hconv && idx >= 1) {
            T_HIP(tr, launch_act_fwd(act_params(tr, s, 0, ACT_LEAKY, rate(up_rate(tr, idx)), LAYER_UP + idx), tr->cv[idx],
                                     training ? tr->cvmax[idx] : nullptr, PL(tr->h_cv[idx]), st));
        } else {
            T_HIP(tr, launch_act_fwd(act_params(tr, s, 0, ACT_LEAKY, rate(up_rate(tr, idx)), LAYER_UP + idx), tr->cv[idx],
                                     training ? tr->cvmax[idx] : nullptr, nullptr, nullptr, 0, flagp, st));
        }
        cur = tr->cv[idx];
        hcur = tr->hconv && idx >= 1 ? &tr->h_cv[idx] : nullptr;
    }
#undef PL
    BnSite& t = tr->bn_t;
    T_HIP(tr, launch_head_fwd(tr->cv[0], (size_t)tr->B * tr->P * tr->P, tr->n[1], tr->K, tr->d_w + tr->o_lt, t.z, st));
    T_TRY(bn_forward_stats(tr, t, update, training));
    return UMX_OK;
}

int enqueue_step(umx_trainer* tr, const float* data, const float* labels, const float* weights, bool update) {
    const int L = tr->L, K = tr->K, B = tr->B, P = tr->P;
    hipStream_t st = tr->stream;
    const umx_train_options& o = tr->o;
    const size_t Npix = (size_t)B * P * P;
    if (update && tr->step > 0 && tr->wscale_every > 0 && tr->step % tr->wscale_every == 0) T_TRY(refresh_wscales(tr));
    if (tr->prof) T_HIP(tr, hipEventRecord(tr->ev[0], st));
    T_HIP(tr, hipMemsetAsync(tr->d_loss, 0, 2 * sizeof(double), st));
    // (the range flag behind the max words is NOT cleared here: it stays up until umx_trainer_loss / umx_trainer_eval has
    // reported it, so an overflow in a step whose loss is never read is not lost under the next step -- ADVICE r4)
    T_HIP(tr, hipMemsetAsync(tr->d_maxw, 0, tr->n_maxw * sizeof(unsigned), st));
    if (tr->overlap) {
        // the side stream is idle until the first weight gradient: it packs the backward pass's operands and sums the regularisation
        // loss (both need nothing but this step's weights) while the main stream runs the forward pass
        T_HIP(tr, hipEventRecord(tr->ev_begin, st));
        T_HIP(tr, hipStreamWaitEvent(tr->side, tr->ev_begin, 0));
        T_TRY(pack_all(tr, st, 0));
        T_TRY(pack_all(tr, tr->side, 1));
        if (o.reg_kind != UMX_REG_NONE)
            T_HIP(tr, launch_reg_loss(tr->d_regsegs, tr->n_regsegs, o.reg_kind, tr->d_part2, tr->d_loss, 1, tr->side));
        T_HIP(tr, hipEventRecord(tr->ev_packed, tr->side));
    } else {
        T_TRY(pack_all(tr, st, 2));
    }

    T_TRY(forward_pass(tr, data, true, update));
    BnSite& t = tr->bn_t;
    {
        const int nblk = loss_blocks(Npix);
        T_HIP(tr, launch_softmax_loss(t.z, t.stat, labels, weights, Npix, K, o.clip_eps, tr->d_probs, tr->d_dt, tr->d_part,
                                      nblk, st));
        T_HIP(tr, launch_sum_to_scalar(tr->d_part, nblk, 1.0 / (double)Npix, tr->d_loss, 0, 0, st));
    }
    if (o.reg_kind != UMX_REG_NONE && !tr->overlap)
        T_HIP(tr, launch_reg_loss(tr->d_regsegs, tr->n_regsegs, o.reg_kind, tr->d_part, tr->d_loss, 1, st));
    if (tr->overlap) T_HIP(tr, hipStreamWaitEvent(st, tr->ev_packed, 0));   // the backward pass's operands are in place
    if (tr->prof) T_HIP(tr, hipEventRecord(tr->ev[1], st));

    // ------------------------------------------------------------------ backward
    {   // top layer: softmax/CE gradient -> BN -> 1x1 conv
        ActParams a = act_params(tr, t, 0, ACT_NONE, 0.f, 0);
        T_TRY(bn_backward(tr, t, a, tr->d_dt, nullptr, tr->d_dt));   // in place: dt -> dt0
        const int nblk = chan_blocks(Npix, tr->n[1]);
        T_HIP(tr, launch_head_bwd(tr->cv[0], tr->d_dt, tr->d_w + tr->o_lt, Npix, tr->n[1], K, tr->DA, tr->d_part, nblk, st));
        T_HIP(tr, launch_reduce_partials(tr->d_part, nblk, tr->n[1] * K, 1.0, tr->d_g + tr->o_lt, tr->d_w + tr->o_lt,
                                         o.reg_top > 0.f ? o.reg_kind : 0, o.reg_top, st));
    }
    // The weight gradients hang off the critical path (gradient w.r.t. conv output -> input gradient -> next layer):
    // they run on a side stream against double-buffered dz / gS while the main stream continues.  Ordering is by events
    // only; every kernel still has a fixed summation order, so the step stays bit-reproducible.
    // (two side streams, one per dz / gS slot: consecutive layers' weight gradients -- small grids that leave most CUs idle at a batch
    // of 8 -- run next to each other as well as next to the main stream)
    hipStream_t wss[2] = {tr->overlap ? tr->side : st, tr->overlap ? (tr->side2 ? tr->side2 : tr->side) : st};
#define ws wss[slot & 1]
    int slot = 0;
    bool used[kSlots] = {}, aux_used[kSlots] = {};
    const bool use_aux = tr->overlap && tr->aux && tr->hconv;
    auto dz_begin = [&](int sl) -> int {      // main: the slot's previous consumers on the side stream are done
        if (tr->overlap && used[sl]) T_HIP(tr, hipStreamWaitEvent(st, tr->ev_side[sl], 0));
        if (aux_used[sl]) { T_HIP(tr, hipStreamWaitEvent(st, tr->ev_aux[sl], 0)); aux_used[sl] = false; }
        return UMX_OK;
    };
    auto dz_ready = [&](int sl) -> int {      // main has written DZ2[sl]; the side stream may read it
        if (tr->overlap) {
            T_HIP(tr, hipEventRecord(tr->ev_dz[sl], st));
            T_HIP(tr, hipStreamWaitEvent(wss[sl & 1], tr->ev_dz[sl], 0));
        }
        return UMX_OK;
    };
    auto side_done = [&](int sl) -> int {
        if (tr->overlap) T_HIP(tr, hipEventRecord(tr->ev_side[sl], wss[sl & 1]));
        used[sl] = true;
        return UMX_OK;
    };
    auto dgrad = [&](TConv& tc, const float* x, const H16& h, const float* xinv, float* dst) -> int {
        if (tc.hidx >= 0) return run_hconv(tr, tc, h, nullptr, dst, xinv, st);
        return run_conv(tr, tc, x, nullptr, dst);
    };
    int S = P;
    for (int idx = 0; idx < L; ++idx) {    // up layers, output side first
        BnSite& s = tr->bn_u[idx];
        const int Cskip = tr->n[idx], Cup = tr->n[idx + 1];
        const float* layer_in = idx == L - 1 ? tr->act_b : tr->cv[idx + 1];
        float* dz = tr->DZ2[slot];
        float* gs = tr->GS2[slot];
        ActParams a = act_params(tr, s, 0, ACT_LEAKY, up_rate(tr, idx), LAYER_UP + idx);
        T_TRY(dz_begin(slot));
        tr->h_dz[slot].Cs = round_up(Cup, 8);
        T_TRY(bn_backward(tr, s, a, tr->DA, nullptr, dz, tr->hconv ? &tr->h_dz[slot] : nullptr, tr->d_xinv + slot));
        T_TRY(dz_ready(slot));
        const H16* const pdz = tr->hconv ? &tr->h_dz[slot] : nullptr;
        T_TRY(run_wgrad(tr, tr->wg_u0[idx], tr->ds[idx], dz, Cskip + Cup, 0, tr->o_w2[idx], o.reg_up, SIZE_MAX, tr->dsmax[idx], s.gmax, ws,
                        tr->hconv ? &tr->h_ds[idx] : nullptr, nullptr, pdz, tr->d_xinv + slot));
        T_TRY(run_wgrad(tr, tr->wg_u1[idx], tr->us[idx], dz, Cskip + Cup, Cskip, tr->o_w2[idx], o.reg_up, SIZE_MAX, tr->usmax[idx], s.gmax, ws,
                        tr->hconv ? &tr->h_us[idx] : nullptr, nullptr, pdz, tr->d_xinv + slot));
        T_TRY(dgrad(tr->c_dg_us[idx], dz, tr->h_dz[slot], tr->d_xinv + slot, tr->DB));
        if (idx >= 1 && use_aux && tr->c_dg_skip[idx].hidx >= 0) {   // (on the weight gradients' stream it was 2 % slower; this stream is its own)
            T_HIP(tr, hipStreamWaitEvent(tr->aux, tr->ev_dz[slot], 0));
            T_TRY(run_hconv(tr, tr->c_dg_skip[idx], tr->h_dz[slot], nullptr, tr->dskip[idx], tr->d_xinv + slot, tr->aux));
            T_HIP(tr, hipEventRecord(tr->ev_aux[slot], tr->aux));
            T_HIP(tr, hipEventRecord(tr->ev_skip[idx], tr->aux));
            aux_used[slot] = true;
        } else if (idx >= 1) {
            T_TRY(dgrad(tr->c_dg_skip[idx], dz, tr->h_dz[slot], tr->d_xinv + slot, tr->dskip[idx]));
        }
        T_HIP(tr, launch_leaky_bwd_s2d_max(tr->DB, tr->us[idx], B, S / 2, Cup, gs, tr->smax[idx], st));
        const bool gs_planes = tr->hconv && (tr->c_dg_T[idx].hidx >= 0 || tr->wg_planes);
        if (gs_planes) {   // (in front of the event: the transposed convolution's weight gradient stages from these planes too)
            tr->h_gs[slot].Cs = round_up(4 * Cup, 8);
            T_TRY(to_h16(tr, gs, (size_t)B * (S / 2) * (S / 2), 4 * Cup, tr->h_gs[slot], tr->smax[idx], tr->d_xinv + kSlots + slot, st));
        }
        if (tr->overlap) {
            T_HIP(tr, hipEventRecord(tr->ev_gs[slot], st));
            T_HIP(tr, hipStreamWaitEvent(ws, tr->ev_gs[slot], 0));
        }
        {
            const H16* const pin = !tr->hconv ? nullptr : idx == L - 1 ? &tr->h_b : &tr->h_cv[idx + 1];
            T_TRY(run_wgrad(tr, tr->wg_T[idx], gs, layer_in, Cup, 0, tr->o_wt[idx], o.reg_up, SIZE_MAX, tr->smax[idx],
                            idx == L - 1 ? tr->bmax : tr->cvmax[idx + 1], ws, gs_planes ? &tr->h_gs[slot] : nullptr, tr->d_xinv + kSlots + slot, pin,
                            nullptr));
        }
        T_TRY(side_done(slot));
        T_TRY(dgrad(tr->c_dg_T[idx], gs, tr->h_gs[slot], tr->d_xinv + kSlots + slot, tr->DA));
        slot = (slot + 1) % tr->nslots;
        S /= 2;
    }
    {   // bottom layer
        float* dz = tr->DZ2[slot];
        ActParams a = act_params(tr, tr->bn_b, 0, ACT_LEAKY, o.drop_bottom, LAYER_BOTTOM);
        T_TRY(dz_begin(slot));
        tr->h_dz[slot].Cs = round_up(tr->n[L + 1], 8);
        T_TRY(bn_backward(tr, tr->bn_b, a, tr->DA, nullptr, dz, tr->hconv ? &tr->h_dz[slot] : nullptr, tr->d_xinv + slot));
        T_TRY(dz_ready(slot));
        T_TRY(run_wgrad(tr, tr->wg_b, tr->ds[L], dz, tr->n[L], 0, tr->o_lb, o.reg_bottom, SIZE_MAX, tr->dsmax[L], tr->bn_b.gmax, ws,
                        tr->hconv ? &tr->h_ds[L] : nullptr, nullptr, tr->hconv ? &tr->h_dz[slot] : nullptr, tr->d_xinv + slot));
        T_TRY(side_done(slot));
        T_TRY(dgrad(tr->c_dg_b, dz, tr->h_dz[slot], tr->d_xinv + slot, tr->DB));
        slot = (slot + 1) % tr->nslots;
    }
    for (int i = L - 1; i >= 0; --i) {     // down layers
        BnSite& s = tr->bn_d[i];
        float* dz = tr->DZ2[slot];
        ActParams a = act_params(tr, s, 1, ACT_LEAKY, down_rate(tr, i), LAYER_DOWN + i);
        const float* dy1 = (i + 1 <= L - 1) ? tr->dskip[i + 1] : nullptr;
        if (dy1 && use_aux && tr->c_dg_skip[i + 1].hidx >= 0) T_HIP(tr, hipStreamWaitEvent(st, tr->ev_skip[i + 1], 0));
        T_TRY(dz_begin(slot));
        tr->h_dz[slot].Cs = round_up(tr->n[i + 1], 8);
        T_TRY(bn_backward(tr, s, a, tr->DB, dy1, dz, tr->hconv && i >= 1 ? &tr->h_dz[slot] : nullptr, tr->d_xinv + slot));
        T_TRY(dz_ready(slot));
        // c00 + shortcut = conv(x, W1 + Wshort): both filters receive the same data gradient (UnMicst1-5.py:102-114)
        T_TRY(run_wgrad(tr, tr->wg_d[i], tr->ds[i], dz, tr->n[i], 0, tr->o_ws[i], o.reg_down, tr->o_w1[i], tr->dsmax[i], s.gmax, ws,
                        tr->hconv && i >= 1 ? &tr->h_ds[i] : nullptr, nullptr, tr->hconv && i >= 1 ? &tr->h_dz[slot] : nullptr,
                        tr->d_xinv + slot));
        T_TRY(side_done(slot));
        S *= 2;
        if (i >= 1) T_TRY(dgrad(tr->c_dg_d[i], dz, tr->h_dz[slot], tr->d_xinv + slot, tr->DB));
        slot = (slot + 1) % tr->nslots;
    }
#undef ws
    if (tr->overlap) {   // join: the optimiser (and the caller) see every gradient
        T_HIP(tr, hipEventRecord(tr->ev_join, wss[0]));
        T_HIP(tr, hipStreamWaitEvent(st, tr->ev_join, 0));
        if (wss[1] != wss[0]) {
            T_HIP(tr, hipEventRecord(tr->ev_join2, wss[1]));
            T_HIP(tr, hipStreamWaitEvent(st, tr->ev_join2, 0));
        }
    }
    if (tr->prof) T_HIP(tr, hipEventRecord(tr->ev[2], st));

    // ------------------------------------------------------------------ update (UnMicst1-5.py:361-380)
    if (update) {
        OptParams op;
        op.kind = o.optimizer;
        const double lr = (double)o.lr0 * std::pow((double)o.decay_rate, (double)(tr->step / std::max(1, o.decay_steps)));
        const double tt = (double)(tr->step + 1);
        op.lr = (float)lr;
        op.lr_t = (float)(lr * std::sqrt(1.0 - std::pow((double)o.beta2, tt)) / (1.0 - std::pow((double)o.beta1, tt)));
        op.beta1 = o.beta1; op.beta2 = o.beta2; op.eps = o.adam_eps; op.momentum = o.momentum;
        T_HIP(tr, launch_optimizer(op, tr->d_w, tr->d_g, tr->d_m, tr->d_v, tr->nparams, st));
        tr->step += 1;
    }
    if (tr->prof) {
        T_HIP(tr, hipEventRecord(tr->ev[3], st));
        tr->pending = true;
    }
    return UMX_OK;
}

int fold_profile(umx_trainer* tr) {
    if (!tr->pending) return UMX_OK;
    T_HIP(tr, hipEventSynchronize(tr->ev[3]));
    float a = 0, b = 0, c = 0;
    T_HIP(tr, hipEventElapsedTime(&a, tr->ev[0], tr->ev[1]));
    T_HIP(tr, hipEventElapsedTime(&b, tr->ev[1], tr->ev[2]));
    T_HIP(tr, hipEventElapsedTime(&c, tr->ev[2], tr->ev[3]));
    tr->t_fwd += a; tr->t_bwd += b; tr->t_opt += c;
    tr->t_steps += 1;
    tr->pending = false;
    return UMX_OK;
}

void fill_common(umx_train_options* o) {
    memset(o, 0, sizeof *o);
    o->optimizer = UMX_OPT_ADAM;
    o->momentum = 0.9f;
    o->beta1 = 0.9f; o->beta2 = 0.999f; o->adam_eps = 1e-8f;
    o->bn_momentum = 0.99f;
    o->seed = 1234;
}

int build_trainer(umx_trainer* tr, const float* blob, size_t blob_floats) {
    const umx_hparams& hp = tr->hp;
    const int L = hp.nLayers, ks = hp.ks, P = hp.imSize, K = hp.nClasses, B = tr->B;
    tr->L = L; tr->K = K; tr->P = P;
    tr->n = {hp.nChannels, hp.nOut0};
    for (int i = 0; i < L; ++i) tr->n.push_back(tr->n.back() * hp.featMapsFact);
    const std::vector<int>& n = tr->n;
    const umx_train_options& o = tr->o;

    // ---- parameter vector layout (== unmicst_amd/model.py tensor_specs, nExtraConvs == 0)
    size_t pos = 0;
    auto seg = [&](const std::string& name, size_t cnt, float reg) { tr->segs.push_back({name, pos, cnt, reg}); pos += cnt; return pos - cnt; };
    std::vector<size_t> bn_d_off(L), bn_u_off(L);
    size_t bn_b_off, bn_t_off;
    tr->o_w1.resize(L); tr->o_ws.resize(L); tr->o_wt.resize(L); tr->o_w2.resize(L);
    char nm[64];
    for (int i = 0; i < L; ++i) {
        snprintf(nm, sizeof nm, "ld%d", i);
        tr->o_w1[i] = seg(std::string(nm) + ".w1", (size_t)ks * ks * n[i] * n[i + 1], 0.f);
        tr->o_ws[i] = seg(std::string(nm) + ".wshort", (size_t)ks * ks * n[i] * n[i + 1], o.reg_down);
        bn_d_off[i] = seg(std::string(nm) + ".bn", 4 * (size_t)n[i + 1], 0.f);
    }
    tr->o_lb = seg("lb.w", (size_t)ks * ks * n[L] * n[L + 1], o.reg_bottom);
    bn_b_off = seg("lb.bn", 4 * (size_t)n[L + 1], 0.f);
    for (int idx = L - 1; idx >= 0; --idx) {
        snprintf(nm, sizeof nm, "lu%d", idx);
        tr->o_wt[idx] = seg(std::string(nm) + ".wt", (size_t)ks * ks * n[idx + 1] * n[idx + 2], o.reg_up);
        tr->o_w2[idx] = seg(std::string(nm) + ".w2", (size_t)ks * ks * (n[idx] + n[idx + 1]) * n[idx + 1], o.reg_up);
        bn_u_off[idx] = seg(std::string(nm) + ".bn", 4 * (size_t)n[idx + 1], 0.f);
    }
    tr->o_lt = seg("lt.w", (size_t)n[1] * K, o.reg_top);
    bn_t_off = seg("lt.bn", 4 * (size_t)K, 0.f);
    tr->nparams = pos;
    if (blob_floats != pos)
        return tfail(tr, UMX_ERR_BLOB, "weight blob has %zu floats, the graph needs %zu", blob_floats, pos);

    T_TRY(talloc(tr, &tr->d_w, pos));
    T_HIP(tr, hipMemcpy(tr->d_w, blob, pos * sizeof(float), hipMemcpyHostToDevice));
    T_TRY(tzero(tr, &tr->d_g, pos));
    T_TRY(tzero(tr, &tr->d_m, pos));
    T_TRY(tzero(tr, &tr->d_v, pos));

    // ---- activations
    tr->ds.assign(L + 1, nullptr);
    tr->bn_d.resize(L); tr->bn_u.resize(L);
    tr->us.assign(L, nullptr); tr->cv.assign(L, nullptr); tr->dskip.assign(L, nullptr);
    size_t max_act = (size_t)B * P * P * std::max(n[0], K);
    int S = P;
    for (int i = 0; i < L; ++i) {
        T_TRY(bn_alloc(tr, tr->bn_d[i], n[i + 1], S, bn_d_off[i]));
        max_act = std::max(max_act, (size_t)B * S * S * n[i + 1]);
        T_TRY(talloc(tr, &tr->ds[i + 1], (size_t)B * (S / 2) * (S / 2) * n[i + 1]));
        if (i + 1 <= L - 1) T_TRY(talloc(tr, &tr->dskip[i + 1], (size_t)B * (S / 2) * (S / 2) * n[i + 1]));
        S /= 2;
    }
    T_TRY(bn_alloc(tr, tr->bn_b, n[L + 1], S, bn_b_off));
    T_TRY(talloc(tr, &tr->act_b, (size_t)B * S * S * n[L + 1]));
    max_act = std::max(max_act, (size_t)B * S * S * n[L + 1]);
    for (int idx = L - 1; idx >= 0; --idx) {
        S *= 2;
        T_TRY(talloc(tr, &tr->us[idx], (size_t)B * S * S * n[idx + 1]));
        T_TRY(bn_alloc(tr, tr->bn_u[idx], n[idx + 1], S, bn_u_off[idx]));
        T_TRY(talloc(tr, &tr->cv[idx], (size_t)B * S * S * n[idx + 1]));
        max_act = std::max(max_act, (size_t)B * S * S * n[idx + 1]);
    }
    T_TRY(bn_alloc(tr, tr->bn_t, K, P, bn_t_off));
    T_TRY(talloc(tr, &tr->d_labels, (size_t)B * P * P * K));
    T_TRY(talloc(tr, &tr->d_weights, (size_t)B * P * P * K));
    T_TRY(talloc(tr, &tr->d_probs, (size_t)B * P * P * K));
    T_TRY(talloc(tr, &tr->d_dt, (size_t)B * P * P * K));
    T_TRY(talloc(tr, &tr->ds[0], (size_t)B * P * P * n[0]));
    T_TRY(talloc(tr, &tr->DA, max_act));
    T_TRY(talloc(tr, &tr->DB, max_act));
    T_TRY(talloc(tr, &tr->DZ, max_act));
    T_TRY(talloc(tr, &tr->GS, max_act));
    tr->DZ2[0] = tr->DZ; tr->GS2[0] = tr->GS;
    for (int sl = 1; sl < tr->nslots; ++sl) {
        T_TRY(talloc(tr, &tr->DZ2[sl], max_act));
        T_TRY(talloc(tr, &tr->GS2[sl], max_act));
    }
    int maxC = K;
    for (int v : n) maxC = std::max(maxC, v);
    tr->part_doubles = std::max<size_t>((size_t)1024 * 2 * maxC, (size_t)1024 * n[1] * K) + 1024;
    T_TRY(talloc(tr, &tr->d_part, tr->part_doubles));
    T_TRY(talloc(tr, &tr->d_part2, tr->part_doubles));
    T_TRY(tzero(tr, &tr->d_loss, 2));
    {   // max-|gradient| words: one per batch-normalised tensor and per up layer, + the range flag
        tr->n_maxw = 6 * L + 4 + 3 * (2 * L + 2);
        T_TRY(tzero(tr, &tr->d_maxw, (size_t)tr->n_maxw + 1));
        int k = 0;
        for (int i = 0; i < L; ++i) tr->bn_d[i].gmax = tr->d_maxw + k++;
        for (int i = 0; i < L; ++i) tr->bn_u[i].gmax = tr->d_maxw + k++;
        tr->bn_b.gmax = tr->d_maxw + k++;
        tr->bn_t.gmax = tr->d_maxw + k++;
        tr->smax.assign(L, nullptr);
        for (int i = 0; i < L; ++i) tr->smax[i] = tr->d_maxw + k++;
        tr->dsmax.assign(L + 1, nullptr); tr->usmax.assign(L, nullptr); tr->cvmax.assign(L, nullptr);
        for (int i = 0; i <= L; ++i) tr->dsmax[i] = tr->d_maxw + k++;
        for (int i = 0; i < L; ++i) tr->usmax[i] = tr->d_maxw + k++;
        for (int i = 0; i < L; ++i) tr->cvmax[i] = tr->d_maxw + k++;
        tr->bmax = tr->d_maxw + k++;
        for (int i = 0; i < L; ++i) { tr->bn_d[i].bw = tr->d_maxw + k; k += 3; }
        for (int i = 0; i < L; ++i) { tr->bn_u[i].bw = tr->d_maxw + k; k += 3; }
        tr->bn_b.bw = tr->d_maxw + k; k += 3;
        tr->bn_t.bw = tr->d_maxw + k; k += 3;
    }

    // ---- split-precision route of the forward / input-gradient convolutions: planes of every tensor they read
    tr->hconv = !getenv("UMX_TRAIN_CONV_F32");
    if (const char* e = getenv("UMX_TRAIN_WSCALE_EVERY")) tr->wscale_every = atoi(e);
    tr->wg_planes = tr->hconv;
    tr->h_blob = blob;
    if (tr->hconv) {
        tr->pctx = new umx_ctx();
        tr->pctx->device = tr->device;
        tr->pctx->hp = hp;
        tr->h_ds.resize(L + 1); tr->h_us.resize(L); tr->h_cv.resize(L);
        int S2 = P;
        size_t max_dz = 0, max_gs = 0;
        for (int i = 0; i <= L; ++i) {
            T_TRY(alloc_h16(tr, tr->h_ds[i], (size_t)B * S2 * S2, n[i]));
            if (i < L) max_dz = std::max(max_dz, (size_t)B * S2 * S2 * round_up(n[i + 1], 8));
            if (i < L) S2 /= 2;
        }
        T_TRY(alloc_h16(tr, tr->h_b, (size_t)B * S2 * S2, n[L + 1]));
        max_dz = std::max(max_dz, (size_t)B * S2 * S2 * round_up(n[L + 1], 8));
        for (int idx = L - 1; idx >= 0; --idx) {
            max_gs = std::max(max_gs, (size_t)B * S2 * S2 * round_up(4 * n[idx + 1], 8));
            S2 *= 2;
            T_TRY(alloc_h16(tr, tr->h_us[idx], (size_t)B * S2 * S2, n[idx + 1]));
            if (idx >= 1) T_TRY(alloc_h16(tr, tr->h_cv[idx], (size_t)B * S2 * S2, n[idx + 1]));
            max_dz = std::max(max_dz, (size_t)B * S2 * S2 * round_up(n[idx + 1], 8));
        }
        for (int sl = 0; sl < tr->nslots; ++sl) {   // gradient planes, one set per dz / gS slot; Cs is set per use
            T_TRY(tzero(tr, &tr->h_dz[sl].hi, max_dz)); T_TRY(tzero(tr, &tr->h_dz[sl].lo, max_dz));
            T_TRY(tzero(tr, &tr->h_gs[sl].hi, max_gs)); T_TRY(tzero(tr, &tr->h_gs[sl].lo, max_gs));
        }
        T_TRY(tzero(tr, &tr->d_xinv, 2 * kSlots));
    }

    // ---- conv launches
    tr->c_fwd_d.resize(L); tr->c_dg_d.resize(L); tr->c_T.resize(L); tr->c_fwd_u.resize(L);
    tr->c_dg_us.resize(L); tr->c_dg_skip.resize(L); tr->c_dg_T.resize(L);
    tr->wg_d.resize(L); tr->wg_u0.resize(L); tr->wg_u1.resize(L); tr->wg_T.resize(L);
    const TapSet fwd = same_taps(ks, false), flp = same_taps(ks, true);
    auto group = [&](int C, size_t w_off, size_t w2_off, int d2, int d3, int transpose, int c_off, const TapSet& ts) {
        GroupSpec g;
        g.C = C; g.w_off = w_off; g.w2_off = w2_off; g.d2 = d2; g.d3 = d3; g.transpose = transpose; g.c_off = c_off;
        g.npar = 1; g.Cblk = std::max(C, 1);
        g.taps[0] = ts;
        return g;
    };
    double mac_total = 0.0;
    S = P;
    for (int i = 0; i < L; ++i) {
        snprintf(nm, sizeof nm, "ld%d", i);
        GroupSpec g = group(n[i], tr->o_w1[i], tr->o_ws[i], n[i], n[i + 1], 0, 0, fwd);
        T_TRY(setup_conv(tr, tr->c_fwd_d[i], nm, S, S, n[i + 1], ACT_NONE, 1, 1, nullptr, nullptr, 1, &g));
        mac_total += tr->c_fwd_d[i].mac;
        if (i >= 1) {
            GroupSpec gd = group(n[i + 1], tr->o_w1[i], tr->o_ws[i], n[i], n[i + 1], 1, 0, flp);
            tr->cur_bwd = true; T_TRY(setup_conv(tr, tr->c_dg_d[i], nm, S, S, n[i], ACT_NONE, 1, 1, nullptr, nullptr, 1, &gd)); tr->cur_bwd = false;
            mac_total += tr->c_dg_d[i].mac;
        }
        T_TRY(setup_wgrad(tr, tr->wg_d[i], nm, S, n[i], n[i], n[i + 1], fwd, {}));
        mac_total += (double)S * S * ks * ks * n[i] * n[i + 1];
        S /= 2;
    }
    {
        GroupSpec g = group(n[L], tr->o_lb, SIZE_MAX, n[L], n[L + 1], 0, 0, fwd);
        T_TRY(setup_conv(tr, tr->c_fwd_b, "lb", S, S, n[L + 1], ACT_NONE, 1, 1, nullptr, nullptr, 1, &g));
        GroupSpec gd = group(n[L + 1], tr->o_lb, SIZE_MAX, n[L], n[L + 1], 1, 0, flp);
        tr->cur_bwd = true; T_TRY(setup_conv(tr, tr->c_dg_b, "lb", S, S, n[L], ACT_NONE, 1, 1, nullptr, nullptr, 1, &gd)); tr->cur_bwd = false;
        T_TRY(setup_wgrad(tr, tr->wg_b, "lb", S, n[L], n[L], n[L + 1], fwd, {}));
        mac_total += tr->c_fwd_b.mac + tr->c_dg_b.mac + (double)S * S * ks * ks * n[L] * n[L + 1];
    }
    const int pb = (ks - 2) / 2;   // pad_before of the stride-2 SAME conv whose gradient the transposed conv is
    for (int idx = L - 1; idx >= 0; --idx) {
        snprintf(nm, sizeof nm, "lu%d", idx);
        const int Cskip = n[idx], Cup = n[idx + 1], Cin = n[idx + 2];
        {   // transposed conv forward: 4 sub-pixel phases (out[2i+a-pb] += in[i] * Wt[a]); LeakyReLU fused
            GroupSpec g;
            g.C = Cin; g.w_off = tr->o_wt[idx]; g.w2_off = SIZE_MAX; g.d2 = Cup; g.d3 = Cin; g.transpose = 1; g.c_off = 0;
            g.npar = 1; g.Cblk = Cin;
            int oy[4], ox[4];
            for (int p = 0; p < 4; ++p) {
                const int pu = p >> 1, pv = p & 1;
                oy[p] = pu; ox[p] = pv;
                for (int a = 0; a < ks; ++a) {
                    if (((a - pb - pu) & 1) != 0) continue;
                    for (int b = 0; b < ks; ++b) {
                        if (((b - pb - pv) & 1) != 0) continue;
                        g.taps[p].off.push_back({(pu + pb - a) / 2, (pv + pb - b) / 2});
                        g.taps[p].m.push_back(a * ks + b);
                    }
                }
            }
            T_TRY(setup_conv(tr, tr->c_T[idx], nm, S, S, Cup, ACT_LEAKY, 4, 2, oy, ox, 1, &g));
            mac_total += (double)S * S * ks * ks * Cin * Cup;
        }
        {   // transposed conv backward on the space-to-depth gradient gS[.., (pa,pb)*Cup + c] = dY[2i+pa, 2j+pb, c]:
            // dX[i] = sum_a dY[2i + a - pb] Wt[a]  ->  q = a - pb = 2*di + pa
            auto split = [](int q, int* d, int* par) { *par = ((q % 2) + 2) % 2; *d = (q - *par) / 2; };
            std::vector<std::pair<int, int>> offs;
            for (int a = 0; a < ks; ++a)
                for (int b = 0; b < ks; ++b) {
                    int di, pa, dj, pbb;
                    split(a - pb, &di, &pa);
                    split(b - pb, &dj, &pbb);
                    std::pair<int, int> od{di, dj};
                    if (std::find(offs.begin(), offs.end(), od) == offs.end()) offs.push_back(od);
                }
            GroupSpec g;
            g.C = 4 * Cup; g.w_off = tr->o_wt[idx]; g.w2_off = SIZE_MAX; g.d2 = Cup; g.d3 = Cin; g.transpose = 0; g.c_off = 0;
            g.npar = 4; g.Cblk = Cup;
            g.taps[0].off = offs;
            g.taps[0].m.assign(offs.size() * 4, -1);
            TapSet slabs;                 // weight-gradient slabs, grouped by parity block
            std::vector<int> coff;
            for (int par = 0; par < 4; ++par)
                for (int a = 0; a < ks; ++a)
                    for (int b = 0; b < ks; ++b) {
                        int di, pa, dj, pbb;
                        split(a - pb, &di, &pa);
                        split(b - pb, &dj, &pbb);
                        if (pa * 2 + pbb != par) continue;
                        const size_t t = std::find(offs.begin(), offs.end(), std::make_pair(di, dj)) - offs.begin();
                        g.taps[0].m[t * 4 + par] = a * ks + b;
                        slabs.off.push_back({di, dj});
                        slabs.m.push_back(a * ks + b);
                        coff.push_back(par * Cup);
                    }
            tr->cur_bwd = true; T_TRY(setup_conv(tr, tr->c_dg_T[idx], nm, S, S, Cin, ACT_NONE, 1, 1, nullptr, nullptr, 1, &g)); tr->cur_bwd = false;
            T_TRY(setup_wgrad(tr, tr->wg_T[idx], nm, S, 4 * Cup, Cup, Cin, slabs, coff));
            mac_total += 2.0 * S * S * ks * ks * Cin * Cup;
        }
        S *= 2;
        {
            GroupSpec g2[2] = {group(Cskip, tr->o_w2[idx], SIZE_MAX, Cskip + Cup, Cup, 0, 0, fwd),
                               group(Cup, tr->o_w2[idx], SIZE_MAX, Cskip + Cup, Cup, 0, Cskip, fwd)};
            T_TRY(setup_conv(tr, tr->c_fwd_u[idx], nm, S, S, Cup, ACT_NONE, 1, 1, nullptr, nullptr, 2, g2));
            mac_total += tr->c_fwd_u[idx].mac;
            GroupSpec gu = group(Cup, tr->o_w2[idx], SIZE_MAX, Cskip + Cup, Cup, 1, Cskip, flp);
            tr->cur_bwd = true; T_TRY(setup_conv(tr, tr->c_dg_us[idx], nm, S, S, Cup, ACT_NONE, 1, 1, nullptr, nullptr, 1, &gu)); tr->cur_bwd = false;
            mac_total += tr->c_dg_us[idx].mac;
            if (idx >= 1) {
                GroupSpec gk = group(Cup, tr->o_w2[idx], SIZE_MAX, Cskip + Cup, Cup, 1, 0, flp);
                tr->cur_bwd = true; T_TRY(setup_conv(tr, tr->c_dg_skip[idx], nm, S, S, Cskip, ACT_NONE, 1, 1, nullptr, nullptr, 1, &gk)); tr->cur_bwd = false;
                mac_total += tr->c_dg_skip[idx].mac;
            }
            T_TRY(setup_wgrad(tr, tr->wg_u0[idx], nm, S, Cskip, Cskip, Cup, fwd, {}));
            T_TRY(setup_wgrad(tr, tr->wg_u1[idx], nm, S, Cup, Cup, Cup, fwd, {}));
            mac_total += (double)S * S * ks * ks * (Cskip + Cup) * Cup;
        }
    }
    mac_total += 3.0 * P * P * n[1] * K;
    tr->flops_per_image = 2.0 * mac_total;
    T_TRY(talloc(tr, &tr->d_ws, tr->ws_floats));
    T_TRY(talloc(tr, &tr->d_ws2, tr->ws_floats));
    T_TRY(talloc(tr, &tr->d_split, tr->split_floats));
    T_TRY(talloc(tr, &tr->d_split2, tr->split_floats));
    T_TRY(talloc(tr, &tr->d_split3, tr->split_floats));
    {
        std::vector<RegSeg> rs;
        for (const Seg& sg : tr->segs)
            if (sg.reg > 0.f) rs.push_back(RegSeg{tr->d_w + sg.off, sg.n, sg.reg});
        tr->n_regsegs = (int)rs.size();
        T_TRY(talloc(tr, &tr->d_regsegs, rs.size()));
        if (!rs.empty()) T_HIP(tr, hipMemcpy(tr->d_regsegs, rs.data(), rs.size() * sizeof(RegSeg), hipMemcpyHostToDevice));
    }
    // forward operands first: the backward-only ones are packed on the side stream while the forward pass runs
    std::stable_partition(tr->packs.begin(), tr->packs.end(), [](const PackDesc& d) { return d.bwd == 0; });
    std::stable_partition(tr->rdescs.begin(), tr->rdescs.end(), [](const RepackDesc& d) { return d.bwd == 0; });
    tr->n_fwd_packs = (int)std::count_if(tr->packs.begin(), tr->packs.end(), [](const PackDesc& d) { return d.bwd == 0; });
    tr->n_fwd_rdescs = (int)std::count_if(tr->rdescs.begin(), tr->rdescs.end(), [](const RepackDesc& d) { return d.bwd == 0; });
    T_TRY(talloc(tr, &tr->d_packs, tr->packs.size()));
    T_HIP(tr, hipMemcpy(tr->d_packs, tr->packs.data(), tr->packs.size() * sizeof(PackDesc), hipMemcpyHostToDevice));
    if (!tr->rdescs.empty()) {
        T_TRY(talloc(tr, &tr->d_rdescs, tr->rdescs.size()));
        T_HIP(tr, hipMemcpy(tr->d_rdescs, tr->rdescs.data(), tr->rdescs.size() * sizeof(RepackDesc), hipMemcpyHostToDevice));
    }
    tr->h_blob = nullptr;
    return UMX_OK;
}

// the step's weights -> the operands of every convolution: fp32 [tap][Cp][Np] (both routes), then conv_f16x3's weight images
// (part 0: the forward pass's operands, 1: the backward-only ones, 2: all)
// Every `wscale_every` steps the per-layer weight scales are re-derived from the parameters as they are now (one copy of the vector
// to the host: 124 MB for the synthetic-256 graph, < 1 % of 256 steps), so that a long run cannot grow a filter out of the
// 32 x headroom the scale of step 0 left it.  Called with nothing in flight that reads the scales.
int refresh_wscales(umx_trainer* tr) {
    if (tr->hconvs.empty()) return UMX_OK;
    T_HIP(tr, hipStreamSynchronize(tr->stream));
    tr->h_params.resize(tr->nparams);
    T_HIP(tr, hipMemcpy(tr->h_params.data(), tr->d_w, tr->nparams * sizeof(float), hipMemcpyDeviceToHost));
    bool changed = false;
    for (size_t k = 0; k < tr->hconvs.size(); ++k) {
        TConv& tc = *tr->hconvs[k];
        const int sh = wscale_shift(operand_wmax(tc, tr->h_params.data()));
        if (sh == tc.wsh) continue;
        tc.wsh = sh;
        const float inv = std::ldexp(1.f, -sh);
        T_HIP(tr, hipMemcpy(tc.winv, &inv, sizeof inv, hipMemcpyHostToDevice));
        for (RepackDesc& rd : tr->rdescs)
            if (rd.owner == (int)k) rd.scale = std::ldexp(1.f, sh);
        changed = true;
    }
    if (changed)
        T_HIP(tr, hipMemcpy(tr->d_rdescs, tr->rdescs.data(), tr->rdescs.size() * sizeof(RepackDesc), hipMemcpyHostToDevice));
    return UMX_OK;
}

int pack_all(umx_trainer* tr, hipStream_t st, int part) {
    const int p0 = part == 1 ? tr->n_fwd_packs : 0, p1 = part == 0 ? tr->n_fwd_packs : (int)tr->packs.size();
    const int r0 = part == 1 ? tr->n_fwd_rdescs : 0, r1 = part == 0 ? tr->n_fwd_rdescs : (int)tr->rdescs.size();
    if (p1 > p0) T_HIP(tr, launch_pack_weights(tr->d_packs + p0, p1 - p0, tr->max_pack, st));
    if (r1 > r0)
        T_HIP(tr, launch_repack_f16x3(tr->d_rdescs + r0, r1 - r0, tr->max_refs, reinterpret_cast<int*>(tr->d_maxw + tr->n_maxw), st));
    return UMX_OK;
}

}  // namespace

extern "C" {

void umx_train_options_solo(umx_train_options* o) {
    fill_common(o);
    o->lr0 = 5e-5f; o->decay_steps = 5000; o->decay_rate = 0.98f;          // UnMicst1-5.py:362-365
    o->reg_kind = UMX_REG_L1;                                               // :84,125,160,213
    o->reg_down = o->reg_bottom = o->reg_up = o->reg_top = 0.00008f;
    o->clip_eps = 1e-7f;                                                    // :369-370
    o->drop_bottom = 0.35f;                                                 // :139
}

void umx_train_options_duo(umx_train_options* o) {
    fill_common(o);
    o->lr0 = 0.00006f; o->decay_steps = 4000; o->decay_rate = 0.99f;       // UnMicst2.py:357-360
    o->reg_kind = UMX_REG_L2;                                               // :82,123,158,211
    o->reg_down = o->reg_bottom = 0.01f;
    o->reg_up = o->reg_top = 0.005f;
    o->clip_eps = 0.f;                                                      // :364-366 (no clip)
    o->drop_down_step = 0.05f; o->drop_bottom = 0.3f;                       // :114,137
    o->drop_up0 = 0.25f; o->drop_up_step = 0.05f;                           // :203
}

const char* umx_trainer_last_error(const umx_trainer* tr) { return tr ? tr->err.c_str() : g_terr.c_str(); }

int umx_trainer_create(const umx_hparams* hp, const float* weight_blob, size_t blob_floats, const umx_train_options* opts,
                       umx_trainer** out) {
    if (!hp || !weight_blob || !opts || !out) return tfail(nullptr, UMX_ERR_INVALID, "null argument");
    *out = nullptr;
    if (hp->graph != UMX_GRAPH_V2 || hp->nExtraConvs != 0)
        return tfail(nullptr, UMX_ERR_INVALID, "the training step covers the v2 graph with nExtraConvs == 0");
    if (hp->nClasses < 1 || hp->nClasses > 8) return tfail(nullptr, UMX_ERR_INVALID, "nClasses must be 1..8");
    if (hp->nLayers < 1 || hp->nLayers > 8 || (hp->ks != 3 && hp->ks != 5) || hp->featMapsFact < 1 ||
        hp->nChannels < 1 || hp->nOut0 < 1)
        return tfail(nullptr, UMX_ERR_INVALID, "unsupported hyper-parameters");
    if ((hp->imSize & (hp->imSize - 1)) || (hp->imSize >> hp->nLayers) < 1)
        return tfail(nullptr, UMX_ERR_INVALID, "imSize must be a power of two, at least 2^nLayers");
    for (int i = 0; i < 8; ++i)
        if (opts->reserved[i]) return tfail(nullptr, UMX_ERR_INVALID, "umx_train_options.reserved must be zero");
    auto bad_rate = [](float r) { return !(r < 1.0f); };
    if (bad_rate(opts->drop_bottom) || bad_rate(opts->drop_up0) || bad_rate(opts->drop_down_step * hp->nLayers))
        return tfail(nullptr, UMX_ERR_INVALID, "dropout rates must be below 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return tfail(nullptr, UMX_ERR_NO_DEVICE, "no HIP device");
    if (opts->device_ordinal < 0 || opts->device_ordinal >= ndev)
        return tfail(nullptr, UMX_ERR_INVALID, "device ordinal %d out of range", opts->device_ordinal);
    umx_trainer* tr = new umx_trainer();
    tr->hp = *hp;
    tr->o = *opts;
    tr->device = opts->device_ordinal;
    tr->B = opts->batch > 0 ? opts->batch : 8;
    int rc = UMX_OK;
    if (hipSetDevice(tr->device) != hipSuccess) rc = tfail(tr, UMX_ERR_HIP, "hipSetDevice failed");
    if (rc == UMX_OK && hipStreamCreateWithFlags(&tr->stream, hipStreamNonBlocking) != hipSuccess)
        rc = tfail(tr, UMX_ERR_HIP, "hipStreamCreate failed");
    if (rc == UMX_OK) rc = build_trainer(tr, weight_blob, blob_floats);
    if (rc == UMX_OK)
        for (int i = 0; i < 4; ++i)
            if (hipEventCreate(&tr->ev[i]) != hipSuccess) rc = tfail(tr, UMX_ERR_HIP, "hipEventCreate failed");
    if (rc == UMX_OK) {
        tr->overlap = true;
        if (hipStreamCreateWithFlags(&tr->side, hipStreamNonBlocking) != hipSuccess) rc = tfail(tr, UMX_ERR_HIP, "hipStreamCreate failed");
        if (hipStreamCreateWithFlags(&tr->side2, hipStreamNonBlocking) != hipSuccess)
            rc = tfail(tr, UMX_ERR_HIP, "hipStreamCreate failed");
        if (hipStreamCreateWithFlags(&tr->aux, hipStreamNonBlocking) != hipSuccess)
            rc = tfail(tr, UMX_ERR_HIP, "hipStreamCreate failed");
        tr->ev_skip.assign(32, nullptr);
        std::vector<hipEvent_t*> evs = {&tr->ev_join, &tr->ev_begin, &tr->ev_packed, &tr->ev_join2};
        for (hipEvent_t& e : tr->ev_skip) evs.push_back(&e);
        for (int sl = 0; sl < kSlots; ++sl) evs.push_back(&tr->ev_aux[sl]);
        for (int sl = 0; sl < kSlots; ++sl) { evs.push_back(&tr->ev_dz[sl]); evs.push_back(&tr->ev_gs[sl]); evs.push_back(&tr->ev_side[sl]); }
        for (hipEvent_t* e : evs)
            if (rc == UMX_OK && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess)
                rc = tfail(tr, UMX_ERR_HIP, "hipEventCreate failed");
    }
    if (rc != UMX_OK) {
        g_terr = tr->err;
        umx_trainer_destroy(tr);
        return rc;
    }
    *out = tr;
    return UMX_OK;
}

void umx_trainer_destroy(umx_trainer* tr) {
    if (!tr) return;
    (void)hipSetDevice(tr->device);
    if (tr->stream) (void)hipStreamSynchronize(tr->stream);
    for (void* p : tr->allocs) (void)hipFree(p);
    if (tr->pctx) {
        for (void* p : tr->pctx->allocs) (void)hipFree(p);
        delete tr->pctx;
    }
    for (int i = 0; i < 4; ++i)
        if (tr->ev[i]) (void)hipEventDestroy(tr->ev[i]);
    if (tr->side) { (void)hipStreamSynchronize(tr->side); (void)hipStreamDestroy(tr->side); }
    if (tr->side2) { (void)hipStreamSynchronize(tr->side2); (void)hipStreamDestroy(tr->side2); }
    if (tr->aux) { (void)hipStreamSynchronize(tr->aux); (void)hipStreamDestroy(tr->aux); }
    for (hipEvent_t e : {tr->ev_join, tr->ev_begin, tr->ev_packed, tr->ev_join2})
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : tr->ev_skip)
        if (e) (void)hipEventDestroy(e);
    for (int sl = 0; sl < kSlots; ++sl)
        if (tr->ev_aux[sl]) (void)hipEventDestroy(tr->ev_aux[sl]);
    for (int sl = 0; sl < kSlots; ++sl)
        for (hipEvent_t e : {tr->ev_dz[sl], tr->ev_gs[sl], tr->ev_side[sl]})
            if (e) (void)hipEventDestroy(e);
    if (tr->stream) (void)hipStreamDestroy(tr->stream);
    delete tr;
}

int umx_train_step_dev(umx_trainer* tr, const float* data_dev, const float* labels_dev, const float* weights_dev,
                       int apply_update) {
    if (!tr || !data_dev || !labels_dev || !weights_dev) return tfail(tr, UMX_ERR_INVALID, "null argument");
    T_HIP(tr, hipSetDevice(tr->device));
    T_TRY(fold_profile(tr));
    float* own = tr->ds[0];
    const int rc = enqueue_step(tr, data_dev, labels_dev, weights_dev, apply_update != 0);
    tr->ds[0] = own;
    return rc;
}

int umx_trainer_loss(umx_trainer* tr, double* loss3) {
    if (!tr || !loss3) return tfail(tr, UMX_ERR_INVALID, "null argument");
    T_HIP(tr, hipSetDevice(tr->device));
    double h[2] = {0, 0};
    unsigned flag = 0;
    T_HIP(tr, hipMemcpyAsync(h, tr->d_loss, sizeof h, hipMemcpyDeviceToHost, tr->stream));
    T_HIP(tr, hipMemcpyAsync(&flag, tr->d_maxw + tr->n_maxw, sizeof flag, hipMemcpyDeviceToHost, tr->stream));
    T_HIP(tr, hipStreamSynchronize(tr->stream));
    if (flag || tr->range_pending) {   // sticky: raised by any step since the last report; reported once, then cleared
        flag = 1;
        tr->range_pending = false;
        T_HIP(tr, hipMemsetAsync(tr->d_maxw + tr->n_maxw, 0, sizeof(unsigned), tr->stream));   // (in the order of the work that can raise the flag)
    }
    if (flag)
        return tfail(tr, UMX_ERR_RANGE, "an operand of a split-precision convolution or weight gradient left the binary16 range "
                                        "(|v| >= 6e4 after scaling or not finite): a diverging run, or a filter that outgrew its scale "
                                        "between two refreshes (UMX_TRAIN_WSCALE_EVERY); UMX_TRAIN_CONV_F32=1 / UMX_TRAIN_WGRAD_F32=1 "
                                        "select the exact-fp32 kernels");
    loss3[0] = h[0] + h[1];
    loss3[1] = h[0];
    loss3[2] = h[1];
    return UMX_OK;
}

int umx_train_step(umx_trainer* tr, const float* data, const float* labels, const float* weights, int apply_update,
                   double* loss3) {
    if (!tr || !data || !labels || !weights) return tfail(tr, UMX_ERR_INVALID, "null argument");
    T_HIP(tr, hipSetDevice(tr->device));
    const size_t npx = (size_t)tr->B * tr->P * tr->P;
    T_HIP(tr, hipMemcpyAsync(tr->ds[0], data, npx * tr->n[0] * sizeof(float), hipMemcpyHostToDevice, tr->stream));
    T_HIP(tr, hipMemcpyAsync(tr->d_labels, labels, npx * tr->K * sizeof(float), hipMemcpyHostToDevice, tr->stream));
    T_HIP(tr, hipMemcpyAsync(tr->d_weights, weights, npx * tr->K * sizeof(float), hipMemcpyHostToDevice, tr->stream));
    T_TRY(umx_train_step_dev(tr, tr->ds[0], tr->d_labels, tr->d_weights, apply_update));
    double l[3];
    T_TRY(umx_trainer_loss(tr, l));
    if (loss3) { loss3[0] = l[0]; loss3[1] = l[1]; loss3[2] = l[2]; }
    return UMX_OK;
}

int umx_trainer_eval(umx_trainer* tr, const float* data, float* probs_host) {
    if (!tr || !data || !probs_host) return tfail(tr, UMX_ERR_INVALID, "null argument");
    T_HIP(tr, hipSetDevice(tr->device));
    T_TRY(fold_profile(tr));
    const size_t npx = (size_t)tr->B * tr->P * tr->P;
    float* own = tr->ds[0];
    // the range flag is shared with the training steps: what is up now belongs to a step whose loss was not read yet -- remember it
    // for umx_trainer_loss, so that the flag this pass raises (repack + split-precision forward) is this pass's own
    unsigned before = 0, flag = 0;
    T_HIP(tr, hipMemcpyAsync(&before, tr->d_maxw + tr->n_maxw, sizeof before, hipMemcpyDeviceToHost, tr->stream));
    T_HIP(tr, hipStreamSynchronize(tr->stream));
    if (before) {
        tr->range_pending = true;
        T_HIP(tr, hipMemsetAsync(tr->d_maxw + tr->n_maxw, 0, sizeof(unsigned), tr->stream));   // (in the order of the work that can raise the flag)
    }
    T_HIP(tr, hipMemcpyAsync(own, data, npx * tr->n[0] * sizeof(float), hipMemcpyHostToDevice, tr->stream));
    T_TRY(pack_all(tr, tr->stream, 0));
    const int rc = forward_pass(tr, own, false, false);
    tr->ds[0] = own;
    T_TRY(rc);
    T_HIP(tr, launch_softmax_only(tr->bn_t.z, tr->bn_t.stat, npx, tr->K, tr->d_probs, tr->stream));
    T_HIP(tr, hipMemcpyAsync(probs_host, tr->d_probs, npx * tr->K * sizeof(float), hipMemcpyDeviceToHost, tr->stream));
    T_HIP(tr, hipMemcpyAsync(&flag, tr->d_maxw + tr->n_maxw, sizeof flag, hipMemcpyDeviceToHost, tr->stream));
    T_HIP(tr, hipStreamSynchronize(tr->stream));
    if (flag) {
        T_HIP(tr, hipMemsetAsync(tr->d_maxw + tr->n_maxw, 0, sizeof(unsigned), tr->stream));   // (in the order of the work that can raise the flag)
        return tfail(tr, UMX_ERR_RANGE, "an operand of the split-precision forward pass of umx_trainer_eval left the binary16 range (|v| >= 6e4 or "
                                        "not finite): the probabilities are not valid; UMX_TRAIN_CONV_F32=1 selects the exact-fp32 kernels");
    }
    return UMX_OK;
}

int umx_trainer_read(umx_trainer* tr, int which, float* out, size_t n_floats) {
    if (!tr || !out) return tfail(tr, UMX_ERR_INVALID, "null argument");
    if (n_floats != tr->nparams) return tfail(tr, UMX_ERR_INVALID, "vector has %zu floats, asked for %zu", tr->nparams, n_floats);
    const float* src = which == UMX_TV_PARAMS ? tr->d_w : which == UMX_TV_GRADS ? tr->d_g : which == UMX_TV_SLOT_M ? tr->d_m
                     : which == UMX_TV_SLOT_V ? tr->d_v : nullptr;
    if (!src) return tfail(tr, UMX_ERR_INVALID, "unknown vector %d", which);
    T_HIP(tr, hipSetDevice(tr->device));
    T_HIP(tr, hipStreamSynchronize(tr->stream));
    T_HIP(tr, hipMemcpy(out, src, n_floats * sizeof(float), hipMemcpyDeviceToHost));
    return UMX_OK;
}

int umx_trainer_read_tensor(umx_trainer* tr, const char* name, float* out, size_t* n_floats) {
    if (!tr || !name || !n_floats) return tfail(tr, UMX_ERR_INVALID, "null argument");
    const std::string nm(name);
    const size_t dot = nm.find('.');
    const std::string layer = nm.substr(0, dot), what = dot == std::string::npos ? "" : nm.substr(dot + 1);
    const float* src = nullptr;
    size_t n = 0;
    auto index_of = [&](const char* prefix, int count) -> int {   // "<prefix><i>" -> i, or -1
        const size_t pl = strlen(prefix);
        if (layer.compare(0, pl, prefix) != 0 || layer.size() == pl) return -1;
        for (size_t k = pl; k < layer.size(); ++k) if (layer[k] < '0' || layer[k] > '9') return -1;
        const int i = atoi(layer.c_str() + pl);
        return i < count ? i : -1;
    };
    const BnSite* site = nullptr;
    int i;
    if (layer == "lb") site = &tr->bn_b;
    else if (layer == "lt") site = &tr->bn_t;
    else if ((i = index_of("ld", tr->L)) >= 0) site = &tr->bn_d[i];
    else if ((i = index_of("lu", tr->L)) >= 0) site = &tr->bn_u[i];
    if (site && what == "z") { src = site->z; n = (size_t)tr->B * site->H * site->W * site->C; }
    else if (site && what == "stat") { src = site->stat; n = 4 * (size_t)site->C; }
    else if (what == "us" && (i = index_of("lu", tr->L)) >= 0) {
        const int S = tr->P >> i;
        src = tr->us[i]; n = (size_t)tr->B * S * S * tr->n[i + 1];
    } else if (what.empty() && (i = index_of("ds", tr->L + 1)) >= 0) {
        const int S = tr->P >> i;
        src = tr->ds[i]; n = (size_t)tr->B * S * S * tr->n[i];
    }
    if (!src) return tfail(tr, UMX_ERR_INVALID, "unknown tensor %s", name);
    const size_t cap = *n_floats;
    *n_floats = n;
    if (!out) return UMX_OK;
    if (cap < n) return tfail(tr, UMX_ERR_INVALID, "%s has %zu floats, the buffer holds %zu", name, n, cap);
    T_HIP(tr, hipSetDevice(tr->device));
    T_HIP(tr, hipStreamSynchronize(tr->stream));
    T_HIP(tr, hipMemcpy(out, src, n * sizeof(float), hipMemcpyDeviceToHost));
    return UMX_OK;
}

int umx_trainer_probs(umx_trainer* tr, float* probs_host) {
    if (!tr || !probs_host) return tfail(tr, UMX_ERR_INVALID, "null argument");
    T_HIP(tr, hipSetDevice(tr->device));
    T_HIP(tr, hipStreamSynchronize(tr->stream));
    T_HIP(tr, hipMemcpy(probs_host, tr->d_probs, (size_t)tr->B * tr->P * tr->P * tr->K * sizeof(float), hipMemcpyDeviceToHost));
    return UMX_OK;
}

int64_t umx_trainer_step_count(const umx_trainer* tr) { return tr ? tr->step : -1; }
int umx_trainer_batch(const umx_trainer* tr) { return tr ? tr->B : 0; }
double umx_trainer_flops_per_image(const umx_trainer* tr) { return tr ? tr->flops_per_image : 0.0; }

int umx_trainer_profile(umx_trainer* tr, int enable, double* fwd_ms, double* bwd_ms, double* opt_ms, int* steps) {
    if (!tr) return tfail(tr, UMX_ERR_INVALID, "null argument");
    T_HIP(tr, hipSetDevice(tr->device));
    T_TRY(fold_profile(tr));
    if (fwd_ms) *fwd_ms = tr->t_fwd;
    if (bwd_ms) *bwd_ms = tr->t_bwd;
    if (opt_ms) *opt_ms = tr->t_opt;
    if (steps) *steps = tr->t_steps;
    tr->t_fwd = tr->t_bwd = tr->t_opt = 0.0;
    tr->t_steps = 0;
    tr->prof = enable != 0;
    return UMX_OK;
}

}  // extern "C"
