// Host side of libumx: graph construction from the reference's hyper-parameters, weight folding / packing,
// device memory plan, launch sequencing and the C ABI declared in include/umx.h.  gfx950 (MI355X) only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/umx.h"
#include "umx_kernels.h"

using namespace umx;

namespace {

thread_local std::string g_err;

struct HostTensor {
    const float* p;
    int d0, d1, d2, d3;  // [kh,kw,a,b]
    float at(int i, int j, int a, int b) const { return p[(((size_t)i * d1 + j) * d2 + a) * d3 + b]; }
};

struct BN {
    const float *g, *b, *m, *v;
};

struct Group {           // one operand group of a launch, host description
    int src;             // buffer id
    int C;               // channels
    std::vector<std::pair<int, int>> taps[4];  // per phase: (dy, dx) input offsets
    std::vector<float> packed[4];              // per phase: [ntaps][Cp][Np]
};

struct Launch {
    std::string name;
    bool head = false;
    int ngroups = 0;
    Group g[2];
    int nphase = 1, o_mul = 1;
    int oy_off[4] = {0, 0, 0, 0}, ox_off[4] = {0, 0, 0, 0};
    int H = 0, W = 0, Cout = 0;
    int dst = -1, outH = 0, outW = 0, pool = 0, act = 0;
    std::vector<float> pre_s, pre_b, post_s, post_b;  // size Cout or empty
    // head only
    std::vector<float> head_w;  // [C][K]
    int head_C = 0, head_K = 0;
    // derived
    // split-precision plan only: after the epilogue, copy `app_C` (<= 2) channels of buffer `app_src` (same pixel grid) into
    // the spare channels [app_c0, app_c0 + app_C) of this launch's last stored octet -- the raw-input skip of the top
    // up-layer rides in the up-sampled tensor, so its convolution reads one 5-octet tensor instead of 1 + 5 octets
    int app_src = -1, app_C = 0, app_c0 = 0;
    int nt = 1, Np = 16, hpix = 2;
    double flops = 0.0;       // algorithmic FLOPs per tile (per image of the batch)
    double exec_flops = 0.0;  // executed incl. channel/N padding
    double bytes = 0.0;       // compulsory HBM bytes per tile: sources + destination (weights excluded)
    // device
    ConvParams cp;
    // split-precision plan (UMX_PREC_F16X3)
    HConvParams hcp;
    RwParams rw;              // register-resident-weight plan (use_rw): the narrow full-resolution layers
    bool use_rw = false;
    int nt16 = 1;             // N-tiles per workgroup of the split-precision kernel
    int wshift = 0;           // weights are stored times 2^wshift
    int n_ksteps = 0;         // K-slots of 32 executed per output tile, all phases (for the executed-FLOP figure)
    float* d_head_w = nullptr;
    float *d_pre_s = nullptr, *d_pre_b = nullptr, *d_post_s = nullptr, *d_post_b = nullptr;
};

struct Buffer {
    size_t floats_per_tile = 0;
    int S = 0, C = 0;        // spatial size and real channels of the tensor
    int Cs = 0;              // stored channels of the (hi, lo) binary16 form
    bool as_f32 = true;      // fp32 NHWC (f32 path, and the head's input in the f16 path) or (hi, lo) binary16 planes
    bool planar = false;     // (hi, lo) planes stored per image as [octet][pixel][8] instead of NHWC (tensors of >= 16 x 16 pixels)
    float* d = nullptr;
};

struct ProfSite {
    std::string name, kernel;
    int64_t launches = 0;
    double total_ms = 0.0, flops = 0.0, bytes = 0.0, exec = 0.0;
};

struct PendingEvent {
    int site;
    hipEvent_t a, b;
};

}  // namespace

struct umx_ctx {
    umx_hparams hp;
    int device = 0;
    int max_batch = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::vector<Launch> plan;
    std::vector<Buffer> bufs;   // bufs[0] = input tiles
    // Second "lane": the tile batches of one band alternate between two activation-buffer sets on two streams, so that
    // the kernels of batch i+1 fill the CUs the tail of batch i's current layer leaves idle and MFMA-bound layers of one
    // batch share a CU with the load-bound full-resolution layers of the other.
    std::vector<Buffer> bufs2;
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    float* d_tiles32_2 = nullptr;
    int nlanes = 1, lane = 0;
    int ncu = 256;
    // host entry points: uploads / downloads on their own streams, slab by slab, under the tile kernels
    hipStream_t up_stream = nullptr, dn_stream = nullptr;
    struct HostSlot {   // device buffers + events of one in-flight host call (two slots: slide i+1 uploads while slide i computes)
        double* d_image = nullptr;  size_t image_cap = 0;
        float* d_probs = nullptr;   size_t probs_cap = 0;
        void* d_out = nullptr;      size_t out_cap = 0;
        std::vector<hipEvent_t> events;
        hipEvent_t done = nullptr;
        int* flag_host = nullptr;   // pinned copy of the range flag, read back behind the slot's last download
        bool busy = false;
    } hs[2];
    std::vector<void*> allocs;
    std::string err;
    // whole-image scratch (grown on demand)
    double* d_image = nullptr;  size_t image_cap = 0;
    float* d_probs = nullptr;   size_t probs_cap = 0;
    void* d_out = nullptr;      size_t out_cap = 0;
    float* d_io_tiles = nullptr; size_t io_tiles_cap = 0;
    float* d_io_probs = nullptr; size_t io_probs_cap = 0;
    // profiling
    bool prof = false;
    std::vector<ProfSite> sites;
    std::vector<PendingEvent> pending;
    std::vector<hipEvent_t> free_events;
    int site_gather = -1, site_stitch = -1, site_split = -1;
    // precision
    int precision = UMX_PREC_F16X3;
    int act_shift = 0;          // activations are stored times 2^act_shift in the (hi, lo) binary16 form
    float* d_tiles32 = nullptr; // fp32 staging of gathered tiles before the split (f16 path)
    int* d_flag = nullptr;      // binary16 range overflow flag
    uint4* d_zeros = nullptr;
    bool head_fused = false;
    Launch split_launch;
};

namespace {

inline std::vector<Buffer>& cur_bufs(umx_ctx* ctx) { return ctx->lane ? ctx->bufs2 : ctx->bufs; }
inline hipStream_t run_stream(umx_ctx* ctx) { return ctx->lane ? ctx->stream2 : ctx->stream; }

int fail(umx_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    g_err = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                                  \
    do {                                                                                                    \
        hipError_t e__ = (expr);                                                                            \
        if (e__ != hipSuccess)                                                                              \
            return fail(ctx, e__ == hipErrorOutOfMemory ? UMX_ERR_OOM : UMX_ERR_HIP, "%s failed: %s", #expr, \
                        hipGetErrorString(e__));                                                            \
    } while (0)

int check_hp(const umx_hparams* hp, std::string* why) {
    if (!hp) { *why = "hp is NULL"; return UMX_ERR_INVALID; }
    if (hp->graph != UMX_GRAPH_LEGACY && hp->graph != UMX_GRAPH_V2) { *why = "unknown graph kind"; return UMX_ERR_INVALID; }
    if (hp->nLayers < 1 || hp->nLayers > 8) { *why = "nLayers must be in [1,8]"; return UMX_ERR_INVALID; }
    if (hp->ks < 1 || hp->ks > 7 || !(hp->ks & 1)) { *why = "ks must be odd and <= 7"; return UMX_ERR_INVALID; }
    if (hp->nExtraConvs < 0 || hp->nExtraConvs > 4) { *why = "nExtraConvs must be in [0,4]"; return UMX_ERR_INVALID; }
    if (hp->nClasses < 2 || hp->nClasses > 4) { *why = "nClasses must be 2..4"; return UMX_ERR_INVALID; }
    if (hp->nChannels < 1 || hp->nOut0 < 1 || hp->featMapsFact < 1) { *why = "bad channel counts"; return UMX_ERR_INVALID; }
    if (hp->imSize < 8 || (hp->imSize & (hp->imSize - 1))) { *why = "imSize must be a power of two >= 8"; return UMX_ERR_INVALID; }
    if ((hp->imSize >> hp->nLayers) < 1) { *why = "imSize too small for nLayers"; return UMX_ERR_INVALID; }
    return UMX_OK;
}

std::vector<int> widths(const umx_hparams& hp) {
    std::vector<int> n = {hp.nChannels, hp.nOut0};
    for (int i = 0; i < hp.nLayers; ++i) n.push_back(n.back() * hp.featMapsFact);
    return n;
}

size_t blob_floats_needed(const umx_hparams& hp) {
    const auto n = widths(hp);
    const int ks = hp.ks, L = hp.nLayers, nx = hp.nExtraConvs;
    const bool v2 = hp.graph == UMX_GRAPH_V2;
    const int kss = v2 ? ks : 1;
    size_t t = 0;
    for (int i = 0; i < L; ++i) {
        t += (size_t)ks * ks * n[i] * n[i + 1] + (size_t)nx * ks * ks * n[i + 1] * n[i + 1] +
             (size_t)kss * kss * n[i] * n[i + 1] + 4 * (size_t)n[i + 1];
    }
    t += (size_t)ks * ks * n[L] * n[L + 1] + (v2 ? 4 * (size_t)n[L + 1] : 0);
    for (int i = L - 1; i >= 0; --i) {
        t += (size_t)ks * ks * n[i + 1] * n[i + 2] + (size_t)ks * ks * (n[i] + n[i + 1]) * n[i + 1] +
             (v2 ? 4 * (size_t)n[i + 1] : 0) + (size_t)nx * ks * ks * n[i + 1] * n[i + 1];
    }
    t += (size_t)n[1] * hp.nClasses + (v2 ? 4 * (size_t)hp.nClasses : 0);
    return t;
}

int round_up(int a, int b) { return (a + b - 1) / b * b; }

// choose N tiles per workgroup: minimise padded N, prefer wide tiles
void choose_nt(int Cout, int* nt, int* Np) {
    const int t16 = (Cout + 15) / 16;
    int best = 1, best_pad = 1 << 30;
    for (int c = 1; c <= kMaxNT; ++c) {
        const int padded = round_up(t16, c);
        if (padded < best_pad || (padded == best_pad && c > best)) { best = c; best_pad = padded; }
    }
    *nt = best;
    *Np = best_pad * 16;
}

void fold_bn(const BN& bn, int C, std::vector<float>* s, std::vector<float>* b) {
    // tf.layers.batch_normalization(training=False): gamma*(x-mean)/sqrt(var+eps)+beta, eps = 1e-3
    s->resize(C);
    b->resize(C);
    for (int c = 0; c < C; ++c) {
        const double sc = (double)bn.g[c] / std::sqrt((double)bn.v[c] + 0.001);
        (*s)[c] = (float)sc;
        (*b)[c] = (float)((double)bn.b[c] - (double)bn.m[c] * sc);
    }
}

struct Builder {
    const umx_hparams& hp;
    const float* blob;   // may be NULL (describe only)
    size_t pos = 0;
    std::vector<Launch> plan;
    std::vector<size_t> buf_floats;  // per tile
    std::vector<std::pair<int, int>> buf_geom;  // (spatial size, channels) per buffer

    bool fold_top_skip = false;   // split-precision plan: see Launch::app_src
    explicit Builder(const umx_hparams& h, const float* b) : hp(h), blob(b) {}

    const float* take(size_t n) {
        const float* r = blob ? blob + pos : nullptr;
        pos += n;
        return r;
    }
    HostTensor take_filter(int kh, int kw, int a, int b) { return HostTensor{take((size_t)kh * kw * a * b), kh, kw, a, b}; }
    BN take_bn(int C) { BN r; r.g = take(C); r.b = take(C); r.m = take(C); r.v = take(C); return r; }
    int new_buf(int S, int C) {
        buf_floats.push_back((size_t)S * S * C);
        buf_geom.push_back({S, C});
        return (int)buf_floats.size() - 1;
    }

    // pack filter channels [c0, c0+C) of w [kh,kw,Cin,Cout] for the taps of a stride-1 SAME conv
    // (cmap: input channel c of the group reads filter channel cmap[c] instead of c0 + c)
    void add_conv_group(Launch& L, int src, const HostTensor& w, int c0, int C, const HostTensor* add = nullptr,
                        const std::vector<int>* cmap = nullptr) {
        Group& g = L.g[L.ngroups++];
        g.src = src;
        g.C = C;
        const int ph = (w.d0 - 1) / 2, pw = (w.d1 - 1) / 2;
        for (int a = 0; a < w.d0; ++a)
            for (int b = 0; b < w.d1; ++b) g.taps[0].push_back({a - ph, b - pw});
        if (!blob) return;
        const int Cp = round_up(C, 4);
        g.packed[0].assign((size_t)w.d0 * w.d1 * Cp * L.Np, 0.f);
        for (int a = 0; a < w.d0; ++a)
            for (int b = 0; b < w.d1; ++b)
                for (int c = 0; c < C; ++c)
                    for (int o = 0; o < L.Cout; ++o) {
                        float v = w.at(a, b, cmap ? (*cmap)[c] : c0 + c, o);
                        if (add) {
                            // same-source shortcut folded into the main filter (exact algebra):
                            // ks x ks shortcut -> element-wise sum; 1x1 shortcut -> centre tap
                            if (add->d0 == w.d0) v += add->at(a, b, c0 + c, o);
                            else if (a == ph && b == pw) v += add->at(0, 0, c0 + c, o);
                        }
                        g.packed[0][(((size_t)a * w.d1 + b) * Cp + c) * L.Np + o] = v;
                    }
    }

    // stride-2 SAME transposed conv as 4 sub-pixel phases; wt [kh,kw,Cout,Cin] (TF conv2d_transpose layout)
    void add_convT_group(Launch& L, int src, const HostTensor& wt) {
        Group& g = L.g[L.ngroups++];
        g.src = src;
        g.C = wt.d3;
        const int Cp = round_up(g.C, 4);
        const int pbh = (wt.d0 - 2) / 2, pbw = (wt.d1 - 2) / 2;  // pad_before of the forward stride-2 SAME conv
        L.nphase = 4;
        L.o_mul = 2;
        for (int p = 0; p < 4; ++p) {
            const int pu = p >> 1, pv = p & 1;
            L.oy_off[p] = pu;
            L.ox_off[p] = pv;
            std::vector<std::pair<int, int>> ab;
            for (int a = 0; a < wt.d0; ++a) {
                if (((a - pbh - pu) & 1) != 0) continue;
                for (int b = 0; b < wt.d1; ++b) {
                    if (((b - pbw - pv) & 1) != 0) continue;
                    ab.push_back({a, b});
                    // out[2i'+pu] += in[i] * W[a] with 2i + a - pb = 2i' + pu  ->  i = i' + (pu + pb - a)/2
                    g.taps[p].push_back({(pu + pbh - a) / 2, (pv + pbw - b) / 2});
                }
            }
            if (!blob) continue;
            g.packed[p].assign(ab.size() * (size_t)Cp * L.Np, 0.f);
            for (size_t t = 0; t < ab.size(); ++t)
                for (int c = 0; c < g.C; ++c)
                    for (int o = 0; o < L.Cout; ++o)
                        g.packed[p][(t * Cp + c) * L.Np + o] = wt.at(ab[t].first, ab[t].second, o, c);
        }
    }

    Launch make(const std::string& name, int H, int Cout, int dst, int pool, int act) {
        Launch L;
        L.name = name;
        L.H = L.W = H;
        L.Cout = Cout;
        L.dst = dst;
        L.pool = pool;
        L.act = act;
        choose_nt(Cout, &L.nt, &L.Np);
        return L;
    }

    void finish(Launch& L) {
        L.outH = L.pool ? L.H / 2 : L.H * L.o_mul;
        L.outW = L.pool ? L.W / 2 : L.W * L.o_mul;
        double mac = 0.0, emac = 0.0, src_bytes = 0.0;
        for (int gi = 0; gi < L.ngroups; ++gi) {
            size_t nt = 0;
            for (int p = 0; p < L.nphase; ++p) nt += L.g[gi].taps[p].size();
            mac += (double)L.H * L.W * nt * L.g[gi].C * L.Cout;
            emac += (double)L.H * L.W * nt * round_up(L.g[gi].C, 4) * L.Np;
            src_bytes += 4.0 * L.H * L.W * L.g[gi].C;
        }
        L.flops = 2.0 * mac;
        L.exec_flops = 2.0 * emac;
        L.bytes = src_bytes + 4.0 * L.outH * L.outW * L.Cout;
        plan.push_back(std::move(L));
    }

    int build() {
        const auto n = widths(hp);
        const int L = hp.nLayers, ks = hp.ks, nx = hp.nExtraConvs, P = hp.imSize;
        const bool v2 = hp.graph == UMX_GRAPH_V2;
        const int kss = v2 ? ks : 1;
        const int act = v2 ? ACT_LEAKY : ACT_RELU;
        std::vector<int> ds(L + 1);
        ds[0] = new_buf(P, n[0]);  // buffer 0: normalised input tiles
        int S = P;
        char nm[64];
        for (int i = 0; i < L; ++i) {
            const int Ci = n[i], Co = n[i + 1];
            HostTensor w1 = take_filter(ks, ks, Ci, Co);
            std::vector<HostTensor> wx;
            for (int e = 0; e < nx; ++e) wx.push_back(take_filter(ks, ks, Co, Co));
            HostTensor wsc = take_filter(kss, kss, Ci, Co);
            BN bn = take_bn(Co);
            ds[i + 1] = new_buf(S / 2, Co);
            auto epilogue = [&](Launch& Lh) {
                // v2: leaky(BN(sum)) (UnMicst1-5.py:114);  legacy: BN(relu(sum)) (UnMicst.py:99); then 2x2 max-pool
                if (!blob) return;
                if (v2) fold_bn(bn, Co, &Lh.pre_s, &Lh.pre_b);
                else fold_bn(bn, Co, &Lh.post_s, &Lh.post_b);
            };
            if (nx == 0) {
                snprintf(nm, sizeof nm, "ld%d.conv", i);
                Launch Lh = make(nm, S, Co, ds[i + 1], 1, act);
                add_conv_group(Lh, ds[i], w1, 0, Ci, &wsc);
                epilogue(Lh);
                finish(Lh);
            } else {
                int t = new_buf(S, Co), t2 = nx > 1 ? new_buf(S, Co) : -1;
                snprintf(nm, sizeof nm, "ld%d.conv1", i);
                Launch L1 = make(nm, S, Co, t, 0, act);  // act fused: the next conv consumes act(c00)
                add_conv_group(L1, ds[i], w1, 0, Ci);
                finish(L1);
                for (int e = 0; e < nx; ++e) {
                    const bool last = e == nx - 1;
                    snprintf(nm, sizeof nm, "ld%d.extra%d", i, e);
                    Launch Le = make(nm, S, Co, last ? ds[i + 1] : t2, last ? 1 : 0, act);
                    add_conv_group(Le, t, wx[e], 0, Co);
                    if (last) {
                        add_conv_group(Le, ds[i], wsc, 0, Ci);  // shortcut of the block input as a second K slab
                        epilogue(Le);
                    }
                    finish(Le);
                    std::swap(t, t2);
                }
            }
            S /= 2;
        }
        int cur;
        {
            const int Ci = n[L], Co = n[L + 1];
            HostTensor w = take_filter(ks, ks, Ci, Co);
            cur = new_buf(S, Co);
            Launch Lb = make("lb.conv", S, Co, cur, 0, act);
            add_conv_group(Lb, ds[L], w, 0, Ci);
            if (v2) {
                BN bn = take_bn(Co);
                if (blob) fold_bn(bn, Co, &Lb.pre_s, &Lb.pre_b);
            }
            finish(Lb);
        }
        for (int idx = L - 1; idx >= 0; --idx) {
            const int Cskip = n[idx], Cup = n[idx + 1], Cin = n[idx + 2];
            HostTensor wt = take_filter(ks, ks, Cup, Cin);
            HostTensor w2 = take_filter(ks, ks, Cskip + Cup, Cup);
            BN bn{};
            if (v2) bn = take_bn(Cup);
            std::vector<HostTensor> wx;
            for (int e = 0; e < nx; ++e) wx.push_back(take_filter(ks, ks, Cup, Cup));
            const int S2 = S * 2;
            const int us = new_buf(S2, Cup);
            snprintf(nm, sizeof nm, "lu%d.convT", idx);
            Launch Lt = make(nm, S, Cup, us, 0, act);
            add_convT_group(Lt, cur, wt);
            finish(Lt);
            int cv = new_buf(S2, Cup);
            snprintf(nm, sizeof nm, "lu%d.conv", idx);
            Launch Lc = make(nm, S2, Cup, cv, 0, act);
            const bool fold = fold_top_skip && idx == 0 && Cskip <= 2 && (Cup % 8) != 0 && (Cup % 8) % 2 == 0 &&
                              (Cup % 8) + Cskip <= 8 && S2 >= 16;
            if (fold) {
                // the transposed convolution's epilogue drops the raw input channels into the spare channels of its last
                // octet; this convolution then reads [us | skip] as one tensor (filter channels permuted accordingly)
                Launch& Ltp = plan.back();
                Ltp.app_src = ds[idx]; Ltp.app_C = Cskip; Ltp.app_c0 = Cup;
                std::vector<int> cmap(Cup + Cskip);
                for (int c = 0; c < Cup; ++c) cmap[c] = Cskip + c;
                for (int c = 0; c < Cskip; ++c) cmap[Cup + c] = c;
                add_conv_group(Lc, us, w2, 0, Cup + Cskip, nullptr, &cmap);
            } else {
                add_conv_group(Lc, ds[idx], w2, 0, Cskip);   // concat3([dsX[index], us]): skip channels first
                add_conv_group(Lc, us, w2, Cskip, Cup);
            }
            if (v2 && blob) fold_bn(bn, Cup, &Lc.pre_s, &Lc.pre_b);
            finish(Lc);
            int other = nx > 0 ? new_buf(S2, Cup) : -1;
            for (int e = 0; e < nx; ++e) {
                snprintf(nm, sizeof nm, "lu%d.extra%d", idx, e);
                // the tensor the softmax head reads gets a buffer of its own: it is never an intermediate, so the
                // split-precision path can keep it fp32 while every other tensor is a (hi, lo) binary16 pair
                const int dstb = (idx == 0 && e == nx - 1) ? new_buf(S2, Cup) : other;
                Launch Le = make(nm, S2, Cup, dstb, 0, act);
                add_conv_group(Le, cv, wx[e], 0, Cup);
                finish(Le);
                other = cv;
                cv = dstb;
            }
            cur = cv;
            S = S2;
        }
        {
            Launch Lh;
            Lh.name = "lt.head";
            Lh.head = true;
            Lh.H = Lh.W = S;
            Lh.head_C = n[1];
            Lh.head_K = hp.nClasses;
            Lh.ngroups = 1;
            Lh.g[0].src = cur;
            Lh.g[0].C = n[1];
            const float* w = take((size_t)n[1] * hp.nClasses);
            if (blob) Lh.head_w.assign(w, w + (size_t)n[1] * hp.nClasses);
            if (v2) {
                BN bn = take_bn(hp.nClasses);
                if (blob) fold_bn(bn, hp.nClasses, &Lh.pre_s, &Lh.pre_b);
            }
            Lh.flops = Lh.exec_flops = 2.0 * S * S * n[1] * hp.nClasses;
            Lh.bytes = 4.0 * S * S * (n[1] + hp.nClasses);
            plan.push_back(std::move(Lh));
        }
        return UMX_OK;
    }
};

// conv geometry (tile shape, LDS halo) for one launch; returns false if unsupported
bool conv_geometry(Launch& L, std::string* why) {
    ConvParams& p = L.cp;
    memset(&p, 0, sizeof p);
    int ymin = 0, ymax = 0, xmin = 0, xmax = 0, ntaps_total = 0;
    for (int gi = 0; gi < L.ngroups; ++gi)
        for (int ph = 0; ph < L.nphase; ++ph)
            for (auto& t : L.g[gi].taps[ph]) {
                ymin = std::min(ymin, t.first); ymax = std::max(ymax, t.first);
                xmin = std::min(xmin, t.second); xmax = std::max(xmax, t.second);
                ++ntaps_total;
            }
    if (ntaps_total > kMaxTaps) { *why = "too many filter taps"; return false; }
    auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    const int TWm = std::min(16, L.W), TH = std::min(16, L.H);
    if ((TWm & (TWm - 1)) || (TH & (TH - 1))) { *why = "layer size must be a power of two"; return false; }
    p.twm_log2 = lg2(TWm);
    p.th_log2 = lg2(TH);
    p.nimg_m = 16 / TWm;
    p.imgs = p.nimg_m * (16 / TH);
    p.hh = TH + ymax - ymin;
    p.hw = TWm + xmax - xmin;
    p.imgplane = p.hh * p.hw;
    int plane = p.imgs * p.imgplane;
    plane = round_up(plane, 32) + 16;  // = 16 (mod 32): conflict-free A-fragment reads
    if (plane - 32 >= p.imgs * p.imgplane) plane -= 32;
    p.plane = plane;
    p.ymin = ymin;
    p.xmin = xmin;
    p.tiles_y = L.H / TH;
    p.tiles_x = L.W / TWm;
    L.hpix = (p.imgs * p.imgplane + 255) / 256;
    if (L.hpix > 4) { *why = "halo too large for the staging registers"; return false; }
    L.hpix = L.hpix <= 2 ? 2 : 4;
    if (L.pool && (TH < 2 || TWm < 2)) { *why = "cannot pool a 1-pixel layer"; return false; }
    p.ngroups = L.ngroups;
    p.H = L.H; p.W = L.W; p.Cout = L.Cout; p.Np = L.Np;
    p.nphase = L.nphase; p.o_mul = L.o_mul;
    p.outH = L.outH; p.outW = L.outW; p.pool = L.pool; p.act = L.act;
    int tpos = 0;
    for (int ph = 0; ph < L.nphase; ++ph) {
        p.ph[ph].oy_off = L.oy_off[ph];
        p.ph[ph].ox_off = L.ox_off[ph];
        for (int gi = 0; gi < L.ngroups; ++gi) {
            p.ph[ph].tap0[gi] = tpos;
            p.ph[ph].ntaps[gi] = (int)L.g[gi].taps[ph].size();
            for (auto& t : L.g[gi].taps[ph]) p.tapoff[tpos++] = (short)((t.first - ymin) * p.hw + (t.second - xmin));
        }
    }
    for (int gi = 0; gi < L.ngroups; ++gi) {
        p.C[gi] = L.g[gi].C;
        p.Cp[gi] = round_up(L.g[gi].C, 4);
        p.vec4[gi] = (L.g[gi].C % 4) == 0;
    }
    if (conv_lds_bytes(L.nt, p.plane) > 160 * 1024) { *why = "LDS footprint too large"; return false; }
    return true;
}

int dev_alloc(umx_ctx* ctx, void** out, size_t bytes) {
    void* d = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, bytes ? bytes : 16));
    ctx->allocs.push_back(d);
    *out = d;
    return UMX_OK;
}

int upload(umx_ctx* ctx, const std::vector<float>& h, float** out) {
    *out = nullptr;
    if (h.empty()) return UMX_OK;
    void* d = nullptr;
    int rc = dev_alloc(ctx, &d, h.size() * sizeof(float));
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = (float*)d;
    return UMX_OK;
}


template <typename T>
int upload_raw(umx_ctx* ctx, const std::vector<T>& h, T** out) {
    *out = nullptr;
    if (h.empty()) return UMX_OK;
    void* d = nullptr;
    int rc = dev_alloc(ctx, &d, h.size() * sizeof(T));
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (T*)d;
    return UMX_OK;
}

// ---- register-resident-weight plan (umx_conv_rw.hip) for a launch plan_f16 has just planned: a plain 3x3/5x5 convolution
// at >= 16x16 resolution with a fused softmax head, one N-block, and a packed weight set small enough for the register
// file.  All (tap, octet) pairs of all operand groups form ONE k-step list; the LDS image of a tile holds every octet of
// every group side by side (pixel pitch OCT*16 bytes, OCT odd: conflict-free fragment reads).
int plan_rw(umx_ctx* ctx, Launch& L, float wscale, const Launch* head, std::string* why) {
    (void)why;
    L.use_rw = false;
    const HConvParams& h = L.hcp;
    {   // measured on MI355X: equal to conv_f16x3 on the layers it covers -- opt-in until it wins
        const char* e = getenv("UMX_RW");
        if (!e || atoi(e) == 0) return UMX_OK;
    }
    if (h.fused_phases || L.nphase != 1 || h.nblocks != 1 || h.head_K <= 0) return UMX_OK;
    if (L.H < 16 || L.W < 16 || (L.H & 15) || (L.W & 15) || h.imgs != 1 || h.twm_log2 != 4 || h.th_log2 != 4) return UMX_OK;
    auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    const int tx = L.W / 16, ty = L.H / 16;
    if ((tx & (tx - 1)) || (ty & (ty - 1))) return UMX_OK;
    RwParams& r = L.rw;
    memset(&r, 0, sizeof r);
    int oct_total = 0, npairs = 0;
    for (int gi = 0; gi < L.ngroups; ++gi) {
        r.noct[gi] = round_up(L.g[gi].C, 8) / 8;
        r.goct[gi] = oct_total;
        oct_total += r.noct[gi];
        npairs += (int)L.g[gi].taps[0].size() * r.noct[gi];
    }
    const int nk = (npairs + 3) / 4;
    if (!conv_rw_supported(h.NT, nk)) return UMX_OK;
    r.ngroups = L.ngroups;
    r.H = L.H; r.W = L.W;
    r.hh = h.hh; r.hw = h.hw; r.nhalo = h.nhalo; r.ymin = h.ymin; r.xmin = h.xmin;
    r.inv_hw = 1.f / (float)h.hw;
    r.OCT = oct_total | 1;   // odd pixel pitch (in 16-byte slots): 16 consecutive pixels at one octet hit 16 distinct bank groups
    r.pix_bytes = r.OCT * 16;
    r.PP = 64 / r.OCT;
    r.nact = r.PP * r.OCT;
    r.ninst = (r.nhalo + r.PP - 1) / r.PP;
    if ((r.ninst + 3) / 4 > 12) return UMX_OK;
    r.piece_bytes = r.nact * 16;
    r.inv_oct_q16 = 65536 / r.OCT + 1;
    r.plane_bytes = round_up(r.nhalo * r.pix_bytes, 16);
    r.ec_off = 4 * r.plane_bytes;
    r.ec_units = (4 + h.head_K) * (h.NT * 4) + 4;
    r.lds_bytes = r.ec_off + (8 * h.NT * 16 + 16) * 4;   // the kernel's LDS copy holds four head rows
    if (r.lds_bytes > 160 * 1024 || r.nhalo * r.OCT > 65535) return UMX_OK;
    r.tx_log2 = lg2(tx); r.ty_log2 = lg2(ty);
    r.nk = nk;
    r.act = L.act; r.head_K = h.head_K;
    r.post_affine = (!L.post_s.empty() || !L.post_b.empty()) ? 1 : 0;
    r.econst = h.econst;
    {   // the fused 1x1 head as MFMA A-fragments (rows = classes): ceil(NT/2) k-steps, k-step s covers N-tiles (2s, 2s+1);
        // lane (q, class) element j < 4 -> channel 16*(2s) + 4q + j, j >= 4 -> channel 16*(2s+1) + 4q + j - 4: exactly the
        // channels a lane of group q holds in its accumulators of those two N-tiles, so the activations need no shuffle.
        // Weights are scaled by 2^hs (their lo parts stay normal binary16); the head's BN scale absorbs 2^-hs.
        if (!head || (int)head->head_w.size() != L.Cout * head->head_K) return UMX_OK;
        const int K = head->head_K, hks = (h.NT + 1) / 2;
        float mx = 0.f;
        for (float v : head->head_w) mx = std::max(mx, std::fabs(v));
        int hsft = 0;
        if (mx > 0.f && std::isfinite(mx)) { int e; std::frexp(mx, &e); hsft = std::max(-24, std::min(30, 14 - e)); }
        const float hscale = std::ldexp(1.f, hsft);
        std::vector<_Float16> HF((size_t)hks * 2 * 512, (_Float16)0.f);
        for (int s2 = 0; s2 < hks; ++s2)
            for (int lane = 0; lane < 64; ++lane) {
                const int cls = lane & 15, qq = lane >> 4;
                if (cls >= K) continue;
                for (int j = 0; j < 8; ++j) {
                    const int nt = 2 * s2 + (j >> 2);
                    const int c = nt * 16 + 4 * qq + (j & 3);
                    if (nt >= h.NT || c >= L.Cout) continue;
                    const float v = head->head_w[(size_t)c * K + cls] * hscale;
                    const _Float16 hi = (_Float16)v;
                    HF[((size_t)s2 * 2 + 0) * 512 + lane * 8 + j] = hi;
                    HF[((size_t)s2 * 2 + 1) * 512 + lane * 8 + j] = (_Float16)(v - (float)hi);
                }
            }
        _Float16* dh = nullptr;
        int rc3;
        if ((rc3 = upload_raw(ctx, HF, &dh))) return rc3;
        r.head_frag = reinterpret_cast<const uint4*>(dh);
        r.head_unscale = std::ldexp(1.f, -hsft);
    }
    // k-step list: group-major, tap-major, octet-minor; the tail is padded with zero-weight pairs on a loaded slot
    struct P { int gi, tap, oct; };
    std::vector<P> pairs;
    for (int gi = 0; gi < L.ngroups; ++gi)
        for (int t = 0; t < (int)L.g[gi].taps[0].size(); ++t)
            for (int o = 0; o < r.noct[gi]; ++o) pairs.push_back({gi, t, o});
    while (pairs.size() % 4) pairs.push_back({0, -1, 0});
    std::vector<unsigned short> kmap((size_t)nk * 4);
    std::vector<_Float16> W((size_t)nk * h.NT * 2 * 512, (_Float16)0.f);
    for (int j = 0; j < nk; ++j) {
        for (int qq = 0; qq < 4; ++qq) {
            const P& pr = pairs[(size_t)j * 4 + qq];
            const auto& tp = L.g[pr.gi].taps[0][pr.tap < 0 ? 0 : pr.tap];
            kmap[(size_t)j * 4 + qq] = (unsigned short)(((tp.first - r.ymin) * r.hw + (tp.second - r.xmin)) * r.OCT +
                                                        r.goct[pr.gi] + pr.oct);
        }
        for (int n = 0; n < h.NT; ++n)
            for (int lane = 0; lane < 64; ++lane) {
                const P& pr = pairs[(size_t)j * 4 + (lane >> 4)];
                if (pr.tap < 0) continue;
                const Group& G = L.g[pr.gi];
                const int Cp = round_up(G.C, 4);
                const int co = n * 16 + (lane & 15);
                const size_t base = (((size_t)j * h.NT + n) * 2) * 512 + (size_t)lane * 8;
                for (int e = 0; e < 8; ++e) {
                    const int c = pr.oct * 8 + e;
                    if (c >= G.C || co >= L.Cout) continue;
                    const float v = G.packed[0][((size_t)pr.tap * Cp + c) * L.Np + co] * wscale;
                    const _Float16 hi = (_Float16)v;
                    W[base + e] = hi;
                    W[base + 512 + e] = (_Float16)(v - (float)hi);
                }
            }
    }
    int rc;
    unsigned short* dk = nullptr;
    _Float16* dw = nullptr;
    if ((rc = upload_raw(ctx, kmap, &dk)) || (rc = upload_raw(ctx, W, &dw))) return rc;
    r.kmap = dk;
    r.w = reinterpret_cast<const uint4*>(dw);
    L.use_rw = true;
    L.exec_flops = 2.0 * 3.0 * (double)nk * 32.0 * (h.NT * 16) * L.H * L.W;
    if (getenv("UMX_DEBUG_PLAN"))
        fprintf(stderr, "[umx plan] %-12s register-resident weights: NT %d, %d k-steps, OCT %d, %d pieces per tile, LDS %d B\n",
                L.name.c_str(), h.NT, nk, r.OCT, r.ninst, r.lds_bytes);
    return UMX_OK;
}

// ---- split-precision plan of one conv launch: chunking of the input octets, k-step table, stage table, weight images
// (layout documented in umx_conv_f16.hip).  Reads the fp32 packing [tap][Cp][Np] produced by the Builder.
int plan_f16(umx_ctx* ctx, Launch& L, int act_shift, bool out_f32, const Launch* head, std::string* why) {
    const ConvParams& g = L.cp;   // tile geometry shared with the fp32 kernel
    HConvParams& h = L.hcp;
    memset(&h, 0, sizeof h);
    const int t16 = (L.Cout + 15) / 16;
    // stride-2 transposed convolution with few output channels: all four sub-pixel phases in one workgroup (the input
    // halo is read once instead of four times; 4 accumulator sets limit it to 5 N-tiles and 128 input pixels)
    const bool fused = L.nphase == 4 && L.o_mul == 2 && L.ngroups == 1 && !out_f32 && t16 <= 5 && L.H >= 8 && L.W >= 16 &&
                       !getenv("UMX_NO_FUSED_CONVT");
    h.fused_phases = fused ? 1 : 0;
    if (fused) {
        const int THg = 1 << g.th_log2, TWg = 1 << g.twm_log2;   // >= 8 and == 16 under the conditions above
        h.twm_log2 = 4; h.th_log2 = 3; h.nimg_m = 1; h.imgs = 1;
        h.hh = 8 + (g.hh - THg); h.hw = 16 + (g.hw - TWg);
        h.tiles_y = L.H / 8; h.tiles_x = L.W / 16;
    } else {
        h.twm_log2 = g.twm_log2; h.th_log2 = g.th_log2; h.nimg_m = g.nimg_m; h.imgs = g.imgs;
        h.hh = g.hh; h.hw = g.hw; h.tiles_y = g.tiles_y; h.tiles_x = g.tiles_x;
    }
    // narrow layers on full 16 x 16 tiles: 16 x 32 tiles, 8 M-tiles per wave.  Opt-in (UMX_TALL=1): half the weight traffic
    // and stage overhead per MFMA, but 215 VGPRs = 2 resident workgroups instead of 4 -- measured lu0.conv +14 %, ld0.conv +15 %
    static const bool tall_ok = getenv("UMX_TALL") && atoi(getenv("UMX_TALL")) != 0;
    const bool tall = tall_ok && !fused && t16 <= 3 && g.twm_log2 == 4 && g.th_log2 == 4 && g.imgs == 1 && L.H >= 32 && (L.H & 31) == 0;
    if (tall) {
        h.th_log2 = 5;
        h.hh = 32 + (g.hh - 16);
        h.tiles_y = L.H / 32;
    }
    h.imgplane = h.hh * h.hw; h.nhalo = h.imgs * h.imgplane;
    h.ymin = g.ymin; h.xmin = g.xmin;
    h.nphase = L.nphase; h.o_mul = L.o_mul;
    h.H = L.H; h.W = L.W; h.Cout = L.Cout; h.Cds = round_up(L.Cout, 8);
    int nt16 = 1, Np16 = 16;
    {   // N-tiles per workgroup: minimise padded N, prefer wide workgroups (fewer re-reads of the input halo)
        int best_pad = 1 << 30;
        for (int c = 1; c <= (fused ? 5 : kMaxNT16); ++c) {
            const int padded = round_up(t16, c);
            if (padded < best_pad || (padded == best_pad && c > nt16)) { nt16 = c; best_pad = padded; }
        }
        Np16 = best_pad * 16;
    }
    L.nt16 = nt16;
    h.NT = nt16; h.nblocks = Np16 / (16 * nt16);
    h.outH = L.outH; h.outW = L.outW; h.pool = L.pool; h.act = L.act;
    if (h.nhalo > kHaloChunks * 64) { *why = "halo too large for the split-precision kernel"; return UMX_ERR_INVALID; }
    h.plane_slots = round_up(h.nhalo, 16);
    const int plane_pair = h.plane_slots * 16 * 2;   // hi + lo bytes of one octet plane

    // weight shift: largest |w| lands in [2^13, 2^14) so that the lo parts stay in binary16's normal range
    float maxabs = 0.f;
    for (int ph = 0; ph < L.nphase; ++ph)
        for (int gi = 0; gi < L.ngroups; ++gi)
            for (float v : L.g[gi].packed[ph]) maxabs = std::max(maxabs, std::fabs(v));
    L.wshift = 0;
    if (maxabs > 0.f && std::isfinite(maxabs)) {
        int e;
        std::frexp(maxabs, &e);           // maxabs = m * 2^e, m in [0.5, 1)
        L.wshift = std::max(-24, std::min(30, 14 - e));
    }
    const float wscale = std::ldexp(1.f, L.wshift);

    // ---- chunking.  A chunk = up to OC octets of one operand group, resident in LDS while its (tap, octet) pairs are
    // consumed 4 per k-step; a stage = up to S k-steps = one weight block.  Consecutive chunks alternate between two
    // halo slots (even chunks at plane 0, odd chunks behind them) so that chunk c+1 loads while chunk c computes.
    // Per-phase kernels walk their own chunk list (groups with taps in that phase); the fused kernel walks one list
    // and, inside each chunk, the phases one after the other.
    int noct[2] = {0, 0};
    for (int gi = 0; gi < L.ngroups; ++gi) noct[gi] = round_up(L.g[gi].C, 8) / 8;
    struct Chunk { int gi, o0, o1; };
    auto chunks_for = [&](int OC, int ph /* -1: every group */) {
        std::vector<Chunk> out;
        for (int gi = 0; gi < L.ngroups; ++gi) {
            if (ph >= 0 && L.g[gi].taps[ph].empty()) continue;
            const int nchunk = (noct[gi] + OC - 1) / OC;
            for (int c = 0; c < nchunk; ++c) out.push_back({gi, c * noct[gi] / nchunk, (c + 1) * noct[gi] / nchunk});
        }
        return out;
    };
    const int nlists = fused ? 1 : L.nphase;   // independent stage lists (= kernel phases)
    auto phases_of = [&](int list) { return fused ? std::make_pair(0, L.nphase) : std::make_pair(list, list + 1); };
    // Search attempts, in order of preference: (pieces per wave and chunk the kernel instantiation indexes, LDS budget per
    // workgroup).  80 KiB = 2 workgroups per CU; narrow layers (few accumulators -> few VGPRs) first try the budgets that let
    // 4 (<= 3 N-tiles, 128 VGPRs) or 3 workgroups per CU cover each other, and the 4-piece instantiation (8 VGPRs fewer).
    struct Attempt { int maxp, cap, max_chunks; };
    std::vector<Attempt> attempts;
    {
        const char* e = getenv("UMX_LDS_CAP_NARROW");
        const int narrow = e ? atoi(e) : 53 * 1024;
        const char* e2 = getenv("UMX_NARROW_NT");
        const int narrow_nt = e2 ? atoi(e2) : 5;   // the kernels of <= 5 N-tiles fit 3 waves per SIMD
        const char* e3 = getenv("UMX_LDS_CAP_NT3");
        const int nt3 = e3 ? atoi(e3) : 40 * 1024;
        if (fused) attempts = {{nt16 <= 3 ? 12 : 4, kMaxLdsPerWG, 1 << 30}};   // (the fused kernels exist in one piece count each)
        else if (tall) attempts = {{4, kMaxLdsPerWG, 1 << 30}};   // (192+ VGPRs: two workgroups per CU whatever the LDS)
        else {
            for (int maxp : {4, 12}) {
                if (maxp == 12 && nt16 > 5) break;
                // (4-5 N-tiles: the third workgroup per CU pays only while the smaller chunks stay few -- ld1.conv, 6 chunks:
                // -12 %; lu1.conv, 24 chunks: +1 %)
                const char* e4 = getenv("UMX_NARROW_MAXCHUNKS");
                const int few = nt16 <= 3 ? 1 << 30 : e4 ? atoi(e4) : 8;
                if (nt16 <= 3 && maxp == 4 && nt3 >= 16 * 1024) attempts.push_back({maxp, std::min(nt3, kMaxLdsPerWG), 1 << 30});
                if (nt16 <= narrow_nt && narrow >= 16 * 1024) attempts.push_back({maxp, std::min(narrow, kMaxLdsPerWG), few});
                attempts.push_back({maxp, kMaxLdsPerWG, 1 << 30});
            }
        }
    }
    const int stage_rows = fused ? 32 : 16;
    const int nwaves = kWaves;
    h.kmt = fused ? 2 : tall ? 8 : kMT;
    const int epi_bytes = nwaves * 2 * stage_rows * (nt16 * 32 + 16);   // epilogue transpose staging
    // dynamic LDS of a plan: halo slots (hi + lo) | weight buffer 0 | weight buffer 1 | the staging area, unless it fits below
    // buffer 1 (the epilogue constants sit in buffer 1 while the staged rows are written: the kernel orders the buffers so)
    auto lds_total = [&](int nslots, int oc, int ss, int plane_pair_bytes) {
        const int below = nslots * oc * plane_pair_bytes, wb = 64 + ss * nt16 * 2048;
        return below + 2 * wb + (epi_bytes <= below + wb ? 0 : epi_bytes);
    };

    // One stage list (= one kernel phase, or the whole fused transposed convolution) for a given (OC, S, slot base E):
    // (tap, octet) pairs of every chunk -> k-steps of 4 -> stages of <= S k-steps.  Pairs left over when a chunk's
    // pair count is not a multiple of 4 are carried into the first k-step of the next chunk instead of being padded:
    // the previous chunk's halo slot is still resident then (its reload is issued at the start of the next chunk's LAST
    // stage, so the next chunk must have >= 2 stages).  Not across the phases of the fused kernel (other accumulators).
    struct Pair { int gi, ph, tap, oct, slot, k; };   // slot: halo slot (0/1) of the chunk; k: octet inside the chunk
    const bool carry_ok = !fused && !getenv("UMX_NO_KSTEP_CARRY");
    auto npairs_of = [&](const Chunk& k, int ph) { return (int)L.g[k.gi].taps[ph].size() * (k.o1 - k.o0); };
    auto plan_list = [&](int OC, int S, int list, std::vector<HStage>* stages_out,
                         std::vector<std::vector<Pair>>* steps_out, int* nchunks) {
        const auto pr = phases_of(list);
        const auto ch = chunks_for(OC, fused ? -1 : list);
        if (nchunks) *nchunks = (int)ch.size();
        int nsteps = 0;
        std::vector<Pair> carry;
        const size_t list_base = stages_out ? stages_out->size() : 0;
        std::vector<int> first_stage(ch.size(), -1);      // per chunk: index (in stages_out) of its first stage
        std::vector<char> carried_in(ch.size(), 0);       // per chunk: its first k-step still reads the previous chunk's slot
        for (size_t c = 0; c < ch.size(); ++c) {
            const int gi = ch[c].gi, o0 = ch[c].o0, o1 = ch[c].o1;
            const int slot = (int)(c & 1);   // consecutive chunks alternate between the two halo slots
            bool first = true;   // the chunk's first stage carries its halo load
            carried_in[c] = !carry.empty();
            if (stages_out) first_stage[c] = (int)stages_out->size();
            for (int ph = pr.first; ph < pr.second; ++ph) {
                const int nt = (int)L.g[gi].taps[ph].size();
                if (!nt) continue;
                std::vector<Pair> pairs = carry;
                carry.clear();
                for (int t = 0; t < nt; ++t)
                    for (int o = o0; o < o1; ++o) pairs.push_back({gi, ph, t, o, slot, o - o0});
                const int rem = (int)pairs.size() % 4;
                if (rem && carry_ok && c + 1 < ch.size() && (int)pairs.size() >= 4) {
                    // stages the next chunk will have if it takes the remainder (it pads or carries on in turn)
                    const int next_k = (rem + npairs_of(ch[c + 1], ph) + (c + 2 < ch.size() ? 0 : 3)) / 4;
                    if ((next_k + S - 1) / S >= 2) {
                        carry.assign(pairs.end() - rem, pairs.end());
                        pairs.resize(pairs.size() - rem);
                    }
                }
                while (pairs.size() % 4) pairs.push_back({gi, ph, -1, o0, slot, 0});   // zero-weight filler on a loaded slot
                const int nk_chunk = (int)pairs.size() / 4;
                nsteps += nk_chunk;
                for (int k = 0; k < nk_chunk; k += S) {
                    HStage st;
                    memset(&st, 0, sizeof st);
                    st.group = first ? (short)gi : (short)-1;
                    first = false;
                    st.oct0 = (short)o0;
                    st.noct = (short)(o1 - o0);
                    st.plane0 = (short)slot;
                    st.phase = (short)ph;
                    st.nk = (short)std::min(S, nk_chunk - k);
                    if (steps_out)
                        for (int j = 0; j < st.nk; ++j)
                            steps_out->push_back(std::vector<Pair>(pairs.begin() + (k + j) * 4, pairs.begin() + (k + j) * 4 + 4));
                    if (stages_out) stages_out->push_back(st);
                }
            }
        }
        // Early halo issue.  The kernel issues the load a stage record carries at the top of the stage BEFORE it, i.e. by
        // default one stage ahead of the chunk's first k-step -- less than a load's round trip under load on the short
        // stages of the narrow layers.  The slot of chunk c is free as soon as chunk c-2 is done with it: after the barrier
        // that opens chunk c-1's first stage, or its second one when chunk c-1's first k-step carries pairs of chunk c-2.
        // The load record of chunk c therefore moves to the stage after that barrier's stage (never past its own chunk).
        static const bool early = getenv("UMX_EARLY_HALO") != nullptr;   // opt-in: neutral on the whole step (lu0.conv +6 %, lu2.convT -6 %)
        if (stages_out && early)
            for (size_t c = 1; c < ch.size(); ++c) {
                const int own = first_stage[c], prev = first_stage[c - 1];
                if (own < 0 || prev < 0) continue;
                const int tgt = std::max((int)list_base + 1, prev + (carried_in[c - 1] ? 2 : 1));
                if (tgt >= own) continue;
                HStage& from = (*stages_out)[own];
                HStage& to = (*stages_out)[tgt];
                if (from.group < 0 || to.group >= 0) continue;   // (one load per record)
                to.group = from.group; to.oct0 = from.oct0; to.noct = from.noct; to.plane0 = from.plane0;
                from.group = -1;
            }
        return nsteps;
    };

    // (OC, S) search.  OC = octets per staged pixel (pixel pitch OC*16 B in the LDS image).  Odd OC maps 16 consecutive pixels
    // at one octet to 16 distinct 16-byte bank groups (conflict-free fragment reads); even OC costs 2- to 4-way conflicts
    // on those reads, which the kernels tolerate (LDS reads are not their limit) -- a mild penalty only.
    int bestOC = 0, bestS = 0, bestSlots = 1, maxp = 4;
    double bestCost = 1e30;
    for (size_t at = 0; at < attempts.size() && !bestOC; ++at) {
    maxp = attempts[at].maxp;
    const int lds_cap = attempts[at].cap;
    for (int OC = 1; OC <= 9; ++OC) {
        int nslots = 1;
        double sectors = 0;   // 64-byte memory requests of the halo loads of one workgroup
        for (int list = 0; list < nlists; ++list) {
            const auto ch = chunks_for(OC, fused ? -1 : list);
            if (ch.size() >= 2) nslots = 2;
            for (const auto& c : ch) sectors += 2.0 * h.nhalo * ((c.o1 - c.o0 + 3) / 4);
        }
        // the kernel keeps one pixel index per (wave, piece of a chunk) in registers
        if (((h.nhalo + 64 / OC - 1) / (64 / OC) + nwaves - 1) / nwaves > maxp) continue;
        for (int S = 1; S <= kStageK; ++S) {
            const int lds = lds_total(nslots, OC, S, plane_pair);
            if (lds > lds_cap) continue;
            int ksteps = 0, nchunks = 0;
            for (int list = 0; list < nlists; ++list) {
                int nc = 0;
                ksteps += plan_list(OC, S, list, nullptr, nullptr, &nc);
                nchunks += nc;
            }
            // executed k-steps (exact) with a barrier/latency charge per stage, a charge per halo chunk load (measured
            // ~0.35 k-steps on the deep layers) and per 64-byte halo request (halo reloads measured at 8-22 % of a layer)
            const double cost = (ksteps * (1.0 + 0.30 / S) + 0.35 * nchunks + sectors / 1500.0) * ((OC & 1) ? 1.0 : 1.03);
            if (nchunks > attempts[at].max_chunks) continue;
            if (cost < bestCost) { bestCost = cost; bestOC = OC; bestS = S; bestSlots = nslots; }
        }
    }
    }
    if (!bestOC) { *why = "LDS footprint too large for the split-precision kernel"; return UMX_ERR_INVALID; }
    if (const char* e = getenv("UMX_PLAN_OVERRIDE")) {   // tuning aid: "layer:OC:S[,layer:OC:S...]" forces a layer's (OC, S)
        std::string spec(e);
        size_t pos = 0;
        while (pos < spec.size()) {
            const size_t end = spec.find(',', pos);
            const std::string item = spec.substr(pos, end == std::string::npos ? std::string::npos : end - pos);
            char nm[64];
            int oc = 0, ss = 0, mp = 0;
            const int nf = sscanf(item.c_str(), "%63[^:]:%d:%d:%d", nm, &oc, &ss, &mp);
            if (nf >= 3 && L.name == nm && oc >= 1 && oc <= 9 && ss >= 1 && ss <= kStageK) {
                if (nf == 4 && (mp == 4 || (mp == 12 && nt16 <= 5)) && !fused) maxp = mp;
                int nslots = 1;
                for (int list = 0; list < nlists; ++list)
                    if (chunks_for(oc, fused ? -1 : list).size() >= 2) nslots = 2;
                const int lds = lds_total(nslots, oc, ss, plane_pair);
                const bool pieces_ok = ((h.nhalo + 64 / oc - 1) / (64 / oc) + nwaves - 1) / nwaves <= maxp;
                if (lds <= kMaxLdsPerWG && pieces_ok) { bestOC = oc; bestS = ss; bestSlots = nslots; }
                else fprintf(stderr, "[umx plan] override %s ignored (LDS %d B)\n", item.c_str(), lds);
            }
            if (end == std::string::npos) break;
            pos = end + 1;
        }
    }
    const int OC = bestOC, S = bestS;
    h.OC = OC;
    h.inv_OC = 1.f / (float)OC;
    h.PP = 64 / OC;
    h.maxp = maxp;
    h.nact = h.PP * OC;
    h.ninst = (h.nhalo + h.PP - 1) / h.PP;
    h.piece_bytes = h.nact * 16;
    h.inv_oc_q16 = 65536 / OC + 1;
    h.pix_bytes = OC * 16;
    h.slot_bytes = h.plane_slots * OC * 16;
    h.lo_off = bestSlots * h.slot_bytes;
    h.b_off = 2 * h.lo_off;
    h.wbuf_bytes = 64 + S * nt16 * 2048;
    {   // XCD-aware tile order (default on; UMX_XCD_ORDER=0 off, or a comma list of layer-name prefixes to limit it)
        h.xcd_order = 1;
        if (const char* e = getenv("UMX_XCD_ORDER")) {
            std::string spec(e);
            h.xcd_order = (spec == "1" || spec == "all") ? 1 : 0;
            size_t pos = 0;
            while (pos < spec.size()) {
                const size_t end = spec.find(',', pos);
                const std::string tok = spec.substr(pos, end == std::string::npos ? std::string::npos : end - pos);
                if (tok.size() > 1 && L.name.compare(0, tok.size(), tok) == 0) h.xcd_order = 1;
                if (end == std::string::npos) break;
                pos = end + 1;
            }
        }
    }
    {   // de-phasing of the workgroups that share a CU (see conv_f16x3): UMX_STAGGER = "cycles" or "layer:cycles,..."
        h.stagger = 0;
        h.nres = std::max(1, std::min((fused || tall) ? 2 : (nt16 <= 3 && maxp == 4) ? 4 : (nt16 <= 3 || (nt16 <= 5 && maxp == 4)) ? 3 : 2, (160 * 1024) / std::max(1, lds_total(bestSlots, OC, S, plane_pair))));
        h.first_gen = h.nres * ctx->ncu;
        if (const char* e = getenv("UMX_STAGGER")) {
            std::string spec(e);
            if (spec.find(':') == std::string::npos) h.stagger = atoi(e);
            else {
                size_t pos = 0;
                while (pos < spec.size()) {
                    const size_t end = spec.find(',', pos);
                    const std::string item = spec.substr(pos, end == std::string::npos ? std::string::npos : end - pos);
                    const size_t c = item.find(':');
                    if (c != std::string::npos && item.substr(0, c) == L.name) h.stagger = atoi(item.c_str() + c + 1);
                    if (end == std::string::npos) break;
                    pos = end + 1;
                }
            }
        }
    }
    // epilogue transpose staging: below weight buffer 1 when it fits there (the constants sit in buffer 1), else above it
    h.stg_off = epi_bytes <= h.b_off + h.wbuf_bytes ? 0 : h.b_off + 2 * h.wbuf_bytes;
    h.lds_bytes = std::max(h.b_off + 2 * h.wbuf_bytes, h.stg_off + epi_bytes);
    if (h.lds_bytes > kMaxLdsPerWG) { *why = "epilogue staging exceeds the LDS budget"; return UMX_ERR_INVALID; }

    std::vector<HStage> stages;
    std::vector<std::vector<_Float16>> wimg(nlists);   // per stage list: [nblk][stage blocks] halves
    L.n_ksteps = 0;
    for (int list = 0; list < nlists; ++list) {
        h.ph[list].oy_off = L.oy_off[list];
        h.ph[list].ox_off = L.ox_off[list];
        h.ph[list].stage0 = (int)stages.size();
        std::vector<std::vector<Pair>> steps;   // k-steps of this list, each 4 pairs (padded ones have tap = -1)
        plan_list(OC, S, list, &stages, &steps, nullptr);
        h.ph[list].nstages = (int)stages.size() - h.ph[list].stage0;
        L.n_ksteps += (int)steps.size();
        // weight slab of one N-block: per stage a block = 64-byte header (k-map) + nk * NT * (hi, lo) images
        size_t per_blk = 0;   // halves
        for (int si = h.ph[list].stage0; si < (int)stages.size(); ++si) {
            stages[si].woff = (int)(per_blk / 8);
            per_blk += 32 + (size_t)stages[si].nk * nt16 * 2 * 512;
        }
        h.ph[list].wblk_stride = (int)(per_blk / 8);
        std::vector<_Float16>& W = wimg[list];
        W.assign(per_blk * h.nblocks, (_Float16)0.f);
        for (int nb = 0; nb < h.nblocks; ++nb) {
            size_t ks = 0;
            for (int si = h.ph[list].stage0; si < (int)stages.size(); ++si) {
                const size_t blk = nb * per_blk + (size_t)stages[si].woff * 8;
                unsigned short* const hdr = reinterpret_cast<unsigned short*>(&W[blk]);
                for (int j = 0; j < stages[si].nk; ++j, ++ks) {
                    for (int qq = 0; qq < 4; ++qq) {
                        const Pair& pr2 = steps[ks][qq];
                        const auto& tp = L.g[pr2.gi].taps[pr2.ph][pr2.tap < 0 ? 0 : pr2.tap];
                        // 16-byte LDS slot of (halo pixel at this tap, octet k) in the pixel-major image of halo slot `slot`
                        const int slot = pr2.slot * (h.plane_slots * OC) + ((tp.first - g.ymin) * h.hw + (tp.second - g.xmin)) * OC + pr2.k;
                        if (slot < 0 || slot >= bestSlots * h.plane_slots * OC || slot > 65535) {
                            *why = "internal: k-map slot out of range";
                            return UMX_ERR_INVALID;
                        }
                        hdr[j * 4 + qq] = (unsigned short)slot;
                    }
                    for (int n = 0; n < nt16; ++n)
                        for (int lane = 0; lane < 64; ++lane) {
                            const Pair& pr2 = steps[ks][lane >> 4];
                            if (pr2.tap < 0) continue;
                            const Group& G = L.g[pr2.gi];
                            const int Cp = round_up(G.C, 4);
                            const int co = nb * nt16 * 16 + n * 16 + (lane & 15);
                            const size_t base = blk + 32 + (((size_t)j * nt16 + n) * 2) * 512 + (size_t)lane * 8;
                            for (int e = 0; e < 8; ++e) {
                                const int c = pr2.oct * 8 + e;
                                if (c >= G.C || co >= L.Cout) continue;
                                const float v = G.packed[pr2.ph][((size_t)pr2.tap * Cp + c) * L.Np + co] * wscale;
                                const _Float16 hi = (_Float16)v;
                                W[base + e] = hi;
                                W[base + 512 + e] = (_Float16)(v - (float)hi);
                            }
                        }
                }
            }
        }
    }

    // epilogue constants per N-block: pre_s absorbs 2^-(weight shift + input activation shift), post_* the output's
    // 2^(activation shift); padded channels get pre_s = post_s = 0 so that they store exact zeros
    {
        const float unshift = std::ldexp(1.f, -(L.wshift + act_shift));
        const float oscale = out_f32 ? 1.f : std::ldexp(1.f, act_shift);
        const int nb16 = nt16 * 16;
        // a fused softmax head needs every channel of a pixel in one workgroup; it replaces the fp32 store of this layer
        const bool fuse_head = head && out_f32 && h.nblocks == 1 && head->head_K <= 4 && !fused && !getenv("UMX_NO_FUSED_HEAD");
        h.head_K = fuse_head ? head->head_K : 0;
        const size_t per_blk = fuse_head ? (size_t)(4 + h.head_K) * nb16 + 16 : (size_t)4 * nb16;
        std::vector<float> ec((size_t)h.nblocks * per_blk, 0.f);
        for (int nb = 0; nb < h.nblocks; ++nb)
            for (int i = 0; i < nb16; ++i) {
                const int c = nb * nb16 + i;
                if (c >= L.Cout) continue;
                float* e = &ec[(size_t)nb * per_blk];
                e[0 * nb16 + i] = (L.pre_s.empty() ? 1.f : L.pre_s[c]) * unshift;
                e[1 * nb16 + i] = L.pre_b.empty() ? 0.f : L.pre_b[c];
                e[2 * nb16 + i] = (L.post_s.empty() ? 1.f : L.post_s[c]) * oscale;
                e[3 * nb16 + i] = (L.post_b.empty() ? 0.f : L.post_b[c]) * oscale;
                for (int k = 0; k < h.head_K; ++k) e[(4 + k) * nb16 + i] = head->head_w[(size_t)c * head->head_K + k];
            }
        if (fuse_head) {
            float* e = &ec[(size_t)(4 + h.head_K) * nb16];
            for (int k = 0; k < h.head_K; ++k) {
                e[k] = head->pre_s.empty() ? 1.f : head->pre_s[k];
                e[8 + k] = head->pre_b.empty() ? 0.f : head->pre_b[k];
            }
        }
        h.post_affine = 0;
        for (int nb = 0; nb < h.nblocks; ++nb)
            for (int i = 0; i < nb16; ++i) {
                if (nb * nb16 + i >= L.Cout) continue;
                const float* e = &ec[(size_t)nb * per_blk];
                if (e[2 * nb16 + i] != 1.f || e[3 * nb16 + i] != 0.f) h.post_affine = 1;
            }
        float* d = nullptr;
        int rc2 = upload(ctx, ec, &d);
        if (rc2) return rc2;
        h.econst = reinterpret_cast<const uint4*>(d);
    }
    if (getenv("UMX_DEBUG_PLAN"))
        fprintf(stderr, "[umx plan] %-12s %sNT %d x %d blocks, OC %d x %d halo slot(s), S %d, LDS %d B, k-steps %d, wshift %d\n",
                L.name.c_str(), fused ? "fused-phase " : "", nt16, h.nblocks, OC, bestSlots, S, h.lds_bytes, L.n_ksteps,
                L.wshift);
    h.inv_imgplane = 1.f / (float)h.imgplane;
    h.inv_hw = 1.f / (float)h.hw;
    int rc;
    HStage* d_st = nullptr;
    {
        HStage dummy;   // the kernel reads stages[stage0] before looking at nstages
        memset(&dummy, 0, sizeof dummy);
        dummy.group = -1;
        stages.push_back(dummy);
    }
    if ((rc = upload_raw(ctx, stages, &d_st))) return rc;
    h.stages = d_st;
    for (int list = 0; list < nlists; ++list) {
        _Float16* d = nullptr;
        if ((rc = upload_raw(ctx, wimg[list], &d))) return rc;
        h.ph[list].w = reinterpret_cast<const uint4*>(d);
    }
    L.exec_flops = 2.0 * 3.0 * (double)L.n_ksteps * 32.0 * Np16 * L.H * L.W;   // MFMA work incl. split and padding
    return plan_rw(ctx, L, wscale, head, why);
}

int grow(umx_ctx* ctx, void** buf, size_t* cap, size_t bytes) {
    if (*cap >= bytes) return UMX_OK;
    if (*buf) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(*buf));
        *buf = nullptr;
        *cap = 0;
    }
    HIP_TRY(ctx, hipMalloc(buf, bytes));
    *cap = bytes;
    return UMX_OK;
}

int site_of(umx_ctx* ctx, const std::string& name, const std::string& kernel) {
    for (size_t i = 0; i < ctx->sites.size(); ++i)
        if (ctx->sites[i].name == name) return (int)i;
    ProfSite s;
    s.name = name;
    s.kernel = kernel;
    ctx->sites.push_back(s);
    return (int)ctx->sites.size() - 1;
}

int prof_fold(umx_ctx* ctx) {
    if (ctx->pending.empty()) return UMX_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->stream2) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream2));
    for (auto& pe : ctx->pending) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, pe.a, pe.b));
        ctx->sites[pe.site].total_ms += ms;
        ctx->free_events.push_back(pe.a);
        ctx->free_events.push_back(pe.b);
    }
    ctx->pending.clear();
    return UMX_OK;
}

struct ProfScope {
    umx_ctx* ctx;
    int site;
    hipEvent_t a = nullptr, b = nullptr;
    bool on;
    ProfScope(umx_ctx* c, int s, double flops, double bytes, double exec = 0.0) : ctx(c), site(s), on(c->prof && s >= 0) {
        if (!on) return;
        auto get = [&](hipEvent_t* e) {
            if (!ctx->free_events.empty()) { *e = ctx->free_events.back(); ctx->free_events.pop_back(); }
            else if (hipEventCreate(e) != hipSuccess) *e = nullptr;
        };
        get(&a);
        get(&b);
        if (!a || !b) { on = false; return; }
        ctx->sites[site].launches += 1;
        ctx->sites[site].flops += flops;
        ctx->sites[site].bytes += bytes;
        ctx->sites[site].exec += exec;
        hipEventRecord(a, run_stream(ctx));
    }
    ~ProfScope() {
        if (!on) return;
        hipEventRecord(b, run_stream(ctx));
        ctx->pending.push_back({site, a, b});
    }
};

// run the UNet on n tiles already in bufs[0] layout at `tiles` -> probs
// the (hi, lo) planes of buffer b for a batch of n tiles
inline _Float16* hi_of(const Buffer& b) { return reinterpret_cast<_Float16*>(b.d); }
inline _Float16* lo_of(const Buffer& b, int n) { return reinterpret_cast<_Float16*>(b.d) + (size_t)n * b.S * b.S * b.Cs; }

// One launch of the split-precision plan on tiles [k0, k0+ns) of a batch of n: every tensor lives in its full-batch
// buffer (hi plane of n tiles, then lo plane of n tiles), a sub-batch is a slice of both planes.
int run_launch_f16(umx_ctx* ctx, Launch& L, const float* tiles, int n, int k0, int ns, float* probs) {
    auto hi_at = [&](const Buffer& b) { return hi_of(b) + (size_t)k0 * b.S * b.S * b.Cs; };
    auto lo_at = [&](const Buffer& b) { return lo_of(b, n) + (size_t)k0 * b.S * b.S * b.Cs; };
    if (L.name == "input.split") {   // fp32 tiles -> (hi, lo) input planes (2 -> 8 channels, scaled by 2^act_shift)
        if (!tiles) return UMX_OK;   // the gather kernel already wrote the (hi, lo) planes
        const Buffer& b0 = cur_bufs(ctx)[0];
        if (ctx->site_split < 0) ctx->site_split = site_of(ctx, "input.split", "split_f32");
        ProfScope ps(ctx, ctx->site_split, 0.0, (double)ns * b0.S * b0.S * (4.0 * b0.C + 4.0 * b0.Cs));
        HIP_TRY(ctx, launch_split_f32(tiles + (size_t)k0 * b0.floats_per_tile, (size_t)ns * b0.S * b0.S, b0.C, b0.Cs,
                                      std::ldexp(1.f, ctx->act_shift), hi_at(b0), lo_at(b0), run_stream(ctx)));
        return UMX_OK;
    }
    if (L.head) {
        if (ctx->head_fused) return UMX_OK;   // computed in the epilogue of the last convolution
        const Buffer& sb = cur_bufs(ctx)[L.g[0].src];
        const size_t npix = (size_t)ns * L.H * L.W;
        ProfScope ps(ctx, site_of(ctx, L.name, "head_softmax"), L.flops * ns, L.bytes * ns);
        HIP_TRY(ctx, launch_head_softmax(sb.d + (size_t)k0 * sb.floats_per_tile, npix, L.head_C, L.head_K, L.d_head_w,
                                         L.d_pre_s, L.d_pre_b, probs + (size_t)k0 * L.H * L.W * L.head_K, run_stream(ctx)));
        return UMX_OK;
    }
    HConvParams p = L.hcp;
    p.B = ns;
    bool any_planar = false;
    for (int gi = 0; gi < L.ngroups; ++gi) {
        const Buffer& sb = cur_bufs(ctx)[L.g[gi].src];
        p.src_hi[gi] = hi_at(sb);
        p.src_lo[gi] = lo_at(sb);
        p.Cs[gi] = sb.Cs;
        p.srcA[gi] = sb.planar ? 16 : sb.Cs * 2;
        p.srcB[gi] = sb.planar ? sb.S * sb.S * 16 : 16;
        if (sb.planar) any_planar = true;
    }
    if (L.ngroups < 2) { p.srcA[1] = p.srcA[0]; p.srcB[1] = p.srcB[0]; }
    const Buffer& db = cur_bufs(ctx)[L.dst];
    if (p.head_K > 0) p.probs = probs + (size_t)k0 * L.H * L.W * p.head_K;   // fused softmax head
    else if (db.as_f32) p.dst_f32 = db.d + (size_t)k0 * db.floats_per_tile;
    else { p.dst_hi = hi_at(db); p.dst_lo = lo_at(db); p.dst_planar = db.planar ? 1 : 0; }
    if (L.app_src >= 0) {
        const Buffer& ab = cur_bufs(ctx)[L.app_src];
        p.app_hi = hi_at(ab); p.app_lo = lo_at(ab); p.app_Cs = ab.Cs;
        p.app_c0 = (L.app_c0 / 8) * 8;            // first channel of the destination octet
        p.app_word = (L.app_c0 % 8) / 2;          // 32-bit word of that octet the two appended binary16 values fill
    }
    char kn[48];
    // the instantiation as rocprofv3 names it (<NT, KMT, NPH>): bench.py groups the timed sites by kernel
    snprintf(kn, sizeof kn, "conv_f16x3<%d, %d, %d, false, %d>", L.nt16, p.kmt, p.fused_phases ? 4 : 1, p.maxp);   // as rocprofv3 prints it
    {
        // diagnostic: UMX_DEBUG_STAMPS=<layer name> prints the mean s_memtime segments of that layer's workgroups
        static const char* dbg_layer = getenv("UMX_DEBUG_STAMPS");
        if (dbg_layer && L.name == dbg_layer) {
            const size_t nwg = (size_t)((ns + p.imgs - 1) / p.imgs) * p.tiles_y * p.tiles_x * p.nblocks * p.nphase;
            long long* d = nullptr;
            HIP_TRY(ctx, hipMalloc((void**)&d, nwg * 7 * sizeof(long long)));
            p.dbg = d;
            HIP_TRY(ctx, launch_conv_f16(p, run_stream(ctx)));
            HIP_TRY(ctx, hipStreamSynchronize(run_stream(ctx)));
            std::vector<long long> hst(nwg * 7);
            HIP_TRY(ctx, hipMemcpy(hst.data(), d, hst.size() * sizeof(long long), hipMemcpyDeviceToHost));
            hipFree(d);
            double m[7] = {0, 0, 0, 0, 0, 0, 0};
            for (size_t w = 0; w < nwg; ++w)
                for (int k = 0; k < 7; ++k) m[k] += (double)hst[w * 7 + k] / nwg;
            fprintf(stderr, "[umx stamps] %s: %zu workgroups, LDS %d B, wbuf %d B | shader cycles: prologue %.0f, stage waits %.0f "
                            "(own loads %.0f, barrier %.0f), issuing the next stage's loads %.0f, MFMA blocks %.0f, epilogue %.0f, total %.0f\n",
                    L.name.c_str(), nwg, p.lds_bytes, p.wbuf_bytes, m[0], m[1], m[5], m[1] - m[5], m[6], m[2] - m[6], m[3], m[4]);
            return UMX_OK;
        }
    }
    if (L.use_rw && !any_planar && (size_t)ns * L.H * L.W * std::max(p.Cs[0], p.Cs[1]) * 2 < 0x7fffffffu) {
        RwParams r = L.rw;
        r.B = ns;
        for (int gi = 0; gi < L.ngroups; ++gi) { r.src_hi[gi] = p.src_hi[gi]; r.src_lo[gi] = p.src_lo[gi]; r.Cs[gi] = p.Cs[gi]; }
        r.probs = p.probs;
        r.overflow_flag = ctx->d_flag;
        r.ntiles = ns << (r.tx_log2 + r.ty_log2);
        snprintf(kn, sizeof kn, "conv_rw<%d, %d>", L.nt16, r.nk);
        ProfScope ps(ctx, site_of(ctx, L.name, kn), L.flops * ns, L.bytes * ns, L.exec_flops * ns);
        HIP_TRY(ctx, launch_conv_rw(r, L.nt16, ctx->ncu, run_stream(ctx)));
        return UMX_OK;
    }
    ProfScope ps(ctx, site_of(ctx, L.name, kn), L.flops * ns, L.bytes * ns, L.exec_flops * ns);
    HIP_TRY(ctx, launch_conv_f16(p, run_stream(ctx)));
    return UMX_OK;
}

// The split-precision plan: input split (unless the gather kernel already wrote the planes), then every launch once
// over the whole batch.  (Running the full-resolution layers as chains over sub-batches, to keep producer -> consumer
// tensors in the 256 MiB Infinity Cache, was measured slower at every sub-batch size -- DESIGN.md section 4.)
int run_unet_f16(umx_ctx* ctx, const float* tiles, int n, float* probs) {
    int rc = run_launch_f16(ctx, ctx->split_launch, tiles, n, 0, n, probs);
    if (rc) return rc;
    for (auto& L : ctx->plan)
        if ((rc = run_launch_f16(ctx, L, tiles, n, 0, n, probs))) return rc;
    if (ctx->prof && ctx->pending.size() > 4096) return prof_fold(ctx);
    return UMX_OK;
}

int check_range_flag(umx_ctx* ctx) {   // call with the stream idle
    if (!ctx->d_flag) return UMX_OK;
    int f = 0;
    HIP_TRY(ctx, hipMemcpy(&f, ctx->d_flag, sizeof f, hipMemcpyDeviceToHost));
    if (!f) return UMX_OK;
    HIP_TRY(ctx, hipMemset(ctx->d_flag, 0, sizeof f));
    return fail(ctx, UMX_ERR_RANGE, "an activation left the binary16 range of the split-precision path; "
                                    "create the context with UMX_PREC_F32 (or UMX_PRECISION=f32)");
}

int run_unet(umx_ctx* ctx, const float* tiles, int n, float* probs) {
    if (ctx->precision == UMX_PREC_F16X3) return run_unet_f16(ctx, tiles, n, probs);
    const umx_hparams& hp = ctx->hp;
    for (auto& L : ctx->plan) {
        const float* src0 = L.g[0].src == 0 ? tiles : cur_bufs(ctx)[L.g[0].src].d;
        if (L.head) {
            const size_t npix = (size_t)n * L.H * L.W;
            ProfScope ps(ctx, site_of(ctx, L.name, "head_softmax"), L.flops * n, L.bytes * n);
            HIP_TRY(ctx, launch_head_softmax(src0, npix, L.head_C, L.head_K, L.d_head_w, L.d_pre_s, L.d_pre_b, probs,
                                             run_stream(ctx)));
            continue;
        }
        ConvParams p = L.cp;
        p.B = n;
        p.src[0] = src0;
        if (L.ngroups > 1) p.src[1] = L.g[1].src == 0 ? tiles : cur_bufs(ctx)[L.g[1].src].d;
        p.dst = cur_bufs(ctx)[L.dst].d;
        char kn[48];
        snprintf(kn, sizeof kn, "conv_mfma_f32<%d, %d>", L.nt, L.hpix <= 2 ? 2 : 4);
        ProfScope ps(ctx, site_of(ctx, L.name, kn), L.flops * n, L.bytes * n, L.exec_flops * n);
        HIP_TRY(ctx, launch_conv(p, L.nt, L.hpix, run_stream(ctx)));
    }
    if (ctx->prof && ctx->pending.size() > 4096) return prof_fold(ctx);
    (void)hp;
    return UMX_OK;
}

// Batching of `total` tiles over the lanes of a context: equal batches (<= max_batch), as many as a multiple of the lane
// count when there is more than one batch, consecutive batches on alternating lanes.  The second lane's stream is forked
// from the context's stream by an event at the first use and joined back in join(); with one lane (or one batch) every
// launch stays on the context's stream.
struct LaneLoop {
    umx_ctx* ctx;
    int batch, k = 0;
    bool forked = false;
    LaneLoop(umx_ctx* c, int total) : ctx(c) {
        int nbatch = (total + c->max_batch - 1) / c->max_batch;
        if (c->nlanes > 1 && nbatch > 1) nbatch = (nbatch + c->nlanes - 1) / c->nlanes * c->nlanes;
        batch = nbatch > 0 ? (total + nbatch - 1) / nbatch : 1;
        ctx->lane = 0;
    }
    int next(int left) {
        ctx->lane = ctx->nlanes > 1 ? k % ctx->nlanes : 0;
        if (ctx->lane == 1 && !forked) {
            forked = true;
            if (hipEventRecord(ctx->ev_fork, ctx->stream) != hipSuccess ||
                hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0) != hipSuccess) ctx->lane = 0;   // degrade to one lane
        }
        ++k;
        return std::min(batch, left);
    }
    int join() {
        ctx->lane = 0;
        if (!forked) return UMX_OK;
        forked = false;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        return UMX_OK;
    }
    ~LaneLoop() { join(); }   // error paths: the side stream is still joined into the context's stream
};

TileGeom geom_of(const umx_hparams& hp, int H, int W) {
    TileGeom g;
    g.H = H; g.W = W;
    g.P = hp.imSize;
    g.margin = hp.imSize / 8;              // int(imSize/8), UnMicst1-5.py:694
    g.sub = g.P - 2 * g.margin;
    g.npr = (H + g.sub - 1) / g.sub;       // ceil, PartitionOfImage.py:49-50
    g.npc = (W + g.sub - 1) / g.sub;
    return g;
}

}  // namespace

// ---- accessors for umx_shard.hip (same shared object; hidden visibility)
static void (*g_destroy_hook)(umx_ctx*) = nullptr;
hipStream_t umx_internal_stream(umx_ctx* ctx) { return ctx->stream; }
int umx_internal_device(umx_ctx* ctx) { return ctx->device; }
void umx_internal_hp(const umx_ctx* ctx, umx_hparams* out) { *out = ctx->hp; }
int umx_internal_fail(umx_ctx* ctx, int code, const char* msg) { return fail(ctx, code, "%s", msg); }
void umx_internal_set_destroy_hook(void (*hook)(umx_ctx*)) { g_destroy_hook = hook; }

extern "C" {

const char* umx_version(void) { return "umx 0.1 (gfx950)"; }

int umx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int umx_device_mem_info(int device_ordinal, size_t* free_bytes, size_t* total_bytes) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, UMX_ERR_NO_DEVICE, "no HIP device available (libumx has no CPU fallback)");
    if (device_ordinal < 0 || device_ordinal >= ndev)
        return fail(nullptr, UMX_ERR_INVALID, "device ordinal %d out of range (%d devices)", device_ordinal, ndev);
    int prev = 0;
    HIP_TRY(nullptr, hipGetDevice(&prev));
    HIP_TRY(nullptr, hipSetDevice(device_ordinal));
    size_t f = 0, t = 0;
    hipError_t e = hipMemGetInfo(&f, &t);
    hipSetDevice(prev);
    if (e != hipSuccess) return fail(nullptr, UMX_ERR_HIP, "hipMemGetInfo failed: %s", hipGetErrorString(e));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return UMX_OK;
}

const char* umx_last_error(const umx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

void umx_test_double_to_half(const double* in, uint16_t* out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = double_to_half_rne(in[i]);
}

int umx_describe(const umx_hparams* hp, int* n_launches, double* flops_per_tile, double* executed_flops_per_tile) {
    std::string why;
    int rc = check_hp(hp, &why);
    if (rc) return fail(nullptr, rc, "%s", why.c_str());
    Builder b(*hp, nullptr);
    b.build();
    double f = 0, e = 0;
    for (auto& L : b.plan) { f += L.flops; e += L.exec_flops; }
    if (n_launches) *n_launches = (int)b.plan.size();
    if (flops_per_tile) *flops_per_tile = f;
    if (executed_flops_per_tile) *executed_flops_per_tile = e;
    return UMX_OK;
}

int umx_create(const umx_hparams* hp, const float* weight_blob, size_t blob_floats, int device_ordinal, int max_batch,
               umx_ctx** out) {
    umx_options o;
    memset(&o, 0, sizeof o);
    o.device_ordinal = device_ordinal;
    o.max_batch = max_batch;
    o.precision = UMX_PREC_DEFAULT;
    o.act_shift = -1;
    return umx_create_opts(hp, weight_blob, blob_floats, &o, out);
}

int umx_create_opts(const umx_hparams* hp, const float* weight_blob, size_t blob_floats, const umx_options* opts,
                    umx_ctx** out) {
    if (!out) return fail(nullptr, UMX_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!opts) return fail(nullptr, UMX_ERR_INVALID, "opts is NULL");
    if (opts->precision == UMX_PREC_DEFAULT && !getenv("UMX_PRECISION")) {
        // default = split precision where its planner covers every layer, else the exact-fp32 MFMA kernels (tiny layers
        // under big filters exceed the LDS image of conv_f16x3).  Both are HIP paths; there is no CPU fallback.
        umx_options o2 = *opts;
        o2.precision = UMX_PREC_F16X3;
        int rc = umx_create_opts(hp, weight_blob, blob_floats, &o2, out);
        if (rc != UMX_ERR_INVALID) return rc;
        const std::string first = g_err;
        o2.precision = UMX_PREC_F32;
        rc = umx_create_opts(hp, weight_blob, blob_floats, &o2, out);
        if (rc) g_err = first + "; fp32 kernels: " + g_err;
        return rc;
    }
    const int device_ordinal = opts->device_ordinal, max_batch = opts->max_batch;
    int precision = opts->precision;
    if (precision == UMX_PREC_DEFAULT) {
        const char* e = getenv("UMX_PRECISION");
        precision = (e && !strcmp(e, "f32")) ? UMX_PREC_F32 : UMX_PREC_F16X3;
    }
    if (precision != UMX_PREC_F32 && precision != UMX_PREC_F16X3)
        return fail(nullptr, UMX_ERR_INVALID, "unknown precision %d", precision);
    int act_shift = opts->act_shift;
    if (act_shift < 0) {
        const char* e = getenv("UMX_ACT_SHIFT");
        act_shift = e ? atoi(e) : 0;
    }
    if (act_shift < 0 || act_shift > 8) return fail(nullptr, UMX_ERR_INVALID, "act_shift must be in [0,8]");
    std::string why;
    int rc = check_hp(hp, &why);
    if (rc) return fail(nullptr, rc, "%s", why.c_str());
    if (!weight_blob) return fail(nullptr, UMX_ERR_INVALID, "weight_blob is NULL");
    if (max_batch < 1) return fail(nullptr, UMX_ERR_INVALID, "max_batch must be >= 1");
    const size_t need = blob_floats_needed(*hp);
    if (need != blob_floats)
        return fail(nullptr, UMX_ERR_BLOB, "weight blob has %zu floats, the graph needs %zu", blob_floats, need);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, UMX_ERR_NO_DEVICE, "no HIP device available (libumx has no CPU fallback)");
    if (device_ordinal < 0 || device_ordinal >= ndev)
        return fail(nullptr, UMX_ERR_INVALID, "device ordinal %d out of range (%d devices)", device_ordinal, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess)
        return fail(nullptr, UMX_ERR_HIP, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, UMX_ERR_NO_DEVICE, "device %d is %s; libumx is built for gfx950 only", device_ordinal,
                    prop.gcnArchName);

    std::unique_ptr<umx_ctx> ctx(new umx_ctx());
    ctx->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->hp = *hp;
    ctx->device = device_ordinal;
    ctx->max_batch = max_batch;
    ctx->precision = precision;
    ctx->act_shift = act_shift;
    umx_ctx* c = ctx.get();
    HIP_TRY(nullptr, hipSetDevice(device_ordinal));
    HIP_TRY(nullptr, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;

    Builder b(*hp, weight_blob);
    {   // opt-in: the append costs the transposed convolution more than the convolution gains
        const char* e = getenv("UMX_FOLD");
        b.fold_top_skip = precision == UMX_PREC_F16X3 && e && atoi(e) != 0;
    }
    b.build();
    if (b.pos != blob_floats) { umx_destroy(ctx.release()); return fail(nullptr, UMX_ERR_BLOB, "internal blob walk mismatch"); }
    c->plan = std::move(b.plan);
    c->bufs.resize(b.buf_floats.size());
    auto bail = [&](int code) { std::string m = c->err; umx_destroy(ctx.release()); g_err = m; return code; };
    const bool f16 = precision == UMX_PREC_F16X3;
    const int head_src = c->plan.back().g[0].src;   // the tensor the softmax head reads stays fp32
    for (size_t i = 0; i < c->bufs.size(); ++i) {
        Buffer& B = c->bufs[i];
        B.floats_per_tile = b.buf_floats[i];
        B.S = b.buf_geom[i].first;
        B.C = b.buf_geom[i].second;
        B.Cs = round_up(B.C, 8);
        B.as_f32 = !f16 || (int)i == head_src;
        // Octet-planar storage for the tensors whose consumers work on whole 16 x 16 (or 8 x 16) tiles of ONE image: a
        // workgroup reads its halo once per octet chunk, and in NHWC every chunk touches every 128-byte line of the
        // footprint (lu0.conv fetched 8.9 GB per launch through L2 for a 3 GB tensor); planar, a line belongs to one chunk.
        // Same bytes per image either way, so batches slice identically.  With 8 stored channels the two forms coincide.
        // Not for the output of a transposed convolution that runs one sub-pixel phase per workgroup (> 5 N-tiles): its
        // stores are every second pixel of a row, 16 bytes at a 32-byte stride in the planar form (lu2.convT +19 %).
        static const int planar_mode = getenv("UMX_PLANAR") ? atoi(getenv("UMX_PLANAR")) : 2;   // 0: NHWC everywhere, 1: planar wherever eligible, 2: the rule below
        bool phase_written = false;
        for (const Launch& Lp : c->plan)
            if (Lp.dst == (int)i && Lp.nphase == 4 && !(Lp.o_mul == 2 && Lp.ngroups == 1 && (Lp.Cout + 15) / 16 <= 5 && Lp.H >= 8 && Lp.W >= 16))
                phase_written = true;
        B.planar = planar_mode != 0 && !(planar_mode == 2 && phase_written) && !B.as_f32 && B.S >= 16 && B.Cs > 8 && !b.fold_top_skip &&
                   (size_t)B.S * B.S * 16 < (1u << 24);   // (the kernels form octet offsets with 24-bit multiplies)
        // (hi, lo) binary16 planes with Cs channels take 4*Cs bytes per pixel
        const size_t bytes_per_tile = B.as_f32 ? B.floats_per_tile * sizeof(float) : (size_t)B.S * B.S * B.Cs * 4;
        void* d = nullptr;
        if ((rc = dev_alloc(c, &d, bytes_per_tile * (size_t)max_batch))) return bail(rc);
        B.d = (float*)d;
    }
    {
        int lanes = opts->lanes;
        if (const char* e = getenv("UMX_LANES")) lanes = atoi(e);
        if (lanes == 0) lanes = 1;   // two lanes measured neutral on MI355X: off by default
        if (lanes < 1 || lanes > 2) { c->err = "lanes must be 1 or 2"; return bail(UMX_ERR_INVALID); }
        c->nlanes = lanes;
    }
    if (c->nlanes == 2) {
        if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
            c->err = "creating the second lane's stream/events failed";
            return bail(UMX_ERR_HIP);
        }
        c->bufs2 = c->bufs;
        for (size_t i = 0; i < c->bufs2.size(); ++i) {
            Buffer& B = c->bufs2[i];
            const size_t bytes_per_tile = B.as_f32 ? B.floats_per_tile * sizeof(float) : (size_t)B.S * B.S * B.Cs * 4;
            void* d = nullptr;
            if ((rc = dev_alloc(c, &d, bytes_per_tile * (size_t)max_batch))) return bail(rc);
            B.d = (float*)d;
        }
    }
    if (f16) {
        void* d = nullptr;
        if ((rc = dev_alloc(c, &d, c->bufs[0].floats_per_tile * sizeof(float) * (size_t)max_batch))) return bail(rc);
        c->d_tiles32 = (float*)d;
        if (c->nlanes == 2) {
            if ((rc = dev_alloc(c, &d, c->bufs[0].floats_per_tile * sizeof(float) * (size_t)max_batch))) return bail(rc);
            c->d_tiles32_2 = (float*)d;
        }
        if ((rc = dev_alloc(c, &d, 256))) return bail(rc);
        c->d_zeros = (uint4*)d;
        if ((rc = dev_alloc(c, &d, 256))) return bail(rc);
        c->d_flag = (int*)d;
        if (hipMemset(c->d_zeros, 0, 256) != hipSuccess || hipMemset(c->d_flag, 0, 256) != hipSuccess) {
            c->err = "hipMemset failed";
            return bail(UMX_ERR_HIP);
        }
    }
    for (auto& L : c->plan) {
        if ((rc = upload(c, L.pre_s, &L.d_pre_s)) || (rc = upload(c, L.pre_b, &L.d_pre_b)) ||
            (rc = upload(c, L.post_s, &L.d_post_s)) || (rc = upload(c, L.post_b, &L.d_post_b)))
            return bail(rc);
        if (L.head) {
            if ((rc = upload(c, L.head_w, &L.d_head_w))) return bail(rc);
            continue;
        }
        if (!conv_geometry(L, &why)) { c->err = L.name + ": " + why; return bail(UMX_ERR_INVALID); }
        if (f16) {
            if ((rc = plan_f16(c, L, act_shift, L.dst == head_src, &c->plan.back(), &why))) {
                if (c->err.empty() || !why.empty()) c->err = L.name + ": " + why;
                return bail(rc);
            }
            L.hcp.zeros = c->d_zeros;
            L.hcp.overflow_flag = c->d_flag;
            if (L.hcp.head_K > 0) {
                c->head_fused = true;
                const Launch& Hd = c->plan.back();
                L.flops += Hd.flops;
                L.bytes += 4.0 * Hd.H * Hd.W * Hd.head_K - 4.0 * L.outH * L.outW * L.Cout;   // probabilities out, no fp32 tensor
            }
            for (int ph = 0; ph < L.nphase; ++ph)
                for (int gi = 0; gi < L.ngroups; ++gi) std::vector<float>().swap(L.g[gi].packed[ph]);
            continue;
        }
        for (int ph = 0; ph < L.nphase; ++ph)
            for (int gi = 0; gi < L.ngroups; ++gi) {
                float* d = nullptr;
                if ((rc = upload(c, L.g[gi].packed[ph], &d))) return bail(rc);
                L.cp.ph[ph].w[gi] = d;
                std::vector<float>().swap(L.g[gi].packed[ph]);
            }
        L.cp.pre_s = L.d_pre_s; L.cp.pre_b = L.d_pre_b; L.cp.post_s = L.d_post_s; L.cp.post_b = L.d_post_b;
    }
    if (f16) c->split_launch.name = "input.split";
    *out = ctx.release();
    return UMX_OK;
}

int umx_precision_of(const umx_ctx* ctx) { return ctx ? ctx->precision : UMX_PREC_DEFAULT; }

void umx_destroy(umx_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (g_destroy_hook) g_destroy_hook(ctx);   // a communicator / buffers umx_shard_init attached to this context
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    for (auto& pe : ctx->pending) { hipEventDestroy(pe.a); hipEventDestroy(pe.b); }
    for (auto e : ctx->free_events) hipEventDestroy(e);
    for (void* d : ctx->allocs) hipFree(d);
    if (ctx->d_image) hipFree(ctx->d_image);
    if (ctx->d_probs) hipFree(ctx->d_probs);
    if (ctx->d_out) hipFree(ctx->d_out);
    if (ctx->d_io_tiles) hipFree(ctx->d_io_tiles);
    if (ctx->d_io_probs) hipFree(ctx->d_io_probs);
    if (ctx->up_stream) { hipStreamSynchronize(ctx->up_stream); hipStreamDestroy(ctx->up_stream); }
    if (ctx->dn_stream) { hipStreamSynchronize(ctx->dn_stream); hipStreamDestroy(ctx->dn_stream); }
    for (auto& h : ctx->hs) {
        for (auto e : h.events) hipEventDestroy(e);
        if (h.done) hipEventDestroy(h.done);
        if (h.flag_host) hipHostFree(h.flag_host);
        if (h.d_image) hipFree(h.d_image);
        if (h.d_probs) hipFree(h.d_probs);
        if (h.d_out) hipFree(h.d_out);
    }
    if (ctx->stream2) { hipStreamSynchronize(ctx->stream2); hipStreamDestroy(ctx->stream2); }
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int umx_set_stream(umx_ctx* ctx, void* hip_stream) {
    if (!ctx) return fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return UMX_OK;
}

int umx_synchronize(umx_ctx* ctx) {
    if (!ctx) return fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return check_range_flag(ctx);
}

int umx_forward_tiles_dev(umx_ctx* ctx, const float* tiles_dev, int n, float* probs_dev) {
    if (!ctx) return fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    if (n < 0 || (n > 0 && (!tiles_dev || !probs_dev))) return fail(ctx, UMX_ERR_INVALID, "bad tiles/probs/n");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t P = ctx->hp.imSize;
    const size_t tile_f = P * P * ctx->hp.nChannels, prob_f = P * P * ctx->hp.nClasses;
    LaneLoop ll(ctx, n);
    for (int i = 0, nb; i < n; i += nb) {
        nb = ll.next(n - i);
        int rc = run_unet(ctx, tiles_dev + (size_t)i * tile_f, nb, probs_dev + (size_t)i * prob_f);
        if (rc) return rc;
    }
    return ll.join();
}

int umx_forward_tiles(umx_ctx* ctx, const float* tiles_host, int n, float* probs_host) {
    if (!ctx) return fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    if (n < 0 || (n > 0 && (!tiles_host || !probs_host))) return fail(ctx, UMX_ERR_INVALID, "bad tiles/probs/n");
    if (n == 0) return UMX_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t P = ctx->hp.imSize;
    const size_t tile_b = P * P * ctx->hp.nChannels * sizeof(float), prob_b = P * P * ctx->hp.nClasses * sizeof(float);
    int rc;
    if ((rc = grow(ctx, (void**)&ctx->d_io_tiles, &ctx->io_tiles_cap, tile_b * n))) return rc;
    if ((rc = grow(ctx, (void**)&ctx->d_io_probs, &ctx->io_probs_cap, prob_b * n))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_io_tiles, tiles_host, tile_b * n, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = umx_forward_tiles_dev(ctx, ctx->d_io_tiles, n, ctx->d_io_probs))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(probs_host, ctx->d_io_probs, prob_b * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return check_range_flag(ctx);
}

int umx_tile_grid(const umx_ctx* ctx, int H, int W, int* patch_rows, int* patch_cols, int* padded_rows,
                  int* padded_cols) {
    if (!ctx || H < 1 || W < 1) return fail(nullptr, UMX_ERR_INVALID, "bad ctx/H/W");
    const TileGeom g = geom_of(ctx->hp, H, W);
    if (patch_rows) *patch_rows = g.npr;
    if (patch_cols) *patch_cols = g.npc;
    if (padded_rows) *padded_rows = g.npr * g.sub + 2 * g.margin;
    if (padded_cols) *padded_cols = g.npc * g.sub + 2 * g.margin;
    return UMX_OK;
}

// tiles [t0, t1) of the slide (row-major tile index) -> probs_dev (tile t0 first): gather + normalise + UNet, in launch
// groups of <= max_batch tiles
static int tiles_range(umx_ctx* ctx, const double* image_dev, int C_img, const TileGeom& g, int band_row0, int band_rows,
                       double mean, double stdv, int t0, int t1, float* probs_dev) {
    const size_t prob_f = (size_t)g.P * g.P * ctx->hp.nClasses;
    if (ctx->site_gather < 0) ctx->site_gather = site_of(ctx, "pi2d.gather_normalise", "gather_normalise");
    const bool direct16 = ctx->precision == UMX_PREC_F16X3 && ctx->hp.nChannels <= 8 && ctx->bufs[0].Cs == 8;
    LaneLoop ll(ctx, t1 - t0);
    for (int t = t0, nb; t < t1; t += nb) {
        nb = ll.next(t1 - t);
        float* const tiles32 = ctx->precision == UMX_PREC_F16X3 ? (ctx->lane ? ctx->d_tiles32_2 : ctx->d_tiles32)
                                                                : cur_bufs(ctx)[0].d;
        {
            ProfScope ps(ctx, ctx->site_gather, 0.0,
                         (double)nb * g.P * g.P * (8.0 + (direct16 ? 32.0 : 4.0 * ctx->hp.nChannels)));
            if (direct16) {   // gather + normalise + (hi, lo) split in one pass
                const Buffer& b0 = cur_bufs(ctx)[0];
                HIP_TRY(ctx, launch_gather_split(image_dev, C_img, band_row0, band_rows, g, ctx->hp.nChannels, mean, stdv, t, nb,
                                                 std::ldexp(1.f, ctx->act_shift), hi_of(b0), lo_of(b0, nb), run_stream(ctx)));
            } else {
                HIP_TRY(ctx, launch_gather_normalise(image_dev, C_img, band_row0, band_rows, g, ctx->hp.nChannels, mean, stdv, t,
                                                     nb, tiles32, run_stream(ctx)));
            }
        }
        int rc = run_unet(ctx, direct16 ? nullptr : tiles32, nb, probs_dev + (size_t)(t - t0) * prob_f);
        if (rc) return rc;
    }
    return ll.join();
}

int umx_band_tiles_dev(umx_ctx* ctx, const double* image_dev, int C_img, int H, int W, int band_row0, int band_rows,
                       double mean, double stdv, int pr0, int pr1, float* probs_dev) {
    if (!ctx) return fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    if (!image_dev || !probs_dev || H < 1 || W < 1) return fail(ctx, UMX_ERR_INVALID, "bad image/probs/H/W");
    if (C_img != 1 && C_img != ctx->hp.nChannels)
        return fail(ctx, UMX_ERR_INVALID, "image has %d channels, model wants 1 or %d", C_img, ctx->hp.nChannels);
    if (!(stdv != 0.0)) return fail(ctx, UMX_ERR_INVALID, "std must be non-zero");
    const TileGeom g = geom_of(ctx->hp, H, W);
    if (pr0 < 0 || pr1 > g.npr || pr0 > pr1) return fail(ctx, UMX_ERR_INVALID, "patch rows [%d,%d) outside [0,%d)", pr0, pr1, g.npr);
    if (pr0 == pr1) return UMX_OK;
    // image rows the patch rows touch: [pr0*sub - m, (pr1-1)*sub + P - m) clipped to the image
    const int need0 = std::max(0, pr0 * g.sub - g.margin), need1 = std::min(H, (pr1 - 1) * g.sub + g.P - g.margin);
    if (band_row0 < 0 || band_rows < 0 || (need1 > need0 && (band_row0 > need0 || band_row0 + band_rows < need1)))
        return fail(ctx, UMX_ERR_INVALID, "band rows [%d,%d) do not cover the rows [%d,%d) the patch rows need",
                    band_row0, band_row0 + band_rows, need0, need1);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return tiles_range(ctx, image_dev, C_img, g, band_row0, band_rows, mean, stdv, pr0 * g.npc, pr1 * g.npc, probs_dev);
}

int umx_stitch_dev(umx_ctx* ctx, const float* probs_dev, int tpr0, int tpr1, int H, int W, int mode, int stitch, int y0,
                   int y1, void* out_dev) {
    if (!ctx) return fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    if (!probs_dev || !out_dev || H < 1 || W < 1) return fail(ctx, UMX_ERR_INVALID, "bad probs/out/H/W");
    if (mode != UMX_MODE_ACCUMULATE && mode != UMX_MODE_REPLACE) return fail(ctx, UMX_ERR_INVALID, "bad mode %d", mode);
    if (stitch != UMX_STITCH_FP16_COMPAT && stitch != UMX_STITCH_FP32) return fail(ctx, UMX_ERR_INVALID, "bad stitch %d", stitch);
    const TileGeom g = geom_of(ctx->hp, H, W);
    if (y0 < 0 || y1 > H || y0 > y1) return fail(ctx, UMX_ERR_INVALID, "rows [%d,%d) outside the image", y0, y1);
    if (y0 == y1) return UMX_OK;
    // patch rows touching image rows [y0,y1): padded rows R = y + m; pr*sub <= R < pr*sub + P
    const int R0 = y0 + g.margin, R1 = y1 - 1 + g.margin;
    const int need_lo = (R0 - g.P + 1 <= 0) ? 0 : (R0 - g.P + g.sub) / g.sub;  // ceil((R0-P+1)/sub)
    const int need_hi = std::min(g.npr - 1, R1 / g.sub);
    if (tpr0 > need_lo || tpr1 <= need_hi || tpr0 < 0 || tpr1 > g.npr)
        return fail(ctx, UMX_ERR_INVALID, "tile rows [%d,%d) do not cover the patch rows [%d,%d] touching image rows [%d,%d)",
                    tpr0, tpr1, need_lo, need_hi, y0, y1);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->site_stitch < 0) ctx->site_stitch = site_of(ctx, "pi2d.stitch", "stitch");
    const int K = ctx->hp.nClasses;
    ProfScope ps(ctx, ctx->site_stitch, 0.0,
                 (double)(y1 - y0) * W * K * (4.0 * ((double)g.P / g.sub) * ((double)g.P / g.sub) + (stitch == 0 ? 2.0 : 4.0)));
    HIP_TRY(ctx, launch_stitch(probs_dev, tpr0, tpr1, g, K, mode, stitch, y0, y1, out_dev, ctx->stream));
    return UMX_OK;
}

int umx_infer_image_dev(umx_ctx* ctx, const double* image_dev, int C_img, int H, int W, double mean, double stdv,
                        int mode, int stitch, void* out_dev) {
    if (!ctx) return fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    if (H < 1 || W < 1) return fail(ctx, UMX_ERR_INVALID, "bad H/W");
    const TileGeom g = geom_of(ctx->hp, H, W);
    const size_t prob_b = (size_t)g.npr * g.npc * g.P * g.P * ctx->hp.nClasses * sizeof(float);
    int rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if ((rc = grow(ctx, (void**)&ctx->d_probs, &ctx->probs_cap, prob_b))) return rc;
    if ((rc = umx_band_tiles_dev(ctx, image_dev, C_img, H, W, 0, H, mean, stdv, 0, g.npr, ctx->d_probs))) return rc;
    return umx_stitch_dev(ctx, ctx->d_probs, 0, g.npr, H, W, mode, stitch, 0, H, out_dev);
}

// ---- host entry points.  The reference hands host arrays across its seam (UnMicst1-5.py:687-710); here the slide goes up
// and the probability stack comes down in row slabs on two copy streams while the tile kernels of the neighbouring slabs
// run: slab s = patch rows [cut[s], cut[s+1]); its upload covers the image rows its tiles read that are not on the device
// yet, its download the image rows no later patch row touches.  With pinned host buffers the transfers are true DMA and
// all but the first upload and the last download ride under compute; with pageable buffers HIP stages them (still correct).
// src_bits: 0 = float64 planes (what singleImageInference receives), 8 / 16 = raw integer planes (the driver's file
// contents; im2double and, with `rescale`, rescale_intensity run on the device).  out_u8: the driver's uint8 planes
// instead of the stitch result.  A rescale needs the plane's (min, max) before the first tile: the upload then runs
// ahead of compute (min/max reduced slab by slab as the rows arrive) and only the download is hidden.
static int host_wait(umx_ctx* ctx, int slot) {
    if (slot < 0 || slot > 1) return fail(ctx, UMX_ERR_INVALID, "slot must be 0 or 1");
    umx_ctx::HostSlot& hs = ctx->hs[slot];
    if (!hs.busy) return UMX_OK;
    hs.busy = false;
    HIP_TRY(ctx, hipEventSynchronize(hs.done));
    if (*hs.flag_host) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return check_range_flag(ctx);
    }
    return UMX_OK;
}

static int host_submit(umx_ctx* ctx, int slot, const void* src, int src_bits, int C_img, int H, int W, int rescale, double mean,
                       double stdv, int mode, int stitch, int out_u8, void* out_host) {
    if (slot < 0 || slot > 1) return fail(ctx, UMX_ERR_INVALID, "slot must be 0 or 1");
    umx_ctx::HostSlot& hs = ctx->hs[slot];
    if (hs.busy) return fail(ctx, UMX_ERR_INVALID, "slot %d still holds a submitted call: wait for it first", slot);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!hs.done) {
        HIP_TRY(ctx, hipEventCreateWithFlags(&hs.done, hipEventDisableTiming));
        HIP_TRY(ctx, hipHostMalloc((void**)&hs.flag_host, 64, hipHostMallocDefault));
    }
    const TileGeom g = geom_of(ctx->hp, H, W);
    const size_t plane = (size_t)H * W, K = ctx->hp.nClasses;
    const size_t in_b = src_bits ? (size_t)(src_bits / 8) : sizeof(double);
    const size_t oel = stitch == UMX_STITCH_FP32 ? 4 : 2;
    const size_t pm_b = K * plane * oel, u8_b = out_u8 ? K * plane : 0;
    const size_t raw_off = (pm_b + u8_b + 255) & ~(size_t)255;
    const size_t raw_b = src_bits ? plane * C_img * in_b : 0;
    const size_t mm_off = (raw_off + raw_b + 255) & ~(size_t)255;
    int rc;
    if ((rc = grow(ctx, (void**)&hs.d_image, &hs.image_cap, plane * C_img * sizeof(double)))) return rc;
    if ((rc = grow(ctx, &hs.d_out, &hs.out_cap, mm_off + 64 * (size_t)C_img))) return rc;
    if ((rc = grow(ctx, (void**)&hs.d_probs, &hs.probs_cap, (size_t)g.npr * g.npc * g.P * g.P * K * sizeof(float)))) return rc;
    unsigned char* const base = (unsigned char*)hs.d_out;
    if (!ctx->up_stream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->dn_stream, hipStreamNonBlocking));
    }
    // slabs = the launch groups of the tile loop (equal groups of <= max_batch tiles, exactly what umx_infer_image_dev
    // runs), so that pipelining the transfers does not change a single kernel launch; UMX_HOST_SLABS=1: no overlap
    const int T = g.npr * g.npc;
    int S = std::max(1, (T + ctx->max_batch - 1) / ctx->max_batch);
    if (const char* e = getenv("UMX_HOST_SLABS")) S = std::max(1, std::min(atoi(e), S));
    while ((int)hs.events.size() < 2 * S + 1) {
        hipEvent_t ev;
        HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hs.events.push_back(ev);
    }
    hipEvent_t* const ev_up = hs.events.data();
    hipEvent_t* const ev_dn = hs.events.data() + S;
    hipEvent_t ev_start = hs.events[2 * S];
    (void)ev_start;   // (a slot's buffers are private to it and free again once host_wait has returned: no extra ordering)
    std::vector<int> tcut(S + 1), cut(S + 1);   // tile cuts; cut[i] = patch rows COMPLETE after slab i-1 (cut[S] = npr)
    for (int i = 0; i <= S; ++i) {
        tcut[i] = (int)((long long)T * i / S);
        cut[i] = tcut[i] / g.npc;
    }
    auto rows_needed = [&](int pr1) { return std::min(H, (pr1 - 1) * g.sub + g.P - g.margin); };
    unsigned* const mm = (unsigned*)(base + mm_off);
    unsigned char* const d_raw = base + raw_off;
    auto upload = [&](int r0, int r1) -> int {   // image rows [r0, r1) of every plane
        for (int c = 0; c < C_img && r1 > r0; ++c) {
            const size_t off = ((size_t)c * H + r0) * W * in_b, n = (size_t)(r1 - r0) * W * in_b;
            void* const dst = src_bits ? (void*)(d_raw + off) : (void*)((unsigned char*)hs.d_image + off);
            HIP_TRY(ctx, hipMemcpyAsync(dst, (const unsigned char*)src + off, n, hipMemcpyHostToDevice, ctx->up_stream));
        }
        return UMX_OK;
    };
    auto convert = [&](int r0, int r1) -> int {   // raw rows -> float64 rows (im2double [+ rescale])
        for (int c = 0; c < C_img && src_bits && r1 > r0; ++c) {
            const size_t e0 = ((size_t)c * H + r0) * W;
            HIP_TRY(ctx, launch_raw_convert(d_raw + e0 * in_b, src_bits, (size_t)(r1 - r0) * W, rescale, mm + 16 * c,
                                            hs.d_image + e0, ctx->stream));
        }
        return UMX_OK;
    };
    int up_done = 0;
    if (src_bits) {
        for (int c = 0; c < C_img; ++c) HIP_TRY(ctx, launch_minmax_init(mm + 16 * c, ctx->stream));
        if (rescale) {   // whole planes first: min / max per plane, reduced as the slabs arrive
            for (int s = 0; s < S; ++s) {
                const int r1 = s == S - 1 ? H : rows_needed((tcut[s + 1] - 1) / g.npc + 1);
                if (r1 <= up_done) continue;
                if ((rc = upload(up_done, r1))) return rc;
                HIP_TRY(ctx, hipEventRecord(ev_up[s], ctx->up_stream));
                HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ev_up[s], 0));
                for (int c = 0; c < C_img && r1 > up_done; ++c)
                    HIP_TRY(ctx, launch_minmax(d_raw + ((size_t)c * H + up_done) * W * in_b, src_bits, (size_t)(r1 - up_done) * W,
                                               mm + 16 * c, ctx->stream));
                up_done = r1;
            }
            if ((rc = convert(0, H))) return rc;
        }
    }
    int y_done = 0;
    for (int s = 0; s < S; ++s) {
        // rows the tiles of this slab read: up to the last patch row it touches
        const int r1 = s == S - 1 ? H : rows_needed((tcut[s + 1] - 1) / g.npc + 1);
        if (r1 > up_done) {
            if ((rc = upload(up_done, r1))) return rc;
            HIP_TRY(ctx, hipEventRecord(ev_up[s], ctx->up_stream));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ev_up[s], 0));
            if ((rc = convert(up_done, r1))) return rc;
            up_done = r1;
        }
        if (tcut[s + 1] > tcut[s]) {
            float* const pr = hs.d_probs + (size_t)tcut[s] * g.P * g.P * K;
            if ((rc = tiles_range(ctx, hs.d_image, C_img, g, 0, H, mean, stdv, tcut[s], tcut[s + 1], pr))) return rc;
        }
        // image rows no later tile touches: below the first incomplete patch row
        const int y1 = s == S - 1 ? H : std::max(y_done, std::min(H, cut[s + 1] * g.sub - g.margin));
        if (y1 > y_done && cut[s + 1] > 0) {
            // the stitch writes a compact slab [K][rows][W]; slabs sit one after the other in the device buffer
            const size_t rows = (size_t)(y1 - y_done), slab_e = K * (size_t)y_done * W;
            unsigned char* const d_slab = base + slab_e * oel;
            if ((rc = umx_stitch_dev(ctx, hs.d_probs, 0, cut[s + 1], H, W, mode, stitch, y_done, y1, d_slab))) return rc;
            if (out_u8) HIP_TRY(ctx, launch_half_to_u8(d_slab, K * rows * W, base + pm_b + slab_e, ctx->stream));
            HIP_TRY(ctx, hipEventRecord(ev_dn[s], ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->dn_stream, ev_dn[s], 0));
            const size_t el = out_u8 ? 1 : oel;
            const unsigned char* const dsrc = out_u8 ? base + pm_b + slab_e : d_slab;
            for (size_t k = 0; k < K; ++k)
                HIP_TRY(ctx, hipMemcpyAsync((unsigned char*)out_host + (k * plane + (size_t)y_done * W) * el,
                                            dsrc + k * rows * W * el, rows * W * el, hipMemcpyDeviceToHost, ctx->dn_stream));
            y_done = y1;
        }
    }
    // the range flag of the split-precision path rides down behind the last planes; `done` then says the call is complete
    // (every upload precedes a kernel that precedes a download on the download stream)
    if (ctx->d_flag) {
        hipEvent_t ev_f = hs.events[2 * S];
        HIP_TRY(ctx, hipEventRecord(ev_f, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->dn_stream, ev_f, 0));
        HIP_TRY(ctx, hipMemcpyAsync(hs.flag_host, ctx->d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->dn_stream));
    } else {
        *hs.flag_host = 0;
    }
    HIP_TRY(ctx, hipEventRecord(hs.done, ctx->dn_stream));
    hs.busy = true;
    return UMX_OK;
}

static int infer_host(umx_ctx* ctx, const void* src, int src_bits, int C_img, int H, int W, int rescale, double mean,
                      double stdv, int mode, int stitch, int out_u8, void* out_host) {
    int rc = host_submit(ctx, 0, src, src_bits, C_img, H, W, rescale, mean, stdv, mode, stitch, out_u8, out_host);
    if (rc) return rc;
    return host_wait(ctx, 0);
}



int umx_infer_image(umx_ctx* ctx, const double* image_host, int C_img, int H, int W, double mean, double stdv, int mode,
                    int stitch, void* out_host) {
    if (!ctx) return fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    if (!image_host || !out_host || H < 1 || W < 1 || C_img < 1) return fail(ctx, UMX_ERR_INVALID, "bad image/out/H/W");
    if (C_img != 1 && C_img != ctx->hp.nChannels)
        return fail(ctx, UMX_ERR_INVALID, "image has %d channels, model wants 1 or %d", C_img, ctx->hp.nChannels);
    if (!(stdv != 0.0)) return fail(ctx, UMX_ERR_INVALID, "std must be non-zero");
    if (mode != UMX_MODE_ACCUMULATE && mode != UMX_MODE_REPLACE) return fail(ctx, UMX_ERR_INVALID, "bad mode %d", mode);
    if (stitch != UMX_STITCH_FP16_COMPAT && stitch != UMX_STITCH_FP32) return fail(ctx, UMX_ERR_INVALID, "bad stitch %d", stitch);
    return infer_host(ctx, image_host, 0, C_img, H, W, 0, mean, stdv, mode, stitch, 0, out_host);
}

int umx_infer_image_raw(umx_ctx* ctx, const void* raw_host, int bits, int C_img, int H, int W, int rescale, double mean,
                        double stdv, int mode, uint8_t* out_host) {
    if (!ctx) return fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    if (!raw_host || !out_host || H < 1 || W < 1 || C_img < 1) return fail(ctx, UMX_ERR_INVALID, "bad image/out/H/W");
    if (bits != 8 && bits != 16) return fail(ctx, UMX_ERR_INVALID, "raw planes must be uint8 or uint16 (bits = %d)", bits);
    if (C_img != 1 && C_img != ctx->hp.nChannels)
        return fail(ctx, UMX_ERR_INVALID, "image has %d channels, model wants 1 or %d", C_img, ctx->hp.nChannels);
    if (!(stdv != 0.0)) return fail(ctx, UMX_ERR_INVALID, "std must be non-zero");
    if (mode != UMX_MODE_ACCUMULATE && mode != UMX_MODE_REPLACE) return fail(ctx, UMX_ERR_INVALID, "bad mode %d", mode);
    return infer_host(ctx, raw_host, bits, C_img, H, W, rescale, mean, stdv, mode, UMX_STITCH_FP16_COMPAT, 1, out_host);
}

// ---- the drivers' whole recipe at --scalingFactor != 1 on the device (reference UnMicst1-5.py:807-821,845-854):
// raw planes -> im2double -> resize to (int(H*sf), int(W*sf)) -> [rescale_intensity((min, max) -> (0, 0.983))] -> inference
// -> np.uint8(255 * pm) -> resize back to (H, W) -> np.uint8(255 * .).  One resize = skimage.transform.resize's defaults
// (umx_kernels.hip).  Synchronous; the planes are small next to the tile work, so nothing is pipelined here.
static int resize_plane(umx_ctx* ctx, const double* src, int H, int W, int h, int w, double* tmpA, double* tmpB, double* wdev,
                        unsigned long long* mm64, double* dst, unsigned char* dst_u8) {
    const double* cur = src;
    const double fy = (double)H / h, fx = (double)W / w;
    const double sig[2] = {std::max(0.0, (fy - 1.0) / 2.0), std::max(0.0, (fx - 1.0) / 2.0)};
    if (h < H || w < W) {   // anti-aliasing Gaussian, axis by axis (scipy.ndimage.gaussian_filter: axis 0 first)
        double* bufs[2] = {tmpA, tmpB};
        int which = 0;
        for (int axis = 0; axis < 2; ++axis) {
            if (!(sig[axis] > 1e-15)) continue;   // scipy skips axes with sigma <= 1e-15
            const int radius = (int)(4.0 * sig[axis] + 0.5);
            std::vector<double> wts((size_t)radius + 1);
            double sum = 0.0;
            std::vector<double> full(2 * (size_t)radius + 1);
            for (int x = -radius; x <= radius; ++x) full[x + radius] = std::exp(-0.5 / (sig[axis] * sig[axis]) * (double)x * (double)x);
            for (double v : full) sum += v;
            for (int j = 0; j <= radius; ++j) wts[j] = full[radius + j] / sum;
            if (radius + 1 > 4096) return fail(ctx, UMX_ERR_INVALID, "scaling factor too small for the resize kernel");
            HIP_TRY(ctx, hipMemcpyAsync(wdev + axis * 4096, wts.data(), wts.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // (wts is a stack-lifetime host buffer)
            HIP_TRY(ctx, launch_gauss1d(cur, bufs[which], H, W, axis, radius, wdev + axis * 4096, ctx->stream));
            cur = bufs[which];
            which ^= 1;
        }
    }
    HIP_TRY(ctx, launch_minmax_f64(cur, (size_t)H * W, mm64, ctx->stream));   // resize clips to the (filtered) input's range
    HIP_TRY(ctx, launch_zoom1(cur, H, W, h, w, mm64, dst, dst_u8, ctx->stream));
    return UMX_OK;
}

// outlier < 0: rescale (if set) to the plane's (min, max); outlier in [0, 100]: to (min, np.percentile(plane, outlier))
static int infer_raw_scaled_impl(umx_ctx* ctx, const void* raw_host, int bits, int C_img, int H, int W, double scaling, int rescale,
                                 double outlier, double mean, double stdv, int mode, uint8_t* out_host) {
    if (!ctx) return fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    if (!raw_host || !out_host || H < 1 || W < 1 || C_img < 1) return fail(ctx, UMX_ERR_INVALID, "bad image/out/H/W");
    if (bits != 8 && bits != 16) return fail(ctx, UMX_ERR_INVALID, "raw planes must be uint8 or uint16 (bits = %d)", bits);
    if (C_img != 1 && C_img != ctx->hp.nChannels)
        return fail(ctx, UMX_ERR_INVALID, "image has %d channels, model wants 1 or %d", C_img, ctx->hp.nChannels);
    if (!(stdv != 0.0)) return fail(ctx, UMX_ERR_INVALID, "std must be non-zero");
    if (!(scaling > 0.0)) return fail(ctx, UMX_ERR_INVALID, "scaling factor must be positive");
    const int h = (int)((double)H * scaling), w = (int)((double)W * scaling);   // int(float(I.shape[0]) * float(sf))
    if (h < 1 || w < 1) return fail(ctx, UMX_ERR_INVALID, "scaled image is empty");
    const bool same = h == H && w == W;   // resize(I, I.shape) leaves im2double(I): the pipelined path does all but the percentile
    if (same && outlier < 0) return umx_infer_image_raw(ctx, raw_host, bits, C_img, H, W, rescale, mean, stdv, mode, out_host);
    if (mode != UMX_MODE_ACCUMULATE && mode != UMX_MODE_REPLACE) return fail(ctx, UMX_ERR_INVALID, "bad mode %d", mode);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t big = (size_t)std::max(H, h) * std::max(W, w), plane = (size_t)H * W, sp = (size_t)h * w, K = ctx->hp.nClasses;
    const size_t in_b = bits / 8;
    // scratch: [raw upload | 3 float64 work planes of the larger size | scaled input planes | fp16 result | u8 out | weights | mm]
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_raw = take(plane * C_img * in_b), o_a = take(big * 8), o_b = take(big * 8), o_c = take(big * 8);
    const size_t o_in = take(sp * C_img * 8), o_pm = take(K * sp * 2), o_u8 = take(K * plane), o_w = take(2 * 4096 * 8), o_mm = take(256);
    const size_t o_sel = take(64 + 512 * 4);   // radix-selection state + histograms of the percentile
    int rc;
    umx_ctx::HostSlot& hs = ctx->hs[0];
    if (hs.busy) return fail(ctx, UMX_ERR_INVALID, "slot 0 still holds a submitted call: wait for it first");
    if ((rc = grow(ctx, &hs.d_out, &hs.out_cap, off))) return rc;
    unsigned char* const base = (unsigned char*)hs.d_out;
    double *A = (double*)(base + o_a), *B = (double*)(base + o_b), *Cw = (double*)(base + o_c), *din = (double*)(base + o_in);
    double* const wdev = (double*)(base + o_w);
    unsigned long long* const mm64 = (unsigned long long*)(base + o_mm);
    unsigned* const mm32 = (unsigned*)(base + o_mm + 64);
    HIP_TRY(ctx, hipMemcpyAsync(base + o_raw, raw_host, plane * C_img * in_b, hipMemcpyHostToDevice, ctx->stream));
    for (int c = 0; c < C_img; ++c) {
        HIP_TRY(ctx, launch_minmax_init(mm32, ctx->stream));
        HIP_TRY(ctx, launch_raw_convert(base + o_raw + (size_t)c * plane * in_b, bits, plane, 0, mm32, A, ctx->stream));   // im2double
        if (same) HIP_TRY(ctx, hipMemcpyAsync(din + (size_t)c * sp, A, sp * 8, hipMemcpyDeviceToDevice, ctx->stream));
        else if ((rc = resize_plane(ctx, A, H, W, h, w, B, Cw, wdev, mm64, din + (size_t)c * sp, nullptr))) return rc;
        if (rescale) {   // rescale_intensity(I, (min, max | percentile), (0, 0.983)) of the RESIZED plane (UnMicst1-5.py:817-821)
            HIP_TRY(ctx, launch_minmax_f64(din + (size_t)c * sp, sp, mm64, ctx->stream));
            if (outlier >= 0)
                HIP_TRY(ctx, launch_percentile_f64(din + (size_t)c * sp, sp, outlier, (unsigned long long*)(base + o_sel),
                                                   (unsigned*)(base + o_sel + 64), mm64, ctx->stream));
            HIP_TRY(ctx, launch_rescale_f64(din + (size_t)c * sp, sp, mm64, ctx->stream));
        }
    }
    if ((rc = umx_infer_image_dev(ctx, din, C_img, h, w, mean, stdv, mode, UMX_STITCH_FP16_COMPAT, base + o_pm))) return rc;
    for (size_t k = 0; k < K; ++k) {
        if (same) {   // resize of a uint8 plane to its own shape and back through np.uint8(255 * .): the plane itself
            HIP_TRY(ctx, launch_half_to_u8(base + o_pm + k * sp * 2, sp, base + o_u8 + k * plane, ctx->stream));
            continue;
        }
        HIP_TRY(ctx, launch_half_to_u8_f64(base + o_pm + k * sp * 2, sp, A, ctx->stream));   // np.uint8(255 * pm) as float u8/255
        if ((rc = resize_plane(ctx, A, h, w, H, W, B, Cw, wdev, mm64, nullptr, base + o_u8 + k * plane))) return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(out_host, base + o_u8, K * plane, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return check_range_flag(ctx);
}

int umx_infer_image_raw_scaled(umx_ctx* ctx, const void* raw_host, int bits, int C_img, int H, int W, double scaling, int rescale,
                               double mean, double stdv, int mode, uint8_t* out_host) {
    return infer_raw_scaled_impl(ctx, raw_host, bits, C_img, H, W, scaling, rescale, -1.0, mean, stdv, mode, out_host);
}

int umx_infer_image_raw_outlier(umx_ctx* ctx, const void* raw_host, int bits, int C_img, int H, int W, double scaling, double outlier,
                                double mean, double stdv, int mode, uint8_t* out_host) {
    if (!(outlier >= 0.0 && outlier <= 100.0)) return fail(ctx, UMX_ERR_INVALID, "outlier percentile must be in [0, 100]");
    return infer_raw_scaled_impl(ctx, raw_host, bits, C_img, H, W, scaling, 1, outlier, mean, stdv, mode, out_host);
}

int umx_infer_image_raw_submit(umx_ctx* ctx, int slot, const void* raw_host, int bits, int C_img, int H, int W, int rescale,
                               double mean, double stdv, int mode, uint8_t* out_host) {
    if (!ctx) return fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    if (!raw_host || !out_host || H < 1 || W < 1 || C_img < 1) return fail(ctx, UMX_ERR_INVALID, "bad image/out/H/W");
    if (bits != 8 && bits != 16) return fail(ctx, UMX_ERR_INVALID, "raw planes must be uint8 or uint16 (bits = %d)", bits);
    if (C_img != 1 && C_img != ctx->hp.nChannels)
        return fail(ctx, UMX_ERR_INVALID, "image has %d channels, model wants 1 or %d", C_img, ctx->hp.nChannels);
    if (!(stdv != 0.0)) return fail(ctx, UMX_ERR_INVALID, "std must be non-zero");
    if (mode != UMX_MODE_ACCUMULATE && mode != UMX_MODE_REPLACE) return fail(ctx, UMX_ERR_INVALID, "bad mode %d", mode);
    return host_submit(ctx, slot, raw_host, bits, C_img, H, W, rescale, mean, stdv, mode, UMX_STITCH_FP16_COMPAT, 1, out_host);
}

int umx_infer_image_wait(umx_ctx* ctx, int slot) {
    if (!ctx) return fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    return host_wait(ctx, slot);
}

// ---- TIFF strip / tile decoders for the drivers' file reader (unmicst_amd/tiffio.py).  Host code: file decoding is not on
// the GPU path; it lives in the library so that the reader does not depend on tifffile / imagecodecs (absent here), which
// the reference uses at UnMicst1-5.py:794-797.  Both return the number of bytes written, or -1 on a malformed stream.
long long umx_tiff_lzw_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    // TIFF 6.0 LZW: codes packed MSB first, 9..12 bits, ClearCode 256, EndOfInformation 257, "early change" (the width
    // grows one code before the table fills a power of two), as written by libtiff, Bio-Formats and tifffile
    static thread_local uint16_t prefix[4096];
    static thread_local uint8_t suffix[4096];
    static thread_local uint16_t length[4096];
    if (!src || !dst) return -1;
    for (int i = 0; i < 256; ++i) { prefix[i] = 0; suffix[i] = (uint8_t)i; length[i] = 1; }
    int bits = 9, next = 258, prev = -1;
    uint32_t acc = 0;
    int nacc = 0;
    size_t out = 0, ip = 0;
    for (;;) {
        while (nacc < bits && ip < n) { acc = (acc << 8) | src[ip++]; nacc += 8; }
        if (nacc < bits) break;                       // stream ended without EOI: accept what was decoded
        const int code = (int)((acc >> (nacc - bits)) & ((1u << bits) - 1));
        nacc -= bits;
        if (code == 256) { bits = 9; next = 258; prev = -1; continue; }
        if (code == 257) break;
        if (prev < 0) {
            if (code > 255) return -1;
            if (out < cap) dst[out] = (uint8_t)code;
            ++out;
            prev = code;
            continue;
        }
        int entry;
        uint8_t first;
        if (code < next) {
            entry = code;
        } else if (code == next) {
            entry = prev;                             // KwKwK: the string of prev + its own first character
        } else {
            return -1;
        }
        // first character of `entry`'s string
        int e = entry;
        while (length[e] > 1) e = prefix[e];
        first = suffix[e];
        const size_t len = length[entry] + (code == next ? 1u : 0u);
        if (out + len <= cap) {
            size_t pos = out + length[entry];
            e = entry;
            while (true) {
                dst[--pos] = suffix[e];
                if (length[e] == 1) break;
                e = prefix[e];
            }
            if (code == next) dst[out + len - 1] = first;
        }
        out += len;
        if (next < 4096) {
            prefix[next] = (uint16_t)prev;
            suffix[next] = first;
            length[next] = (uint16_t)(length[prev] + 1);
            ++next;
            if (next >= (1 << bits) - 1 && bits < 12) ++bits;
        }
        prev = code;
    }
    return out <= cap ? (long long)out : -1;
}

long long umx_tiff_packbits_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    if (!src || !dst) return -1;
    size_t ip = 0, out = 0;
    while (ip < n) {
        const int8_t h = (int8_t)src[ip++];
        if (h >= 0) {                                  // h + 1 literal bytes
            const size_t k = (size_t)h + 1;
            if (ip + k > n || out + k > cap) return -1;
            memcpy(dst + out, src + ip, k);
            ip += k; out += k;
        } else if (h != -128) {                        // next byte repeated 1 - h times
            const size_t k = (size_t)(1 - h);
            if (ip >= n || out + k > cap) return -1;
            memset(dst + out, src[ip++], k);
            out += k;
        }
    }
    return (long long)out;
}

int umx_profile_enable(umx_ctx* ctx, int on) {
    if (!ctx) return fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    int rc = prof_fold(ctx);
    if (rc) return rc;
    ctx->prof = on != 0;
    for (auto& s : ctx->sites) { s.launches = 0; s.total_ms = 0; s.flops = 0; s.bytes = 0; s.exec = 0; }
    return UMX_OK;
}

int umx_profile_read(umx_ctx* ctx, umx_prof_entry* entries, int max_entries, int* n_entries) {
    if (!ctx || !n_entries) return fail(ctx, UMX_ERR_INVALID, "bad arguments");
    int rc = prof_fold(ctx);
    if (rc) return rc;
    int n = 0;
    for (auto& s : ctx->sites) {
        if (s.launches == 0) continue;
        if (entries && n < max_entries) {
            umx_prof_entry& e = entries[n];
            memset(&e, 0, sizeof e);
            snprintf(e.name, sizeof e.name, "%s", s.name.c_str());
            snprintf(e.kernel, sizeof e.kernel, "%s", s.kernel.c_str());
            e.launches = s.launches;
            e.total_ms = s.total_ms;
            e.flops_per_launch_sum = s.flops;
            e.bytes_per_launch_sum = s.bytes;
            e.exec_flops_sum = s.exec;
        }
        ++n;
    }
    *n_entries = n;
    return UMX_OK;
}

}  // extern "C"
