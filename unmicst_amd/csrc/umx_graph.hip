// Graph construction of libumx: the reference's hyper-parameters and weight blob -> the list of convolution launches
// (reference UnMicst1-5.py:55-237 for the v2 graph, UnMicst.py:51-187 for the legacy one) with BatchNorm folded, the
// same-source shortcut summed into the main filter, transposed convolutions split into sub-pixel phases and the concat
// replaced by two operand groups.  Host code only.
#include "umx_internal.h"

namespace umx {

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
        const double sc = (double)bn.g[c] / std::sqrt((double)bn.v[c] + kBnEpsilon);
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
                Lh.bn = v2 ? 1 : 2;
                if (!blob) return;
                if (v2) fold_bn(bn, Co, &Lh.pre_s, &Lh.pre_b);
                else fold_bn(bn, Co, &Lh.post_s, &Lh.post_b);
            };
            if (nx == 0) {
                snprintf(nm, sizeof nm, "ld%d.conv", i);
                Launch Lh = make(nm, S, Co, ds[i + 1], 1, act);
                add_conv_group(Lh, ds[i], w1, 0, Ci, &wsc);
                Lh.summed_shortcut = kss;
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
                Lb.bn = 1;
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
            // Raw-skip fold (top layer of the split-precision plan): the up-sampled tensor has spare stored channels (36 real of
            // 40), the raw input 1 - 2: the transposed convolution's epilogue writes them there and this convolution reads
            // [us | skip] as one tensor (filter channels permuted accordingly) -- one operand group, 12 instead of 14 k-steps
            const bool fold = fold_top_skip && idx == 0 && nx == 0 && Cskip <= 2 && (Cup % 8) != 0 && (Cup % 8) % 2 == 0 &&
                              (Cup % 8) + Cskip <= 8 && S2 >= 16 && Cup <= 80;   // (<= 5 N-tiles: the kernels that carry the append)
            if (fold) {
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
            if (v2) Lc.bn = 1;
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
                Lh.bn = 1;
                if (blob) fold_bn(bn, hp.nClasses, &Lh.pre_s, &Lh.pre_b);
            }
            Lh.flops = Lh.exec_flops = 2.0 * S * S * n[1] * hp.nClasses;
            Lh.bytes = 4.0 * S * S * (n[1] + hp.nClasses);
            plan.push_back(std::move(Lh));
        }
        return UMX_OK;
    }
};

int build_graph(const umx_hparams& hp, const float* blob, std::vector<Launch>* plan, std::vector<size_t>* buf_floats,
                std::vector<std::pair<int, int>>* buf_geom, size_t* pos, bool fold_top_skip) {
    Builder b(hp, blob);
    b.fold_top_skip = fold_top_skip;
    const int rc = b.build();
    if (plan) *plan = std::move(b.plan);
    if (buf_floats) *buf_floats = std::move(b.buf_floats);
    if (buf_geom) *buf_geom = std::move(b.buf_geom);
    if (pos) *pos = b.pos;
    return rc;
}

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

}  // namespace umx
