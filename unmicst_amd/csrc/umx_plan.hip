// Split-precision planner of libumx: chunking of the input octets, k-step table, stage table and packed weight images of
// one convolution launch for conv_f16x3 (layout documented in umx_conv_f16.hip).  Host code only.
#include "umx_internal.h"

#include <algorithm>
#include <atomic>
#include <thread>

namespace umx {

// ---- OCP MX fp6 (e2m3) block of 32: shared scale 2^(floor(log2 amax) - 2) as an e8m0 byte, elements round-to-nearest-even and saturating
// at 7.5, element i in bits [6 i, 6 i + 6) of the 24 bytes (the order v_cvt_scalef32_pk32_fp6_f16 writes and the scaled MFMA reads:
// tools/probes/mx_fp6_semantics.hip)
int mx_pack_e2m3(const double (&v)[32], double amax, unsigned char (&out)[24]) {
    memset(out, 0, sizeof out);
    if (!(amax > 0.0) || !std::isfinite(amax)) return 127;
    int e2;
    std::frexp(amax, &e2);                        // amax = m * 2^e2, m in [0.5, 1): floor(log2 amax) = e2 - 1
    int se = std::max(-127, std::min(127, e2 - 1 - 2));
    const double inv = std::ldexp(1.0, -se);
    for (int i = 0; i < 32; ++i) {
        const double a = std::fabs(v[i]) * inv;
        unsigned code;
        if (a >= 7.5) code = 31;
        else {
            int eb = 0;                            // binade: [0, 1) subnormal step 1/8, [1, 2) 1/8, [2, 4) 1/4, [4, 8) 1/2
            if (a >= 4.0) eb = 3; else if (a >= 2.0) eb = 2; else if (a >= 1.0) eb = 1;
            const double step = eb <= 1 ? 0.125 : eb == 2 ? 0.25 : 0.5;
            const double qv = std::nearbyint(a / step) * step;   // (default rounding mode: to nearest even)
            if (qv >= 7.5) code = 31;
            else if (qv < 1.0) code = (unsigned)std::lround(qv * 8.0);
            else {
                int ee = qv >= 4.0 ? 3 : qv >= 2.0 ? 2 : 1;
                code = (unsigned)(ee << 3) | (unsigned)std::lround((qv / std::ldexp(1.0, ee - 1) - 1.0) * 8.0);
            }
        }
        if (std::signbit(v[i]) && code) code |= 32u;
        for (int b = 0; b < 6; ++b)
            if ((code >> b) & 1u) out[(6 * i + b) / 8] |= (unsigned char)(1u << ((6 * i + b) % 8));
    }
    return se + 127;
}

// ---- split-precision plan of one conv launch: chunking of the input octets, k-step table, stage table, weight images
// (layout documented in umx_conv_f16.hip).  Reads the fp32 packing [tap][Cp][Np] produced by the Builder.
int plan_f16(umx_ctx* ctx, Launch& L, int act_shift, bool out_f32, const Launch* head, std::string* why, bool dry) {
    // A per-phase transposed convolution of tiny images (the solo model's 4 x 4 -> 8 x 8 layer: 16 images per tile, one octet per
    // halo chunk next to 8 N-tiles of weights) fills its k-steps with 1 - 2 (tap, octet) pairs of 4 in the phases that have 1 - 2
    // taps: 640 executed k-steps for 360.  Narrower N-blocks leave LDS for two octets per chunk (400 k-steps; lu3.convT -16 %,
    // the solo step -1.6 %): dry-run the search for every N-tile count with the same padding and take a >= 20 % shorter K loop.
    if (!dry && !L.force_nt16 && L.nphase == 4 && !L.d2s && !L.train && !getenv("UMX_PLAN_NT") && !getenv("UMX_PLAN_OVERRIDE")) {
        const int t16 = (L.Cout + 15) / 16;
        int base_k = 0, base_nt = 0, best_k = 0, best_nt = 0;
        for (int c = std::min(t16, kMaxNT16); c >= 4; --c) {   // (narrower than 4 N-tiles re-reads the halo too often)
            L.force_nt16 = c;
            std::string w2;
            if (plan_f16(ctx, L, act_shift, out_f32, head, &w2, true) != UMX_OK || L.hcp.fused_phases) { L.force_nt16 = 0; continue; }
            if (L.nt16 != c) continue;                       // (c does not keep the padded width: the planner chose another count)
            if (!base_nt) { base_nt = c; base_k = L.n_ksteps; best_nt = c; best_k = base_k; }   // the default choice comes first
            else if (L.n_ksteps < best_k) { best_nt = c; best_k = L.n_ksteps; }
        }
        L.force_nt16 = (base_nt && best_nt != base_nt && best_k * 5 <= base_k * 4) ? best_nt : 0;
    }
    const ConvParams& g = L.cp;   // tile geometry shared with the fp32 kernel
    HConvParams& h = L.hcp;
    memset(&h, 0, sizeof h);
    const int t16 = (L.Cout + 15) / 16;
    // stride-2 transposed convolution with few output channels: all four sub-pixel phases in one workgroup (the input
    // halo is read once instead of four times; 4 accumulator sets limit it to 5 N-tiles and 128 input pixels)
    const bool fused = L.nphase == 4 && L.o_mul == 2 && L.ngroups == 1 && !out_f32 && t16 <= 5 && L.H >= 8 && L.W >= 16 &&
                       !L.train;
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
    h.imgplane = h.hh * h.hw; h.nhalo = h.imgs * h.imgplane;
    h.ymin = g.ymin; h.xmin = g.xmin;
    h.nphase = L.nphase; h.o_mul = L.o_mul;
    h.H = L.H; h.W = L.W; h.Cout = L.Cout; h.Cds = round_up(L.d2s ? L.d2s_Cout : L.Cout, 8);   // (stored channels of the destination)
    int nt16 = 1, Np16 = 16;
    {   // N-tiles per workgroup: minimise padded N, prefer wide workgroups (fewer re-reads of the input halo)
        int best_pad = 1 << 30;
        for (int c = 1; c <= (fused ? 5 : kMaxNT16); ++c) {
            const int padded = round_up(t16, c);
            if (padded < best_pad || (padded == best_pad && c > nt16)) { nt16 = c; best_pad = padded; }
        }
        Np16 = best_pad * 16;
        if (L.force_nt16 > 0 && L.force_nt16 <= (fused ? 5 : kMaxNT16) && round_up(t16, L.force_nt16) == best_pad) nt16 = L.force_nt16;
        if (const char* e = getenv("UMX_PLAN_NT")) {   // tuning aid: "layer:NT" forces a layer's N-tiles per workgroup
            char nm[64];
            int c = 0;
            if (sscanf(e, "%63[^:]:%d", nm, &c) == 2 && L.name == nm && c >= 1 && c <= (fused ? 5 : kMaxNT16)) {
                nt16 = c;
                Np16 = round_up(t16, c) * 16;
            }
        }
    }
    L.nt16 = nt16;
    h.NT = nt16; h.nblocks = Np16 / (16 * nt16);
    // packed last N-tile (conv_f16x3's PK form): <= 8 real channels in the last of 2..5 N-tiles of a single N-block
    const int last_real = L.Cout - (nt16 - 1) * 16;
    h.d2s = L.d2s;
    h.d2s_mix = L.d2s && L.d2s_R > 0;
    h.pk = (!L.d2s && !L.train && h.nblocks == 1 && nt16 >= 2 && nt16 <= 5 && last_real >= 1 && last_real <= 8 && !getenv("UMX_DEBUG_STAMPS")) ? 1 : 0;
    // fp6 cross terms (conv_f16x3's F6 form): the 9-tile plain / per-phase kernel on the layers at <= 1/4 of the input resolution --
    // where tests/fp8_cross_term_report.py holds 1e-4 with a 4 x margin on the reference's trained weights (the full-resolution layers do not)
    // ... and, of those, the plain convolutions of >= 2 N-blocks on tiles of one image (>= 16 x 16 pixels): the per-phase transposed
    // convolutions (1 - 4 taps: a handful of stages per workgroup) and the 8 x 8-pixel layers (four images per tile: a 400-pixel halo,
    // chunks of one octet in the two-tile workgroup's LDS) measured 15 - 22 % SLOWER in the form, the single-N-block layers at 1/4
    // resolution level (+0.3 ... +3 %), the long-K convolutions of 2 - 4 N-blocks 5 - 13 % faster (docs/experiments.md); UMX_F6_ALL=1
    // takes every eligible layer (the A/B)
    const bool f6_all = getenv("UMX_F6_ALL") && atoi(getenv("UMX_F6_ALL")) == 1;
    const bool f6 = ctx->f6 && !fused && !L.d2s && !L.train && !out_f32 && nt16 == kMaxNT16 && !h.pk && L.H * 4 <= ctx->hp.imSize &&
                    (f6_all || (L.nphase == 1 && g.imgs == 1 && h.nblocks >= 2)) && !getenv("UMX_DEBUG_STAMPS");
    h.f6 = f6 ? 1 : 0;
    // two tiles per eight-wave workgroup (conv_f16x3's W2 form): the F6 form runs on it.  For the 3-product kernel alone it measured 4 %
    // SLOWER (docs/experiments.md, round 6: two independent four-wave workgroups per CU de-phase and cover each other's stage waits; one
    // eight-wave workgroup runs its waves in lock-step) -- UMX_W2=1 selects it there for that A/B
    const bool w2 = f6 || (getenv("UMX_W2") && atoi(getenv("UMX_W2")) == 1 && !fused && !L.d2s && !L.train && !out_f32 && nt16 == kMaxNT16 &&
                           !h.pk && !getenv("UMX_DEBUG_STAMPS"));
    h.w2 = w2 ? 1 : 0;
    if (f6)
        if (const char* e = getenv("UMX_F6_ABLATE"))
            if (atoi(e) & 3) {   // timing-only ablations of docs/experiments.md (conv_f16x3, f6step): WRONG RESULTS, said so once per process
                static std::atomic<bool> told{false};
                if (!told.exchange(true))
                    fprintf(stderr, "[umx] UMX_F6_ABLATE=%s: parts of the fp6 cross-term stage are switched off -- timing only, the results are WRONG\n", e);
                h.f6 |= (atoi(e) & 3) << 1;
            }
    h.outH = L.outH; h.outW = L.outW; h.pool = L.pool; h.act = L.act;
    if (h.nhalo > kHaloChunks * 64) { *why = "halo too large for the split-precision kernel"; return UMX_ERR_INVALID; }
    h.plane_slots = round_up(h.nhalo, 16);
    const int plane_pair = h.plane_slots * 16 * 2;   // hi + lo bytes of one octet plane

    // weight shift: largest |w| lands in [2^13, 2^14) so that the lo parts stay in binary16's normal range
    float maxabs = 0.f;
    for (int ph = 0; ph < L.nphase; ++ph)
        for (int gi = 0; gi < L.ngroups; ++gi)
            for (float v : L.g[gi].packed[ph]) maxabs = std::max(maxabs, std::fabs(v));
    L.wshift = 0;   // (training plans: the weights' scale is applied when they are repacked, and undone through HConvParams::dyn)
    if (maxabs > 0.f && std::isfinite(maxabs) && !L.train) {
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
        const int narrow = 53 * 1024;   // 3 workgroups per CU
        const int narrow_nt = 5;        // the kernels of <= 5 N-tiles fit 3 waves per SIMD
        const int nt3 = 40 * 1024;      // 4 workgroups per CU (the <= 3-tile kernels are built for 128 VGPRs)
        if (fused) {   // (4 pieces per wave and chunk where the halo fits them: 8 index registers and two thirds of the prologue less)
            attempts.clear();
            if (nt16 <= 3) attempts.push_back({4, narrow, 1 << 30});   // 164 registers: three workgroups per CU where the LDS allows
            attempts.push_back({4, kMaxLdsPerWG, 1 << 30});
            if (nt16 <= 3) attempts.push_back({12, kMaxLdsPerWG, 1 << 30});
        }
        else {
            for (int maxp : {4, 12}) {
                if (maxp == 12 && (nt16 > 5 || L.d2s)) break;   // (the depth-to-space instantiations keep 4 pixel indices)
                // (4-5 N-tiles at three workgroups per CU: round 2 kept them to <= 8 chunks -- lu1.conv, 24 chunks, ran 16 % faster
                // that way but the step did not, at NHWC-era L2 re-fetch rates; with planar activations and the register epilogue
                // the same-box A/B is 49.32 / 48.99 / 49.06 -> 48.70 / 48.77 / 48.74 ms per step: the rule is gone)
                // ... for plain convolutions.  A per-phase transposed convolution keeps it: the solo model's lu2.convT (8 x 8 input,
                // 320 -> 160 channels, 40 chunks) ran 1.84 -> 4.35 ms per launch at the tighter budget.
                const int few = L.nphase == 1 ? 1 << 30 : 8;   // (same-box A/B: solo 81.2 -> 83.1 k tiles/s, synthetic-256 19.12 -> 19.26 k)
                if (nt16 <= 3 && maxp == 4) attempts.push_back({maxp, nt3, 1 << 30});
                if (nt16 <= narrow_nt) attempts.push_back({maxp, narrow, few});
                attempts.push_back({maxp, kMaxLdsPerWG, 1 << 30});
            }
        }
    }
    // last resort: the whole LDS of a CU for one workgroup.  Tiny layers under big filters (2 x 2 / 4 x 4 pixels, 5 x 5 taps, 16 images
    // per tile: a 1024-pixel halo) fit nothing smaller; they are a few percent of such a model's work, and without them the whole model
    // falls back to the exact-fp32 engine (4.6 x slower end to end)
    {
        const int mp_last = attempts.empty() ? 4 : attempts.back().maxp;
        attempts.push_back({mp_last, 160 * 1024 - 512, 1 << 30});
    }
    const int nwaves = kWaves;
    h.kmt = fused ? 2 : kMT;
    // dynamic LDS of a plan: halo slots (hi + lo) | weight buffer 0 | weight buffer 1 (the epilogue stores from registers)
    // (F6 form: a stage holds two k-steps of hi images and one 2-KiB fp6 image per N-tile; a workgroup is two tiles with their own halo
    // slots over one pair of weight buffers, and has its CU's whole LDS)
    auto wbuf_of = [&](int ss) { return 64 + nt16 * (f6 ? 4096 : ss * 2048); };
    auto lds_total = [&](int nslots, int oc, int ss, int plane_pair_bytes) {
        return (w2 ? 2 : 1) * nslots * oc * plane_pair_bytes + 2 * wbuf_of(ss);
    };

    // One stage list (= one kernel phase, or the whole fused transposed convolution) for a given (OC, S, slot base E):
    // (tap, octet) pairs of every chunk -> k-steps of 4 -> stages of <= S k-steps.  Pairs left over when a chunk's
    // pair count is not a multiple of 4 are carried into the first k-step of the next chunk instead of being padded:
    // the previous chunk's halo slot is still resident then (its reload is issued at the start of the next chunk's LAST
    // stage, so the next chunk must have >= 2 stages).  Not across the phases of the fused kernel (other accumulators).
    struct Pair { int gi, ph, tap, oct, slot, k; };   // slot: halo slot (0/1) of the chunk; k: octet inside the chunk
    const bool carry_ok = !fused && !L.train;   // (training plans: every chunk stands alone -- the K split cuts between chunks)
    auto npairs_of = [&](const Chunk& k, int ph) { return (int)L.g[k.gi].taps[ph].size() * (k.o1 - k.o0); };
    auto plan_list = [&](int OC, int S, int list, std::vector<HStage>* stages_out,
                         std::vector<std::vector<Pair>>* steps_out, int* nchunks) {
        const auto pr = phases_of(list);
        const auto ch = chunks_for(OC, fused ? -1 : list);
        if (nchunks) *nchunks = (int)ch.size();
        int nsteps = 0;
        std::vector<Pair> carry;
        for (size_t c = 0; c < ch.size(); ++c) {
            const int gi = ch[c].gi, o0 = ch[c].o0, o1 = ch[c].o1;
            const int slot = (int)(c & 1);   // consecutive chunks alternate between the two halo slots
            bool first = true;   // the chunk's first stage carries its halo load
            for (int ph = pr.first; ph < pr.second; ++ph) {
                const int nt = (int)L.g[gi].taps[ph].size();
                if (!nt) continue;
                std::vector<Pair> pairs = carry;
                carry.clear();
                const std::vector<unsigned char>& dead = L.g[gi].dead[ph];
                const int noct_g = (L.g[gi].C + 7) / 8;
                for (int t = 0; t < nt; ++t)
                    for (int o = o0; o < o1; ++o) {
                        if (!dead.empty() && dead[(size_t)t * noct_g + o]) continue;
                        pairs.push_back({gi, ph, t, o, slot, o - o0});
                    }
                if (pairs.empty()) continue;
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
                    if (f6) st.nk = (short)(st.nk | 0x100);   // (F6 form: the stage ends with the scaled MFMAs of its k-steps' cross terms)
                    if (stages_out) stages_out->push_back(st);
                }
            }
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
    const int lds_cap = w2 ? 160 * 1024 - 512 : attempts[at].cap;
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
        for (int S = (f6 ? 2 : 1); S <= (f6 ? 2 : kStageK); ++S) {
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
            // (F6 form: the chunking the 3-product plan would choose -- its k-steps are cheaper, not its halo)
            const double cost = (ksteps * (1.0 + 0.30 / (f6 ? 1 : S)) + 0.35 * nchunks + sectors / 1500.0) * ((OC & 1) ? 1.0 : 1.03);
            if (nchunks > attempts[at].max_chunks) continue;
            if (cost < bestCost) { bestCost = cost; bestOC = OC; bestS = S; bestSlots = nslots; }
        }
    }
    }
    if (!bestOC) { *why = "LDS footprint too large for the split-precision kernel"; return UMX_ERR_INVALID; }
    if (dry) {   // the planner's own trial: executed k-steps of this N-tile count, nothing built
        L.n_ksteps = 0;
        for (int list = 0; list < nlists; ++list) L.n_ksteps += plan_list(bestOC, bestS, list, nullptr, nullptr, nullptr);
        return UMX_OK;
    }
    if (const char* e = getenv("UMX_PLAN_OVERRIDE")) {   // tuning aid: "layer:OC:S[,layer:OC:S...]" forces a layer's (OC, S)
        std::string spec(e);
        size_t pos = 0;
        while (pos < spec.size()) {
            const size_t end = spec.find(',', pos);
            const std::string item = spec.substr(pos, end == std::string::npos ? std::string::npos : end - pos);
            char nm[64];
            int oc = 0, ss = 0, mp = 0;
            const int nf = sscanf(item.c_str(), "%63[^:]:%d:%d:%d", nm, &oc, &ss, &mp);
            if (nf >= 3 && L.name == nm && oc >= 1 && oc <= 9 && ss >= 1 && ss <= (f6 ? 2 : kStageK)) {
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
    // 16-byte LDS slot of (halo pixel, octet k of the chunk) in halo slot `hs`
    auto lds_slot = [&](int hs, int pixel, int k) {
        return hs * (h.plane_slots * OC) + pixel * OC + k;
    };
    h.slot_bytes = h.plane_slots * OC * 16;
    h.lo_off = bestSlots * h.slot_bytes;
    h.b_off = 2 * h.lo_off;
    h.wbuf_bytes = wbuf_of(S);
    h.xcd_order = 1;   // XCD-aware tile order (run_launch_f16 turns it into order 2 where its rule says so)
    h.lds_bytes = (w2 ? 2 : 1) * h.b_off + 2 * h.wbuf_bytes;

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
            if (!f6) per_blk += 32 + (size_t)stages[si].nk * nt16 * 2 * 512;
            else per_blk += 32 + (size_t)nt16 * 2048;   // two k-steps of hi images (1 KiB each) + one 2-KiB fp6 image per N-tile
        }
        h.ph[list].wblk_stride = (int)(per_blk / 8);
        std::vector<_Float16>& W = wimg[list];
        W.assign(per_blk * h.nblocks, (_Float16)0.f);
        if (L.train) { L.wrefs[list].clear(); L.slab_units[list] = per_blk * h.nblocks / 8; }
        // one task per (N-block, stage): disjoint ranges of W, so the tasks are spread over host threads (the solo model's 117 MB of
        // filters took 0.5 s of a 3 s command-line run on one core); training plans record references and stay on one thread
        std::vector<size_t> ks_of;   // first k-step of each stage of this list
        {
            size_t k0 = 0;
            for (int si = h.ph[list].stage0; si < (int)stages.size(); ++si) { ks_of.push_back(k0); k0 += (size_t)(stages[si].nk & 0xff); }
        }
        const bool d2s_skip = L.d2s && L.d2s_npb == 4;
        std::atomic<bool> bad_slot{false};
        const int nst = (int)stages.size() - h.ph[list].stage0;
        auto fill_task = [&](int task) {
            const int nb = task / nst, si = h.ph[list].stage0 + task % nst;
            size_t ks = ks_of[(size_t)(task % nst)];
            if (f6) {
                // the stage's fp6 images (behind the room of two k-steps of hi images): per N-tile [64 lanes] x {24 bytes of e2m3, scale byte}
                // in two 16-byte planes.  Lane (row, qb): K block = k-step qb & 1 of the stage; qb < 2: w_lo (multiplies x_hi), qb >= 2: w_hi
                // (multiplies x_lo) -- the kernel's pixel operand.  A lone k-step's partner block is zero.
                const int nk = stages[si].nk & 0xff;
                const size_t blk = nb * per_blk + (size_t)stages[si].woff * 8;
                unsigned char* const img = reinterpret_cast<unsigned char*>(&W[blk + 32 + (size_t)2 * nt16 * 512]);
                for (int n = 0; n < nt16; ++n)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int row = lane & 15, qb = lane >> 4, jj = qb & 1;
                        const int co = nb * nt16 * 16 + n * 16 + row;
                        double v32[32];
                        double amax = 0.0;
                        for (int i = 0; i < 32; ++i) v32[i] = 0.0;
                        if (jj < nk && co < L.Cout)
                            for (int pp = 0; pp < 4; ++pp) {
                                const Pair& pr2 = steps[ks + jj][pp];
                                if (pr2.tap < 0) continue;
                                const Group& G = L.g[pr2.gi];
                                const int Cp = round_up(G.C, 4);
                                for (int e = 0; e < 8; ++e) {
                                    const int c = pr2.oct * 8 + e;
                                    if (c >= G.C) continue;
                                    const float v = G.packed[pr2.ph][((size_t)pr2.tap * Cp + c) * L.Np + co] * wscale;
                                    const _Float16 hi = (_Float16)v;
                                    const double t = qb < 2 ? (double)(float)(_Float16)(v - (float)hi) : (double)(float)hi;
                                    v32[pp * 8 + e] = t;
                                    amax = std::max(amax, std::fabs(t));
                                }
                            }
                        unsigned char bytes[24];
                        const int e8 = mx_pack_e2m3(v32, amax, bytes);
                        unsigned char* const p0 = img + (size_t)n * 2048 + (size_t)lane * 16;
                        unsigned char* const p1 = p0 + 1024;
                        memcpy(p0, bytes, 16);
                        memcpy(p1, bytes + 16, 8);
                        p1[8] = (unsigned char)e8; p1[9] = p1[10] = p1[11] = 0;
                        memset(p1 + 12, 0, 4);
                    }
            }
            {
                const size_t blk = nb * per_blk + (size_t)stages[si].woff * 8;
                unsigned short* const hdr = reinterpret_cast<unsigned short*>(&W[blk]);
                const int nk_st = stages[si].nk & 0xff;
                if (f6 && nk_st == 1) {   // (the scaled MFMA's lanes of the absent partner k-step read the lone k-step's slots; their weights are zero)
                    for (int qq = 0; qq < 4; ++qq) {
                        const Pair& pr2 = steps[ks][qq];
                        const auto& tp = L.g[pr2.gi].taps[pr2.ph][pr2.tap < 0 ? 0 : pr2.tap];
                        hdr[4 + qq] = (unsigned short)lds_slot(pr2.slot, (tp.first - g.ymin) * h.hw + (tp.second - g.xmin), pr2.k);
                    }
                }
                for (int j = 0; j < nk_st; ++j, ++ks) {
                    for (int qq = 0; qq < 4; ++qq) {
                        const Pair& pr2 = steps[ks][qq];
                        const auto& tp = L.g[pr2.gi].taps[pr2.ph][pr2.tap < 0 ? 0 : pr2.tap];
                        // 16-byte LDS slot of (halo pixel at this tap, octet k) in the pixel-major image of halo slot `slot`
                        const int slot = lds_slot(pr2.slot, (tp.first - g.ymin) * h.hw + (tp.second - g.xmin), pr2.k);
                        if (slot < 0 || slot >= bestSlots * h.plane_slots * OC || slot > 65535) { bad_slot = true; return; }
                        hdr[j * 4 + qq] = (unsigned short)slot;
                    }
                    if (d2s_skip) {
                        // depth-to-space form, one block of four phases: a k-step made only of taps the odd output rows (phase slots
                        // 2, 3) do not have leaves their N-tiles without weights -- the kernel skips them (flag byte j behind the k-map)
                        bool dead = true;
                        for (int qq = 0; qq < 4; ++qq) {
                            const Pair& pr2 = steps[ks][qq];
                            if (pr2.tap >= 0 && (pr2.tap >= 16 || (L.d2s_tapmask[pr2.ph][pr2.tap] & 0xC))) dead = false;
                        }
                        reinterpret_cast<unsigned char*>(hdr)[32 + j] = dead ? 1 : 0;
                    }
                    for (int n = 0; n < nt16; ++n)
                        for (int lane = 0; lane < 64; ++lane) {
                            const Pair& pr2 = steps[ks][lane >> 4];
                            if (pr2.tap < 0) continue;
                            const Group& G = L.g[pr2.gi];
                            const int Cp = round_up(G.C, 4);
                            const bool packed = h.pk && n == nt16 - 1;   // rows 0..7: w_hi, rows 8..15: w_lo of the same 8 channels
                            const int row = lane & 15;
                            const int co = nb * nt16 * 16 + n * 16 + (packed ? (row & 7) : row);
                            const size_t base = blk + 32 + (((size_t)j * nt16 + n) * (f6 ? 1 : 2)) * 512 + (size_t)lane * 8;
                            if (L.train) {   // by reference: the trainer fills this unit from its device-resident fp32 operand
                                const int nv = std::min(8, G.C - pr2.oct * 8);
                                if (nv > 0 && co < L.Cout)
                                    L.wrefs[list].push_back(HWRef{(int)(base / 8), (int)(((size_t)pr2.tap * Cp + (size_t)pr2.oct * 8) * L.Np + co),
                                                                  (unsigned short)nv, (unsigned short)pr2.gi});
                                continue;
                            }
                            for (int e = 0; e < 8; ++e) {
                                const int c = pr2.oct * 8 + e;
                                if (c >= G.C || co >= L.Cout) continue;
                                const float v = G.packed[pr2.ph][((size_t)pr2.tap * Cp + c) * L.Np + co] * wscale;
                                const _Float16 hi = (_Float16)v;
                                if (packed) { W[base + e] = row < 8 ? hi : (_Float16)(v - (float)hi); continue; }   // (no lo image)
                                W[base + e] = hi;
                                if (!f6) W[base + 512 + e] = (_Float16)(v - (float)hi);   // (F6 form: the lo part lives in the B stage's image)
                            }
                        }
                }
            }
        };
        const int ntasks = h.nblocks * nst;
        const size_t work = per_blk * (size_t)h.nblocks;
        int nthr = L.train || work < (1u << 20) ? 1 : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (const char* e = getenv("UMX_PLAN_THREADS")) nthr = std::max(1, atoi(e));
        if (L.train) nthr = 1;
        nthr = std::min(nthr, std::max(1, ntasks));
        if (nthr <= 1) {
            for (int t = 0; t < ntasks; ++t) fill_task(t);
        } else {
            std::atomic<int> next{0};
            std::vector<std::thread> pool;
            for (int i = 0; i < nthr; ++i)
                pool.emplace_back([&] { for (int t; (t = next.fetch_add(1)) < ntasks;) fill_task(t); });
            for (auto& th : pool) th.join();
        }
        if (bad_slot) { *why = "internal: k-map slot out of range"; return UMX_ERR_INVALID; }
    }

    // epilogue constants per N-block: pre_s absorbs 2^-(weight shift + input activation shift), post_* the output's
    // 2^(activation shift); padded channels get pre_s = post_s = 0 so that they store exact zeros
    {
        const float unshift = std::ldexp(1.f, -(L.wshift + act_shift));
        const float oscale = out_f32 ? 1.f : std::ldexp(1.f, act_shift);
        const int nb16 = nt16 * 16;
        // a fused softmax head needs every channel of a pixel in one workgroup; it replaces the fp32 store of this layer
        const bool fuse_head = head && out_f32 && h.nblocks == 1 && head->head_K <= 4 && !fused;
        h.head_K = fuse_head ? head->head_K : 0;
        const size_t per_blk = (fuse_head ? (size_t)(4 + h.head_K) * nb16 + 16 : (size_t)4 * nb16) + (L.d2s ? 64 : 0);
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
        if (L.d2s) {   // destination table of the depth-to-space epilogue (HConvParams::d2s_mix), in the destination's layout
            const Buffer& db = ctx->bufs[L.dst];
            const int dPix = db.planar ? 8 : h.Cds, dOct = db.planar ? L.outH * L.outW * 8 : 8;
            const int per_z = 2 * nt16 + 8;
            if ((h.nblocks != 1 && L.d2s_R > 0) || 2 * per_z > 64 || L.nphase > 2) { *why = "internal: depth-to-space table"; return UMX_ERR_INVALID; }
            for (int nb = 0; nb < h.nblocks; ++nb) {   // (every N-block carries its own copy behind its constants)
            int* const tab = reinterpret_cast<int*>(&ec[(size_t)nb * per_blk + (size_t)4 * nb16]);
            for (int z = 0; z < L.nphase; ++z) {
                for (int vl = 0; vl < 2 * nt16; ++vl) {
                    const int vo = nb * 2 * nt16 + vl;   // stored octet along the block's N axis
                    int d = -1;
                    if (vo < L.d2s_npb * L.d2s_F) {   // (make_d2s: the octets of slots 2i, 2i + 1 alternate)
                        const int pr = vo / (2 * L.d2s_F), rm = vo % (2 * L.d2s_F), j = 2 * pr + (rm & 1), o = rm / 2;
                        d = (L.d2s_oy[z][j] * L.outW + L.d2s_ox[z][j]) * dPix + o * dOct;
                    }
                    tab[z * per_z + vl] = d;
                }
                for (int j = 0; j < 4; ++j) {
                    const bool on = L.d2s_R > 0 && j < L.d2s_npb;
                    const int pp = on ? L.d2s_oy[z][j] * L.outW + L.d2s_ox[z][j] : 0;
                    tab[z * per_z + 2 * nt16 + j] = on ? pp * dPix + L.d2s_F * dOct : -1;
                    tab[z * per_z + 2 * nt16 + 4 + j] = on ? L.d2s_oy[z][j] * 2 + L.d2s_ox[z][j] : 0;
                }
            }
            }
        }
        if (fuse_head) {
            // The 1x1 head runs on the matrix cores (conv_f16x3's fused-head epilogue): its weights as MFMA A-fragments, rows =
            // classes, scaled by 2^hs so that their lo parts stay normal binary16 numbers; the head's BN scale absorbs 2^-hs.
            // k order of head k-step s2: lane group q, element j <-> channel (2 s2 + (j >> 2)) * 16 + 4 q + (j & 3) -- the four
            // accumulator values of N-tiles 2 s2 and 2 s2 + 1 a lane holds, so activations never leave their lane.
            float hmax = 0.f;
            for (float v : head->head_w) hmax = std::max(hmax, std::fabs(v));
            int hs = 0;
            if (hmax > 0.f && std::isfinite(hmax)) {
                int e2;
                std::frexp(hmax, &e2);
                hs = std::max(-24, std::min(30, 14 - e2));
            }
            const float hscale = std::ldexp(1.f, hs);
            float* e = &ec[(size_t)(4 + h.head_K) * nb16];
            for (int k = 0; k < h.head_K; ++k) {
                e[k] = (head->pre_s.empty() ? 1.f : head->pre_s[k]) * std::ldexp(1.f, -hs);
                e[8 + k] = head->pre_b.empty() ? 0.f : head->pre_b[k];
            }
            const int ns2 = (nt16 + 1) / 2;
            std::vector<_Float16> HF((size_t)ns2 * 2 * 512, (_Float16)0.f);
            for (int s2 = 0; s2 < ns2; ++s2)
                for (int lane = 0; lane < 64; ++lane) {
                    const int row = lane & 15, qq = lane >> 4;
                    if (row >= h.head_K) continue;
                    for (int j = 0; j < 8; ++j) {
                        const int nt = 2 * s2 + (j >> 2), c = nt * 16 + 4 * qq + (j & 3);
                        if (nt >= nt16 || c >= L.Cout) continue;
                        const float v = head->head_w[(size_t)c * head->head_K + row] * hscale;
                        const _Float16 hi = (_Float16)v;
                        HF[((size_t)s2 * 2) * 512 + (size_t)lane * 8 + j] = hi;
                        HF[((size_t)s2 * 2 + 1) * 512 + (size_t)lane * 8 + j] = (_Float16)(v - (float)hi);
                    }
                }
            _Float16* dh = nullptr;
            int rc3 = upload_raw(ctx, HF, &dh);
            if (rc3) return rc3;
            h.head_frag = reinterpret_cast<const uint4*>(dh);
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
                L.name.c_str(), fused ? (h.pk ? "fused-phase packed " : "fused-phase ") : (h.pk ? "packed " : h.f6 ? "fp6-cross " : h.w2 ? "two-tile " : ""), nt16, h.nblocks, OC, bestSlots, S, h.lds_bytes, L.n_ksteps,
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
    if (L.train) L.stages_host = stages;
    for (int list = 0; list < nlists; ++list) {
        _Float16* d = nullptr;
        if ((rc = upload_raw(ctx, wimg[list], &d))) return rc;
        h.ph[list].w = reinterpret_cast<const uint4*>(d);
    }
    L.exec_flops = 2.0 * 3.0 * (double)L.n_ksteps * 32.0 * Np16 * L.H * L.W;   // MFMA work incl. split and padding
    if (h.pk) L.exec_flops -= 2.0 * (double)L.n_ksteps * 32.0 * 16.0 * L.H * L.W;    // (the packed N-tile takes 2 products)
    if (h.f6) L.exec_flops *= 0.5;   // matrix time in binary16-MFMA units: 1 (x_hi * w_hi) + 0.5 (one 16-cycle scaled MFMA per two k-steps)
    return UMX_OK;
}

// ---- dense-K plan of the first down-sampling layer (umx_conv_first.hip).  Applies to a pooled single-source convolution of
// the input tiles (buffer 0) with <= 4 input channels and <= 80 output channels on tiles of >= 16 x 16 pixels whose taps
// x channel slots fit two k-steps; everything else stays on conv_f16x3.  Reuses plan_f16's weight shift and epilogue
// constants, so the two kernels differ only in the summation order inside the (now single) k-step.
bool conv_first_eligible(const umx_hparams& hp) {
    if (hp.nExtraConvs != 0 || hp.nChannels < 1 || hp.nChannels > 4 || hp.imSize < 16 || hp.nOut0 > 80) return false;
    const int CW = hp.nChannels == 1 ? 1 : hp.nChannels == 2 ? 2 : 4;
    return conv_first_supported((hp.nOut0 + 15) / 16, CW, (hp.ks * hp.ks * CW + 31) / 32);
}

int plan_first(umx_ctx* ctx, Launch& L, int act_shift, std::string* why) {
    (void)why;
    L.use_first = false;
    const HConvParams& h = L.hcp;
    if (L.ngroups != 1 || L.g[0].src != 0 || L.nphase != 1 || !L.pool || h.head_K > 0 || h.nblocks != 1) return UMX_OK;
    const int Ci = L.g[0].C, P = L.H;
    if (Ci < 1 || Ci > 4 || P < 16 || L.W != P || (P & (P - 1))) return UMX_OK;
    const auto& taps = L.g[0].taps[0];
    const int ntaps = (int)taps.size();
    int ks = 1;
    while (ks * ks < ntaps) ks += 2;
    if (ks * ks != ntaps) return UMX_OK;
    for (int t = 0; t < ntaps; ++t)
        if (taps[t].first != t / ks - (ks - 1) / 2 || taps[t].second != t % ks - (ks - 1) / 2) return UMX_OK;
    const int CW = Ci == 1 ? 1 : Ci == 2 ? 2 : 4, NKS = (ntaps * CW + 31) / 32, NT = h.NT;
    if (!conv_first_supported(NT, CW, NKS)) return UMX_OK;
    if (L.g[0].packed[0].empty()) return UMX_OK;
    FirstParams& f = L.first;
    memset(&f, 0, sizeof f);
    f.P = P; f.Ci = Ci; f.ks = ks; f.ntaps = ntaps;
    f.NT = NT; f.CW = CW; f.NKS = NKS;
    f.rw_log2 = P >= 64 ? 6 : P >= 32 ? 5 : 4;
    f.hh = 16 + ks - 1;
    f.hw = (1 << f.rw_log2) + ks - 1;
    f.inv_hw = 1.f / (float)f.hw;
    f.lds_bytes = ((f.hh * f.hw * 4 * CW + 15) & ~15) + 4 * NT * 16 * (int)sizeof(float);
    f.act = L.act;
    f.post_affine = h.post_affine;
    f.econst = reinterpret_cast<const float*>(h.econst);
    f.Cds = h.Cds;
    f.outS = P / 2;
    // MFMA A-fragments: lane (q = lane >> 4, row = lane & 15) holds weight[channel n*16 + row][k = 32 s + 8 q + e], e = 0..7,
    // k = tap * CW + c
    const float wscale = std::ldexp(1.f, L.wshift);
    const int Cp = round_up(Ci, 4);
    std::vector<_Float16> W((size_t)NKS * NT * 2 * 512, (_Float16)0.f);
    for (int s = 0; s < NKS; ++s)
        for (int n = 0; n < NT; ++n)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int kk = 32 * s + 8 * (lane >> 4) + e, t = kk / CW, c = kk % CW, co = n * 16 + (lane & 15);
                    if (t >= ntaps || c >= Ci || co >= L.Cout) continue;
                    const float v = L.g[0].packed[0][((size_t)t * Cp + c) * L.Np + co] * wscale;
                    const _Float16 hi = (_Float16)v;
                    const size_t base = (((size_t)s * NT + n) * 2) * 512 + (size_t)lane * 8 + e;
                    W[base] = hi;
                    W[base + 512] = (_Float16)(v - (float)hi);
                }
    _Float16* d = nullptr;
    int rc = upload_raw(ctx, W, &d);
    if (rc) return rc;
    f.w = reinterpret_cast<const uint4*>(d);
    L.use_first = true;
    if (getenv("UMX_DEBUG_PLAN"))
        fprintf(stderr, "[umx plan] %-12s dense-K first layer: NT %d, %d channel slot(s) x %d taps = %d k-step(s), region 16 x %d, LDS %d B\n",
                L.name.c_str(), NT, CW, ntaps, NKS, 1 << f.rw_log2, f.lds_bytes);
    return UMX_OK;
}

// ---- depth-to-space form of a narrow stride-2 transposed convolution.  Output pixel (2y + pu, 2x + pv) of phase (pu, pv) is a
// convolution of the input at (y, x) over that phase's taps; all phases' taps lie in one small window (2 x 2 for a 3 x 3 filter),
// so the four phases are ONE plain convolution over the window whose output channels are [phase][channel] (structural zeros where
// a phase has no tap), followed by a depth-to-space store.  What it buys over the fused-phase kernel (KMT = 2, one phase per
// stage, 4 x ceil(Cout / 16) N-tiles): 4 x 36 channels fill 9 N-tiles exactly instead of 12, the K loop is 9 k-steps of 108 MFMAs
// instead of 22 of 18 (half the barriers, a quarter of the weight-fragment reads per MFMA), workgroups cover 256 input pixels.
// What it costs: the structural zeros are multiplied (81 tile-k-steps instead of 59).  N layout of a block of `npb` phases:
// [slots 0, 1: F octets each, alternating] [slots 2, 3 likewise] [remainder tile: 4 channel slots per phase, R <= 4 real] -- F = Cout / 8,
// R = Cout % 8 -- so that a stored octet belongs to one phase, and the remainder tile's lane group q to phase slot q.
// One block of all four phases if that is <= 9 N-tiles, else two blocks by output row parity (the pu = 1 block has no dy = -1
// taps: half the K loop); wider layers stay on the per-phase / fused-phase forms.
bool make_d2s(Launch& L) {
    if (L.head || L.nphase != 4 || L.o_mul != 2 || L.ngroups != 1 || L.pool || L.H < 2) return false;   // (M-tile pairs = row pairs of an image)
    for (int ph = 0; ph < 4; ++ph)
        if (L.oy_off[ph] != (ph >> 1) || L.ox_off[ph] != (ph & 1) || L.g[0].taps[ph].empty() || L.g[0].packed[ph].empty()) return false;
    const int F = L.Cout / 8, R = L.Cout % 8, rem = R > 0 ? 1 : 0;
    if (R > 4 || F < 1) return false;
    memset(L.d2s_tapmask, 0, sizeof L.d2s_tapmask);
    int nz, npb;
    if (2 * F + rem >= 5 && 2 * F + rem <= 9) { nz = 1; npb = 4; }
    else if (F + rem >= 5 && F + rem <= 9) { nz = 2; npb = 2; }
    // (two N-blocks per row parity -- F = 10, the solo model's top layer 160 -> 80 -- measured 20 % slower than the fused-phase form)
    else return false;
    if (L.app_src >= 0 && (!rem || L.app_c0 / 8 != F || (L.app_c0 % 8) / 2 < 1 || L.W < 16)) return false;   // appended channels ride in the remainder tile
    const int NT = npb * F / 2 + rem, Nv = NT * 16;
    // N index of (phase slot j, channel co).  The full octets of the two phases of an output row (slots 2i, 2i + 1: pixels 2x and
    // 2x + 1) alternate, so that the two octets of an N-tile are the SAME destination octet of neighbouring output pixels and one
    // store instruction writes whole 512-byte runs of an output row (not 16 bytes at a 32-byte stride, completed later)
    auto vchan = [&](int j, int co) {
        return co < 8 * F ? ((j >> 1) * 2 * F + 2 * (co / 8) + (j & 1)) * 8 + (co & 7) : npb * 8 * F + 4 * j + (co - 8 * F);
    };
    Group& g = L.g[0];
    const int Cp = round_up(g.C, 4), Np_old = L.Np;
    std::vector<std::pair<int, int>> taps_v[2];
    std::vector<float> packed_v[2];
    for (int z = 0; z < nz; ++z) {
        std::vector<std::pair<int, int>>& U = taps_v[z];
        for (int j = 0; j < npb; ++j)
            for (auto& t : g.taps[z * npb + j])
                if (std::find(U.begin(), U.end(), t) == U.end()) U.push_back(t);
        std::sort(U.begin(), U.end());
        packed_v[z].assign(U.size() * (size_t)Cp * Nv, 0.f);
        for (int j = 0; j < npb; ++j) {
            const int ph = z * npb + j;
            L.d2s_oy[z][j] = L.oy_off[ph];
            L.d2s_ox[z][j] = L.ox_off[ph];
            for (size_t t = 0; t < g.taps[ph].size(); ++t) {
                const size_t u = std::find(U.begin(), U.end(), g.taps[ph][t]) - U.begin();
                if (u < 16) L.d2s_tapmask[z][u] |= (unsigned char)(1u << j);
                for (int c = 0; c < g.C; ++c)
                    for (int co = 0; co < L.Cout; ++co) {
                        const int v = vchan(j, co);
                        packed_v[z][(u * Cp + c) * Nv + v] = g.packed[ph][(t * Cp + c) * Np_old + co];
                    }
            }
        }
    }
    // epilogue constants in the N order of a block (the same for every block): padding channels store exact zeros
    auto permute = [&](const std::vector<float>& src, float dflt, float pad) {
        std::vector<float> out((size_t)Nv, pad);
        for (int j = 0; j < npb; ++j)
            for (int co = 0; co < L.Cout; ++co) {
                const int v = vchan(j, co);
                out[v] = src.empty() ? dflt : src[co];
            }
        return out;
    };
    L.pre_s = permute(L.pre_s, 1.f, 0.f);
    L.pre_b = permute(L.pre_b, 0.f, 0.f);
    if (!L.post_s.empty() || !L.post_b.empty()) {
        L.post_s = permute(L.post_s, 1.f, 1.f);
        L.post_b = permute(L.post_b, 0.f, 0.f);
    }
    L.d2s = 1; L.d2s_Cout = L.Cout; L.d2s_F = F; L.d2s_R = R; L.d2s_npb = npb;
    for (int ph = 0; ph < 4; ++ph) {
        g.taps[ph].clear();
        std::vector<float>().swap(g.packed[ph]);
        L.oy_off[ph] = L.ox_off[ph] = 0;
    }
    for (int z = 0; z < nz; ++z) { g.taps[z] = taps_v[z]; g.packed[z] = std::move(packed_v[z]); }
    L.nphase = nz;
    L.Cout = Nv;
    L.Np = Nv;
    return true;
}

}  // namespace umx
