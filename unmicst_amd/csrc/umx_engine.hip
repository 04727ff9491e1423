// Host side of libumx: graph construction from the reference's hyper-parameters, weight folding / packing,
// device memory plan, launch sequencing and the C ABI declared in include/umx.h.  gfx950 (MI355X) only.
#include "umx_internal.h"

using namespace umx;

namespace umx {

thread_local std::string g_err;

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
        if (ctx->sites[i].name == name && ctx->sites[i].kernel == kernel) return (int)i;   // (a layer that runs on two kernels -- e.g. the persistent form for large launch groups only -- has two entries)
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

// run the UNet on n tiles already in bufs[0] layout at `tiles` -> probs
// the (hi, lo) planes of buffer b for a batch of n tiles
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
        _Float16* const chi = ctx->in_cw ? reinterpret_cast<_Float16*>(b0.d) + (size_t)k0 * b0.S * b0.S * 2 * ctx->in_cw : hi_at(b0);
        HIP_TRY(ctx, launch_split_f32(tiles + (size_t)k0 * b0.floats_per_tile, (size_t)ns * b0.S * b0.S, b0.C, b0.Cs,
                                      std::ldexp(1.f, ctx->act_shift), chi, lo_at(b0), ctx->in_cw, run_stream(ctx)));
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
    p.overflow_flag = ctx->d_flag + ctx->flag_word;
    for (int gi = 0; gi < L.ngroups; ++gi) {
        const Buffer& sb = cur_bufs(ctx)[L.g[gi].src];
        p.src_hi[gi] = hi_at(sb);
        p.src_lo[gi] = lo_at(sb);
        p.Cs[gi] = sb.Cs;
        p.srcA[gi] = sb.planar ? 16 : sb.Cs * 2;
        p.srcB[gi] = sb.planar ? sb.S * sb.S * 16 : 16;
    }
    if (L.ngroups < 2) { p.srcA[1] = p.srcA[0]; p.srcB[1] = p.srcB[0]; }
    const Buffer& db = cur_bufs(ctx)[L.dst];
    if (p.head_K > 0) p.probs = probs + (size_t)k0 * L.H * L.W * p.head_K;   // fused softmax head
    else if (db.as_f32) p.dst_f32 = db.d + (size_t)k0 * db.floats_per_tile;
    else { p.dst_hi = hi_at(db); p.dst_lo = lo_at(db); p.dst_planar = db.planar ? 1 : 0; }
    if (L.app_src >= 0) {   // the raw-input skip rides in the spare channels of this launch's output (Launch::app_src)
        const Buffer& ab = cur_bufs(ctx)[L.app_src];
        if (L.app_src != 0 || !ctx->in_cw) return fail(ctx, UMX_ERR_INVALID, "internal: the appended tensor is not in the compact form");
        p.app_c = reinterpret_cast<const unsigned char*>(ab.d) + (size_t)k0 * ab.S * ab.S * 4 * ctx->in_cw;
        p.app_cw = ctx->in_cw;
        p.app_oct = L.app_c0 / 8;                 // the stored octet ...
        p.app_word = (L.app_c0 % 8) / 2;          // ... and its 32-bit word the two appended binary16 values fill
    }
    for (int gi = 0; gi < L.ngroups; ++gi)
        if (L.g[gi].src == 0 && ctx->in_cw && !L.use_first)
            return fail(ctx, UMX_ERR_INVALID, "internal: %s reads the compact input tiles through the generic kernel", L.name.c_str());
    {   // workgroup order (HConvParams::xcd_order): (N-block, phase) fastest inside an XCD for the layers that the x-fastest order
        // makes fabric-bound -- every one of a tile's N-blocks x phases workgroups re-reads its halo, and with 16 - 64 tiles resident
        // per XCD none of it is shared: if those bytes over the layer's matrix time (its executed FLOPs at 0.9 PFLOP/s) exceed
        // 4 TB/s the halo goes through ONE L2 instead.  (Settled per launch: the batch decides.  The solo model's 4 x 4 ... 16 x 16
        // pixel transposed convolutions: 5.9 - 7 TB/s, -13 ... -30 %; the 9-tile layers of the 256-pixel graph: 2.4 - 2.8 TB/s, and
        // order 2 costs them 8 - 17 % -- their weight slabs then compete for the L2: hence a rule, not a switch -- the A/B
        // environment variable of rounds 3-4 is retired.)
        const int ntiles = ((ns + p.imgs - 1) / p.imgs) * p.tiles_y * p.tiles_x;
        const int YZ = p.nblocks * (p.fused_phases ? 1 : p.nphase);
        bool want2 = false;
        // (plain convolutions of 2 - 4 N-blocks: the blocks of a tile share its halo through the L2 and only 2 - 4 weight slabs
        // compete for it -- the solo model's lu1.conv / lu2.conv -9 / -7 %, the 256-pixel graph's lu3.conv / lu4.conv -2 / -1 %)
        if (!p.fused_phases && p.nphase == 1 && YZ >= 2 && YZ <= 4 && ntiles >= 64) want2 = true;
        if (!want2 && YZ >= 4 && ntiles >= 64) {
            double octets = 0.0;
            for (int gi = 0; gi < L.ngroups; ++gi) octets += (double)((L.g[gi].C + 7) / 8);
            const double halo_bytes = 1.3 * (double)p.nhalo * octets * 32.0 * (double)ntiles * YZ;   // (1.3: 128-byte lines of short rows)
            const double matrix_s = L.exec_flops * ns / 0.9e15;
            want2 = matrix_s > 0.0 && halo_bytes / matrix_s > 4e12;
        }
        if (p.xcd_order == 1 && YZ > 1 && want2 && !p.w2) {   // (the W2 / F6 forms pair x-adjacent tiles per workgroup: order 1 only)
            p.xcd_order = 2;
            p.ntiles_grid = ntiles;
            p.tiles_per_xcd = (ntiles + 7) / 8;
        }
    }
    char kn[64];
    // the instantiation as rocprofv3 names it (<NT, KMT, NPH>): bench.py groups the timed sites by kernel
    // (<NT, KMT, NPH, DBG, MAXP, PK, D2S, F6, W2>)
    snprintf(kn, sizeof kn, "conv_f16x3<%d, %d, %d, false, %d, %s, %s, %s, %s>", L.nt16, p.kmt, p.fused_phases ? 4 : 1, p.maxp, p.pk ? "true" : "false",
             p.d2s ? "true" : "false", p.f6 ? "true" : "false", p.w2 ? "true" : "false");
    {
        // diagnostic: UMX_DEBUG_STAMPS=<layer name> prints the mean s_memtime segments of that layer's workgroups
        const char* const dbg_layer = getenv("UMX_DEBUG_STAMPS");
        if (dbg_layer && L.name == dbg_layer) {
            size_t nwg = (size_t)((ns + p.imgs - 1) / p.imgs) * p.tiles_y * p.tiles_x;
            if (p.w2) nwg = (nwg + 1) / 2;   // (two tiles per workgroup)
            nwg *= (size_t)p.nblocks * (p.fused_phases ? 1 : p.nphase);
            long long* d = nullptr;
            HIP_TRY(ctx, hipMalloc((void**)&d, nwg * 7 * sizeof(long long)));
            p.dbg = d;
            if (p.xcd_order == 2) p.xcd_order = 1;   // (the stamp records are indexed by the three-dimensional grid)
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
    if (L.use_first) {   // dense-K kernel of the first down-sampling layer
        FirstParams f = L.first;
        const Buffer& sb = cur_bufs(ctx)[0];
        f.B = ns;
        f.src_hi = hi_at(sb); f.src_lo = lo_at(sb);
        f.src_c = ctx->in_cw ? reinterpret_cast<const unsigned char*>(sb.d) + (size_t)k0 * sb.S * sb.S * 4 * ctx->in_cw : nullptr;
        f.dst_hi = p.dst_hi; f.dst_lo = p.dst_lo; f.dst_planar = p.dst_planar;
        f.overflow_flag = p.overflow_flag;
        snprintf(kn, sizeof kn, "conv_first<%d, %d, %d, false>", f.NT, f.CW, f.NKS);
        ProfScope ps(ctx, site_of(ctx, L.name, kn), L.flops * ns, L.bytes * ns, 2.0 * 3.0 * f.NKS * 32.0 * f.NT * 16.0 * L.H * L.W * ns);
        HIP_TRY(ctx, launch_conv_first(f, run_stream(ctx)));
        return UMX_OK;
    }
    const int site = site_of(ctx, L.name, kn);
    ctx->sites[site].xcd_order = p.xcd_order;
    ProfScope ps(ctx, site, L.flops * ns, L.bytes * ns, L.exec_flops * ns);
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

// tiles [t0, t1) of the slide (row-major tile index) -> probs_dev (tile t0 first): gather + normalise + UNet, in launch
// groups of <= max_batch tiles
bool gathers_raw(const umx_ctx* ctx) {
    return ctx->precision == UMX_PREC_F16X3 && ctx->hp.nChannels <= 8 && ctx->bufs[0].Cs == 8;
}

int tiles_range(umx_ctx* ctx, const double* image_dev, int C_img, const TileGeom& g, int band_row0, int band_rows,
                       double mean, double stdv, int t0, int t1, float* probs_dev, const void* raw_dev, int raw_bits,
                       const unsigned* mm_dev) {
    const size_t prob_f = (size_t)g.P * g.P * ctx->hp.nClasses;
    if (ctx->site_gather < 0) ctx->site_gather = site_of(ctx, "pi2d.gather_normalise", "gather_normalise");
    const bool direct16 = ctx->precision == UMX_PREC_F16X3 && ctx->hp.nChannels <= 8 && ctx->bufs[0].Cs == 8;
    if (raw_dev && !direct16) return fail(ctx, UMX_ERR_INVALID, "internal: raw planes handed to an engine that does not gather from them");
    const void* const src = raw_dev ? raw_dev : (const void*)image_dev;
    const double src_b = raw_dev ? raw_bits / 8.0 : 8.0;
    LaneLoop ll(ctx, t1 - t0);
    for (int t = t0, nb; t < t1; t += nb) {
        nb = ll.next(t1 - t);
        float* const tiles32 = ctx->precision == UMX_PREC_F16X3 ? (ctx->lane ? ctx->d_tiles32_2 : ctx->d_tiles32)
                                                                : cur_bufs(ctx)[0].d;
        {
            ProfScope ps(ctx, ctx->site_gather, 0.0,
                         (double)nb * g.P * g.P * (src_b * C_img + (ctx->in_cw ? 4.0 * ctx->in_cw : direct16 ? 32.0 : 4.0 * ctx->hp.nChannels)));
            if (direct16) {   // gather + normalise + (hi, lo) split in one pass
                const Buffer& b0 = cur_bufs(ctx)[0];
                HIP_TRY(ctx, launch_gather_split(src, raw_dev ? raw_bits : 0, C_img, band_row0, band_rows, g, ctx->hp.nChannels, mean, stdv, t, nb,
                                                 std::ldexp(1.f, ctx->act_shift), hi_of(b0), lo_of(b0, nb), ctx->in_cw, run_stream(ctx),
                                                 raw_dev ? mm_dev : nullptr));
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

}  // namespace umx

// ---- accessors for umx_shard.hip (same shared object; hidden visibility)
static void (*g_destroy_hook)(umx_ctx*) = nullptr;
hipStream_t umx_internal_stream(umx_ctx* ctx) { return ctx->stream; }
int umx_internal_device(umx_ctx* ctx) { return ctx->device; }
void umx_internal_hp(const umx_ctx* ctx, umx_hparams* out) { *out = ctx->hp; }
int umx_internal_fail(umx_ctx* ctx, int code, const char* msg) { return fail(ctx, code, "%s", msg); }
void umx_internal_set_destroy_hook(void (*hook)(umx_ctx*)) { g_destroy_hook = hook; }
// (umx_shard.hip) how a submitted call's completion event is waited for on a context that holds a communicator: polling, so that a peer's
// failure or a timeout ends the wait with a status instead of blocking for ever; NULL / no hook: hipEventSynchronize
static int (*g_wait_hook)(umx_ctx*, hipEvent_t) = nullptr;
void umx_internal_set_wait_hook(int (*hook)(umx_ctx*, hipEvent_t)) { g_wait_hook = hook; }
int umx_internal_wait_event(umx_ctx* ctx, hipEvent_t ev) {
    if (g_wait_hook) return g_wait_hook(ctx, ev);
    const hipError_t e = hipEventSynchronize(ev);
    return e == hipSuccess ? UMX_OK : fail(ctx, UMX_ERR_HIP, "hipEventSynchronize failed: %s", hipGetErrorString(e));
}

extern "C" {

const char* umx_version(void) { return "umx 0.4 (gfx950)"; }
int umx_prof_entry_size(void) { return (int)sizeof(umx_prof_entry); }

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

// the planner's OCP MX fp6 packing of one block of 32 (what the F6 form's weight images are made of): 24 bytes + the e8m0 scale byte
int umx_test_mx_pack_e2m3(const double* v32, uint8_t* out24) {
    if (!v32 || !out24) return -1;
    double v[32], amax = 0.0;
    for (int i = 0; i < 32; ++i) { v[i] = v32[i]; amax = std::max(amax, std::fabs(v[i])); }
    unsigned char b[24];
    const int e8 = mx_pack_e2m3(v, amax, b);
    memcpy(out24, b, 24);
    return e8;
}

// the DEVICE routine the stitch kernel converts with (d2h_rne: round-to-odd binary32, then v_cvt_f16_f32) on the same vectors: the
// host routine above states the conversion, this one is what runs
int umx_test_double_to_half_dev(const double* in, uint16_t* out, size_t n) {
    if (!in || !out) return fail(nullptr, UMX_ERR_INVALID, "in / out is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(nullptr, UMX_ERR_NO_DEVICE, "no HIP device available");
    double* din = nullptr;
    uint16_t* dout = nullptr;
    HIP_TRY(nullptr, hipMalloc((void**)&din, n * sizeof(double) + 16));
    if (hipMalloc((void**)&dout, n * sizeof(uint16_t) + 16) != hipSuccess) { hipFree(din); return fail(nullptr, UMX_ERR_OOM, "hipMalloc failed"); }
    hipError_t e = hipMemcpy(din, in, n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_d2h_rne_test(din, dout, n, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, dout, n * sizeof(uint16_t), hipMemcpyDeviceToHost);
    hipFree(din);
    hipFree(dout);
    return e == hipSuccess ? UMX_OK : fail(nullptr, UMX_ERR_HIP, "umx_test_double_to_half_dev: %s", hipGetErrorString(e));
}

int umx_describe(const umx_hparams* hp, int* n_launches, double* flops_per_tile, double* executed_flops_per_tile) {
    std::string why;
    int rc = check_hp(hp, &why);
    if (rc) return fail(nullptr, rc, "%s", why.c_str());
    std::vector<Launch> plan;
    build_graph(*hp, nullptr, &plan, nullptr, nullptr, nullptr);
    double f = 0, e = 0;
    for (auto& L : plan) { f += L.flops; e += L.exec_flops; }
    if (n_launches) *n_launches = (int)plan.size();
    if (flops_per_tile) *flops_per_tile = f;
    if (executed_flops_per_tile) *executed_flops_per_tile = e;
    return UMX_OK;
}

// Would umx_create(UMX_PREC_F16X3) take this model?  Host only (no device): the graph is built with zero weights and every launch goes
// through the split-precision planner's search (plan_f16 in its dry form: geometry, N-tile choice, (chunk, stage) search against the LDS
// budget).  UMX_OK, or UMX_ERR_INVALID with the first refused layer and the planner's reason in umx_last_error(NULL).
int umx_plan_check(const umx_hparams* hp) {
    std::string why;
    int rc = check_hp(hp, &why);
    if (rc) return fail(nullptr, rc, "%s", why.c_str());
    std::unique_ptr<umx_ctx> ctx(new umx_ctx());
    ctx->hp = *hp;
    ctx->precision = UMX_PREC_F16X3;
    std::vector<float> blob(blob_floats_needed(*hp), 0.f);
    struct { std::vector<size_t> buf_floats; std::vector<std::pair<int, int>> buf_geom; size_t pos = 0; } b;
    build_graph(*hp, blob.data(), &ctx->plan, &b.buf_floats, &b.buf_geom, &b.pos, conv_first_eligible(*hp));
    const int head_src = ctx->plan.back().g[0].src;
    for (auto& L : ctx->plan) {
        if (L.head) continue;
        make_d2s(L);
        if (!conv_geometry(L, &why)) return fail(nullptr, UMX_ERR_INVALID, "%s: %s", L.name.c_str(), why.c_str());
        if ((rc = plan_f16(ctx.get(), L, 0, L.dst == head_src, &ctx->plan.back(), &why, true)))
            return fail(nullptr, rc, "%s: %s", L.name.c_str(), why.c_str());
    }
    return UMX_OK;
}

int umx_describe_graph(const umx_hparams* hp, char* json, size_t cap, size_t* needed) {
    std::string why;
    int rc = check_hp(hp, &why);
    if (rc) return fail(nullptr, rc, "%s", why.c_str());
    std::vector<Launch> plan;
    std::vector<std::pair<int, int>> geom;
    build_graph(*hp, nullptr, &plan, nullptr, &geom, nullptr);
    std::string o = "{";
    char b[256];
    snprintf(b, sizeof b, "\"bn_epsilon\": %.9g, \"leaky_slope\": %.9g, \"buffers\": [", kBnEpsilon, (double)kLeakySlope);
    o += b;
    for (size_t i = 0; i < geom.size(); ++i) {
        snprintf(b, sizeof b, "%s{\"id\": %zu, \"size\": %d, \"channels\": %d}", i ? ", " : "", i, geom[i].first, geom[i].second);
        o += b;
    }
    o += "], \"launches\": [";
    for (size_t li = 0; li < plan.size(); ++li) {
        const Launch& L = plan[li];
        static const char* acts[] = {"none", "relu", "leaky_relu"};
        static const char* bns[] = {"none", "before_activation", "after_activation"};
        snprintf(b, sizeof b, "%s{\"name\": \"%s\", \"kind\": \"%s\", \"size\": %d, \"out_channels\": %d, \"dst\": %d, "
                              "\"max_pool\": %d, \"activation\": \"%s\", \"batch_norm\": \"%s\", \"stride\": %d, "
                              "\"summed_shortcut_ks\": %d, \"groups\": [",
                 li ? ", " : "", L.name.c_str(), L.head ? "head_softmax" : L.o_mul == 2 ? "conv_transpose" : "conv", L.H,
                 L.head ? L.head_K : L.Cout, L.dst, L.pool ? 2 : 0, L.head ? "softmax" : acts[L.act], bns[L.bn], L.o_mul, L.summed_shortcut);
        o += b;
        for (int gi = 0; gi < L.ngroups; ++gi) {
            size_t ntaps = 0;
            for (int ph = 0; ph < (L.head ? 0 : L.nphase); ++ph) ntaps += L.g[gi].taps[ph].size();
            int ks = 1;
            while ((size_t)ks * ks < ntaps) ++ks;
            snprintf(b, sizeof b, "%s{\"src\": %d, \"channels\": %d, \"ks\": %d}", gi ? ", " : "", L.g[gi].src, L.g[gi].C, L.head ? 1 : ks);
            o += b;
        }
        o += "]}";
    }
    o += "]}";
    if (needed) *needed = o.size() + 1;
    if (!json || cap < o.size() + 1) return json ? fail(nullptr, UMX_ERR_INVALID, "umx_describe_graph: %zu bytes needed", o.size() + 1) : UMX_OK;
    memcpy(json, o.c_str(), o.size() + 1);
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
        o2.precision = UMX_PREC_F16X3_F6;   // (= UMX_PREC_F16X3 for every model without wide low-resolution layers)
        int rc = umx_create_opts(hp, weight_blob, blob_floats, &o2, out);
        if (rc != UMX_ERR_INVALID) return rc;
        o2.precision = UMX_PREC_F16X3;
        rc = umx_create_opts(hp, weight_blob, blob_floats, &o2, out);
        if (rc != UMX_ERR_INVALID) return rc;
        const std::string first = g_err;
        o2.precision = UMX_PREC_F32;
        rc = umx_create_opts(hp, weight_blob, blob_floats, &o2, out);
        if (rc) g_err = first + "; fp32 kernels: " + g_err;
        // not silently: the exact-fp32 engine is ~4.6 x slower.  The note is what umx_last_error(ctx) returns until an error replaces it
        else (*out)->err = "note: UMX_PREC_DEFAULT runs this model on the exact-fp32 engine, about 4.6 x slower than the split-precision "
                           "kernels, whose planner refused it (" + first + ")";
        return rc;
    }
    const int device_ordinal = opts->device_ordinal, max_batch = opts->max_batch;
    int precision = opts->precision;
    if (precision == UMX_PREC_DEFAULT) {
        const char* e = getenv("UMX_PRECISION");
        precision = (e && !strcmp(e, "f32")) ? UMX_PREC_F32 : (e && !strcmp(e, "f16x3")) ? UMX_PREC_F16X3 : UMX_PREC_F16X3_F6;
    }
    if (precision != UMX_PREC_F32 && precision != UMX_PREC_F16X3 && precision != UMX_PREC_F16X3_F6)
        return fail(nullptr, UMX_ERR_INVALID, "unknown precision %d", precision);
    const bool f6 = precision == UMX_PREC_F16X3_F6;   // (internally: the split-precision engine with a flag)
    if (f6) precision = UMX_PREC_F16X3;
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
    ctx->f6 = f6;
    ctx->act_shift = act_shift;
    umx_ctx* c = ctx.get();
    HIP_TRY(nullptr, hipSetDevice(device_ordinal));
    HIP_TRY(nullptr, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;

    struct { std::vector<size_t> buf_floats; std::vector<std::pair<int, int>> buf_geom; size_t pos = 0; } b;
    // (the raw-skip fold of the split-precision plan:
    // only where the first layer takes the dense-K kernel: the input tiles are then stored once, in the compact form both read)
    static thread_local bool t_no_fold = false;   // (set for the one retry below)
    const bool fold = precision == UMX_PREC_F16X3 && conv_first_eligible(*hp) && !t_no_fold;
    build_graph(*hp, weight_blob, &c->plan, &b.buf_floats, &b.buf_geom, &b.pos, fold);
    if (b.pos != blob_floats) { umx_destroy(ctx.release()); return fail(nullptr, UMX_ERR_BLOB, "internal blob walk mismatch"); }
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
        bool phase_written = false;
        for (const Launch& Lp : c->plan)
            if (Lp.dst == (int)i && Lp.nphase == 4 && !(Lp.o_mul == 2 && Lp.ngroups == 1 && (Lp.Cout + 15) / 16 <= 5 && Lp.H >= 8 && Lp.W >= 16))
                phase_written = true;
        B.planar = !phase_written && !B.as_f32 && i != 0 && B.S >= 16 && B.Cs > 8 &&
                   (size_t)B.S * B.S * 16 < (1u << 24);   // (the kernels form octet offsets with 24-bit multiplies)
        // (hi, lo) binary16 planes with Cs channels take 4*Cs bytes per pixel
        const size_t bytes_per_tile = B.as_f32 ? B.floats_per_tile * sizeof(float) : (size_t)B.S * B.S * B.Cs * 4;
        void* d = nullptr;
        if ((rc = dev_alloc(c, &d, bytes_per_tile * (size_t)max_batch))) return bail(rc);
        B.d = (float*)d;
    }
    {
        int lanes = opts->lanes;
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
        if (f16) make_d2s(L);   // (narrow stride-2 transposed convolutions: depth-to-space form, before the geometry is laid out)
        if (!conv_geometry(L, &why)) { c->err = L.name + ": " + why; return bail(UMX_ERR_INVALID); }
        if (f16) {
            if ((rc = plan_f16(c, L, act_shift, L.dst == head_src, &c->plan.back(), &why))) {
                if (c->err.empty() || !why.empty()) c->err = L.name + ": " + why;
                return bail(rc);
            }
            if ((rc = plan_first(c, L, act_shift, &why))) { c->err = L.name + ": " + why; return bail(rc); }
            L.hcp.zeros = c->d_zeros;
            L.hcp.overflow_flag = c->d_flag;
            if (L.hcp.f6) c->f6_used = true;
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
    {   // compact input tiles: the graph folded the raw skip AND the first layer runs on the dense-K kernel
        bool folded = false, first = false;
        for (const Launch& L : c->plan) { folded = folded || L.app_src == 0; first = first || L.use_first; }
        if (folded && !first) {
            // The fold was decided from the hyper-parameters (conv_first_eligible), the dense-K first layer from the plan: where the
            // two disagree (a forced N-tile count, a first layer the planner split into N-blocks) the graph is rebuilt ONCE with the
            // two-group top convolution -- not failed into the 4 x slower exact-fp32 kernels
            umx_destroy(ctx.release());
            t_no_fold = true;
            const int rc2 = umx_create_opts(hp, weight_blob, blob_floats, opts, out);
            t_no_fold = false;
            return rc2;
        }
        if (folded) c->in_cw = c->hp.nChannels == 1 ? 1 : c->hp.nChannels == 2 ? 2 : 4;
    }
    *out = ctx.release();
    return UMX_OK;
}

int umx_precision_of(const umx_ctx* ctx) { return !ctx ? UMX_PREC_DEFAULT : ctx->f6_used ? UMX_PREC_F16X3_F6 : ctx->precision; }

void umx_destroy(umx_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    // drain everything that may still touch the context's buffers (or a communicator's) before anything is released
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) hipStreamSynchronize(ctx->stream2);
    if (ctx->up_stream) hipStreamSynchronize(ctx->up_stream);
    if (ctx->dn_stream) hipStreamSynchronize(ctx->dn_stream);
    if (g_destroy_hook) g_destroy_hook(ctx);   // a communicator / buffers umx_shard_init attached to this context
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
    if (stitch != UMX_STITCH_FP16_COMPAT && stitch != UMX_STITCH_FP32) return fail(ctx, UMX_ERR_INVALID, "bad stitch %d", stitch);
    return umx::stitch_rows(ctx, probs_dev, tpr0, tpr1, H, W, mode, stitch, y0, y1, out_dev, 0);
}

}  // extern "C"

namespace umx {
// (umx_stitch_dev + the internal forms: stitch = kStitchU8 writes the drivers' uint8 planes, plane_rows > 0 a padded destination)
int stitch_rows(umx_ctx* ctx, const float* probs_dev, int tpr0, int tpr1, int H, int W, int mode, int stitch, int y0, int y1,
                void* out_dev, int plane_rows) {
    if (!probs_dev || !out_dev || H < 1 || W < 1) return fail(ctx, UMX_ERR_INVALID, "bad probs/out/H/W");
    if (mode != UMX_MODE_ACCUMULATE && mode != UMX_MODE_REPLACE) return fail(ctx, UMX_ERR_INVALID, "bad mode %d", mode);
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
                 (double)(y1 - y0) * W * K * (4.0 * ((double)g.P / g.sub) * ((double)g.P / g.sub) + (stitch == 0 ? 2.0 : stitch == kStitchU8 ? 1.0 : 4.0)));
    HIP_TRY(ctx, launch_stitch(probs_dev, tpr0, tpr1, g, K, mode, stitch, y0, y1, out_dev, ctx->stream, plane_rows));
    return UMX_OK;
}
}  // namespace umx

extern "C" {

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

int umx_profile_enable(umx_ctx* ctx, int on) {
    if (!ctx) return fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    int rc = prof_fold(ctx);
    if (rc) return rc;
    ctx->prof = on < 0 ? 0 : on;
    for (auto& s : ctx->sites) { s.launches = 0; s.seen = 0; s.total_ms = 0; s.flops = 0; s.bytes = 0; s.exec = 0; }
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
            e.launches_seen = s.seen;
            e.xcd_order = s.xcd_order;
        }
        ++n;
    }
    *n_entries = n;
    return UMX_OK;
}

}  // extern "C"

