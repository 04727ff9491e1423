// Host pipeline of libumx: the entry points that take host arrays (reference seam UnMicst1-5.py:687-710 and the drivers'
// pre/post-processing around it, :807-854), with uploads / downloads overlapped with the tile kernels.
#include "umx_internal.h"

#include <limits>
#include <thread>

using namespace umx;

// (min, max) of a host plane in one pass on a few threads -- what np.min / np.max over the page cost the reference's driver
// (UnMicst1-5.py:817-821) twice; the loops are plain enough for the host compiler to vectorise (AVX2 where the CPU has it)
namespace {
template <class T>
__attribute__((target("avx2"))) void range_avx2(const T* x, size_t n, uint32_t* lo, uint32_t* hi) {
    T a = std::numeric_limits<T>::max(), b = 0;
    for (size_t i = 0; i < n; ++i) { a = x[i] < a ? x[i] : a; b = x[i] > b ? x[i] : b; }
    *lo = a; *hi = b;
}
template <class T>
void range_plain(const T* x, size_t n, uint32_t* lo, uint32_t* hi) {
    T a = std::numeric_limits<T>::max(), b = 0;
    for (size_t i = 0; i < n; ++i) { a = x[i] < a ? x[i] : a; b = x[i] > b ? x[i] : b; }
    *lo = a; *hi = b;
}
template <class T>
void range_threads(const T* x, size_t n, uint32_t* out) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("UMX_HOST_THREADS")) nt = (unsigned)std::max(1, atoi(e));
    nt = (unsigned)std::max<size_t>(1, std::min<size_t>(nt, n >> 20));   // >= 1 M samples per thread
    std::vector<uint32_t> lo(nt, 0xFFFFFFFFu), hi(nt, 0u);
    auto work = [&](unsigned t) {
        const size_t a = n * t / nt, b = n * (t + 1) / nt;
        if (avx2) range_avx2(x + a, b - a, &lo[t], &hi[t]);
        else range_plain(x + a, b - a, &lo[t], &hi[t]);
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& t : th) t.join();
    out[0] = *std::min_element(lo.begin(), lo.end());
    out[1] = *std::max_element(hi.begin(), hi.end());
}
}  // namespace


int umx_internal_wait_event(umx_ctx* ctx, hipEvent_t ev);   // umx_engine.hip

extern "C" {

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
    if (const int rc = umx_internal_wait_event(ctx, hs.done)) return rc;   // (a sharded context polls its communicator while it waits)
    if (*hs.flag_host)   // this slot's own flag word, cleared by the next submit on the slot in stream order
        return fail(ctx, UMX_ERR_RANGE, "an activation left the binary16 range of the split-precision path; "
                                        "create the context with UMX_PREC_F32 (or UMX_PRECISION=f32)");
    return UMX_OK;
}

static int host_submit_impl(umx_ctx* ctx, int slot, bool sync_call, const void* src, int src_bits, int C_img, int H, int W, int rescale,
                            double mean, double stdv, int mode, int stitch, int out_u8, void* out_host) {
    umx_ctx::HostSlot& hs = ctx->hs[slot];
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
    // raw planes: the tile gather converts (and, with `rescale`, rescales to the plane's (min, max) words, ready in stream order
    // before the first tile) as it reads -- the float64 image, 8 bytes written and 14 read per pixel and channel, is never made, and
    // not allocated either: 4.3 GB for a two-channel 16384 x 16384 slide
    const bool raw_gather = src_bits != 0 && gathers_raw(ctx);
    if (!raw_gather && (rc = grow(ctx, (void**)&hs.d_image, &hs.image_cap, plane * C_img * sizeof(double)))) return rc;
    if ((rc = grow(ctx, &hs.d_out, &hs.out_cap, mm_off + 64 * (size_t)C_img))) return rc;
    if ((rc = grow(ctx, (void**)&hs.d_probs, &hs.probs_cap, (size_t)g.npr * g.npc * g.P * g.P * K * sizeof(float)))) return rc;
    unsigned char* const base = (unsigned char*)hs.d_out;
    if (!ctx->up_stream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->dn_stream, hipStreamNonBlocking));
    }
    // slabs = the launch groups of the tile loop (equal groups of <= max_batch tiles, exactly what umx_infer_image_dev
    // runs), so that pipelining the transfers does not change a single kernel launch
    const int T = g.npr * g.npc;
    const int S = std::max(1, (T + ctx->max_batch - 1) / ctx->max_batch);
    // A synchronous call on a slide of one launch group has nothing to overlap: its upload, kernels and download go down ONE
    // stream in order, without the copy streams and the events that hand slabs from one stream to the next (a 1024 x 1024
    // slide is a 1 - 3 ms call).  Submitted calls keep the copy streams: slide i+1's upload rides under slide i's kernels --
    // unless the slide is so small (< 0.6 TFLOP, about 2 ms of kernels: the legacy model's 1024 x 1024 slide is 1 ms) that the
    // cross-stream events cost more than the overlap returns: that call measured 1.03 ms on some boxes and 1.54 on others through
    // the copy streams, 1.02 as one in-order stream.
    double slide_flops = 0.0;
    for (const Launch& L : ctx->plan) slide_flops += L.flops;
    slide_flops *= (double)T;
    const bool single = S == 1 && (sync_call || slide_flops < 0.6e12);
    const hipStream_t up_s = single ? ctx->stream : ctx->up_stream, dn_s = single ? ctx->stream : ctx->dn_stream;
    // this call's range flag: its own word, cleared in stream order in front of its kernels
    const int fw = 16 * (slot + 1);
    if (ctx->d_flag) HIP_TRY(ctx, hipMemsetAsync(ctx->d_flag + fw, 0, sizeof(int), ctx->stream));
    ctx->flag_word = fw;
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
            HIP_TRY(ctx, hipMemcpyAsync(dst, (const unsigned char*)src + off, n, hipMemcpyHostToDevice, up_s));
        }
        return UMX_OK;
    };
    auto convert = [&](int r0, int r1) -> int {   // raw rows -> float64 rows (im2double [+ rescale])
        for (int c = 0; c < C_img && src_bits && !raw_gather && r1 > r0; ++c) {
            const size_t e0 = ((size_t)c * H + r0) * W;
            HIP_TRY(ctx, launch_raw_convert(d_raw + e0 * in_b, src_bits, (size_t)(r1 - r0) * W, rescale, mm + 16 * c,
                                            hs.d_image + e0, ctx->stream));
        }
        return UMX_OK;
    };
    int up_done = 0;
    // A synchronous rescaled call whose caller handed no range in: every slab's upload is enqueued first (pinned pages: DMA), the
    // planes' (min, max) found by host threads while the rows cross the bus (3 ms per 537 MB plane), and the tile kernels start on
    // the first slab as soon as that pass is done -- instead of behind the whole upload and a device reduction (UMX_HOST_RANGE=0)
    uint32_t own_range[2 * 8];
    const uint32_t* range_in = ctx->range_in;
    std::vector<int> pre_r1;   // rows up after slab s, when the uploads were enqueued ahead
    if (src_bits && rescale && !range_in && sync_call && C_img <= 8 && !(getenv("UMX_HOST_RANGE") && atoi(getenv("UMX_HOST_RANGE")) == 0)) {
        pre_r1.resize(S);
        for (int s = 0; s < S; ++s) {
            const int r1 = s == S - 1 ? H : rows_needed((tcut[s + 1] - 1) / g.npc + 1);
            if (r1 > up_done) {
                if ((rc = upload(up_done, r1))) return rc;
                if (!single) HIP_TRY(ctx, hipEventRecord(ev_up[s], ctx->up_stream));
                up_done = r1;
            }
            pre_r1[s] = up_done;
        }
        for (int c = 0; c < C_img; ++c)
            if (umx_plane_range((const unsigned char*)src + (size_t)c * plane * in_b, src_bits, plane, own_range + 2 * c) != UMX_OK)
                return fail(ctx, UMX_ERR_INVALID, "plane range");
        range_in = own_range;
    }
    if (src_bits) {
        for (int c = 0; c < C_img; ++c) HIP_TRY(ctx, launch_minmax_init(mm + 16 * c, ctx->stream));
        if (rescale && range_in) {
            // the caller's file reader saw every sample and hands the planes' (min, max) in: the words the reduction below would
            // leave, so the slide goes up slab by slab under the tile kernels like an un-rescaled one (a synchronous rescaled
            // call otherwise spends the whole upload -- 21 ms for the 1.07 GB metric slide -- in front of its first tile)
            for (int c = 0; c < C_img; ++c) {
                HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)(mm + 16 * c), (int)range_in[2 * c], 1, ctx->stream));
                HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)(mm + 16 * c + 1), (int)range_in[2 * c + 1], 1, ctx->stream));
            }
        } else if (rescale) {   // whole planes first: min / max per plane, reduced as the slabs arrive
            for (int s = 0; s < S; ++s) {
                const int r1 = s == S - 1 ? H : rows_needed((tcut[s + 1] - 1) / g.npc + 1);
                if (r1 <= up_done) continue;
                if ((rc = upload(up_done, r1))) return rc;
                if (!single) {
                    HIP_TRY(ctx, hipEventRecord(ev_up[s], ctx->up_stream));
                    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ev_up[s], 0));
                }
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
        if (!pre_r1.empty()) {   // uploaded ahead: wait for this slab's rows, convert them
            const int r0 = s ? pre_r1[s - 1] : 0;
            if (pre_r1[s] > r0) {
                if (!single) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ev_up[s], 0));
                if ((rc = convert(r0, pre_r1[s]))) return rc;
            }
        } else if (r1 > up_done) {
            if ((rc = upload(up_done, r1))) return rc;
            if (!single) {
                HIP_TRY(ctx, hipEventRecord(ev_up[s], ctx->up_stream));
                HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ev_up[s], 0));
            }
            if ((rc = convert(up_done, r1))) return rc;
            up_done = r1;
        }
        if (tcut[s + 1] > tcut[s]) {
            float* const pr = hs.d_probs + (size_t)tcut[s] * g.P * g.P * K;
            if ((rc = tiles_range(ctx, raw_gather ? nullptr : hs.d_image, C_img, g, 0, H, mean, stdv, tcut[s], tcut[s + 1], pr, raw_gather ? d_raw : nullptr,
                                  raw_gather ? src_bits : 0, raw_gather && rescale ? mm : nullptr)))
                return rc;
        }
        // image rows no later tile touches: below the first incomplete patch row
        const int y1 = s == S - 1 ? H : std::max(y_done, std::min(H, cut[s + 1] * g.sub - g.margin));
        if (y1 > y_done && cut[s + 1] > 0) {
            // the stitch writes a compact slab [K][rows][W]; slabs sit one after the other in the device buffer
            const size_t rows = (size_t)(y1 - y_done), slab_e = K * (size_t)y_done * W;
            unsigned char* const d_slab = base + slab_e * oel;
            // (uint8 out: the cast rides in the stitch -- no float16 slab is written and read back)
            if ((rc = stitch_rows(ctx, hs.d_probs, 0, cut[s + 1], H, W, mode, out_u8 ? kStitchU8 : stitch, y_done, y1,
                                  out_u8 ? (void*)(base + pm_b + slab_e) : (void*)d_slab, 0)))
                return rc;
            if (!single) {
                HIP_TRY(ctx, hipEventRecord(ev_dn[s], ctx->stream));
                HIP_TRY(ctx, hipStreamWaitEvent(ctx->dn_stream, ev_dn[s], 0));
            }
            const size_t el = out_u8 ? 1 : oel;
            const unsigned char* const dsrc = out_u8 ? base + pm_b + slab_e : d_slab;
            for (size_t k = 0; k < K; ++k)
                HIP_TRY(ctx, hipMemcpyAsync((unsigned char*)out_host + (k * plane + (size_t)y_done * W) * el,
                                            dsrc + k * rows * W * el, rows * W * el, hipMemcpyDeviceToHost, dn_s));
            y_done = y1;
        }
    }
    // the range flag of the split-precision path rides down behind the last planes; `done` then says the call is complete
    // (every upload precedes a kernel that precedes a download on the download stream)
    if (ctx->d_flag) {
        if (!single) {
            hipEvent_t ev_f = hs.events[2 * S];
            HIP_TRY(ctx, hipEventRecord(ev_f, ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->dn_stream, ev_f, 0));
        }
        HIP_TRY(ctx, hipMemcpyAsync(hs.flag_host, ctx->d_flag + fw, sizeof(int), hipMemcpyDeviceToHost, dn_s));
    } else {
        *hs.flag_host = 0;
    }
    HIP_TRY(ctx, hipEventRecord(hs.done, dn_s));
    hs.busy = true;
    return UMX_OK;
}

static int host_submit(umx_ctx* ctx, int slot, bool sync_call, const void* src, int src_bits, int C_img, int H, int W, int rescale,
                       double mean, double stdv, int mode, int stitch, int out_u8, void* out_host) {
    if (slot < 0 || slot > 1) return fail(ctx, UMX_ERR_INVALID, "slot must be 0 or 1");
    if (ctx->hs[slot].busy) return fail(ctx, UMX_ERR_INVALID, "slot %d still holds a submitted call: wait for it first", slot);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rc = host_submit_impl(ctx, slot, sync_call, src, src_bits, C_img, H, W, rescale, mean, stdv, mode, stitch, out_u8, out_host);
    ctx->flag_word = 0;
    if (rc) {
        // an error in the middle of enqueueing: transfers that reference the caller's buffers and this slot's device buffers may
        // be in flight -- drain them before the caller (or the next submit) frees or reuses anything
        const std::string msg = ctx->err;
        if (ctx->up_stream) hipStreamSynchronize(ctx->up_stream);
        hipStreamSynchronize(ctx->stream);
        if (ctx->dn_stream) hipStreamSynchronize(ctx->dn_stream);
        ctx->err = msg;
    }
    return rc;
}

static int infer_host(umx_ctx* ctx, const void* src, int src_bits, int C_img, int H, int W, int rescale, double mean,
                      double stdv, int mode, int stitch, int out_u8, void* out_host) {
    int rc = host_submit(ctx, 0, true, src, src_bits, C_img, H, W, rescale, mean, stdv, mode, stitch, out_u8, out_host);
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

int umx_plane_range(const void* raw_host, int bits, size_t n, uint32_t* range) {
    if (!raw_host || !range || n == 0 || (bits != 8 && bits != 16)) return UMX_ERR_INVALID;
    if (bits == 16) range_threads(static_cast<const uint16_t*>(raw_host), n, range);
    else range_threads(static_cast<const uint8_t*>(raw_host), n, range);
    return UMX_OK;
}

int umx_infer_image_raw_range(umx_ctx* ctx, const void* raw_host, int bits, int C_img, int H, int W, const uint32_t* range,
                              double mean, double stdv, int mode, uint8_t* out_host) {
    if (!ctx) return fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    if (!range) return umx_infer_image_raw(ctx, raw_host, bits, C_img, H, W, 1, mean, stdv, mode, out_host);
    if (C_img < 1) return fail(ctx, UMX_ERR_INVALID, "bad image/out/H/W");
    const uint32_t top = bits == 8 ? 255u : 65535u;
    for (int c = 0; c < C_img; ++c)
        if (range[2 * c] > range[2 * c + 1] || range[2 * c + 1] > top)
            return fail(ctx, UMX_ERR_INVALID, "plane %d: range (%u, %u) is not a (min, max) of %d-bit samples", c, range[2 * c],
                        range[2 * c + 1], bits);
    ctx->range_in = range;
    const int rc = umx_infer_image_raw(ctx, raw_host, bits, C_img, H, W, 1, mean, stdv, mode, out_host);
    ctx->range_in = nullptr;
    return rc;
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
    return host_submit(ctx, slot, false, raw_host, bits, C_img, H, W, rescale, mean, stdv, mode, UMX_STITCH_FP16_COMPAT, 1, out_host);
}

int umx_infer_image_wait(umx_ctx* ctx, int slot) {
    if (!ctx) return fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    return host_wait(ctx, slot);
}

}  // extern "C"
