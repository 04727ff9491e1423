// Internal host-side interface of libumx (gfx950 only): the graph / plan / context structures shared by the graph builder
// (umx_graph.hip), the split-precision planner (umx_plan.hip), the engine and C ABI (umx_engine.hip) and the host pipeline
// (umx_host.hip).  Nothing here is part of the C ABI (include/umx.h).
#pragma once
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


namespace umx {

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
    // (training) per phase, [tap][octet]: 1 = every weight of this (tap, 8 input channels) pair is a structural zero -- the input
    // gradient of a stride-2 transposed convolution on the space-to-depth tensor has 9 real (window position, parity block) pairs
    // of 16 -- and the planner leaves the pair out of the K loop.  Empty: none.
    std::vector<unsigned char> dead[4];
};

// Training (umx_train.hip): where one 16-byte unit of a packed weight image comes from.  The filters change every step, so the
// planner records, instead of values, for every (k-step, N-tile, lane) unit of a launch's weight slab the 8 source elements of the
// fp32 operand the trainer repacks on the device ([tap][Cp][Np], the layout Group::packed has on the host): element e of the unit =
// arr[base + e * stride] for e < nvalid, else 0.
struct HWRef {
    int dst;                   // uint4 index of the unit's hi image inside the (phase list's) slab; its lo image is 64 units on
    int base;                  // first source element
    unsigned short nvalid;     // 1..8 real input channels in the octet
    unsigned short arr;        // operand group the unit reads
};

struct Launch {
    std::string name;
    bool train = false;        // split-precision plan for the trainer: plain / per-phase kernels only, fp32 output, weights by reference
    std::vector<HWRef> wrefs[4];   // (train) per stage list
    size_t slab_units[4] = {0, 0, 0, 0};   // (train) uint4 units of each list's weight slab (all N-blocks)
    std::vector<HStage> stages_host;       // (train) the stage table as uploaded: the trainer cuts it between halo chunks (K split)
    bool head = false;
    int ngroups = 0;
    Group g[2];
    int nphase = 1, o_mul = 1;
    int oy_off[4] = {0, 0, 0, 0}, ox_off[4] = {0, 0, 0, 0};
    int H = 0, W = 0, Cout = 0;
    int dst = -1, outH = 0, outW = 0, pool = 0, act = 0;
    std::vector<float> pre_s, pre_b, post_s, post_b;  // size Cout or empty
    // split-precision plan only: this launch's epilogue also drops `app_C` (<= 2) channels of buffer `app_src` (same pixel grid as
    // its output) into the spare channels [app_c0, app_c0 + app_C) of its last stored octet -- the raw-input skip of the top
    // up-layer rides in the up-sampled tensor, so that layer's convolution reads ONE 5-octet tensor instead of 1 + 5 octets
    int app_src = -1, app_C = 0, app_c0 = 0;
    int bn = 0;               // where the BatchNorm affine sits: 0 none, 1 before the activation (pre_*), 2 after it (post_*)
    int summed_shortcut = 0;  // > 0: a same-source shortcut filter of this size is summed into the main filter (exact algebra)
    // head only
    std::vector<float> head_w;  // [C][K]
    int head_C = 0, head_K = 0;
    // derived
    int nt = 1, Np = 16, hpix = 2;
    double flops = 0.0;       // algorithmic FLOPs per tile (per image of the batch)
    double exec_flops = 0.0;  // executed incl. channel/N padding
    double bytes = 0.0;       // compulsory HBM bytes per tile: sources + destination (weights excluded)
    // device
    ConvParams cp;
    // split-precision plan (UMX_PREC_F16X3)
    HConvParams hcp;
    FirstParams first;        // dense-K plan of the first down-sampling layer (use_first; umx_conv_first.hip)
    bool use_first = false;
    // depth-to-space form of a stride-2 transposed convolution (make_d2s): nphase = number of N-blocks, each a plain convolution
    // over the union of its phases' taps with the N axis [phase slot][full octets] (+ one remainder tile of <= 4 channels per phase)
    int d2s = 0;              // 1: rewritten
    int d2s_Cout = 0;         // the real output channels (Cout holds the N extent of one block)
    int d2s_F = 0, d2s_R = 0; // full octets / remainder channels per phase
    int d2s_npb = 0;          // phases per block (4 or 2)
    int d2s_oy[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, d2s_ox[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};   // sub-pixel offset of (block, phase slot)
    unsigned char d2s_tapmask[2][16] = {{0}, {0}};   // per block and tap of its window: bit j set = phase slot j has that tap
    int nt16 = 1;             // N-tiles per workgroup of the split-precision kernel
    int force_nt16 = 0;       // > 0: the planner's N-tile choice is overridden (its own trial of narrower N-blocks)
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
    int64_t launches = 0, seen = 0;
    double total_ms = 0.0, flops = 0.0, bytes = 0.0, exec = 0.0;
    int xcd_order = 0;   // of the site's last launch (conv_f16x3 only)
};

struct PendingEvent {
    int site;
    hipEvent_t a, b;
};

}  // namespace umx

struct umx_ctx {
    using Launch = umx::Launch;
    using Buffer = umx::Buffer;
    using ProfSite = umx::ProfSite;
    using PendingEvent = umx::PendingEvent;
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
    const uint32_t* range_in = nullptr;   // (during umx_infer_image_raw_range) the planes' (min, max) as the caller's reader found them
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
    int prof = 0;               // 0 off, 1 every launch, N >= 2 every N-th launch of a site bracketed by events
    std::vector<ProfSite> sites;
    std::vector<PendingEvent> pending;
    std::vector<hipEvent_t> free_events;
    int site_gather = -1, site_stitch = -1, site_split = -1;
    // precision
    int precision = UMX_PREC_F16X3;
    bool f6 = false;            // split precision with the cross terms of the deep layers on the block-scaled fp6 matrix instruction (UMX_PREC_F16X3_F6)
    bool f6_used = false;       // ... and at least one layer of this model runs in that form (what umx_precision_of reports)
    int act_shift = 0;          // activations are stored times 2^act_shift in the (hi, lo) binary16 form
    float* d_tiles32 = nullptr; // fp32 staging of gathered tiles before the split (f16 path)
    int* d_flag = nullptr;      // binary16 range overflow flags (64 words): word 0 for the synchronous entry points, words 16 and
                                // 32 for the two slots of the submit / wait API -- a flag is cleared and read in stream order by
                                // the call that owns it, never from the host while another call is in flight
    int flag_word = 0;          // the word the launches being enqueued report to
    int in_cw = 0;              // > 0: the input tiles (buffer 0) are stored in the compact form [pixel]{hi[in_cw] | lo[in_cw]} -- their
                                // only readers are the dense-K first layer and the raw-skip append of the top transposed convolution
    uint4* d_zeros = nullptr;
    bool head_fused = false;
    Launch split_launch;
};

namespace umx {

extern thread_local std::string g_err;

inline std::vector<Buffer>& cur_bufs(umx_ctx* ctx) { return ctx->lane ? ctx->bufs2 : ctx->bufs; }
inline hipStream_t run_stream(umx_ctx* ctx) { return ctx->lane ? ctx->stream2 : ctx->stream; }

int fail(umx_ctx* ctx, int code, const char* fmt, ...);

#define HIP_TRY(ctx, expr)                                                                                  \
    do {                                                                                                    \
        hipError_t e__ = (expr);                                                                            \
        if (e__ != hipSuccess)                                                                              \
            return fail(ctx, e__ == hipErrorOutOfMemory ? UMX_ERR_OOM : UMX_ERR_HIP, "%s failed: %s", #expr, \
                        hipGetErrorString(e__));                                                            \
    } while (0)


inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// ---- umx_graph.hip: hyper-parameters -> launch list (reference UnMicst1-5.py:55-237, UnMicst.py:51-187)
int check_hp(const umx_hparams* hp, std::string* why);
std::vector<int> widths(const umx_hparams& hp);
size_t blob_floats_needed(const umx_hparams& hp);
// builds the launch list and the activation-buffer list (per-tile floats, (spatial size, channels)); blob may be NULL
// (describe only); *pos = floats of the blob consumed
int build_graph(const umx_hparams& hp, const float* blob, std::vector<Launch>* plan, std::vector<size_t>* buf_floats,
                std::vector<std::pair<int, int>>* buf_geom, size_t* pos, bool fold_top_skip = false);
bool conv_geometry(Launch& L, std::string* why);

// ---- umx_plan.hip: split-precision plan of one launch
int mx_pack_e2m3(const double (&v)[32], double amax, unsigned char (&out)[24]);   // umx_plan.hip: one OCP MX fp6 (e2m3) block of 32
int plan_f16(umx_ctx* ctx, Launch& L, int act_shift, bool out_f32, const Launch* head, std::string* why, bool dry = false);
// dense-K plan of the first down-sampling layer, for a launch plan_f16 has just planned (sets L.use_first when it applies)
int plan_first(umx_ctx* ctx, Launch& L, int act_shift, std::string* why);
// depth-to-space rewrite of a narrow stride-2 transposed convolution (before conv_geometry / plan_f16); true if L was rewritten
bool make_d2s(Launch& L);
// the same decision from the hyper-parameters alone (the graph builder folds the raw skip only when the first layer takes this kernel)
bool conv_first_eligible(const umx_hparams& hp);

// ---- umx_engine.hip
int dev_alloc(umx_ctx* ctx, void** out, size_t bytes);
int upload(umx_ctx* ctx, const std::vector<float>& h, float** out);
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

int grow(umx_ctx* ctx, void** buf, size_t* cap, size_t bytes);
int site_of(umx_ctx* ctx, const std::string& name, const std::string& kernel);
int prof_fold(umx_ctx* ctx);
struct ProfScope {
    umx_ctx* ctx;
    int site;
    hipEvent_t a = nullptr, b = nullptr;
    bool on;
    ProfScope(umx_ctx* c, int s, double flops, double bytes, double exec = 0.0) : ctx(c), site(s), on(c->prof && s >= 0) {
        if (!on) return;
        const int64_t k = ctx->sites[site].seen++;
        if (ctx->prof > 1 && k % ctx->prof != 0) { on = false; return; }   // sampling: this launch runs unbracketed
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

inline _Float16* hi_of(const Buffer& b) { return reinterpret_cast<_Float16*>(b.d); }
inline _Float16* lo_of(const Buffer& b, int n) { return reinterpret_cast<_Float16*>(b.d) + (size_t)n * b.S * b.S * b.Cs; }

int check_range_flag(umx_ctx* ctx);   // call with the stream idle
// umx_stitch_dev's body with the internal forms: stitch = kStitchU8 (uint8 planes), plane_rows > 0 (rows per class plane of `out_dev`)
int stitch_rows(umx_ctx* ctx, const float* probs_dev, int tpr0, int tpr1, int H, int W, int mode, int stitch, int y0, int y1,
                void* out_dev, int plane_rows);
TileGeom geom_of(const umx_hparams& hp, int H, int W);
// tiles [t0, t1) of the slide (row-major tile index) -> probs_dev (tile t0 first): gather + normalise + UNet
// (raw_dev / raw_bits: the same planes as raw uint8 / uint16 values, for an engine that gathers from them -- gathers_raw())
int tiles_range(umx_ctx* ctx, const double* image_dev, int C_img, const TileGeom& g, int band_row0, int band_rows,
                double mean, double stdv, int t0, int t1, float* probs_dev, const void* raw_dev = nullptr, int raw_bits = 0,
                const unsigned* mm_dev = nullptr /* raw planes: rescale_intensity to each plane's (min, max) words, 16 words apart */);
// the tile gather of this engine can read raw integer planes (im2double in the gather: no float64 image is written or read)
bool gathers_raw(const umx_ctx* ctx);

}  // namespace umx
