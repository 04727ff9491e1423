// Whole-slide inference sharded over the GPUs of one node INSIDE the library: one process per GPU, RCCL over xGMI.
// (SURVEY.md section 8(b) lists umx_infer_image_sharded in the C ABI; the reference itself is single-device,
// UnMicst1-5.py:769.)  The schedule is the one unmicst_amd/sharding.py runs through torch.distributed -- and which the
// gloo tests (tests/test_sharding_cpu.py, 2 and 3 ranks) and the 2/3-ranks-on-one-GPU test check bit for bit against a
// single process:
//   * patch rows are split into contiguous bands, one per rank; a rank holds only the image rows its tiles read;
//   * the LAST patch row of a band is computed first and sent to the next rank (ncclSend / ncclRecv on a communication
//     stream, hidden under the rest of the band's tiles): the 2*margin image rows below a band boundary are covered by
//     patch rows pb-1 and pb;
//   * the band is computed in slabs of patch rows; as soon as a slab's image rows are final they are stitched (tiles visited
//     in ascending global index: bit-identical to a one-GPU run) and all-gathered (ncclAllGather on equal-size padded slabs)
//     while the next slab's tiles run; the gathered rows are scattered into the [K, H, W] result every rank ends up with.
// RCCL is not linked: the few nccl* entry points are resolved with dlopen / dlsym when a context is given a communicator,
// from the librccl the process already holds (a Python caller has torch's) or from ROCm's.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "umx_internal.h"   // (same shared object: the raw entry stages its band like the host path of umx_host.hip)

// internal accessors exported by umx_engine.hip (hidden visibility: same shared object only)
hipStream_t umx_internal_stream(umx_ctx* ctx);
int umx_internal_device(umx_ctx* ctx);
void umx_internal_hp(const umx_ctx* ctx, umx_hparams* out);
int umx_internal_fail(umx_ctx* ctx, int code, const char* msg);
void umx_internal_set_destroy_hook(void (*hook)(umx_ctx*));
void umx_internal_set_wait_hook(int (*hook)(umx_ctx*, hipEvent_t));

namespace {

struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;   // (optional: failure detection)
    decltype(&ncclCommAbort) CommAbort = nullptr;
    std::string err;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names)
            if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // the copy the process already uses (torch's)
        if (!r.h)
            if (const char* e = getenv("UMX_RCCL_PATH")) r.h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
        for (const char* n : names)
            if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.h) r.h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!r.h) { r.err = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?"); return; }
#define UMX_SYM(f) r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.h, "nccl" #f)); if (!r.f) r.err = "librccl lacks nccl" #f;
        UMX_SYM(GetUniqueId) UMX_SYM(CommInitRank) UMX_SYM(CommDestroy) UMX_SYM(Send) UMX_SYM(Recv) UMX_SYM(AllGather)
        UMX_SYM(GroupStart) UMX_SYM(GroupEnd) UMX_SYM(GetErrorString)
#undef UMX_SYM
        r.CommGetAsyncError = reinterpret_cast<decltype(r.CommGetAsyncError)>(dlsym(r.h, "ncclCommGetAsyncError"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.h, "ncclCommAbort"));
    });
    return &r;
}

struct Buf { void* d = nullptr; size_t cap = 0; };

// Every inter-rank byte of the schedule below goes through this table -- the umx_shard_transport of include/umx.h.  umx_shard_init
// fills it with RCCL (stream-ordered ncclSend / ncclRecv / ncclAllGather on the context's communicator); umx_shard_init_transport
// takes the caller's: how the band / halo / scatter code runs in worlds of 2 and 3 on ONE GPU in the test-suite (RCCL refuses two
// ranks on a device), with the messages staged through host memory.
int rccl_send(void* user, const void* dev, size_t bytes, int peer, void* stream);
int rccl_recv(void* user, void* dev, size_t bytes, int peer, void* stream);
int rccl_all_gather(void* user, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream);
int rccl_group_start(void* user);
int rccl_group_end(void* user);

struct Shard {
    ncclComm_t comm = nullptr;
    umx_shard_transport tp = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool ready = false;
    bool failed = false;          // a transport call failed, a peer's failure was seen or a wait timed out: the communicator (RCCL) is
                                  // aborted and every later call on this context is refused until umx_shard_fini + a new umx_shard_init
    std::string tp_err;           // message of the last failed transport call (RCCL: ncclGetErrorString)
    int rank = 0, world = 1;
    hipStream_t comm_stream = nullptr;
    std::vector<hipEvent_t> events[2];   // per call slot (the raw entry keeps two slides in flight; the device entry uses slot 0)
    Buf probs;                           // written and read in the order of the context's stream only
    Buf gathered[2], full_u8[2];         // per slot: the communication stream of slide i may still read them under slide i+1's tiles
    std::vector<Buf> send[2];
};

std::map<umx_ctx*, Shard> g_shards;
std::mutex g_mu;
int shard_wait(umx_ctx* ctx, hipEvent_t ev);   // (below: the bounded wait of a submitted sharded call)

int fail(umx_ctx* ctx, int code, const std::string& msg) { return umx_internal_fail(ctx, code, msg.c_str()); }

#define S_HIP(ctx, expr)                                                                                    \
    do {                                                                                                    \
        hipError_t e__ = (expr);                                                                            \
        if (e__ != hipSuccess) return fail(ctx, e__ == hipErrorOutOfMemory ? UMX_ERR_OOM : UMX_ERR_HIP,     \
                                           std::string(#expr " failed: ") + hipGetErrorString(e__));        \
    } while (0)
#define S_NCCL(ctx, expr)                                                                                   \
    do {                                                                                                    \
        ncclResult_t r__ = (expr);                                                                          \
        if (r__ != ncclSuccess) return fail(ctx, UMX_ERR_HIP, std::string(#expr " failed: ") + rccl()->GetErrorString(r__)); \
    } while (0)

int grow(umx_ctx* ctx, Buf* b, size_t bytes) {
    if (b->cap >= bytes) return UMX_OK;
    if (b->d) { S_HIP(ctx, hipDeviceSynchronize()); S_HIP(ctx, hipFree(b->d)); b->d = nullptr; b->cap = 0; }
    S_HIP(ctx, hipMalloc(&b->d, bytes ? bytes : 16));
    b->cap = bytes;
    return UMX_OK;
}

// After a failure the collectives already enqueued can wait for peers that will never arrive: ncclCommAbort makes them return, so that
// the streams drain and the peers' own waits see an error instead of blocking (they poll ncclCommGetAsyncError, shard_wait below).
void shard_abort(Shard& s) {
    s.failed = true;
    if (s.comm && rccl()->CommAbort) { rccl()->CommAbort(s.comm); s.comm = nullptr; }
}

void release(Shard& s) {
    if (s.comm && rccl()->CommDestroy) rccl()->CommDestroy(s.comm);
    if (s.comm_stream) hipStreamDestroy(s.comm_stream);
    for (auto& ev : s.events) for (auto e : ev) hipEventDestroy(e);
    for (Buf* b : {&s.probs, &s.gathered[0], &s.gathered[1], &s.full_u8[0], &s.full_u8[1]}) if (b->d) hipFree(b->d);
    for (auto& v : s.send) for (auto& b : v) if (b.d) hipFree(b.d);
}

int rccl_fail(Shard* s, ncclResult_t r, const char* what) {
    s->tp_err = std::string(what) + " failed: " + rccl()->GetErrorString(r);
    return 1;
}
int rccl_send(void* user, const void* dev, size_t bytes, int peer, void* stream) {
    Shard* s = static_cast<Shard*>(user);
    const ncclResult_t r = rccl()->Send(dev, bytes, ncclUint8, peer, s->comm, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : rccl_fail(s, r, "ncclSend");
}
int rccl_recv(void* user, void* dev, size_t bytes, int peer, void* stream) {
    Shard* s = static_cast<Shard*>(user);
    const ncclResult_t r = rccl()->Recv(dev, bytes, ncclUint8, peer, s->comm, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : rccl_fail(s, r, "ncclRecv");
}
int rccl_all_gather(void* user, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream) {
    Shard* s = static_cast<Shard*>(user);
    const ncclResult_t r = rccl()->AllGather(send_dev, recv_dev, bytes_per_rank, ncclUint8, s->comm, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : rccl_fail(s, r, "ncclAllGather");
}
int rccl_group_start(void* user) {
    const ncclResult_t r = rccl()->GroupStart();
    return r == ncclSuccess ? 0 : rccl_fail(static_cast<Shard*>(user), r, "ncclGroupStart");
}
int rccl_group_end(void* user) {
    const ncclResult_t r = rccl()->GroupEnd();
    return r == ncclSuccess ? 0 : rccl_fail(static_cast<Shard*>(user), r, "ncclGroupEnd");
}

void on_destroy(umx_ctx* ctx) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_shards.find(ctx);
    if (it == g_shards.end()) return;
    release(it->second);
    g_shards.erase(it);
}

}  // namespace

// ---- band geometry (the C twin of unmicst_amd/sharding.py: band_partition / owned_rows / needed_image_rows / slab_rows;
// tests/test_abi.py compares the two over a grid of sizes)
namespace umx_geom {

void band(int npr, int world, int rank, int* pa, int* pb) {
    const int active = std::min(world, npr);
    const int base = active ? npr / active : 0, extra = active ? npr % active : 0;
    int start = 0;
    for (int r = 0; r <= rank; ++r) {
        const int n = r < active ? base + (r < extra ? 1 : 0) : 0;
        if (r == rank) { *pa = start; *pb = start + n; }
        start += n;
    }
}

void owned(int pa, int pb, int npr, int sub, int margin, int H, int* y0, int* y1) {
    if (pa >= pb) { *y0 = *y1 = 0; return; }
    *y0 = pa == 0 ? 0 : std::min(H, std::max(0, pa * sub - margin));
    *y1 = pb == npr ? H : std::min(H, std::max(0, pb * sub - margin));
}

void needed(int pa, int pb, int sub, int margin, int patch, int H, int* r0, int* r1) {
    if (pa >= pb) { *r0 = *r1 = 0; return; }
    *r0 = std::max(0, pa * sub - margin);
    *r1 = std::min(H, (pb - 1) * sub + patch - margin);
}

int cut(int pa, int pb, int n, int i) { return pa + (int)(((long long)(pb - pa) * i) / n); }

void slab(int pa, int pb, int npr, int sub, int margin, int H, int n, int i, int* s0, int* s1) {
    if (pa >= pb) { *s0 = *s1 = 0; return; }
    int y0, y1;
    owned(pa, pb, npr, sub, margin, H, &y0, &y1);
    const int a = i == 0 ? y0 : std::min(y1, std::max(y0, cut(pa, pb, n, i) * sub - margin));
    const int b = i == n - 1 ? y1 : std::min(y1, std::max(y0, cut(pa, pb, n, i + 1) * sub - margin));
    *s0 = a;
    *s1 = std::max(a, b);
}

}  // namespace umx_geom

extern "C" {

int umx_shard_unique_id(umx_unique_id* out) {
    if (!out) return umx_internal_fail(nullptr, UMX_ERR_INVALID, "out is NULL");
    Rccl* r = rccl();
    if (!r->err.empty()) return umx_internal_fail(nullptr, UMX_ERR_HIP, r->err.c_str());
    static_assert(sizeof(umx_unique_id) == sizeof(ncclUniqueId), "umx_unique_id must be ncclUniqueId-sized");
    ncclUniqueId id;
    ncclResult_t rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess) return umx_internal_fail(nullptr, UMX_ERR_HIP, r->GetErrorString(rc));
    memcpy(out, &id, sizeof id);
    return UMX_OK;
}

int umx_shard_init(umx_ctx* ctx, const umx_unique_id* id, int rank, int world) {
    if (!ctx || !id) return umx_internal_fail(ctx, UMX_ERR_INVALID, "ctx / id is NULL");
    if (world < 1 || rank < 0 || rank >= world) return umx_internal_fail(ctx, UMX_ERR_INVALID, "bad rank / world");
    Rccl* r = rccl();
    if (!r->err.empty()) return umx_internal_fail(ctx, UMX_ERR_HIP, r->err.c_str());
    S_HIP(ctx, hipSetDevice(umx_internal_device(ctx)));
    std::lock_guard<std::mutex> lk(g_mu);
    umx_internal_set_destroy_hook(on_destroy);
    umx_internal_set_wait_hook(shard_wait);
    Shard& s = g_shards[ctx];
    if (s.ready) { release(s); s = Shard(); }
    ncclUniqueId nid;
    memcpy(&nid, id, sizeof nid);
    S_NCCL(ctx, r->CommInitRank(&s.comm, world, nid, rank));
    s.rank = rank;
    s.world = world;
    s.tp = umx_shard_transport{&s, rccl_send, rccl_recv, rccl_all_gather, rccl_group_start, rccl_group_end};   // (map nodes do not move)
    s.ready = true;
    S_HIP(ctx, hipStreamCreateWithFlags(&s.comm_stream, hipStreamNonBlocking));
    return UMX_OK;
}

int umx_shard_init_transport(umx_ctx* ctx, const umx_shard_transport* tp, int rank, int world) {
    if (!ctx || !tp) return umx_internal_fail(ctx, UMX_ERR_INVALID, "ctx / transport is NULL");
    if (!tp->send || !tp->recv || !tp->all_gather) return umx_internal_fail(ctx, UMX_ERR_INVALID, "the transport lacks send / recv / all_gather");
    if (world < 1 || rank < 0 || rank >= world) return umx_internal_fail(ctx, UMX_ERR_INVALID, "bad rank / world");
    S_HIP(ctx, hipSetDevice(umx_internal_device(ctx)));
    std::lock_guard<std::mutex> lk(g_mu);
    umx_internal_set_destroy_hook(on_destroy);
    umx_internal_set_wait_hook(shard_wait);
    Shard& s = g_shards[ctx];
    if (s.ready) { release(s); s = Shard(); }
    s.tp = *tp;
    s.rank = rank;
    s.world = world;
    s.ready = true;
    S_HIP(ctx, hipStreamCreateWithFlags(&s.comm_stream, hipStreamNonBlocking));
    return UMX_OK;
}

int umx_shard_fini(umx_ctx* ctx) {
    if (!ctx) return UMX_OK;
    on_destroy(ctx);
    return UMX_OK;
}

int umx_shard_plan(const umx_hparams* hpp, int H, int W, int rank, int world, int nslabs, int slab, int* patch_row0,
                   int* patch_row1, int* need_row0, int* need_row1, int* own_row0, int* own_row1, int* slab_row0,
                   int* slab_row1, int* nslabs_used) {
    if (!hpp || hpp->imSize < 8 || H < 1 || W < 1 || world < 1 || rank < 0 || rank >= world)
        return umx_internal_fail(nullptr, UMX_ERR_INVALID, "bad hp / H / W / rank / world");
    const umx_hparams hp = *hpp;
    const int margin = hp.imSize / 8, sub = hp.imSize - 2 * margin;
    const int npr = (H + sub - 1) / sub;           // PI2D.setup, PartitionOfImage.py:49
    int pa, pb;
    umx_geom::band(npr, world, rank, &pa, &pb);
    int n = std::max(1, nslabs);
    for (int r = 0; r < world; ++r) {
        int a, b;
        umx_geom::band(npr, world, r, &a, &b);
        if (b > a) n = std::min(n, b - a);
    }
    n = std::max(1, n);
    if (slab < 0 || slab >= n) return umx_internal_fail(nullptr, UMX_ERR_INVALID, "slab index out of range");
    int v0, v1;
    if (patch_row0) *patch_row0 = pa;
    if (patch_row1) *patch_row1 = pb;
    umx_geom::needed(pa, pb, sub, margin, hp.imSize, H, &v0, &v1);
    if (need_row0) *need_row0 = v0;
    if (need_row1) *need_row1 = v1;
    umx_geom::owned(pa, pb, npr, sub, margin, H, &v0, &v1);
    if (own_row0) *own_row0 = v0;
    if (own_row1) *own_row1 = v1;
    umx_geom::slab(pa, pb, npr, sub, margin, H, n, slab, &v0, &v1);
    if (slab_row0) *slab_row0 = v0;
    if (slab_row1) *slab_row1 = v1;
    if (nslabs_used) *nslabs_used = n;
    return UMX_OK;
}

}  // extern "C"

namespace {

// What one call of the schedule reads and writes.  Source: a float64 band on the device, or a raw integer band (the tile gather
// converts, umx_conv_f16.hip gather_split_kernel) that is still on its way up -- then `band_host` is staged piece by piece on
// `up_s` ahead of the tiles that read it.  Result: the stitched slabs as they are (fp16 / fp32), or cast to the drivers' uint8
// (UnMicst1-5.py:848-854 at the identity grid) before they are gathered; optionally this rank's own rows also go down to
// the host on `dn_s` as soon as their slab is final.
struct RunIO {
    const double* band_f64 = nullptr;
    const void* raw_dev = nullptr;
    int raw_bits = 0;
    const unsigned* mm = nullptr;          // raw source with intensity rescale: the planes' (min, max) words, 16 apart
    const void* band_host = nullptr;       // raw source: host rows to stage into raw_dev (NULL: raw_dev is complete)
    hipStream_t up_s = nullptr, dn_s = nullptr;
    int u8 = 0;
    void* out_full = nullptr;              // [K, H, W] on the device, every rank
    uint8_t* own_host = nullptr;           // u8 only: [K, own rows, W] on the host, this rank's rows
    int slot = 0;
    bool join = true;                      // the context's stream waits for the gathers at the end (device entry)
    hipEvent_t* ev_gathered = nullptr;     // out: recorded on the communication stream behind the last scatter copy
    hipEvent_t* ev_cs_end = nullptr;       // out: recorded on the context's stream behind the call's last kernel
};

int sharded_run_enqueue(umx_ctx* ctx, Shard& s, const RunIO& io, int C_img, int H, int W, int band_row0, int band_rows, double mean,
                        double stdv, int mode, int stitch, int nslabs) {
    const umx_shard_transport& tp = s.tp;
#define S_TP(ctx, expr)                                                                                       \
    do {                                                                                                      \
        s.tp_err.clear();                                                                                     \
        if ((expr) != 0) return fail(ctx, UMX_ERR_HIP, s.tp_err.empty() ? std::string(#expr " failed") : s.tp_err); \
    } while (0)
    hipStream_t cs = umx_internal_stream(ctx), ms = s.comm_stream;
    umx_hparams hp;
    umx_internal_hp(ctx, &hp);
    int npr = 0, npc = 0, rc;
    if ((rc = umx_tile_grid(ctx, H, W, &npr, &npc, nullptr, nullptr))) return rc;
    const int P = hp.imSize, K = hp.nClasses, margin = P / 8, sub = P - 2 * margin;
    const size_t sel = stitch == UMX_STITCH_FP32 ? 4 : 2;   // element of the stitched slab
    const size_t el = io.u8 ? 1 : sel;                      // element that is gathered
    const int world = s.world, rank = s.rank, slot = io.slot;
    std::vector<int> A(world), B(world);
    std::vector<int> active;
    int n = std::max(1, nslabs);
    for (int q = 0; q < world; ++q) {
        umx_geom::band(npr, world, q, &A[q], &B[q]);
        if (B[q] > A[q]) { active.push_back(q); n = std::min(n, B[q] - A[q]); }
    }
    n = std::max(1, n);
    const int pa = A[rank], pb = B[rank];
    if (s.failed) return fail(ctx, UMX_ERR_INVALID, "the sharded world of this context failed earlier (its communicator was aborted): umx_shard_fini, then umx_shard_init in every rank");
    // argument checks of both public entries (the range kernels below index the band unchecked; nothing is enqueued before these)
    if (mode != UMX_MODE_ACCUMULATE && mode != UMX_MODE_REPLACE) return fail(ctx, UMX_ERR_INVALID, "sharded entry: bad stitch mode");
    if (stitch != UMX_STITCH_FP16_COMPAT && stitch != UMX_STITCH_FP32)
        return fail(ctx, UMX_ERR_INVALID, "sharded entry: stitch must be UMX_STITCH_FP16_COMPAT or UMX_STITCH_FP32");
    if (!(stdv != 0.0) || C_img < 1 || (C_img != 1 && C_img != hp.nChannels))
        return fail(ctx, UMX_ERR_INVALID, "sharded entry: std must be non-zero and the band must hold 1 or nChannels planes");
    if (band_rows < 0 || band_row0 < 0 || band_row0 + band_rows > H) return fail(ctx, UMX_ERR_INVALID, "sharded entry: band outside the image");
    if (pa < pb) {
        if (!io.band_f64 && !io.raw_dev) return fail(ctx, UMX_ERR_INVALID, "sharded entry: NULL band");
        int need0 = 0, need1 = 0;
        umx_geom::needed(pa, pb, sub, margin, P, H, &need0, &need1);
        if (band_row0 > need0 || band_row0 + band_rows < need1)
            return fail(ctx, UMX_ERR_INVALID, "sharded entry: the band does not cover the image rows its patch rows read (umx_shard_plan: need_row0 .. need_row1)");
    }
    const bool has_prev = pa < pb && pa > 0, has_next = pa < pb && pb < npr;
    const int lo = has_prev ? pa - 1 : pa;
    const size_t tile_f = (size_t)P * P * K, row_f = tile_f * npc;
    if ((rc = grow(ctx, &s.probs, std::max<size_t>(1, (size_t)std::max(pb - lo, 0)) * row_f * sizeof(float)))) return rc;
    float* const probs = (float*)s.probs.d;
    std::vector<hipEvent_t>& evs = s.events[slot];
    const int nev = 6 + 4 * n;
    while ((int)evs.size() < nev) {
        hipEvent_t e;
        S_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        evs.push_back(e);
    }
    hipEvent_t ev_in = evs[0], ev_last = evs[1], ev_halo = evs[2], ev_done = evs[3], ev_end = evs[4];
    // the communication stream starts behind whatever the caller queued on the context's stream
    S_HIP(ctx, hipEventRecord(ev_in, cs));
    S_HIP(ctx, hipStreamWaitEvent(ms, ev_in, 0));
    // staged upload of a raw band: the rows of the band's LAST patch row go first (its tiles run first), then the rows of the slabs
    // from the top; what is up = [band_row0, head) and [tail, band end)
    const int band_end = band_row0 + band_rows;
    int head = band_row0, tail = band_end, nup = 0;
    const size_t in_b = io.raw_bits ? (size_t)io.raw_bits / 8 : sizeof(double);
    auto stage = [&](int pr0, int pr1) -> int {
        if (!io.band_host || pr1 <= pr0) return UMX_OK;
        int a = std::max(band_row0, std::max(0, pr0 * sub - margin)), b = std::min(band_end, std::min(H, (pr1 - 1) * sub + P - margin));
        a = std::max(a, head);
        b = std::min(b, tail);
        if (b <= a) return UMX_OK;
        if (a > head && b < tail) a = head;          // (never with this schedule: keep the two intervals contiguous anyway)
        for (int c = 0; c < C_img; ++c) {
            const size_t off = ((size_t)c * band_rows + (a - band_row0)) * W * in_b;
            S_HIP(ctx, hipMemcpyAsync((unsigned char*)io.raw_dev + off, (const unsigned char*)io.band_host + off, (size_t)(b - a) * W * in_b,
                                      hipMemcpyHostToDevice, io.up_s));
        }
        if (a == head) head = b; else tail = a;
        if (io.up_s != cs) {
            hipEvent_t e = evs[6 + 2 * n + std::min(nup++, 2 * n - 1)];
            S_HIP(ctx, hipEventRecord(e, io.up_s));
            S_HIP(ctx, hipStreamWaitEvent(cs, e, 0));
        }
        return UMX_OK;
    };
    const umx::TileGeom g = umx::geom_of(hp, H, W);
    // tiles [t0, t1) of the slide (row-major tile index) -> their place in `probs` (tile lo * npc first)
    auto tiles = [&](int t0, int t1) -> int {
        if (t1 <= t0) return UMX_OK;
        if ((rc = stage(t0 / g.npc, (t1 - 1) / g.npc + 1))) return rc;
        float* const dst = probs + (size_t)(t0 - lo * g.npc) * tile_f;
        return umx::tiles_range(ctx, io.raw_dev ? nullptr : io.band_f64, C_img, g, band_row0, band_rows, mean, stdv, t0, t1, dst, io.raw_dev,
                                io.raw_bits, io.mm);
    };
    // Launch groups.  The next rank waits for this band's LAST patch row, so it is computed first -- but not as a launch group of
    // its own (86 tiles of the 16384-wide slide fill a third of the chip's workgroup slots on the deep layers and still cost every
    // layer a generation): the first group is the band's last F tiles, F = the remainder of the band's tile count over the launch
    // group size (raised by whole groups until it holds the last row), and every later group is a full one, whatever slab it
    // belongs to.  11 patch rows of 86 tiles at 256 per group: 178 + 3 x 256 instead of 86 + 4 x 215 -- the same number of
    // workgroup generations as an unsharded band.  The last band (nobody waits) runs in order.
    const int Bt = std::max(1, ctx->max_batch), T0 = pa * g.npc, T1 = pb * g.npc;
    int F = 0;
    if (has_next) {
        F = (T1 - T0) % Bt;
        while (F < g.npc) F += Bt;
        F = std::min(F, T1 - T0);
    }
    if ((rc = tiles(T1 - F, T1))) return rc;
    int done = T0;   // tiles [T0, done) and [T1 - F, T1) are enqueued
    S_HIP(ctx, hipEventRecord(ev_last, cs));
    S_HIP(ctx, hipStreamWaitEvent(ms, ev_last, 0));
    if (has_next || has_prev) {
        const int me = (int)(std::find(active.begin(), active.end(), rank) - active.begin());
        if (tp.group_start) S_TP(ctx, tp.group_start(tp.user));
        if (has_next) S_TP(ctx, tp.send(tp.user, probs + (size_t)(pb - 1 - lo) * row_f, row_f * sizeof(float), active[me + 1], ms));
        if (has_prev) S_TP(ctx, tp.recv(tp.user, probs, row_f * sizeof(float), active[me - 1], ms));
        if (tp.group_end) S_TP(ctx, tp.group_end(tp.user));
    }
    S_HIP(ctx, hipEventRecord(ev_halo, ms));
    std::vector<Buf>& send = s.send[slot];
    Buf& gathered = s.gathered[slot];
    if ((int)send.size() < n) send.resize(n);
    int own0 = 0, own1 = 0;
    umx_geom::owned(pa, pb, npr, sub, margin, H, &own0, &own1);
    {   // size the gather buffers for the largest slab up front (no reallocation between enqueued operations: a grow() in the slab loop
        // would synchronise the device and free memory between enqueued collectives when a later slide is larger).  The stitch writes
        // straight into the padded send buffer [K][mx][W]; the padded tails stay uninitialised: the scatter never reads them.
        int mx_all = 1;
        for (int i = 0; i < n; ++i)
            for (int q = 0; q < world; ++q) {
                int a, b;
                umx_geom::slab(A[q], B[q], npr, sub, margin, H, n, i, &a, &b);
                mx_all = std::max(mx_all, b - a);
            }
        if ((rc = grow(ctx, &gathered, (size_t)world * K * mx_all * W * el))) return rc;
        for (int i = 0; i < n; ++i)
            if ((rc = grow(ctx, &send[i], (size_t)K * mx_all * W * el))) return rc;
    }
    for (int i = 0; i < n; ++i) {
        // the launch groups slab i still lacks: every tile of the patch rows below its last image row (full groups: one may run into the next slab)
        const int need = pa < pb ? std::min(umx_geom::cut(pa, pb, n, i + 1) * g.npc, T1 - F) : T0;
        while (done < need) {
            const int t1 = std::min(done + Bt, T1 - F);
            if ((rc = tiles(done, t1))) return rc;
            done = t1;
        }
        if (i == 0) S_HIP(ctx, hipStreamWaitEvent(cs, ev_halo, 0));   // the previous rank's last patch row feeds this band's first rows
        int s0, s1;
        umx_geom::slab(pa, pb, npr, sub, margin, H, n, i, &s0, &s1);
        std::vector<int> ra(world), rb(world);
        int mx = 1;
        for (int q = 0; q < world; ++q) {
            umx_geom::slab(A[q], B[q], npr, sub, margin, H, n, i, &ra[q], &rb[q]);
            mx = std::max(mx, rb[q] - ra[q]);
        }
        const size_t plane_b = (size_t)mx * W * el, send_b = (size_t)K * plane_b;
        // straight into the padded gather buffer [K][mx][W] (uint8 path: the drivers' cast rides in the stitch)
        if (s1 > s0 && (rc = umx::stitch_rows(ctx, probs, lo, pb, H, W, mode, io.u8 ? umx::kStitchU8 : stitch, s0, s1, send[i].d, mx))) return rc;
        hipEvent_t ev_s = evs[6 + 2 * i];
        S_HIP(ctx, hipEventRecord(ev_s, cs));
        S_HIP(ctx, hipStreamWaitEvent(ms, ev_s, 0));
        if (io.own_host && s1 > s0) {   // this rank's rows of the slab: down to the host under the next slab's tiles
            if (io.dn_s != cs) S_HIP(ctx, hipStreamWaitEvent(io.dn_s, ev_s, 0));
            const size_t own_rows = (size_t)(own1 - own0);
            for (int k = 0; k < K; ++k)
                S_HIP(ctx, hipMemcpyAsync(io.own_host + ((size_t)k * own_rows + (size_t)(s0 - own0)) * W, (char*)send[i].d + k * plane_b,
                                          (size_t)(s1 - s0) * W, hipMemcpyDeviceToHost, io.dn_s));
        }
        // (one gather buffer per slot, reused slab after slab: gather i+1 is queued behind the scatter copies of gather i)
        S_TP(ctx, tp.all_gather(tp.user, send[i].d, gathered.d, send_b, ms));
        for (int q = 0; q < world; ++q)
            for (int k = 0; k < K && rb[q] > ra[q]; ++k)
                S_HIP(ctx, hipMemcpyAsync((char*)io.out_full + ((size_t)k * H + ra[q]) * W * el,
                                          (char*)gathered.d + (size_t)q * send_b + k * plane_b,
                                          (size_t)(rb[q] - ra[q]) * W * el, hipMemcpyDeviceToDevice, ms));
    }
    S_HIP(ctx, hipEventRecord(ev_done, ms));
    if (io.join) S_HIP(ctx, hipStreamWaitEvent(cs, ev_done, 0));   // the result is complete for whatever the caller queues next
    S_HIP(ctx, hipEventRecord(ev_end, cs));
    if (io.ev_gathered) *io.ev_gathered = ev_done;
    if (io.ev_cs_end) *io.ev_cs_end = ev_end;
#undef S_TP
    return UMX_OK;
}

// An error in the middle of enqueueing: gathers and scatter copies into the caller's `out_full` may already sit on the communication
// stream -- drain it (and the context's stream, whose kernels feed it) before the caller gets its buffers back; the message survives.
int sharded_run(umx_ctx* ctx, Shard& s, const RunIO& io, int C_img, int H, int W, int band_row0, int band_rows, double mean,
                double stdv, int mode, int stitch, int nslabs) {
    const int rc = sharded_run_enqueue(ctx, s, io, C_img, H, W, band_row0, band_rows, mean, stdv, mode, stitch, nslabs);
    if (rc && rc != UMX_ERR_INVALID) {   // (UMX_ERR_INVALID: the argument checks, nothing was enqueued)
        const std::string msg = ctx->err;
        shard_abort(s);                  // peers must not wait for this rank's collectives, and ours must not wait for theirs
        hipStreamSynchronize(umx_internal_stream(ctx));
        if (s.comm_stream) hipStreamSynchronize(s.comm_stream);
        ctx->err = msg;
    }
    return rc;
}

// How a submitted sharded call is waited for (umx_infer_image_wait): the completion event is polled, and between polls the
// communicator's asynchronous error state is read (ncclCommGetAsyncError: a peer died, a link failed) and a wall-clock bound is kept
// (UMX_SHARD_TIMEOUT_S, default 600 s; 0 = none) -- either ends the wait with UMX_ERR_HIP after ncclCommAbort, never with a hang.
int shard_wait(umx_ctx* ctx, hipEvent_t ev) {
    Shard* sp = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_shards.find(ctx);
        if (it != g_shards.end()) sp = &it->second;
    }
    if (!sp || !sp->ready) {
        const hipError_t e = hipEventSynchronize(ev);
        return e == hipSuccess ? UMX_OK : fail(ctx, UMX_ERR_HIP, std::string("hipEventSynchronize failed: ") + hipGetErrorString(e));
    }
    Shard& s = *sp;
    double limit = 600.0;
    if (const char* e = getenv("UMX_SHARD_TIMEOUT_S")) limit = atof(e);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) { shard_abort(s); return fail(ctx, UMX_ERR_HIP, std::string("hipEventQuery failed: ") + hipGetErrorString(q)); }
        if ((++spins & 63u) == 0u) {
            if (s.comm && rccl()->CommGetAsyncError) {
                ncclResult_t ae = ncclSuccess;
                const ncclResult_t r = rccl()->CommGetAsyncError(s.comm, &ae);
                if (r != ncclSuccess || (ae != ncclSuccess && ae != ncclInProgress)) {
                    const std::string why = rccl()->GetErrorString(r != ncclSuccess ? r : ae);
                    shard_abort(s);
                    hipStreamSynchronize(s.comm_stream);
                    return fail(ctx, UMX_ERR_HIP, "the sharded world failed while this rank waited (RCCL: " + why + "); the communicator was aborted");
                }
            }
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (limit > 0.0 && waited > limit) {
                shard_abort(s);
                return fail(ctx, UMX_ERR_HIP, "timed out after " + std::to_string((int)waited) + " s waiting for a sharded call (UMX_SHARD_TIMEOUT_S): a peer "
                                              "is not taking part; the communicator was aborted");
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    return UMX_OK;
}

Shard* shard_of(umx_ctx* ctx) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_shards.find(ctx);
    return it == g_shards.end() || !it->second.ready ? nullptr : &it->second;
}

}  // namespace

extern "C" {

int umx_infer_image_sharded_dev(umx_ctx* ctx, const double* band_dev, int C_img, int H, int W, int band_row0, int band_rows,
                                double mean, double stdv, int mode, int stitch, int nslabs, void* out_full_dev) {
    if (!ctx) return umx_internal_fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    if (!out_full_dev || H < 1 || W < 1) return umx_internal_fail(ctx, UMX_ERR_INVALID, "bad out / H / W");
    Shard* sp = shard_of(ctx);
    if (!sp) return umx_internal_fail(ctx, UMX_ERR_INVALID, "call umx_shard_init (or umx_shard_init_transport) on this context first");
    if (ctx->hs[0].busy || ctx->hs[1].busy)   // (slot 0's gather buffers are this entry's too)
        return umx_internal_fail(ctx, UMX_ERR_INVALID, "a submitted call is still in flight on this context: wait for it first");
    S_HIP(ctx, hipSetDevice(umx_internal_device(ctx)));
    RunIO io;
    io.band_f64 = band_dev;
    io.out_full = out_full_dev;
    return sharded_run(ctx, *sp, io, C_img, H, W, band_row0, band_rows, mean, stdv, mode, stitch, nslabs);
}

// The same schedule fed the way the one-GPU line is fed (umx_infer_image_raw_submit): this rank's raw uint8 / uint16 rows come up
// from the host piece by piece on the upload stream ahead of the tiles that read them, the tile gather converts (im2double, and
// with `range` the drivers' rescale_intensity to the whole planes' (min, max), which the caller's reader knows -- a rank sees only
// its band), stitched slabs are cast to the drivers' uint8 before they are gathered (a quarter of the fp32 bytes over xGMI), and
// the rank's own rows go down to the host on the download stream under the next slab's tiles.  Two slides may be in flight.
static int sharded_raw_submit(umx_ctx* ctx, int slot, const void* band_host, int bits, int C_img, int H, int W, int band_row0,
                              int band_rows, const uint32_t* range, double mean, double stdv, int mode, int nslabs,
                              uint8_t* own_out_host, uint8_t* out_full_dev) {
    using namespace umx;
    Shard* sp = shard_of(ctx);
    if (!sp) return fail(ctx, UMX_ERR_INVALID, "call umx_shard_init (or umx_shard_init_transport) on this context first");
    Shard& s = *sp;
    umx_ctx::HostSlot& hs = ctx->hs[slot];
    if (!hs.done) {
        HIP_TRY(ctx, hipEventCreateWithFlags(&hs.done, hipEventDisableTiming));
        HIP_TRY(ctx, hipHostMalloc((void**)&hs.flag_host, 64, hipHostMallocDefault));
    }
    if (!ctx->up_stream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->dn_stream, hipStreamNonBlocking));
    }
    const size_t in_b = (size_t)bits / 8, K = ctx->hp.nClasses;
    const size_t raw_b = (size_t)C_img * std::max(band_rows, 0) * W * in_b;
    const size_t mm_off = (raw_b + 255) & ~(size_t)255;
    int rc;
    if ((rc = grow(ctx, &hs.d_out, &hs.out_cap, mm_off + 64 * (size_t)std::max(C_img, 1) + 256))) return rc;
    unsigned char* const base = (unsigned char*)hs.d_out;
    unsigned* const mm = (unsigned*)(base + mm_off);
    if (!out_full_dev) {
        if ((rc = ::grow(ctx, &s.full_u8[slot], K * (size_t)H * W))) return rc;
        out_full_dev = (uint8_t*)s.full_u8[slot].d;
    }
    const int fw = 16 * (slot + 1);
    if (ctx->d_flag) HIP_TRY(ctx, hipMemsetAsync(ctx->d_flag + fw, 0, sizeof(int), ctx->stream));
    ctx->flag_word = fw;
    RunIO io;
    io.slot = slot;
    io.u8 = 1;
    io.out_full = out_full_dev;
    io.own_host = own_out_host;
    io.up_s = ctx->up_stream;
    io.dn_s = ctx->dn_stream;
    io.join = false;
    hipEvent_t ev_gathered = nullptr, ev_end = nullptr;
    io.ev_gathered = &ev_gathered;
    io.ev_cs_end = &ev_end;
    if (range)
        for (int c = 0; c < C_img; ++c) {
            HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)(mm + 16 * c), (int)range[2 * c], 1, ctx->stream));
            HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)(mm + 16 * c + 1), (int)range[2 * c + 1], 1, ctx->stream));
        }
    if (gathers_raw(ctx)) {
        io.raw_dev = base;
        io.raw_bits = bits;
        io.mm = range ? mm : nullptr;
        io.band_host = band_host;
    } else {
        // an engine whose gather reads float64 (exact-fp32 precision): the band goes up whole and is converted first
        if ((rc = grow(ctx, (void**)&hs.d_image, &hs.image_cap, (size_t)C_img * std::max(band_rows, 1) * W * sizeof(double)))) return rc;
        if (raw_b) HIP_TRY(ctx, hipMemcpyAsync(base, band_host, raw_b, hipMemcpyHostToDevice, ctx->stream));
        for (int c = 0; c < C_img && band_rows > 0; ++c)
            HIP_TRY(ctx, launch_raw_convert(base + (size_t)c * band_rows * W * in_b, bits, (size_t)band_rows * W, range ? 1 : 0, mm + 16 * c,
                                            hs.d_image + (size_t)c * band_rows * W, ctx->stream));
        io.band_f64 = hs.d_image;
    }
    rc = sharded_run(ctx, s, io, C_img, H, W, band_row0, band_rows, mean, stdv, mode, UMX_STITCH_FP16_COMPAT, nslabs);
    ctx->flag_word = 0;
    if (rc) return rc;
    // `done`: the gathers and scatters (communication stream), this rank's downloads and the range flag behind the last kernel
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->dn_stream, ev_gathered, 0));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->dn_stream, ev_end, 0));
    if (ctx->d_flag) HIP_TRY(ctx, hipMemcpyAsync(hs.flag_host, ctx->d_flag + fw, sizeof(int), hipMemcpyDeviceToHost, ctx->dn_stream));
    else *hs.flag_host = 0;
    HIP_TRY(ctx, hipEventRecord(hs.done, ctx->dn_stream));
    hs.busy = true;
    return UMX_OK;
}

int umx_infer_image_sharded_raw_submit(umx_ctx* ctx, int slot, const void* band_host, int bits, int C_img, int H, int W,
                                       int band_row0, int band_rows, const uint32_t* range, double mean, double stdv, int mode,
                                       int nslabs, uint8_t* own_out_host, uint8_t* out_full_dev) {
    using namespace umx;
    if (!ctx) return fail(nullptr, UMX_ERR_INVALID, "ctx is NULL");
    if (slot < 0 || slot > 1) return fail(ctx, UMX_ERR_INVALID, "slot must be 0 or 1");
    if (H < 1 || W < 1 || C_img < 1 || band_rows < 0 || band_row0 < 0 || band_row0 + band_rows > H || (band_rows > 0 && !band_host))
        return fail(ctx, UMX_ERR_INVALID, "bad band / H / W");
    if (bits != 8 && bits != 16) return fail(ctx, UMX_ERR_INVALID, "raw planes must be uint8 or uint16 (bits = %d)", bits);
    if (C_img != 1 && C_img != ctx->hp.nChannels)
        return fail(ctx, UMX_ERR_INVALID, "image has %d channels, model wants 1 or %d", C_img, ctx->hp.nChannels);
    if (!(stdv != 0.0)) return fail(ctx, UMX_ERR_INVALID, "std must be non-zero");
    if (mode != UMX_MODE_ACCUMULATE && mode != UMX_MODE_REPLACE) return fail(ctx, UMX_ERR_INVALID, "bad mode %d", mode);
    const uint32_t top = bits == 8 ? 255u : 65535u;
    for (int c = 0; range && c < C_img; ++c)
        if (range[2 * c] > range[2 * c + 1] || range[2 * c + 1] > top)
            return fail(ctx, UMX_ERR_INVALID, "plane %d: range (%u, %u) is not a (min, max) of %d-bit samples", c, range[2 * c], range[2 * c + 1], bits);
    if (ctx->hs[slot].busy) return fail(ctx, UMX_ERR_INVALID, "slot %d still holds a submitted call: wait for it first", slot);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rc = sharded_raw_submit(ctx, slot, band_host, bits, C_img, H, W, band_row0, band_rows, range, mean, stdv, mode, nslabs,
                                      own_out_host, out_full_dev);
    ctx->flag_word = 0;
    if (rc) {   // an error in the middle of enqueueing: drain what references the caller's buffers before returning
        const std::string msg = ctx->err;
        if (ctx->up_stream) hipStreamSynchronize(ctx->up_stream);
        hipStreamSynchronize(ctx->stream);
        if (ctx->dn_stream) hipStreamSynchronize(ctx->dn_stream);
        ctx->err = msg;
    }
    return rc;
}

int umx_infer_image_sharded_raw(umx_ctx* ctx, const void* band_host, int bits, int C_img, int H, int W, int band_row0, int band_rows,
                                const uint32_t* range, double mean, double stdv, int mode, int nslabs, uint8_t* own_out_host,
                                uint8_t* out_full_dev) {
    const int rc = umx_infer_image_sharded_raw_submit(ctx, 0, band_host, bits, C_img, H, W, band_row0, band_rows, range, mean, stdv, mode,
                                                      nslabs, own_out_host, out_full_dev);
    return rc ? rc : umx_infer_image_wait(ctx, 0);
}

}  // extern "C"
