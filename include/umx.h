/*
 * umx.h -- C ABI of the MI355X-native UnMicst inference engine (libumx.so).
 *
 * The reference (HMS-IDAC/UnMicst) has no FFI layer: its hot path sits behind three static methods of a
 * Python namespace-class and one TensorFlow call.  Each entry point below names the reference interface it
 * replaces (file:line relative to the reference tree).  Conventions: every function returns 0 on success and a
 * non-zero umx_status otherwise (no exceptions cross the ABI); umx_last_error() gives the message; the caller
 * owns every buffer it passes; a ctx owns its device memory and stream; one ctx per host thread (a ctx is not
 * thread-safe, the library is re-entrant across ctxs).  "_dev" variants take DEVICE pointers and only enqueue
 * work on the ctx stream (call umx_synchronize); the plain variants take HOST pointers and are synchronous.
 * There is no CPU fallback: every compute entry point fails with UMX_ERR_NO_DEVICE without a gfx950 GPU.
 */
#ifndef UMX_H
#define UMX_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define UMX_API __attribute__((visibility("default")))
#else
#define UMX_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct umx_ctx umx_ctx;

enum umx_status {
    UMX_OK = 0,
    UMX_ERR_INVALID = 1,     /* bad argument / unsupported hyper-parameters */
    UMX_ERR_BLOB = 2,        /* weight blob size does not match the graph (reference: tf NotFoundError on restore) */
    UMX_ERR_NO_DEVICE = 3,   /* no usable HIP device */
    UMX_ERR_HIP = 4,         /* a HIP runtime call failed */
    UMX_ERR_OOM = 5,
    UMX_ERR_RANGE = 6        /* split-precision path only: an activation left the binary16 range (|v| >= 6e4) */
};

/* Arithmetic of the convolutions.  All hold the 1e-4 tolerance on the probability maps.
 *   UMX_PREC_F32    exact fp32 products on v_mfma_f32_16x16x4_f32 (bit-for-bit an fp32 fma chain).
 *   UMX_PREC_F16X3  every fp32 product as three binary16 MFMA products of a (hi, lo) split of both operands with fp32
 *                   accumulation (~2^-21 relative error per product; 16/3 of the fp32 matrix rate).
 *   UMX_PREC_F16X3_F6  UMX_PREC_F16X3 with the two CROSS terms (x_hi*w_lo + x_lo*w_hi, a 2^-11 correction) of the wide plain
 *                   convolutions at <= 1/4 of the input resolution (>= 2 blocks of 144 output channels: the duo widths' ld3 / ld4 /
 *                   lu3 / lu4 convolutions; no layer of the solo or legacy models) on the block-scaled matrix instruction (OCP MX fp6
 *                   e2m3, 4 x the K per instruction): half the matrix time on those layers, 5 - 13 % of their run time.  1e-6 .. 3e-6
 *                   on the GPU against the oracle, 2e-5 in emulation on the reference's trained weights
 *                   (tests/fp8_cross_term_report.py); every other layer as UMX_PREC_F16X3.
 *   UMX_PREC_DEFAULT  = UMX_PREC_F16X3_F6 where the split-precision planner covers every layer of the graph, else UMX_PREC_F32
 *                   (layers of 4x4 pixels under 7x7 filters exceed the split kernels' halo image) -- said through
 *                   umx_last_error(ctx); UMX_PRECISION=f32|f16x3|f16f6 in the environment pins it.  umx_precision_of() tells
 *                   which one a ctx runs (UMX_PREC_F16X3 when no layer of the model takes the fp6 form). */
enum umx_precision { UMX_PREC_DEFAULT = 0, UMX_PREC_F32 = 1, UMX_PREC_F16X3 = 2, UMX_PREC_F16X3_F6 = 3 };

typedef struct umx_options {
    int32_t device_ordinal;
    int32_t max_batch;       /* tiles per UNet launch group */
    int32_t precision;       /* enum umx_precision */
    int32_t act_shift;       /* F16X3: activations stored times 2^act_shift (0..8); -1 = default (0) */
    int32_t lanes;           /* 0 = default (1), 1 or 2: activation-buffer sets / streams the tile batches of one call
                                alternate between (2: the kernels of consecutive batches overlap; twice the arena) */
    int32_t reserved[11];    /* must be zero */
} umx_options;

enum umx_graph {
    UMX_GRAPH_LEGACY = 0,    /* reference UnMicst.py:51-187 (ReLU, 1x1 shortcut, BN after ReLU, no BN elsewhere) */
    UMX_GRAPH_V2 = 1         /* reference UnMicst1-5.py:55-237 == UnMicst2.py:52-235 at inference */
};

/* The reference's hp dict (UnMicst1-5.py:57-67) + which graph builder consumes it.  downSampFact is fixed at 2. */
typedef struct umx_hparams {
    int32_t graph, imSize, nChannels, nClasses, nOut0, nLayers, ks, nExtraConvs, featMapsFact;
} umx_hparams;

enum umx_mode { UMX_MODE_ACCUMULATE = 0, UMX_MODE_REPLACE = 1 };      /* PI2D.Mode, PartitionOfImage.py:75 */
enum umx_stitch { UMX_STITCH_FP16_COMPAT = 0, UMX_STITCH_FP32 = 1 }; /* fp16-compat reproduces the reference's
    float16 Output/Count accumulators bit for bit (PartitionOfImage.py:84-122); fp32 blends in double and
    returns float32 */

/*
 * Weight blob: one flat little-endian float32 array of the raw TensorFlow tensors in graph-execution order
 * (unmicst_amd/model.py: tensor_specs):
 *   for i in 0..nLayers-1:  ld{i}: w1 [ks,ks,Cin,Cout]; nExtraConvs x wextra [ks,ks,Cout,Cout];
 *                           wshort [s,s,Cin,Cout] (s = 1 legacy, ks v2); BN gamma,beta,moving_mean,moving_var [Cout]
 *   lb: w [ks,ks,Cn,2Cn]; (v2 only) BN x4 [2Cn]
 *   for i in nLayers-1..0:  lu{i}: wt [ks,ks,C(i+1),C(i+2)] (conv2d_transpose filter, OUTPUT channel first);
 *                           w2 [ks,ks,C(i)+C(i+1),C(i+1)]; (v2 only) BN x4; nExtraConvs x wextra
 *   lt: w [1,1,C1,nClasses]; (v2 only) BN x4 [nClasses]
 * All folding/packing happens inside umx_create.
 */

/* Replaces the nvidia-smi / NVML device probing of reference UnMicst1-5.py:750-769, toolbox/GPUselect.py:4-22. */
UMX_API int umx_device_count(void);

/* Free / total HBM bytes of one device -- what toolbox/GPUselect.py:4-22 (pick_gpu_lowest_memory, via NVML) reads to
 * choose a device when --GPU is not given (UnMicst1-5.py:751-754).  Either pointer may be NULL. */
UMX_API int umx_device_mem_info(int device_ordinal, size_t* free_bytes, size_t* total_bytes);

/* Replaces UNet2D.singleImageInferenceSetup's graph build + Saver.restore (UnMicst1-5.py:656-682).
 * max_batch = tiles per UNet launch group (activation arena is sized for it; any n is accepted later). */
UMX_API int umx_create(const umx_hparams* hp, const float* weight_blob, size_t blob_floats, int device_ordinal,
               int max_batch, umx_ctx** out);

/* umx_create with explicit options (precision, ...). */
UMX_API int umx_create_opts(const umx_hparams* hp, const float* weight_blob, size_t blob_floats, const umx_options* opts,
                    umx_ctx** out);
/* The precision a ctx actually runs (enum umx_precision). */
UMX_API int umx_precision_of(const umx_ctx* ctx);

/* Replaces UNet2D.singleImageInferenceCleanup (UnMicst1-5.py:684-685). NULL is a no-op. */
UMX_API void umx_destroy(umx_ctx* ctx);

/* Message of the last failure on this ctx (ctx == NULL: last failure of a ctx-less call on this thread). */
UMX_API const char* umx_last_error(const umx_ctx* ctx);

/* Use the caller's HIP stream (hipStream_t passed as void*; NULL restores the ctx's own stream). */
UMX_API int umx_set_stream(umx_ctx* ctx, void* hip_stream);
UMX_API int umx_synchronize(umx_ctx* ctx);

/* Replaces Session.run(UNet2D.nn, {tfData: tiles, tfTraining: 0}) (UnMicst1-5.py:704):
 * tiles [n,imSize,imSize,nChannels] float32 NHWC (already normalised) -> probs [n,imSize,imSize,nClasses]. */
UMX_API int umx_forward_tiles(umx_ctx* ctx, const float* tiles_host, int n, float* probs_host);
UMX_API int umx_forward_tiles_dev(umx_ctx* ctx, const float* tiles_dev, int n, float* probs_dev);

/* Tile grid of an H x W image for this model (PI2D.setup, PartitionOfImage.py:23-75; margin = imSize/8,
 * UnMicst1-5.py:694).  Any output pointer may be NULL. */
UMX_API int umx_tile_grid(const umx_ctx* ctx, int H, int W, int* patch_rows, int* patch_cols, int* padded_rows,
                  int* padded_cols);

/* Replaces UNet2D.singleImageInference (UnMicst1-5.py:687-710, UnMicst2.py:666-689, UnMicst.py:520-541) for
 * ALL classes in one pass: image float64 [C_img,H,W] (C_img == 1: the plane is copied to every input channel,
 * as solo does; else C_img == nChannels), already resized/rescaled by the driver.
 * out: [nClasses,H,W], uint16-encoded IEEE float16 for UMX_STITCH_FP16_COMPAT, float32 for UMX_STITCH_FP32. */
UMX_API int umx_infer_image(umx_ctx* ctx, const double* image_host, int C_img, int H, int W, double mean, double std,
                    int mode, int stitch, void* out_host);
UMX_API int umx_infer_image_dev(umx_ctx* ctx, const double* image_dev, int C_img, int H, int W, double mean, double std,
                        int mode, int stitch, void* out_dev);

/* The drivers' whole pre/post-processing around singleImageInference at --scalingFactor 1 and --outlier -1 (reference
 * UnMicst1-5.py:807-821 and :848-854; UnMicst2.py:771-788,813-817; UnMicst.py:618-633,654-656), fused with the path:
 *   raw [C_img,H,W] uint8 (bits 8) or uint16 (bits 16)  ->  float64 in [0,1] (skimage's resize at the identity grid:
 *   multiply by 1/255 or 1/65535)  ->  rescale != 0: rescale_intensity(in (min,max) of the plane, out (0, 0.983)) per
 *   plane (legacy / duo / cyto feed this; solo feeds the un-rescaled plane, rescale == 0)  ->  umx_infer_image (fp16-compat
 *   stitch)  ->  uint8(255 * pm) -> resize (identity) -> uint8(255 * .): out [nClasses,H,W] uint8.
 * Bit-identical to the host-side recipe (unmicst_amd/driver.py); saves the float64 upload and the fp16 download.  With rescale the
 * call finds each plane's (min, max) with host threads (umx_plane_range) while the rows are on their way up, and the tile gather
 * applies the rescale to the raw samples: no pass over the slide precedes the first tile. */
UMX_API int umx_infer_image_raw(umx_ctx* ctx, const void* raw_host, int bits, int C_img, int H, int W, int rescale,
                        double mean, double std, int mode, uint8_t* out_host);

/* umx_infer_image_raw with rescale != 0 and the planes' extrema handed in: range[2 * c], range[2 * c + 1] = (min, max) of plane c's
 * raw samples, as the file reader that produced raw_host found them (the reference's drivers read the page with tifffile and call
 * np.min / np.max on it, UnMicst1-5.py:817-821; UnMicst.py:606-610).  The call then needs no pass over the whole slide in front of its
 * first tile: rows go up and planes come down under the tile kernels, as without a rescale.  The values are trusted (they must be
 * the true extrema for the result to equal umx_infer_image_raw's); range == NULL is umx_infer_image_raw(rescale = 1). */
UMX_API int umx_infer_image_raw_range(umx_ctx* ctx, const void* raw_host, int bits, int C_img, int H, int W, const uint32_t* range,
                                      double mean, double std, int mode, uint8_t* out_host);

/* (min, max) of n uint8 / uint16 samples on the host, one pass on up to 16 threads: the pair umx_infer_image_raw_range takes, for a
 * caller whose reader did not keep it (the reference: np.min / np.max over the page, UnMicst1-5.py:817-821).  No context, no GPU. */
UMX_API int umx_plane_range(const void* raw_host, int bits, size_t n, uint32_t* range);

/* The same recipe at --scalingFactor != 1 (reference UnMicst1-5.py:813-816: resize to (int(H*sf), int(W*sf)) before the
 * inference, :850: resize of the uint8 planes back to (H, W)), with skimage.transform.resize's defaults -- order 1, mode
 * 'reflect', anti-aliasing Gaussian on shrinking axes, clip to the input range -- evaluated in float64 on the device.
 * rescale != 0: rescale_intensity of the RESIZED plane to (0, 0.983).  scaling == 1 is umx_infer_image_raw. */
UMX_API int umx_infer_image_raw_scaled(umx_ctx* ctx, const void* raw_host, int bits, int C_img, int H, int W, double scaling,
                                       int rescale, double mean, double std, int mode, uint8_t* out_host);

/* The same with the drivers' --outlier percentile (reference UnMicst1-5.py:817-821, UnMicst2.py:780-788, UnMicst.py:606-610):
 * rescale_intensity of the resized plane to (min, np.percentile(plane, outlier)) -> (0, 0.983), values above the percentile
 * clipped.  The percentile is exact: the two order statistics numpy interpolates between are found on the device by radix
 * selection over the float64 plane, the interpolation is numpy's (method 'linear').  outlier in [0, 100]; scaling may be 1. */
UMX_API int umx_infer_image_raw_outlier(umx_ctx* ctx, const void* raw_host, int bits, int C_img, int H, int W, double scaling,
                                        double outlier, double mean, double std, int mode, uint8_t* out_host);

/* How the host entry points move data: the slide goes up and the planes come down in the launch groups of the tile loop,
 * on two copy streams, under the tile kernels of the neighbouring groups (pinned host buffers make these true DMA; pageable
 * ones are staged by HIP and still correct).  A stream of slides -- the drivers' per-file loop, reference
 * UnMicst1-5.py:781-876 run once per file by MCMICRO -- can keep two calls in flight: umx_infer_image_raw_submit enqueues
 * one slide on `slot` (0 or 1; each slot owns its device buffers) and returns at once; umx_infer_image_wait blocks until
 * that slot's planes are in out_host and reports its errors (UMX_ERR_RANGE included).  raw_host / out_host must stay
 * valid until the wait.  umx_infer_image_raw == submit on slot 0 + wait. */
UMX_API int umx_infer_image_raw_submit(umx_ctx* ctx, int slot, const void* raw_host, int bits, int C_img, int H, int W,
                                       int rescale, double mean, double std, int mode, uint8_t* out_host);
UMX_API int umx_infer_image_wait(umx_ctx* ctx, int slot);

/*
 * Whole-slide inference sharded over the GPUs of one node, RCCL over xGMI inside the library (the reference is
 * single-device: UnMicst1-5.py:769).  One process (or thread with its own ctx) per GPU:
 *   rank 0: umx_shard_unique_id(&id); hand the 128 bytes to the other ranks over any channel (file, socket, MPI, ...);
 *   every rank: umx_shard_init(ctx, &id, rank, world)                      -- collective (ncclCommInitRank);
 *               umx_shard_plan(&hp, H, W, rank, world, nslabs, 0, ...)     -- which image rows this rank must hold;
 *               umx_infer_image_sharded_dev(ctx, band_dev, ...)            -- collective; every rank ends up with the full
 *                                                                             [nClasses,H,W] result in out_full_dev;
 *               umx_synchronize(ctx); ... ; umx_shard_fini(ctx) (or umx_destroy).
 * Patch rows are split into contiguous bands; a rank computes the tiles of its band from the image rows [need_row0,
 * need_row1) it holds (band_dev = those rows, float64 [C_img, band_rows, W], band_row0 = first row held), sends the
 * probabilities of its last patch row to the next rank, stitches the rows it owns -- visiting tiles in ascending global
 * index, so the fp16 result is bit-identical to a one-GPU run -- and all-gathers them slab by slab under the tiles of
 * the next slab.  RCCL is resolved at run time (dlopen of the librccl the process already holds, else ROCm's;
 * UMX_RCCL_PATH overrides).  unmicst_amd/sharding.py is the same schedule over torch.distributed.
 */
typedef struct umx_unique_id { char internal[128]; } umx_unique_id;   /* == ncclUniqueId */
UMX_API int umx_shard_unique_id(umx_unique_id* out);
UMX_API int umx_shard_init(umx_ctx* ctx, const umx_unique_id* id, int rank, int world);
/* The inter-rank operations of the schedule as a table: umx_shard_init fills it with RCCL; umx_shard_init_transport takes the
 * caller's instead (MPI, sockets, a host-staged stand-in -- how tests/test_gpu_parity.py runs the band / halo / scatter code of
 * umx_infer_image_sharded_dev in worlds of 2 and 3 on one GPU, where RCCL refuses to put two ranks on a device).  Semantics are
 * RCCL's: `stream` is a hipStream_t; an operation takes effect in stream order (a blocking implementation synchronises the stream,
 * moves the bytes and returns); send / recv between group_start and group_end may be matched in any order (either may be NULL);
 * all_gather: every rank contributes bytes_per_rank at send_dev and receives world * bytes_per_rank at recv_dev, rank order;
 * non-zero return = failure (UMX_ERR_HIP).  `peer` is a rank of the world given to the init call. */
typedef struct umx_shard_transport {
    void* user;
    int (*send)(void* user, const void* dev, size_t bytes, int peer, void* stream);
    int (*recv)(void* user, void* dev, size_t bytes, int peer, void* stream);
    int (*all_gather)(void* user, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream);
    int (*group_start)(void* user);
    int (*group_end)(void* user);
} umx_shard_transport;
UMX_API int umx_shard_init_transport(umx_ctx* ctx, const umx_shard_transport* transport, int rank, int world);
UMX_API int umx_shard_fini(umx_ctx* ctx);
/* Geometry of rank `rank` of `world` for an H x W image cut into `nslabs` slabs per band (the count actually used --
 * no more than the smallest band's patch rows -- comes back in nslabs_used): its patch rows, the image rows its tiles read
 * (what band_dev must cover), the image rows it stitches, and the rows of slab `slab`.  Any output may be NULL. */
UMX_API int umx_shard_plan(const umx_hparams* hp, int H, int W, int rank, int world, int nslabs, int slab, int* patch_row0,
                           int* patch_row1, int* need_row0, int* need_row1, int* own_row0, int* own_row1, int* slab_row0,
                           int* slab_row1, int* nslabs_used);
UMX_API int umx_infer_image_sharded_dev(umx_ctx* ctx, const double* band_dev, int C_img, int H, int W, int band_row0,
                                        int band_rows, double mean, double std, int mode, int stitch, int nslabs,
                                        void* out_full_dev);

/* The sharded schedule fed like the one-GPU line (umx_infer_image_raw_submit; SURVEY section 8(e), no reference counterpart:
 * UnMicst1-5.py:769 is single-device).  band_host = this rank's raw uint8 / uint16 rows [C_img, band_rows, W] (image rows
 * [band_row0, band_row0 + band_rows) = need_row0 .. need_row1 of umx_shard_plan), staged piece by piece on the upload stream ahead
 * of the tiles that read them; im2double happens in the tile gather.  range: NULL = no intensity rescale (the solo driver,
 * UnMicst1-5.py:816-821); else the (min, max) of EACH WHOLE plane (2 words per plane, as for umx_infer_image_raw_range -- a rank
 * holds only its band, the caller's reader saw the file) and the gather applies rescale_intensity to (0, 0.983).  Stitched slabs
 * are cast to the drivers' uint8 (UnMicst1-5.py:848-854 at the identity grid) and all-gathered as uint8; every rank ends up with
 * the full [nClasses, H, W] uint8 stack in out_full_dev (NULL: a buffer the library owns), and this rank's own rows
 * [own_row0, own_row1) go down to own_out_host [nClasses, own rows, W] (NULL: no download) on the download stream under the next
 * slab's tiles.  _submit enqueues on `slot` (0 or 1) and returns; umx_infer_image_wait(ctx, slot) completes it (errors,
 * UMX_ERR_RANGE included); band_host / own_out_host / out_full_dev must stay valid until then.  Collective over the world of
 * umx_shard_init.  Bit-identical, row for row, to umx_infer_image_raw[_range] of the whole slide on one GPU. */
UMX_API int umx_infer_image_sharded_raw_submit(umx_ctx* ctx, int slot, const void* band_host, int bits, int C_img, int H, int W,
                                               int band_row0, int band_rows, const uint32_t* range, double mean, double std,
                                               int mode, int nslabs, uint8_t* own_out_host, uint8_t* out_full_dev);
UMX_API int umx_infer_image_sharded_raw(umx_ctx* ctx, const void* band_host, int bits, int C_img, int H, int W, int band_row0,
                                        int band_rows, const uint32_t* range, double mean, double std, int mode, int nslabs,
                                        uint8_t* own_out_host, uint8_t* out_full_dev);

/* Strip / tile decoders of the drivers' own TIFF reader (unmicst_amd/tiffio.py; the reference reads through tifffile /
 * imagecodecs, UnMicst1-5.py:794-797): TIFF 6.0 LZW (compression 5, the OME-TIFF / Bio-Formats default) and PackBits
 * (32773).  Host code, no device needed.  Return the decoded byte count (<= cap) or -1 on a malformed stream. */
UMX_API long long umx_tiff_lzw_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap);
UMX_API long long umx_tiff_packbits_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap);

/* The two halves of umx_infer_image, exposed for band sharding across GPUs (one process per GPU):
 * umx_band_tiles_dev: PI2D.getPatch + normalise + UNet for patch rows [pr0,pr1) -> probs
 *   [(pr1-pr0)*patch_cols, P,P,K] float32.  image_dev holds image rows [band_row0, band_row0+band_rows) of the
 *   full H x W image ([C_img, band_rows, W]); rows the requested patch rows need but the band lacks are an error.
 * umx_stitch_dev: PI2D.patchOutput/getValidOutput restricted to image rows [y0,y1), given the probabilities of
 *   patch rows [tpr0,tpr1) (every tile touching those rows must be inside that range) -> out [K, y1-y0, W]. */
UMX_API int umx_band_tiles_dev(umx_ctx* ctx, const double* image_dev, int C_img, int H, int W, int band_row0,
                       int band_rows, double mean, double std, int pr0, int pr1, float* probs_dev);
UMX_API int umx_stitch_dev(umx_ctx* ctx, const float* probs_dev, int tpr0, int tpr1, int H, int W, int mode, int stitch,
                   int y0, int y1, void* out_dev);

/* Per-launch-site timing with HIP events on the ctx stream (used by bench.py for the roofline object). */
typedef struct umx_prof_entry {
    char name[48];          /* layer name, e.g. "lu1.conv" */
    char kernel[64];        /* kernel instantiation as rocprofv3 names it, e.g. "conv_f16x3<9, 4, 1, false, 4, false, false, false>" */
    int64_t launches;
    double total_ms;
    double flops_per_launch_sum;  /* sum over launches of algorithmic FLOPs (no padded work counted) */
    double bytes_per_launch_sum;  /* sum over launches of compulsory HBM bytes */
    double exec_flops_sum;        /* sum over launches of FLOPs issued to the matrix cores (padding and, for
                                     UMX_PREC_F16X3, the three products per fp32 product included) */
    int64_t launches_seen;        /* every launch of the site while profiling was on; `launches` and the sums above cover the
                                     launches that were bracketed by events (all of them unless sampling) */
    int32_t xcd_order;            /* conv_f16x3 sites, last launch: workgroup order (0 plain, 1 contiguous tile run per XCD, 2 (N-block,
                                     phase) fastest inside an XCD) */
    int32_t reserved;
} umx_prof_entry;
/* on = 0: off; 1: bracket every launch with two events; N >= 2: bracket every N-th launch of each site (two events per launch
 * cost the synthetic-256 step 1.1 % -- profiles/r03/prof_event_overhead.txt -- so bench.py samples).  Resets the counters. */
UMX_API int umx_profile_enable(umx_ctx* ctx, int on);
UMX_API int umx_profile_read(umx_ctx* ctx, umx_prof_entry* entries, int max_entries, int* n_entries);
/* sizeof(umx_prof_entry) of THIS build: a binding whose mirror of the struct has another size must not call umx_profile_read
 * (the struct grew in rounds 3 and 4; unmicst_amd/umx.py checks it when it loads the library, so that two builds compared
 * through UMX_LIB cannot be decoded with the wrong stride). */
UMX_API int umx_prof_entry_size(void);

/* Host-side helpers exported for the CPU test-suite (no GPU needed): the double->float16 round-to-nearest-even
 * used by the fp16-compat stitch, and the library's view of a model (layer count, packed weight bytes, FLOPs). */
UMX_API void umx_test_double_to_half(const double* in, uint16_t* out, size_t n);
/* the same conversion by the DEVICE routine the stitch kernel uses (needs a GPU): must equal the host routine bit for bit */
UMX_API int umx_test_double_to_half_dev(const double* in, uint16_t* out, size_t n);
UMX_API int umx_describe(const umx_hparams* hp, int* n_launches, double* flops_per_tile, double* executed_flops_per_tile);
/* The library's wiring of a model as JSON text (host only): activation buffers (spatial size, channels), the launch list in
 * execution order -- per launch its operand groups (source buffer, channels, filter taps: concat order = group order), output
 * buffer, fused max-pool, activation, where the BatchNorm affine sits, stride-2 transposed convolution -- and the constants the
 * kernels use (BatchNorm epsilon, LeakyReLU slope).  What a reviewer compares with the graph the reference builds
 * (UnMicst1-5.py:55-237; the op graph it saves: models/<name>/model.ckpt.meta).  Writes at most cap bytes (NUL-terminated) and
 * returns in *needed the bytes the whole text takes; UMX_ERR_INVALID if cap is too small. */
UMX_API int umx_describe_graph(const umx_hparams* hp, char* json, size_t cap, size_t* needed);
/* test entry (host): one OCP MX fp6 block as the planner packs the F6 form's weights -- 32 doubles in, 24 bytes out (element i in bits
 * [6 i, 6 i + 6): e2m3, round-to-nearest-even, saturating at 7.5), returns the block's e8m0 scale byte (2^(floor(log2 max) - 2)) */
UMX_API int umx_test_mx_pack_e2m3(const double* v32, uint8_t* out24);
/* Host only (no device needed): would umx_create with UMX_PREC_F16X3 take this model?  UMX_OK, or UMX_ERR_INVALID with the first
 * refused layer and the split-precision planner's reason in umx_last_error(NULL) -- what UMX_PREC_DEFAULT falls back to the exact-fp32
 * engine on (and warns about). */
UMX_API int umx_plan_check(const umx_hparams* hp);
UMX_API const char* umx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* UMX_H */
