/*
 * umx_train.h -- C ABI of the training step of libumx.so (MI355X / gfx950 only).
 *
 * One call of umx_train_step == one `sess.run([optOp, loss], feed_dict={tfData, tfLabels, tfWeights, tfTraining: 1})`
 * of the reference's training loop (reference UnMicst1-5.py:483-484 solo, UnMicst2.py:471-472 duo): forward of the v2
 * graph with batch-statistics BN and dropout (UnMicst1-5.py:83-237), weighted cross-entropy + regularisation loss
 * (:367-373), gradients of every trainable variable, the optimiser update (:378-380) and the BN moving-average update
 * (UPDATE_OPS, :375,379).  Parameters live in the same flat blob layout umx_create takes, so a trained blob loads into
 * the inference engine unchanged.  Covered: UMX_GRAPH_V2 with nExtraConvs == 0 and 3x3 or 5x5 filters (every v2 model the
 * reference ships is 3x3).
 * Conventions as in umx.h: 0 = ok, umx_trainer_last_error() gives the message, the caller owns host buffers, one
 * trainer per host thread.  There is no CPU fallback.
 */
#ifndef UMX_TRAIN_H
#define UMX_TRAIN_H

#include "umx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct umx_trainer umx_trainer;

enum umx_optimizer { UMX_OPT_ADAM = 0, UMX_OPT_MOMENTUM = 1 };
enum umx_regulariser { UMX_REG_NONE = 0, UMX_REG_L1 = 1, UMX_REG_L2 = 2 };

/* The knobs the reference hard-codes in its scripts; umx_train_options_solo/_duo fill in those values. */
typedef struct umx_train_options {
    int32_t device_ordinal;
    int32_t batch;            /* images per step; 0 = hp.batchSize */
    int32_t optimizer;        /* enum umx_optimizer */
    int32_t decay_steps;      /* lr = lr0 * decay_rate^floor(step / decay_steps)  (tf.train.exponential_decay, staircase) */
    float lr0, decay_rate;
    float momentum;           /* MomentumOptimizer */
    float beta1, beta2, adam_eps;
    int32_t reg_kind;         /* enum umx_regulariser */
    float reg_down, reg_bottom, reg_up, reg_top;   /* coefficient on: shortcut kernels | lb kernel | lu kernels | lt kernel */
    float clip_eps;           /* > 0: log(clip(p, eps, 1-eps)) (solo, UnMicst1-5.py:369-370); 0: log(p) (duo) */
    float drop_down_step;     /* dropout rate of down layer i = drop_down_step * i          (UnMicst2.py:114) */
    float drop_bottom;        /*                 bottom layer                               (UnMicst1-5.py:139) */
    float drop_up0, drop_up_step; /*             up layer idx = drop_up0 - drop_up_step*idx (UnMicst2.py:203) */
    float bn_momentum;        /* moving-average momentum of tf.layers.batch_normalization: 0.99 */
    uint64_t seed;            /* dropout stream (counter-based hash of seed, step, layer, element) */
    int32_t reserved[8];      /* must be zero */
} umx_train_options;

UMX_API void umx_train_options_solo(umx_train_options* o);   /* UnMicst1-5.py: Adam 5e-5 x0.98/5000, l1(8e-5), bottom dropout 0.35 */
UMX_API void umx_train_options_duo(umx_train_options* o);    /* UnMicst2.py:  Adam 6e-5 x0.99/4000, l2(0.01/0.005), dropout everywhere */

/* replaces UNet2D.setup + tf.global_variables_initializer / saver.restore (UnMicst1-5.py:55-237,445-449): the blob holds
 * the initial (or restored) variables incl. BN moving statistics; optimiser slots start at zero, step at 0. */
UMX_API int umx_trainer_create(const umx_hparams* hp, const float* weight_blob, size_t blob_floats,
                               const umx_train_options* opts, umx_trainer** out);
UMX_API void umx_trainer_destroy(umx_trainer* tr);
UMX_API const char* umx_trainer_last_error(const umx_trainer* tr);

/* One step on HOST buffers: data [B,P,P,nChannels], labels and weights [B,P,P,nClasses], float32 NHWC (the reference's
 * batchData / batchLabels / batchWeights, UnMicst1-5.py:455-457,483).  apply_update 0: loss and gradients only
 * (parameters, slots, moving statistics and the step counter stay).  loss3 = {total, data term, regularisation}. */
UMX_API int umx_train_step(umx_trainer* tr, const float* data, const float* labels, const float* weights,
                           int apply_update, double* loss3);
/* Same on DEVICE buffers; only enqueues on the trainer's stream.  umx_trainer_loss synchronises and reads the loss. */
UMX_API int umx_train_step_dev(umx_trainer* tr, const float* data_dev, const float* labels_dev, const float* weights_dev,
                               int apply_update);
UMX_API int umx_trainer_loss(umx_trainer* tr, double* loss3);

enum umx_trainer_vector { UMX_TV_PARAMS = 0, UMX_TV_GRADS = 1, UMX_TV_SLOT_M = 2, UMX_TV_SLOT_V = 3 };
/* copy one blob-shaped vector to the host (saver.save of the variables / the gradients of the last step) */
UMX_API int umx_trainer_read(umx_trainer* tr, int which, float* out, size_t n_floats);
/* softmax output of the last step's forward pass [B,P,P,nClasses] (the reference evaluates its pixel errors on it,
 * UnMicst1-5.py:386-397) */
UMX_API int umx_trainer_probs(umx_trainer* tr, float* probs_host);
/* Diagnostics of the last step's forward pass (what sess.run would fetch by tensor name): `name` is
 *   "ld<i>.z" | "lb.z" | "lu<i>.z" | "lt.z"   the convolution output in front of that layer's BatchNorm, [B,H,W,C];
 *   "<layer>.stat"                            its batch statistics [4][C] = mean | rstd | scale = gamma rstd | shift = beta - mean scale
 *                                             (BN output = z * scale + shift: the value the LeakyReLU / max-pool decisions are taken on);
 *   "lu<i>.us"                                the up-sampled tensor behind its LeakyReLU (UnMicst1-5.py:192-195), [B,2h,2h,C];
 *   "ds<i>"                                   input of down layer i (ds0 = the batch; pooled + dropped output of layer i-1).
 * *n_floats in: capacity of `out`; out: the tensor's size (out == NULL just queries it).  Synchronises the trainer's stream.
 * The parity tests use it to take the oracle's gradient at the SAME activation / pooling decisions (tests/test_gpu_train.py). */
UMX_API int umx_trainer_read_tensor(umx_trainer* tr, const char* name, float* out, size_t* n_floats);
/* Session.run(UNet2D.nn / errors, feed_dict={tfData: batchData, tfTraining: 0}) with the trainer's current variables
 * (the validation and test passes of UNet2D.train, UnMicst1-5.py:501-502,564-565): moving statistics, no dropout.
 * data [B,P,P,nChannels] and probs [B,P,P,nClasses] are HOST buffers; synchronous. */
UMX_API int umx_trainer_eval(umx_trainer* tr, const float* data, float* probs_host);
UMX_API int64_t umx_trainer_step_count(const umx_trainer* tr);
UMX_API int umx_trainer_batch(const umx_trainer* tr);
/* algorithmic FLOPs of one step per image: forward + input gradients + weight gradients of every convolution */
UMX_API double umx_trainer_flops_per_image(const umx_trainer* tr);
/* per-phase time of the steps since the last call (ms, accumulated with HIP events when enabled) */
UMX_API int umx_trainer_profile(umx_trainer* tr, int enable, double* fwd_ms, double* bwd_ms, double* opt_ms, int* steps);

#ifdef __cplusplus
}
#endif
#endif
