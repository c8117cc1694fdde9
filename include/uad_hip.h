/* uad_hip.h — C-ABI of libuad_hip.so: the MI355X (gfx950) conv autoencoder train / reconstruct hot path.
 *
 * This is the drop-in boundary for the reference's TF-1.15 `sess.run` calls (the reference has no native code and
 * no FFI of its own; every entry below names the reference lines whose device work it replaces):
 *
 *   uad_forward / uad_backward / uad_adam_step / uad_train_step
 *       <- trainers/VAE.py:83-96, trainers/AE.py:70-83  (one sess.run({reconstruction, **losses, optimizer}))
 *          (+ trainers/ceVAE.py:86-117 on models/context_encoder_variational_autoencoder.py:9-59, losses
 *          trainers/ceVAE.py:38-51)
 *          graph = models/variational_autoencoder.py:9-47 | models/autoencoder.py:9-40 on
 *                  models/customlayers.py:16-38; losses trainers/VAE.py:36-42 | AE.py:28-29;
 *                  optimizer trainers/DLMODEL.py:112-131 (tf.train.AdamOptimizer, beta1 from
 *                  utils/default_config_setup.py:257)
 *   uad_forward(want_backward = 0)
 *       <- trainers/VAE.py:105-118, AE.py:92-105 (reconstruct: sess.run({'reconstruction'})) and the VAL pass
 *   uad_restore_step
 *       <- trainers/GMVAE_spatial.py:178-190 (150 x sess.run({'grads'}) + host `restored -= restore_lr * grads`), graph
 *          models/gaussian_mixture_variational_autoencoder_spatial.py:9-65, losses trainers/GMVAE_spatial.py:61-92
 *   uad_residual
 *       <- utils/Evaluation.py:282-289 (residual map, brain mask, hyper-intensity prior) and
 *          trainers/VAE.py:120 (l1err)
 *   uad_erode_cross / uad_median3d / uad_scores_*
 *       <- utils/Evaluation.py:84-89 (scipy binary_erosion), :108-110 (scipy median_filter), trainers/Metrics.py:17-19,45-47,
 *          67-72,138-162 (sklearn AUPRC / AUROC, Dice threshold sweep)
 *   uad_set_params / uad_get_params / uad_tensor_info
 *       <- tf.global_variables_initializer / tf.train.Saver variable access (trainers/DLMODEL.py:63-110)
 *   uad_op_*  — single-kernel entry points used by the parity tests (no reference counterpart).
 *
 * Conventions: plain C, no torch types.  All tensor arguments are DEVICE pointers to fp32 NHWC buffers owned by the
 * caller unless stated otherwise; `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are
 * asynchronous on `stream` except the ones documented as synchronous.  Every function returns 0 on success or a
 * UAD_ERR_* code; uad_last_error() returns the thread-local message of the last failure.  A handle is not thread-safe.
 */
#ifndef UAD_HIP_H
#define UAD_HIP_H

#ifdef __cplusplus
extern "C" {
#endif
/* libuad_hip.so is built with -fvisibility=hidden (build.py): everything declared between this push and the pop at the end of the file -- and
 * nothing else -- is exported, so `nm -D --defined-only libuad_hip.so` lists exactly this header's functions (tests/test_host_cpu.py). */
#pragma GCC visibility push(default)

enum { UAD_OK = 0, UAD_ERR_INVALID = 1, UAD_ERR_HIP = 2, UAD_ERR_UNSUPPORTED = 3 };
enum { UAD_ARCH_AE = 0, UAD_ARCH_VAE = 1, UAD_ARCH_CEVAE = 2, UAD_ARCH_GMVAE_SPATIAL = 3,
       UAD_ARCH_AE_SPATIAL = 4 };   /* models/autoencoder_spatial.py:7-27: no dense bottleneck; the latent is the encoder feature map
                                       [n,r,r,C]: io.mask_mu is its dropout keep-mask and io.z_mu receives it (both that shape) */
enum { UAD_BUF_PARAMS = 0, UAD_BUF_GRADS = 1, UAD_BUF_ADAM_M = 2, UAD_BUF_ADAM_V = 3 };
/* ENCODER_HI + ENCODER_LO partition ENCODER: HI = the deep blocks' variables (complete after the blocks >= 2 are back-propagated: ~93 % of the
 * segment at the default depth), LO = the first two blocks' kernels -- the data-parallel layer's last, exposed all-reduce then moves 0.2 MB
 * instead of 2.7 MB */
enum { UAD_SEG_DECODER = 0, UAD_SEG_BOTTLENECK = 1, UAD_SEG_ENCODER = 2, UAD_SEG_ENCODER_HI = 3, UAD_SEG_ENCODER_LO = 4, UAD_SEG_ALL = -1 };
/* arithmetic of the k5 s2 forward / data-gradient contractions: exact fp32 MFMA (default), or split-bf16 (x = hi + lo,
 * hi*hi + hi*lo + lo*hi on the bf16 matrix cores, fp32 accumulate: ~2^-17 relative error per product) */
enum { UAD_MATH_F32 = 0, UAD_MATH_BF16X3 = 1,
       UAD_MATH_BF16X3_ALL = 2 };   /* uad_gan_* only: also the generic k3 / k1 contractions of the ResNet graph in bf16x3.  Each
                                       contraction stays inside 1e-4, but through the 20-layer ResNet critic the penalty scalar
                                       drifts to ~3e-4 of its value, so for THAT graph this mode is opt-in and not parity-rated.
                                       The shorter generic-kernel graphs (aae_kind 4-6: Zimmerer VAE / ceVAE, GMVAE (You)) pass the
                                       same 1e-4 gradient checks in this mode as in fp32; their trainers default to it */
enum { UAD_MATH_BF16X6 = 3 };       /* uad_create handles (AE / VAE / ceVAE / spatial GMVAE) only, round 6: THREE bf16 planes per operand (x = h + m + l),
                                       six products per multiply on the bf16 matrix cores, fp32 accumulate -- fp32-grade results (the dropped terms are
                                       below 2^-26 of a product; tests hold 1e-5 against the fp64 oracle) at 6 x 32 instead of 8 x 64 matrix-pipe cycles
                                       per K = 16 of an fp32 MFMA.  The k5 s2 spatial kernels carry it; launches they do not take run exact fp32 */

typedef struct uad_model uad_model_t;

typedef struct {
    int arch;       /* UAD_ARCH_*                                              */
    int height;     /* config.outputHeight  (trainers/AEMODEL.py:18-19)         */
    int width;      /* config.outputWidth   (must equal height; power of two)   */
    int channels;   /* config.numChannels   (1 supported)                       */
    int inter_res;  /* config.intermediateResolutions[0]                        */
    int zdim;       /* config.zDim                                              */
    int max_batch;  /* largest n any later call will pass                       */
    /* spatial GMVAE only (trainers/GMVAE_spatial.py:12-21); ignored otherwise */
    int dim_c;      /* config.dim_c  mixture components                         */
    int dim_z;      /* config.dim_z                                             */
    int dim_w;      /* config.dim_w                                             */
    float c_lambda; /* config.c_lambda                                          */
} uad_config_t;

typedef struct {
    const float* x;          /* in  [n,H,W,C]                                                        */
    const float* eps;        /* in  [n,zdim] N(0,1) noise (VAE); NULL = 0                            */
    const float* mask_mu;    /* in  [n,zdim] keep-mask pre-scaled by 1/(1-rate), NULL = no dropout.
                                     VAE: on z_mu (variational_autoencoder.py:31); AE: on z (autoencoder.py:29) */
    const float* mask_sigma; /* in  [n,zdim] VAE: on z_log_sigma (:32)                               */
    const float* mask_dec;   /* in  [n,flat] VAE: on dec_dense output (:35); ignored for AE (autoencoder.py:30) */
    float* x_hat;            /* out [n,H,W,C] reconstruction                                         */
    float* l1_map;           /* out [n,H,W,C] |x_hat - x|, may be NULL                               */
    float* z_mu;             /* out [n,zdim] (AE: z), may be NULL                                    */
    float* z_log_sigma;      /* out [n,zdim] VAE, may be NULL                                        */
    float* z_sigma;          /* out [n,zdim] VAE, may be NULL                                        */
    float* scalars;          /* out [8] reconstructionLoss, kl, loss, 0, Rec_vae, Rec_ce, loss_vae, 0
                                     (means over the n samples; [4..6] are written for ceVAE only)       */
    float* rec_per_sample;   /* out [n] sum_hwc L1 (ceVAE: [2n], VAE branch then context branch), may be NULL */
    /* ---- ceVAE only (models/context_encoder_variational_autoencoder.py:9-59); NULL / ignored otherwise ---- */
    const float* x_ce;       /* in  [n,H,W,C] context-masked input (trainers/CE.py:123-139); NULL = x.  ceVAE: the second branch's input.
                                AE handles (dense / spatial): context-encoder training (trainers/CE.py:19-21,87-92): the network reads x_ce, the L1
                                term compares x_hat with x */
    const float* mask_mu_ce; /* in  [n,zdim] keep-mask on z_mu_ce (:37); given iff mask_mu is         */
    const float* mask_dec_ce;/* in  [n,flat] keep-mask on dec_dense(z_mu_ce) (:43); given iff mask_dec is */
    float* x_hat_ce;         /* out [n,H,W,C] reconstruction of the context branch, may be NULL      */
    float* l1_map_ce;        /* out [n,H,W,C] |x_hat_ce - x_ce|, may be NULL                         */
    float* anomaly;          /* out [n,H,W,C] L1_vae * |d loss_vae / d x| (trainers/ceVAE.py:51); written by
                                     uad_backward's ENCODER segment, may be NULL                          */
    /* ---- spatial GMVAE only (models/gaussian_mixture_variational_autoencoder_spatial.py:9-65) ----
     * x_hat = xz_mu; z_mu / z_log_sigma are the [n,r,r,dim_z] maps (r = inter_res); scalars = {mean_p_loss
     * (= reconstructionLoss), conditional_prior_loss, loss, w_prior_loss, c_prior_loss, 0, 0, 0} */
    const float* eps_w;      /* in  [n,r,r,dim_w] N(0,1) noise of w_sampled (:27); NULL = 0          */
    const float* eps_z;      /* in  [n,r,r,dim_z] N(0,1) noise of z_sampled (:32); NULL = 0          */
    float* w_mu;             /* out [n,r,r,dim_w], may be NULL                                       */
    float* w_log_sigma;      /* out [n,r,r,dim_w], may be NULL                                       */
    float* pc;               /* out [n,r,r,dim_c] softmax mixture posterior (:63), may be NULL       */
} uad_io_t;

const char* uad_last_error(void);
const char* uad_version(void);

int uad_create(const uad_config_t* cfg, uad_model_t** out);
int uad_destroy(uad_model_t* m);

long long uad_param_count(const uad_model_t* m);
int uad_num_tensors(const uad_model_t* m);
/* name (NUL-terminated, truncated to name_cap), flat offset, rank and shape (padded with 1s to 4) of tensor idx,
 * in TF variable-creation order (Encoder, Bottleneck, Decoder). */
int uad_tensor_info(const uad_model_t* m, int idx, char* name, int name_cap, long long* offset, int* rank, int* shape4);
/* device pointer to a handle-owned flat fp32 buffer of uad_param_count() elements (UAD_BUF_*) */
float* uad_buffer(uad_model_t* m, int which);
/* flat [offset, offset+count) range of one gradient segment; segments complete in the order DECODER, BOTTLENECK,
 * ENCODER (= ENCODER_HI, then ENCODER_LO) during uad_backward — the data-parallel layer all-reduces each as soon as it is done.
 * A segment may be empty (count 0). */
int uad_grad_segment(const uad_model_t* m, int segment, long long* offset, long long* count);

/* synchronous host<->device copies of the flat parameter vector */
int uad_set_params(uad_model_t* m, const float* host, long long count);
int uad_get_params(uad_model_t* m, float* host, long long count);
int uad_get_buffer(uad_model_t* m, int which, float* host, long long count);
int uad_set_buffer(uad_model_t* m, int which, const float* host, long long count);
int uad_reset_optimizer(uad_model_t* m);            /* zero Adam slots, t = 0 (synchronous) */
long long uad_get_step(const uad_model_t* m);
int uad_set_step(uad_model_t* m, long long t);

/* forward pass + losses.  want_backward != 0 additionally keeps what uad_backward needs and starts the backward
 * (d loss / d c of the last decoder block is produced by the fused loss kernel).  want_backward == 2 asks for the
 * data-gradient chain only (ceVAE validation / reconstruct: the anomaly map needs d loss_vae / d x but no parameter
 * gradient); the following uad_backward then leaves UAD_BUF_GRADS untouched.
 * ceVAE runs both branches as one 2n-sample pass through the shared layers (VAE-branch samples first). */
int uad_forward(uad_model_t* m, const uad_io_t* io, int n, int want_backward, void* stream);
/* gradient of `loss` w.r.t. every parameter into the UAD_BUF_GRADS buffer; segment = UAD_SEG_* (call DECODER,
 * BOTTLENECK, ENCODER in that order -- ENCODER may be given as ENCODER_HI followed by ENCODER_LO -- or UAD_SEG_ALL). */
int uad_backward(uad_model_t* m, int segment, void* stream);
/* Data-parallel form of a segmented uad_backward: the same launches, but a segment whose parameter gradients are all written on the handle's
 * own side stream (DECODER, BOTTLENECK where it runs fused, ENCODER_HI) returns WITHOUT making `stream` wait for that side stream -- three stalls of
 * `stream` per step (each exposing the tail of a slab reduction) that only the collective needs, not the next segment's kernels.  *ready_stream
 * receives the stream in whose order the segment's gradient slice is complete: the side stream where the join was skipped, else `stream`.  The caller
 * issues the slice's all-reduce in THAT stream's order (torch: dist.all_reduce under torch.cuda.stream(ExternalStream(ready))) and goes on with the
 * next segment on `stream`.  The last segment (ENCODER / ENCODER_LO) always joins: after it every gradient is complete in `stream`'s order too.
 * The reference has no counterpart (single process); parallel.DataParallelStep is the caller. */
int uad_backward_deferred(uad_model_t* m, int segment, void* stream, void** ready_stream);
/* ---- library-issued gradient all-reduce (RCCL over xGMI) -------------------------------------------------------------
 * SURVEY.md section 8b's `allreduce_attach(comm)`.  The reference is single-process; under slice-batch data parallelism (one process per GPU) the
 * flat fp32 gradient buffer is summed over the ranks once per step.  With a communicator attached the LIBRARY enqueues ncclAllReduce itself --
 * in place on its own gradient buffer, on its side stream right behind the slab reductions that complete the bucket (the last bucket: on the
 * caller's stream in front of the optimizer step) -- while the caller's stream runs on with the next backward segment: no process-group hand-off per collective (torch.distributed costs an event
 * record + wait on both sides of every all-reduce, 40-80 us per step at four buckets).  librccl.so.1 is bound at run time (dlopen; the copy the
 * process already holds, e.g. PyTorch-ROCm's, else the loader path or UAD_RCCL_LIB): libuad_hip.so does not link it.
 *   uad_rccl_unique_id      rank 0: a fresh ncclUniqueId (128 bytes) -- hand it to the other ranks by any channel (parallel.py: one broadcast over
 *                           the existing torch.distributed group)
 *   uad_rccl_comm_create    every rank, on its own device: ncclCommInitRank(world, id, rank) -> opaque communicator (collective call)
 *   uad_rccl_allreduce      in-place float sum over the ranks of any device buffer, enqueued on `stream` (the GAN handle's per-phase group gradient)
 *   uad_allreduce_attach    comm + bucket plan: bucket i = gradient elements [offset[i], offset[i] + count[i]), all-reduced right after backward segment
 *                           after_segment[i] (UAD_SEG_DECODER | BOTTLENECK | ENCODER_HI | ENCODER_LO); comm == NULL detaches.  The caller applies
 *                           grad_scale = 1 / world in uad_adam_step (the sum of per-rank batch-mean gradients / world = the global batch mean).
 *   uad_backward_allreduce  one backward segment (as uad_backward_deferred: DECODER, BOTTLENECK, ENCODER_HI, ENCODER_LO in this order) + its buckets'
 *                           all-reduces; after ENCODER_LO `stream` has been made to wait for every bucket, so the optimizer step can follow on it. */
int uad_rccl_unique_id(void* id_out, int cap);
int uad_rccl_comm_create(const void* id_bytes, int world, int rank, void** comm_out);
int uad_rccl_comm_destroy(void* comm);
int uad_rccl_allreduce(void* comm, float* buf, long long count, void* stream);
int uad_allreduce_attach(uad_model_t* m, void* comm, int world, int nbuckets, const int* after_segment, const long long* offset, const long long* count);
int uad_backward_allreduce(uad_model_t* m, int segment, void* stream);
/* TF-1.15 Adam: t += 1; lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps); grads scaled by grad_scale first */
int uad_adam_step(uad_model_t* m, float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);
/* the other optimizers of DLMODEL.create_optimizer (trainers/DLMODEL.py:113-123) with TF-1.15's update rules; the two slot buffers are the
 * Adam slots' storage (UAD_BUF_ADAM_M = Momentum's accumulator / RMSProp's `momentum` slot, UAD_BUF_ADAM_V = RMSProp's `rms` slot, which
 * TensorFlow initialises to ONE: set it before the first UAD_OPT_RMS step).  RMSProp: decay 0.9, eps 1e-10 are TF's defaults. */
enum { UAD_OPT_SGD = 1, UAD_OPT_MOMENTUM = 2, UAD_OPT_RMS = 3 };
int uad_optimizer_step(uad_model_t* m, int kind, float lr, float momentum, float decay, float eps, float grad_scale, void* stream);
/* Fault word of the fused bottleneck kernels (their sibling-workgroup exchange is bounded: uad_bott.hip).  A fault makes every optimizer launch
 * behind it a no-op ON THE DEVICE (parameters and slots stay those of the last good step) and is reported -- once, with the step counter rolled
 * back by the number of skipped updates (every optimizer call since the FIRST faulted launch) -- by the next uad_forward / uad_get_buffer /
 * uad_check_fault on the handle.  synchronize != 0 waits
 * for `stream` first (call it so at the end of an epoch, before a checkpoint, and -- under data parallelism -- before agreeing on the flag
 * across ranks: trainers/AEMODEL.py).  Returns UAD_OK when no fault is pending. */
int uad_check_fault(uad_model_t* m, int synchronize, void* stream);
/* on != 0: uad_forward / uad_get_buffer no longer report a pending fault -- only uad_check_fault does.  For data-parallel runs: a rank that raised
 * alone in the middle of an epoch would leave the other ranks blocked in the next gradient all-reduce; with deferred reporting every rank keeps
 * issuing the epoch's collectives (the faulted rank's optimizer launches stay no-ops on the device), all ranks agree on the word in the epoch's
 * scalar all-reduce (trainers/AEMODEL.py: process) and raise TOGETHER; the run restarts from the last checkpoint.  Default off. */
int uad_set_fault_deferred(uad_model_t* m, int on);
/* uad_forward(want_backward=1) + uad_backward(ALL) + uad_adam_step */
int uad_train_step(uad_model_t* m, const uad_io_t* io, int n, float lr, float beta1, float beta2, float eps,
                   void* stream);

/* spatial GMVAE restoration (trainers/GMVAE_spatial.py:178-190): one `sess.run(grads)` + host update, on device.
 * grads = d( n * loss + sum_n tv_lambda * TV_n(x - xz_mu) ) / d x  at the current x_restored (tf.gradients of the [n]-shaped
 * `loss + restore` sums its elements: each slice sees its own loss terms with weight 1, as if restored alone); then
 * x_restored -= restore_lr * grads in place.  grads_out (may be NULL) receives the gradient.  No parameter gradient is
 * computed and no host synchronisation happens: the caller enqueues restore_steps calls back to back.
 * On a UAD_ARCH_VAE handle this is trainers/VAE_You.py:52-53,133-144 instead: grads = d( rec_n + kl_n + tv_lambda * TV_n(x - x_hat) ) / d x
 * per sample (no 1/n: `pixel_loss` is not averaged), eps_z is the [n,zDim] reparameterisation noise, eps_w is ignored. */
int uad_restore_step(uad_model_t* m, float* x_restored, const float* eps_w, const float* eps_z, int n, float tv_lambda,
                     float restore_lr, float* grads_out, void* stream);

int uad_set_math_mode(uad_model_t* m, int mode);   /* UAD_MATH_* ; takes effect at the next uad_forward */
int uad_get_math_mode(const uad_model_t* m);
/* tests: device pointer + element count (for the batch of the last uad_forward; 2n rows inside a ceVAE handle) of a named
 * intermediate: "enc_c<i>" / "dec_c<i>" = pre-BN output of encoder / decoder block i (the activation pattern of the step is
 * sign(gamma' c + beta); the last decoder block's is not written when its epilogue is fused -- read "G0" = d loss / d c of that block
 * right after a want_backward forward instead -- or, when the step keeps that gradient in its compressed form, "fin_bits" = one
 * 32-bit word per output pixel (bit ch = channel ch's BN output > 0; read the floats' bit patterns) and "fin_dxhat" =
 * sign(x_hat - x) / n per pixel; both report count 0 otherwise), "dec_in" = the decoder's pre-BN input, "G0" / "G1" = gradient
 * ping-pong buffers, "fused_final" = flag in *count. */
int uad_debug_buffer(uad_model_t* m, const char* name, float** ptr, long long* count);

/* per-launch-group HIP-event profiler (bench.py's roofline leg).  While enabled, every launch group of
 * uad_forward / uad_backward / uad_adam_step is bracketed by hipEventRecord on the caller's stream.
 * uad_profile_report synchronises the device and writes lines "tag count total_ms\n" into buf, then clears. */
int uad_profile_enable(uad_model_t* m, int on);
int uad_profile_report(uad_model_t* m, char* buf, int cap);

/* residual anomaly map: out = mask * (pos_only ? max(x - xr, 0) : |x - xr|), zero where x < prior_thresh
 * (pass -INFINITY to disable); l1err[n] = sum |x - xr| per sample (may be NULL); mask may be NULL. hw = H*W*C. */
int uad_residual(const float* x, const float* xr, const float* mask, int n, int hw, int pos_only, float prior_thresh,
                 float* out, float* l1err, void* stream);

/* ---- residual-map scoring on the device (utils/Evaluation.py:84-127, 238-312; trainers/Metrics.py) ----------------------
 * uad_erode_cross: scipy.ndimage.binary_erosion(mask, generate_binary_structure(2,1), iterations) per [H,W] slice, zero
 *   border (utils/Evaluation.py:84-89 with iterations = 12); mask/out are fp32 [n,H,W], out = 1.0 where the pixel survives.
 * uad_median3d: scipy.ndimage.median_filter(vol, (5,5,5)) with the default 'reflect' boundary (utils/Evaluation.py:108-110).
 * uad_scores_*: every threshold metric of trainers/Metrics.py from ONE descending device sort of the n voxel scores:
 *   AUROC (sklearn roc_curve + auc, Metrics.py:45-47), AUPRC (sklearn average_precision_score, :17-19) and
 *   dice(score > t, label) for batches of thresholds (the inner step of compute_dice_curve_recursive, :138-162).
 *   label: fp32, nonzero = positive.  create is synchronous (allocates; one 32-bit radix sort + scans). */
typedef struct uad_scores uad_scores_t;
int uad_erode_cross(const float* mask, int n, int H, int W, int iterations, float* out, void* stream);
int uad_median3d(const float* vol, int D, int H, int W, int ksize, float* out, void* stream);
/* Monte-Carlo dropout statistics of K reconstructions (utils/Evaluation.py:238-266 with numMonteCarloSamples > 1): recs [K, total],
 * optional mask [total] (the eroded brain mask is applied to every sample first, as the reference does); mean [total] = E[mask*rec],
 * var [total] (may be NULL) = E[(mask*rec)^2] - E[mask*rec]^2 = Metrics.combined_predictive_uncertainty(x_recs, 0) (:170-173) */
int uad_mc_stats(const float* recs, const float* mask, int K, long long total, float* mean, float* var, void* stream);
/* utils/Evaluation.py:113-127 filter_3d_connected_components: out = vol with every 26-connected component of non-zero voxels that has
 * at most max_voxels (reference: 7) voxels set to 0 (bounded flood fill per voxel, exact; max_voxels < 16; not in place) */
int uad_cc_filter(const float* vol, int D, int H, int W, int max_voxels, float* out, void* stream);
int uad_scores_create(const float* pred, const float* label, long long n, uad_scores_t** out, void* stream);
int uad_scores_auc(const uad_scores_t* s, double* auroc, double* auprc, double* positives);
int uad_scores_dice(uad_scores_t* s, const double* thresholds_host, int k, double* dice_host, void* stream);
int uad_scores_destroy(uad_scores_t* s);

/* ---- noise of one step, drawn on the device --------------------------------------------------------------------
 * The reference draws eps (tf.random_normal, models/variational_autoencoder.py:34) and the dropout masks (keras Dropout(rate)(x, training),
 * e.g. :28-30) inside the graph on every sess.run; here they are explicit inputs of uad_forward, and this call fills them without a host
 * round trip.  Philox4x32-10, key = seed, counter = (element / 4, GLOBAL sample index sample0 + i, step, stream id): sample i of the
 * call gets the same numbers whichever rank / batch split draws it (data-parallel runs reproduce the single-process run).
 *   kind UAD_RNG_NORMAL:    out[i][e] ~ N(0, 1) (Box-Muller on 24-bit uniforms)
 *   kind UAD_RNG_KEEP_MASK: out[i][e] = u >= rate ? 1 / (1 - rate) : 0        (nn.dropout's rule, pre-scaled)
 * jobs: up to 8 arrays [n, per_sample] filled by ONE launch; `stream` tells independent arrays of a step apart (mu / sigma / dec masks). */
enum { UAD_RNG_NORMAL = 0, UAD_RNG_KEEP_MASK = 1 };
typedef struct { float* out; int per_sample; int kind; float rate; int stream; } uad_rng_job_t;
int uad_rng_fill(const uad_rng_job_t* jobs, int njobs, int n, unsigned long long seed, unsigned long long step, long long sample0, void* stream);

/* ---- measurement: the shader clock under load ---------------------------------------------------------------------
 * One wave, launched on `stream` (give it a stream of its own, beside the workload), samples s_memtime (shader cycles) and s_memrealtime
 * (100 MHz) for ticks_100mhz ticks and writes out2[0] = shader cycles, out2[1] = 100 MHz ticks (device memory): clock [GHz] =
 * out2[0] / out2[1] / 10.  bench.py prices `roofline.frac` at the 2.4 GHz spec clock and reports `frac_at_measured_clock` beside it. */
int uad_clock_probe(unsigned long long* out2, unsigned long long ticks_100mhz, void* stream);

/* ---- batch assembly from an HBM-resident slice cache ------------------------------------------------------------
 * Replaces the host-side batch slicing of dataloaders/BRAINWEB.py:411-478 (`next_batch`: images[images_in_set[start:end]], the label
 * -> brain-mask mapping :466-476) when the whole slice set lives in device memory (288 GB HBM): no H2D copy per step.
 *   uad_gather_slices: out[b] = src[idx[b]]                      src [N, slice_elems] fp32, idx device int32 [n]
 *   uad_gather_mask:   out[b][p] = lut256[labels[idx[b]][p]]     labels [N, slice_px] u8; lut256 NULL = the label value as float */
int uad_gather_slices(const float* src, const int* idx, int n, long long slice_elems, float* out, void* stream);
int uad_gather_mask(const unsigned char* labels, const int* idx, int n, long long slice_px, const unsigned char* lut256, float* out,
                    void* stream);

/* ---- f-AnoGAN (unified graph) ------------------------------------------------------------------------------
 * Replaces models/fanogan.py:11-84 (encoder + generator + critic graph) and the three optimisation phases of
 * trainers/fAnoGAN.py:45-77 (losses :50-66, the WGAN-GP penalty's tf.gradients :55-57, three Adams :71-77);
 * uad_gan_reconstruct replaces fAnoGAN.reconstruct :220-239.  One handle owns one flat fp32 parameter buffer in TF
 * variable-creation order (Encoder, Generator, Discriminator groups), its gradient and Adam slots.
 * A phase runs the forward graph the reference's sess.run of that phase needs, the phase's loss, and (want_backward)
 * the gradient of that loss w.r.t. the phase's variable group into the gradient buffer; uad_gan_adam then applies
 * TF-Adam to that group only (each group keeps its own step counter).  All calls are asynchronous on `stream`. */
enum { UAD_GAN_ENCODER = 0, UAD_GAN_GENERATOR = 1, UAD_GAN_DISCRIMINATOR = 2 };
enum { UAD_GAN_UNIFIED = 0, UAD_GAN_RESNET = 1,
       UAD_GAN_ANOVAEGAN = 2 };   /* models/anovaegan.py:10-80 + trainers/AnoVAEGAN.py:45-86 on the unified blocks: the encoder ends in mu /
                                     log-sigma heads (io.mask_z / io.mask_sigma, io.eps), the generator decodes z_vae with a linear output and
                                     the critic judges the reconstruction.  Phases: UAD_GAN_ENCODER = optim_vae (reconstructionLoss +
                                     kl_weight * kl over Encoder + Generator variables; uad_gan_adam(UAD_GAN_ENCODER) steps both, the
                                     Generator with its own second pair of slots, UAD_BUF_ADAM_M2 / _V2), UAD_GAN_GENERATOR = optim_gen,
                                     UAD_GAN_DISCRIMINATOR = optim_dis; io.z is unused */
enum { UAD_GAN_AAE = 3 };        /* dense-bottleneck BN autoencoder + re-encoding constraint and / or latent WGAN-GP critic; cfg.aae_kind:
                                     0 = models/constrained_autoencoder.py:9-48 + trainers/ConstrainedAE.py:36-45,
                                     1 = models/adversarial_autoencoder.py:10-72 + trainers/AAE.py:40-67,
                                     2 = models/constrained_adversarial_autoencoder.py:10-79 + trainers/ConstrainedAAE.py:44-70.
                                     Phases / groups: UAD_GAN_GENERATOR (1) = optim_ae (loss = mean(L2 [+ rho Rec_z]), every autoencoder variable),
                                     UAD_GAN_DISCRIMINATOR (2) = optim_dis (io.z = prior sample, io.alpha = eps of z_hat = z + eps (z - z_)),
                                     UAD_GAN_ENCODER (0) = optim_gen (-mean d_, the variables named 'Encoder/...', own Adam slots);
                                     io.mask_z / mask_g / mask_sigma = dropout masks of z_, dec_dense, z_rec.
                                     3 = the dense GMVAE, models/gaussian_mixture_variational_autoencoder.py:11-76 + trainers/GMVAE.py:56-101:
                                     cfg.zdim = dim_z, cfg.dim = dim_c, cfg.dim_w, cfg.c_lambda; one phase, UAD_GAN_GENERATOR = the optimizer over every
                                     variable (group 1); io.eps_w [n,dim_w] / io.eps [n,dim_z] = reparameterisation noise, io.mask_w_mu / mask_w_ls [n,dim_w],
                                     io.mask_z [n,dim_z] (z_mu; z_log_sigma has no dropout, model :42), io.mask_g [n,flat] (dec_dense);
                                     scalars: UAD_GAN_S_GM_*; restoration through uad_gan_restore_step.
                                     4 = the Zimmerer VAE, models/variational_autoencoder_Zimmerer.py:7-32 under trainers/VAE.py:36-42 (k4 s2 convolutions
                                     16-64-256-1024 + leaky_relu 0.2, no normalisation / dropout; inter_res must be height / 16); one phase,
                                     UAD_GAN_GENERATOR; io.eps [n,zDim]; scalars UAD_GAN_S_REC_LOSS, UAD_GAN_S_KL, UAD_GAN_S_ENC_LOSS (= loss).
                                     5 = the context-encoding VAE on the same stack, models/context_encoder_variational_autoencoder_Zimmerer.py:8-45 under
                                     trainers/ceVAE.py:38-51: io.x_ce, both branches as one 2n-sample pass; want_backward 2 = data-gradient chain only
                                     (io.anomaly without parameter gradients); scalars UAD_GAN_S_LOSS_IMG = Rec_vae, UAD_GAN_S_LOSS_FTS = Rec_ce, UAD_GAN_S_KL,
                                     UAD_GAN_S_REC_LOSS, UAD_GAN_S_ENC_LOSS = loss, UAD_GAN_S_GM_LOSS = loss_vae.
                                     6 = the original-architecture spatial GMVAE, models/gaussian_mixture_variational_autoencoder_You.py:8-85 under
                                     trainers/GMVAE_spatial.py: inter_res = height / 4 (the latent map), cfg as for kind 3, io.eps_w / io.eps =
                                     [n,r,r,dim_w] / [n,r,r,dim_z], no masks; scalars UAD_GAN_S_GM_*; uad_gan_restore_step.
                                     7 = the constrained adversarial autoencoder on residual blocks, models/constrained_adversarial_autoencoder_Chen.py:11-162
                                     under trainers/ConstrainedAAE.py:44-70: cfg.dim (32 | 64), height = 8 * inter_res; phases / groups / io as kind 2
                                     (no dropout masks; io.alpha = the run's scalar eps repeated n times, z_hat = eps z + (1 - eps) z_) */
enum { UAD_GAN_GROUP_VAE = 3 };   /* uad_gan_group only: the contiguous Encoder + Generator slice (AnoVAE-GAN's optim_vae) */
enum { UAD_BUF_ADAM_M2 = 4, UAD_BUF_ADAM_V2 = 5 };
/* scalars[16] written by uad_gan_phase (entries a phase does not compute are left untouched) */
enum { UAD_GAN_S_GEN_LOSS = 0, UAD_GAN_S_DISC_FAKE = 1, UAD_GAN_S_DISC_REAL = 2, UAD_GAN_S_PENALTY = 3, UAD_GAN_S_DISC_LOSS = 4,
       UAD_GAN_S_LOSS_IMG = 5, UAD_GAN_S_LOSS_FTS = 6, UAD_GAN_S_ENC_LOSS = 7, UAD_GAN_S_REC_LOSS = 8, UAD_GAN_S_KL = 9,
       UAD_GAN_S_GM_LOSS = 10, UAD_GAN_S_GM_CON = 11, UAD_GAN_S_GM_W = 12, UAD_GAN_S_GM_C = 13 };   /* dense GMVAE: loss, conditional_prior_loss,
                                                                                                       w_prior_loss, c_prior_loss (mean_p_loss = S_REC_LOSS) */
typedef struct uad_gan uad_gan_t;
typedef struct {
    int height, width, channels;   /* square power-of-two slices, 1 channel */
    int inter_res;                 /* config.intermediateResolutions[0] */
    int zdim;                      /* config.zDim */
    int max_batch;
    float scale, kappa;            /* fAnoGAN.Config :15-16 (gradient-penalty weight, feature-loss weight) */
    int variant;                   /* UAD_GAN_UNIFIED: models/fanogan.py:11-84; UAD_GAN_RESNET: models/fanogan_schlegl.py:11-161
                                      (pre-activation residual blocks, k3 convolutions, avg-pool / k1 s2 shortcuts, tanh output;
                                      height must be 8 * inter_res; no dropout in that graph: mask_z / mask_g are ignored) */
    int dim;                       /* RESNET only: base width (fanogan_schlegl.py:13: 64); 0 = 64 */
    float kl_weight;               /* ANOVAEGAN only: AnoVAEGAN.Config.kl_weight (:17) */
    int aae_kind;                  /* AAE only: 0 constrained AE, 1 AAE, 2 constrained AAE, 3 dense GMVAE, 4 Zimmerer VAE, 5 Zimmerer ceVAE, 6 GMVAE (You), 7 constrained AAE (Chen) */
    float rho;                     /* AAE only: weight of the latent re-encoding term (ConstrainedAE.Config.rho :15) */
    int dim_w;                     /* dense GMVAE only: GMVAE.Config.dim_w (:17); dim_z = zdim, dim_c = dim */
    float c_lambda;                /* dense GMVAE only: GMVAE.Config.c_lambda (:18) */
} uad_gan_config_t;
typedef struct {
    const float* x;                /* [n,H,W,1] batch (critic and encoder phases, reconstruct) */
    const float* z;                /* [n,zDim] sample_z() (generator and critic phases) */
    const float* alpha;            /* [n] the critic phase's interpolation coefficients (tf.random_uniform, fanogan.py:67) */
    const float* mask_z;           /* optional [n,zDim] inverted-dropout keep mask on the encoder's latent (fanogan.py:30) */
    const float* mask_g;           /* optional [n,flat] mask on the generator's dense output (fanogan.py:39,44) */
    float* generated;              /* optional out [n,H,W,1]: x_ (generator / critic phases) */
    float* reconstruction;         /* optional out [n,H,W,1]: x_enc (encoder phase, reconstruct) */
    float* z_enc;                  /* optional out [n,zDim] */
    float* l1_map;                 /* optional out [n,H,W,1]: |x - x_enc| (losses['L1']) */
    float* scalars;                /* optional out [16], UAD_GAN_S_* */
    const float* eps;              /* ANOVAEGAN: [n,zDim] N(0,1) reparameterisation noise (NULL = 0) */
    const float* mask_sigma;       /* ANOVAEGAN: optional keep mask of the log-sigma head (mask_z is the mu head's) */
    const float* eps_w;            /* dense GMVAE: [n,dim_w] N(0,1) noise of w_sampled (NULL = 0); io.eps is z_sampled's */
    const float* mask_w_mu;        /* dense GMVAE: optional [n,dim_w] keep masks of the w_mu / w_log_sigma heads */
    const float* mask_w_ls;
    const float* x_ce;             /* ceVAE on the Zimmerer stack (aae_kind 5): [n,H,W,1] context-masked input (NULL = x) */
    float* l1_map_ce;              /* ... optional out [n,H,W,1]: |x_ce - x_hat_ce| (io.l1_map is the VAE branch's, io.generated = x_hat_ce) */
    float* anomaly;                /* ... optional out [n,H,W,1]: L1_vae * |d loss_vae / d x| (written when want_backward != 0) */
} uad_gan_io_t;
int uad_gan_create(const uad_gan_config_t* cfg, uad_gan_t** out);
int uad_gan_destroy(uad_gan_t* g);
long long uad_gan_param_count(const uad_gan_t* g);
int uad_gan_num_tensors(const uad_gan_t* g);
int uad_gan_tensor_info(const uad_gan_t* g, int idx, char* name, int name_cap, long long* offset, int* rank, int* shape4);
float* uad_gan_buffer(uad_gan_t* g, int which);                         /* UAD_BUF_* device pointers */
int uad_gan_group(const uad_gan_t* g, int group, long long* offset, long long* count);   /* UAD_GAN_* slice of the buffers */
int uad_gan_set_buffer(uad_gan_t* g, int which, const float* host, long long count);
int uad_gan_get_buffer(uad_gan_t* g, int which, float* host, long long count);
int uad_gan_set_math_mode(uad_gan_t* g, int mode);                      /* UAD_MATH_* */
long long uad_gan_get_step(const uad_gan_t* g, int group);
int uad_gan_set_step(uad_gan_t* g, int group, long long t);
/* phase = the variable group being trained: GENERATOR (gen_loss), DISCRIMINATOR (disc_loss incl. penalty), ENCODER (enc_loss) */
int uad_gan_phase(uad_gan_t* g, int phase, const uad_gan_io_t* io, int n, int want_backward, void* stream);
int uad_gan_adam(uad_gan_t* g, int group, float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);
/* Data parallelism of the GAN handles, library-issued (round 6; the reference is single-process: trainers/fAnoGAN.py:70-77,96-129 run one sess.run per phase).
 * With a communicator attached (uad_rccl_comm_create) uad_gan_phase(..., want_backward = 1) all-reduces the TRAINED group's gradient slice itself: the slice is
 * cut into up to four buckets on tensor boundaries and a bucket's ncclAllReduce is enqueued on a collective stream as soon as the last kernel that writes one of
 * its tensors is enqueued -- i.e. while the backward of the earlier layers still runs (ResNet graph: per residual block; the other f-AnoGAN graphs: at the end
 * of the phase) -- and the phase's stream waits for the last bucket before the call returns to the caller, whose next launch is uad_gan_adam with
 * grad_scale = 1 / world.  comm == NULL detaches.  AAE-family handles (aae_kind != 0): UAD_ERR_UNSUPPORTED -- the caller all-reduces the slice with
 * uad_rccl_allreduce as before. */
int uad_gan_allreduce_attach(uad_gan_t* g, void* comm, int world);
int uad_gan_reconstruct(uad_gan_t* g, const uad_gan_io_t* io, int n, void* stream);
/* dense GMVAE restoration (trainers/GMVAE.py:172-184): one `sess.run(grads)` + the host update, on device.  grads = d( n * loss +
 * sum_n tv_lambda * TV_n(x - xz_mu) ) / d x at the current x_restored (= io->x is ignored; tf.gradients sums the [n]-shaped `loss + restore`,
 * so every slice sees its own loss terms with weight 1); then x_restored -= restore_lr * grads in place.  grads_out may be NULL.
 * io supplies the noise / dropout masks of this run.  No parameter gradient is produced. */
int uad_gan_restore_step(uad_gan_t* g, float* x_restored, const uad_gan_io_t* io, int n, float tv_lambda, float restore_lr, float* grads_out,
                         void* stream);
/* HIP-event profiler of the k3 spatial kernels of the ResNet graph (csrc/uad_convk16.inc; process-wide, bench.py's roofline leg for BASELINE
 * configs[3]): while enabled every launch is bracketed by two events on its stream.  uad_k3_profile_read synchronises the device and writes one
 * text line per launch shape -- "kind p1 p2 ntaps planes N MH MW CA Nn calls total_ms" (kind 0: tap-list F / D kernel, p1 = input stride, p2 =
 * halo extent, MH x MW = iteration grid, CA -> Nn channels; kind 1: filter gradient, p1 = stride, MH x MW = small grid, CA = CB, Nn = CS) --
 * and returns the bytes the whole table needs. */
int uad_k3_profile_enable(int on);
int uad_k3_profile_read(char* buf, int cap);
/* tests: device pointer + element count of a named intermediate of the last phase (NULL name table entry -> error) */
int uad_gan_debug_buffer(uad_gan_t* g, const char* name, float** ptr, long long* count);

/* ---- single-kernel entry points (parity tests) ------------------------------------------------------------
 * geometry of one strided-conv relation: big pixel (S*i-P+ky, S*j-P+kx) <-> small pixel (i,j); weights W[tap][cb][cs]
 * (= HWIO for Conv2D with big=input, [kh,kw,Cout,Cin] for Conv2DTranspose with big=output). */
typedef struct { int N, HB, WB, CB, HS, WS, CS, KS, S, P; } uad_conv_desc_t;
/* activation-on-load v -> lrelu_alpha(scale[c]*v + shift[c]); scale == NULL = identity */
typedef struct { const float* scale; const float* shift; float alpha; } uad_xform_t;

int uad_op_conv_f(const uad_conv_desc_t* d, const float* big_in, const uad_xform_t* xf, const float* W,
                  const float* bias, const float* mul, const float* add, float* small_out, void* stream);
int uad_op_conv_d(const uad_conv_desc_t* d, const float* small_in, const uad_xform_t* xf, const float* W,
                  const float* bias, const float* mul, const float* add, float* big_out, void* stream);
/* data-gradient forms with the fused activation backward: out = acc*lrelu'(es*cprev+eh)*es; s1[c]=sum d_bn,
 * s2[c]=sum d_bn*cprev (synchronous: allocates scratch) */
int uad_op_conv_f_bwdact(const uad_conv_desc_t* d, const float* big_in, const float* W, const float* cprev,
                         const uad_xform_t* act, float* small_out, float* s1, float* s2, void* stream);
int uad_op_conv_d_bwdact(const uad_conv_desc_t* d, const float* small_in, const float* W, const float* cprev,
                         const uad_xform_t* act, float* big_out, float* s1, float* s2, void* stream);
int uad_op_conv_w(const uad_conv_desc_t* d, const float* big, const uad_xform_t* xfb, const float* small_,
                  const uad_xform_t* xfs, float* dW, void* stream);
int uad_op_conv_first_fwd(const uad_conv_desc_t* d, const float* x, const float* W, const float* bias, float* out,
                          void* stream);
int uad_op_conv_first_wgrad(const uad_conv_desc_t* d, const float* x, const float* g, float* dW, void* stream);
int uad_op_adam(float* p, const float* g, float* m, float* v, long long n, float lr_t, float beta1, float beta2,
                float eps, float gscale, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* UAD_HIP_H */
