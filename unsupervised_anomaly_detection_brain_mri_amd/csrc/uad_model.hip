// Model-level orchestration + the C-ABI (include/uad_hip.h) for the dense-bottleneck AE / VAE.
// One handle owns the flat fp32 parameter / gradient / Adam-slot buffers, every pre-BN activation of the step and
// the reduction scratch; a train step is ~60 asynchronous kernel launches on the caller's stream, no host sync.
//
// Graph restated (reference): models/customlayers.py:16-38, models/autoencoder.py:9-40,
// models/variational_autoencoder.py:9-47; losses trainers/AE.py:28-29, trainers/VAE.py:36-42;
// Adam trainers/DLMODEL.py:112-131.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include <dlfcn.h>
#include <unistd.h>
// RCCL: types only -- the library is bound at run time (rccl_api below), libuad_hip.so does not link it.  Without the development header (a single-GPU ROCm
// install) the few opaque types are declared here and the uad_rccl_* entry points still work whenever librccl.so.1 itself can be loaded.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclHalf = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
#endif
#include "../../include/uad_hip.h"
#include "uad_kernels.h"

static thread_local std::string g_err;

// records the message uad_last_error() returns; shared with uad_eval.hip
int uad_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define fail uad_fail

namespace {

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(UAD_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr float kBnEps = 1e-3f;     // tf.layers.BatchNormalization default epsilon
constexpr float kLrelu = 0.3f;      // keras LeakyReLU() default (customlayers.py:23,36)

struct Tensor {
    std::string name;
    long long off;
    int rank;
    int shape[4];
    long long count() const { return (long long)shape[0] * shape[1] * shape[2] * shape[3]; }
};

struct ConvLayer {        // conv / convT block followed by BN + (Leaky)ReLU
    UadConvDesc d;        // geometry at batch 1 (N filled per call)
    long long w, b, gamma, beta;   // flat offsets
    float* c;             // pre-BN output [N, ., ., C]
};

}  // namespace

struct uad_model {
    uad_config_t cfg;
    int n_pool, cenc, cmid, flat;
    std::vector<Tensor> tensors;
    long long nparams;
    long long seg_off[5], seg_cnt[5];
    float *params, *grads, *adam_m, *adam_v;
    float *wpack_f, *wpack_d;          // k-quad-interleaved copies of the 5x5 kernels (refreshed once per forward)
    float *wpack16_f, *wpack16_d;      // bf16 hi|lo planes of the same kernels (bf16x3 math mode); 2*nparams ushorts each
    int math;                          // UAD_MATH_F32 | UAD_MATH_BF16X3 | UAD_MATH_BF16X6
    float *wpack3_f, *wpack3_d;        // bf16x6: three bf16 planes of the 5x5 kernels, 4 * nparams ushorts each (allocated when the mode is first set)
    UadGemmWs ws;                      // split-K slabs for the GEMMs that cannot fill the chip on their own
    bool packed_valid;
    bool pack_inflight;               // the repack of the updated parameters was launched on SIDE by the optimizer step (ev_pack)
    hipEvent_t ev_opt, ev_pack, ev_pack_head; bool pack_head; bool pack_head_main; hipStream_t pack_head_stream;      // ev_pack_head: the first packed consumer's tensor is ready (the rest: ev_pack)
    long long step;
    // layers
    std::vector<ConvLayer> enc, dec;
    long long bw, bb;                 // Bottleneck/conv2d
    long long muw, mub, sgw, sgb;     // dense mu / sigma (AE: muw/mub = dense_z)
    long long dw, db;                 // dense_dec
    long long rw, rb;                 // Bottleneck/conv2d_1
    long long dbn_g, dbn_b;           // Decoder/batch_normalization (input BN + ReLU)
    long long fw, fb;                 // dec_Conv2D_final
    // activations
    float *t, *mu_raw, *ls_raw, *mu, *ls, *sigma, *z, *dvec, *cb, *kl;
    float *xhat_own;
    float *wT_d, *wT_mu, *wT_sg;       // transposed copies of the dense kernels (fused bottleneck backward), refreshed with the packs
    // ceVAE: both branches run as one 2n-sample pass; staging for the concatenated inputs / outputs
    float *xcat, *mdec_cat, *l1_own;
    int nmul;                          // samples per user sample inside the handle (2 for ceVAE)
    // spatial GMVAE: latent heads (15 contiguous tensors), their per-location scratch, TV-restore state
    long long gm_off[15];              // wmu_k,wmu_b,wls_k,wls_b,zmu_k,zmu_b,zls_k,zls_b,c7_k,c7_b,m_k,m_b,l_k,l_b,var
    long long gm_total;                // elements of the heads segment
    float *gm_h, *gm_loc_loss, *gm_dheads, *gm_da7, *gm_mid, *gm_dM, *gm_dLq, *gm_ws, *gm_partial, *gm_dxhat;
    const float* dec_in0;              // input of the first decoder ConvT: cb (AE family) or gm_h (GMVAE)
    bool restore;                      // last forward was a restoration pass (TV term in the objective)
    bool fb_on_load;                   // ... whose d loss / d c of the last block is formed inside dec.back()'s data-gradient kernel
    float restore_tv, restore_lr;
    float restore_scale;               // weight of every sample's loss terms in the restore objective (1: tf.gradients sums the [n]-shaped ys)
    float* restore_x;                  // x_restored (updated in place by the backward) or null
    float* restore_grads;              // optional gradient output
    // gradient ping-pong + small grads
    float *G0, *G1;
    float* dcb_keep;                  // copy of d loss / d cb for conv2d_1's kernel gradient (SIDE)
    float* bott_xch; unsigned* bott_flags; unsigned bott_epoch;
    unsigned* bott_err_host; unsigned* bott_err_dev;      // pinned word the fused bottleneck kernels report a timed-out sibling exchange through
    unsigned* bott_fault;                                 // the same word in device memory: the optimizer kernels read it and skip their update
    std::vector<unsigned> opt_epochs;                     // bottleneck launch epoch at every optimizer call since the last fault check: how many updates a fault skipped
    bool fault_deferred;                                  // uad_set_fault_deferred: only uad_check_fault reports (data-parallel runs agree on the word first)
    float* bnfin_scratch;             // counters + partials of the 2-D BN-gradient finalize (SIDE stream only)
    float* bott_wpart;                // [4 * max_batch][2*cenc*cmid + cmid] shares of conv2d / conv2d_1's parameter gradients
    float *g_small[6];                // d_cb-side temporaries: dd, dz, dmu_raw, dls_raw, dflat, dflat2
    // scratch
    float *colpart, *wpartial, *colscratch, *red_partial, *rec_partial, *rec_ps, *scalars_own;
    size_t colpart_cap, wpartial_cap;
    // state of the last forward
    int last_n, last_nuser;            // samples inside the handle / samples the caller passed
    uad_io_t last_io;
    const float* x_eff;                // [last_n] input of the last forward (xcat for ceVAE)
    const float* mask_dec_eff;         // [last_n, flat] or null
    bool have_fwd;
    bool data_only;                    // uad_forward(want_backward = 2): no parameter gradients
    unsigned* fin_bits;                // training step: the last block's d loss / d c in compressed form -- activation-pattern word per output
    float* fin_dxh;                    //   pixel + sign(x_hat - x) / n per pixel (UadEpilogue::fin_bits, UadXform::fb_bits)
    bool last_fin_bits;                // the last forward left its loss gradient in that form (G0 was not written)
    bool fwd_tail_is_loss;            // the last operation the last forward enqueued was its loss.finalize kernel (nothing the backward reads)
    bool joined;                       // uad_backward_deferred: the segment just run ended with the side stream joined into the caller's
    bool last_fused_final;             // the last forward ran the last block's BN / final conv / loss inside the ConvT epilogue (its c is not written)
    std::vector<void*> allocs;
    // library-issued gradient all-reduce (uad_allreduce_attach): RCCL communicator of this rank, the stream the collectives run on, the bucket plan
    void* ar_comm; int ar_world; hipStream_t ar_stream; bool ar_own_stream;
    int ar_nb; int ar_after[4]; long long ar_off[4], ar_cnt[4];
    hipEvent_t ar_ev_in[4], ar_ev_out; bool ar_pending;
    // second stream + events of the backward pass; per-layer scratch touched by that stream
    hipStream_t side;
    std::vector<hipEvent_t> sync_events;
    size_t ev_next;
    float* cp_slot[16];
    float* wp_slot[16];
    // optional per-launch-group HIP-event profiler (uad_profile_*)
    bool prof_on;
    struct ProfRec { const char* tag; hipEvent_t a, b; };
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> ev_pool;
};

namespace {

struct ProfScope {
    uad_model* m; hipStream_t st; hipEvent_t a, b; const char* tag; bool on;
    static hipEvent_t get(uad_model* m) {
        if (!m->ev_pool.empty()) { hipEvent_t e = m->ev_pool.back(); m->ev_pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
    ProfScope(uad_model* m_, const char* tag_, hipStream_t st_) : m(m_), st(st_), tag(tag_), on(m_->prof_on) {
        if (on) { a = get(m); b = get(m); (void)hipEventRecord(a, st); }
    }
    ~ProfScope() {
        if (on) { (void)hipEventRecord(b, st); m->prof.push_back({tag, a, b}); }
    }
};
#define PROF(tag) ProfScope prof_scope_##__LINE__(m, tag, st)

float* P(uad_model* m, long long off) { return m->params + off; }
// packed-weight views of one 5x5 tensor for the current math mode (the unused one is null)
// (bf16x6 keeps BOTH: the three planes for the k5 spatial kernels and the fp32 pack for the launches those do not take)
inline bool bf_mode(const uad_model* m) { return m->math == UAD_MATH_BF16X3 || m->math == UAD_MATH_BF16X6; }
inline int planes_of(const uad_model* m) { return m->math == UAD_MATH_BF16X6 ? 3 : 2; }
const float* PKF(uad_model* m, long long off) { return m->math != UAD_MATH_BF16X3 ? m->wpack_f + off : nullptr; }
const float* PKD(uad_model* m, long long off) { return m->math != UAD_MATH_BF16X3 ? m->wpack_d + off : nullptr; }
const unsigned short* PK16F(uad_model* m, long long off) {
    return m->math == UAD_MATH_BF16X3 ? (const unsigned short*)m->wpack16_f + 2 * off : m->math == UAD_MATH_BF16X6 ? (const unsigned short*)m->wpack3_f + 4 * off : nullptr;
}
const unsigned short* PK16D(uad_model* m, long long off) {
    return m->math == UAD_MATH_BF16X3 ? (const unsigned short*)m->wpack16_d + 2 * off : m->math == UAD_MATH_BF16X6 ? (const unsigned short*)m->wpack3_d + 4 * off : nullptr;
}
long long PLANE(const ConvLayer& L) { return (long long)L.d.KS * L.d.KS * L.d.CB * L.d.CS; }
float* Gr(uad_model* m, long long off) { return m->grads + off; }

UadXform bn_xform(uad_model* m, long long gamma, long long beta, float alpha) {
    UadXform x;
    x.scale = P(m, gamma); x.shift = P(m, beta); x.alpha = alpha; x.mult = 1.0f / sqrtf(1.0f + kBnEps);
    return x;
}
UadXform no_xform() { UadXform x; x.scale = nullptr; x.shift = nullptr; x.alpha = 1.f; x.mult = 1.f; return x; }
// restoration: d loss / d c of the last block is formed while the data-gradient kernel stages its input (no final<BWD> pass)
bool restore_fb_on_load(uad_model* m, int n) {
    if (!m->restore || !bf_mode(m)) return false;
    UadConvDesc d = m->dec.back().d; d.N = n;
    return uad_conv_f_supports_final_bwd(d, true, m->ws.floats);
}

UadEpilogue epi_bias(const float* bias, const float* mul = nullptr, const float* add = nullptr) {
    UadEpilogue e;
    memset(&e, 0, sizeof e);
    e.kind = UAD_EPI_BIAS; e.bias = bias; e.mul = mul; e.add = add;
    return e;
}
UadEpilogue epi_bwd(uad_model* m, const float* cprev, long long gamma, long long beta, float alpha) {
    UadEpilogue e;
    memset(&e, 0, sizeof e);
    e.kind = UAD_EPI_BWD_ACT; e.cprev = cprev; e.escale = P(m, gamma); e.eshift = P(m, beta); e.ealpha = alpha;
    e.emult = 1.0f / sqrtf(1.0f + kBnEps); e.colpart = m->colpart;
    return e;
}

long long add_tensor(uad_model* m, const std::string& name, int rank, int s0, int s1, int s2, int s3) {
    Tensor t;
    t.name = name; t.off = m->nparams; t.rank = rank;
    t.shape[0] = s0; t.shape[1] = s1; t.shape[2] = s2; t.shape[3] = s3;
    m->nparams += t.count();
    m->tensors.push_back(t);
    return t.off;
}

int dev_alloc(uad_model* m, float** p, size_t floats) {
    void* q = nullptr;
    if (floats == 0) floats = 4;
    HIP_TRY(hipMalloc(&q, floats * sizeof(float)));
    HIP_TRY(hipMemset(q, 0, floats * sizeof(float)));
    m->allocs.push_back(q);
    *p = (float*)q;
    return UAD_OK;
}

// latent-head launch arguments of the spatial GMVAE (parameters + the last encoder block's activation-on-load)
UadGmArgs gm_args(uad_model* m, const float* eps_w, const float* eps_z, float inv_batch) {
    UadGmArgs a;
    memset(&a, 0, sizeof a);
    const ConvLayer& EL = m->enc.back();
    a.cenc = m->cenc; a.W = m->cfg.dim_w; a.Z = m->cfg.dim_z; a.C = m->cfg.dim_c;
    a.c_lambda = m->cfg.c_lambda; a.inv_batch = inv_batch;
    a.c_enc = EL.c; a.scale = P(m, EL.gamma); a.shift = P(m, EL.beta); a.alpha = kLrelu; a.mult = 1.0f / sqrtf(1.0f + kBnEps);
    const float** slots[15] = {&a.wmu_k, &a.wmu_b, &a.wls_k, &a.wls_b, &a.zmu_k, &a.zmu_b, &a.zls_k, &a.zls_b,
                               &a.c7_k, &a.c7_b, &a.m_k, &a.m_b, &a.l_k, &a.l_b, &a.var};
    for (int k = 0; k < 15; ++k) *slots[k] = P(m, m->gm_off[k]);
    a.eps_w = eps_w; a.eps_z = eps_z;
    a.h_out = m->gm_h; a.loc_loss = m->gm_loc_loss;
    return a;
}

// launch arguments of the per-sample fused bottleneck (AE / VAE / ceVAE)
UadBottArgs bott_args(uad_model* m, const uad_io_t& io, const float* mask_dec, int nu) {
    UadBottArgs a;
    memset(&a, 0, sizeof a);
    const ConvLayer& EL = m->enc.back();
    const bool vae = m->sgw >= 0;
    a.cenc = m->cenc; a.cmid = m->cmid; a.npos = m->cfg.inter_res * m->cfg.inter_res; a.zdim = m->cfg.zdim;
    a.n_vae = nu; a.alpha = kLrelu; a.mult = 1.0f / sqrtf(1.0f + kBnEps); a.inv_batch = 1.0f / (float)nu;
    a.c_enc = EL.c; a.scale = P(m, EL.gamma); a.shift = P(m, EL.beta);
    a.Wb = P(m, m->bw); a.bb = P(m, m->bb); a.Wmu = P(m, m->muw); a.bmu = P(m, m->mub);
    a.Wsg = vae ? P(m, m->sgw) : nullptr; a.bsg = vae ? P(m, m->sgb) : nullptr;
    a.Wd = P(m, m->dw); a.bd = P(m, m->db); a.Wr = P(m, m->rw); a.br = P(m, m->rb);
    a.WdT = m->wT_d; a.WmuT = m->wT_mu; a.WsgT = vae ? m->wT_sg : nullptr;
    a.eps = io.eps; a.mask_mu = io.mask_mu; a.mask_ls = io.mask_sigma;
    a.mask_mu_ce = m->cfg.arch == UAD_ARCH_CEVAE ? io.mask_mu_ce : nullptr;
    a.mask_dec = vae ? mask_dec : nullptr;        // AE: the dec_dense dropout is never active (autoencoder.py:30)
    a.t = m->t; a.mu = m->mu; a.ls = m->ls; a.sigma = m->sigma; a.z = m->z; a.kl = m->kl; a.dvec = m->dvec; a.cb = m->cb;
    a.xch = m->bott_xch; a.flags = m->bott_flags; a.xw = 2 * m->cfg.zdim; a.epoch = 0;      // epoch: set at each launch
    a.err = m->bott_err_dev; a.err_dev = m->bott_fault;
    return a;
}

UadConvDesc dense_desc(int n, int in, int out) { return UadConvDesc{n, 1, 1, in, 1, 1, out, 1, 1, 0}; }
UadConvDesc conv1x1_desc(int n, int h, int w, int cin, int cout) { return UadConvDesc{n, h, w, cin, h, w, cout, 1, 1, 0}; }

int ilog2i(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

}  // namespace

void uad_conv_any_order_next(bool on);      // uad_gemm.hip: the next spatial F / D launch (filter-gradient launch) leaves the AQL barrier bit clear
void uad_conv_w_any_order_next(bool on);

extern "C" {

const char* uad_last_error(void) { return g_err.c_str(); }
const char* uad_version(void) { return "uad_hip 0.1 (gfx950, fp32 MFMA)"; }

int uad_create(const uad_config_t* cfg, uad_model_t** out) {
    if (!cfg || !out) return fail(UAD_ERR_INVALID, "null argument");
    const int H = cfg->height, Wd = cfg->width;
    if (H != Wd || H <= 0 || (H & (H - 1))) return fail(UAD_ERR_INVALID, "height/width must be equal powers of two");
    if (cfg->inter_res <= 0 || (cfg->inter_res & (cfg->inter_res - 1)) || cfg->inter_res >= H)
        return fail(UAD_ERR_INVALID, "inter_res must be a power of two smaller than height");
    if (cfg->channels != 1) return fail(UAD_ERR_UNSUPPORTED, "numChannels=%d: only 1 is supported", cfg->channels);
    if (cfg->arch < UAD_ARCH_AE || cfg->arch > UAD_ARCH_AE_SPATIAL) return fail(UAD_ERR_INVALID, "bad arch");
    const bool gm = cfg->arch == UAD_ARCH_GMVAE_SPATIAL;
    const bool sp = cfg->arch == UAD_ARCH_AE_SPATIAL;        // spatial AE: encoder feature map -> decoder, nothing in between
    if (gm && (cfg->dim_c < 1 || cfg->dim_c > 64 || cfg->dim_z < 1 || cfg->dim_w < 1 || cfg->dim_z * cfg->dim_c > 4096 || cfg->dim_w > 64))
        return fail(UAD_ERR_UNSUPPORTED, "GMVAE: need 1 <= dim_c <= 64, dim_z*dim_c <= 4096, 1 <= dim_w <= 64");
    if (!gm && !sp && (cfg->zdim <= 0 || cfg->zdim % 8)) return fail(UAD_ERR_UNSUPPORTED, "zDim must be a positive multiple of 8");
    if (cfg->max_batch <= 0) return fail(UAD_ERR_INVALID, "max_batch must be positive");

    uad_model* m = new uad_model();
    m->cfg = *cfg;
    m->nparams = 0;
    m->step = 0;
    m->have_fwd = false;
    m->prof_on = false;
    m->ar_comm = nullptr; m->ar_world = 1; m->ar_stream = nullptr; m->ar_own_stream = false; m->ar_nb = 0; m->ar_ev_out = nullptr; m->ar_pending = false;
    for (int i = 0; i < 4; ++i) m->ar_ev_in[i] = nullptr;
    const int npool = ilog2i(H) - ilog2i(cfg->inter_res);
    m->n_pool = npool;
    if (npool < 1 || npool > 7) { delete m; return fail(UAD_ERR_UNSUPPORTED, "log2(height/inter_res) = %d: 1..7 conv blocks supported", npool); }
    const bool vae = cfg->arch == UAD_ARCH_VAE || cfg->arch == UAD_ARCH_CEVAE;
    const bool cevae = cfg->arch == UAD_ARCH_CEVAE;
    m->nmul = cevae ? 2 : 1;
    m->data_only = false;
    m->restore = false; m->fb_on_load = false; m->restore_x = nullptr; m->restore_grads = nullptr; m->restore_tv = 0.f; m->restore_lr = 0.f;
    m->restore_scale = 1.0f;
    m->gm_total = 0;
    char nm[128];
    // the GMVAE graph opens no variable scope: plain layer names, BN layers numbered across encoder and decoder
    const char* ENC = gm ? "" : "Encoder/";
    const char* DEC = gm ? "" : "Decoder/";
    int bn_idx = 0;
    auto bn_scope = [&](const char* scope, int ae_idx) {
        std::string r = scope;
        if (gm) { r += bn_idx == 0 ? "batch_normalization" : "batch_normalization_" + std::to_string(bn_idx); ++bn_idx; }
        else r += ae_idx < 0 ? "batch_normalization" : "batch_normalization_" + std::to_string(ae_idx);
        return r;
    };

    // ---- parameter table in TF variable-creation order ----
    int cin = cfg->channels, res = H;
    for (int i = 0; i < npool; ++i) {
        const int f = (32 << i) < 128 ? (32 << i) : 128;
        ConvLayer L;
        L.d = UadConvDesc{1, res, res, cin, res / 2, res / 2, f, 5, 2, 1};
        snprintf(nm, sizeof nm, "%senc_conv2D_%d/kernel", ENC, i); L.w = add_tensor(m, nm, 4, 5, 5, cin, f);
        snprintf(nm, sizeof nm, "%senc_conv2D_%d/bias", ENC, i); L.b = add_tensor(m, nm, 1, f, 1, 1, 1);
        const std::string bs = bn_scope(ENC, i);
        L.gamma = add_tensor(m, bs + "/gamma", 1, f, 1, 1, 1);
        L.beta = add_tensor(m, bs + "/beta", 1, f, 1, 1, 1);
        L.c = nullptr;
        m->enc.push_back(L);
        cin = f; res /= 2;
    }
    m->seg_off[UAD_SEG_ENCODER] = 0; m->seg_cnt[UAD_SEG_ENCODER] = m->nparams;
    {   // ENCODER_LO = [enc0.kernel .. enc1.kernel] (finished by the last two blocks' backward), ENCODER_HI = the rest (enc1.bias ..): within a
        // block the order is kernel, bias, gamma, beta, and block i's backward finishes enc[i].kernel and enc[i-1].{bias, gamma, beta}
        const long long split = m->enc.size() >= 3 ? m->enc[1].b : m->nparams;
        m->seg_off[UAD_SEG_ENCODER_LO] = 0; m->seg_cnt[UAD_SEG_ENCODER_LO] = split;
        m->seg_off[UAD_SEG_ENCODER_HI] = split; m->seg_cnt[UAD_SEG_ENCODER_HI] = m->nparams - split;
    }
    m->cenc = cin; m->cmid = cin / 8;
    const int ir = cfg->inter_res;
    m->flat = ir * ir * m->cmid;
    if (m->cmid % 8 || m->flat % 8) { delete m; return fail(UAD_ERR_UNSUPPORTED, "bottleneck channels must be a multiple of 8"); }
    m->bw = m->bb = m->muw = m->mub = m->sgw = m->sgb = m->dw = m->db = m->rw = m->rb = -1;
    if (gm) {
        // models/gaussian_mixture_variational_autoencoder_spatial.py:14-43, creation (= first call) order
        const int W = cfg->dim_w, Z = cfg->dim_z, Q = cfg->dim_z * cfg->dim_c;
        const char* hn[4] = {"q_wz_x/w_mu", "q_wz_x/w_log_sigma", "q_wz_x/z_mu", "q_wz_x/z_log_sigma"};
        const int hd[4] = {W, W, Z, Z};
        for (int k = 0; k < 4; ++k) {
            m->gm_off[2 * k] = add_tensor(m, std::string(hn[k]) + "/kernel", 4, 1, 1, m->cenc, hd[k]);
            m->gm_off[2 * k + 1] = add_tensor(m, std::string(hn[k]) + "/bias", 1, hd[k], 1, 1, 1);
        }
        m->gm_off[8] = add_tensor(m, "p_z_wc/1x1convlayer/kernel", 4, 1, 1, W, 64);
        m->gm_off[9] = add_tensor(m, "p_z_wc/1x1convlayer/bias", 1, 64, 1, 1, 1);
        m->gm_off[10] = add_tensor(m, "p_z_wc/z_wc_mu/kernel", 4, 1, 1, 64, Q);
        m->gm_off[11] = add_tensor(m, "p_z_wc/z_wc_mu/bias", 1, Q, 1, 1, 1);
        m->gm_off[12] = add_tensor(m, "p_z_wc/z_wc_log_sigma/kernel", 4, 1, 1, 64, Q);
        m->gm_off[13] = add_tensor(m, "p_z_wc/z_wc_log_sigma/bias", 1, Q, 1, 1, 1);
        m->gm_off[14] = add_tensor(m, "Variable", 1, Q, 1, 1, 1);
        m->gm_total = m->nparams - m->gm_off[0];
    } else if (sp) {
        // no bottleneck variables (autoencoder_spatial.py:11-17)
    } else {
    m->bw = add_tensor(m, "Bottleneck/conv2d/kernel", 4, 1, 1, m->cenc, m->cmid);
    m->bb = add_tensor(m, "Bottleneck/conv2d/bias", 1, m->cmid, 1, 1, 1);
    // the ceVAE graph leaves its Dense layers unnamed (context_encoder_variational_autoencoder.py:30-32): keras numbers them
    const char* n_mu = cevae ? "Bottleneck/dense" : "Bottleneck/dense_mu";
    const char* n_sg = cevae ? "Bottleneck/dense_1" : "Bottleneck/dense_sigma";
    const char* n_dec = cevae ? "Bottleneck/dense_2" : "Bottleneck/dense_dec";
    if (vae) {
        m->muw = add_tensor(m, std::string(n_mu) + "/kernel", 2, m->flat, cfg->zdim, 1, 1);
        m->mub = add_tensor(m, std::string(n_mu) + "/bias", 1, cfg->zdim, 1, 1, 1);
        m->sgw = add_tensor(m, std::string(n_sg) + "/kernel", 2, m->flat, cfg->zdim, 1, 1);
        m->sgb = add_tensor(m, std::string(n_sg) + "/bias", 1, cfg->zdim, 1, 1, 1);
    } else {
        m->muw = add_tensor(m, "Bottleneck/dense_z/kernel", 2, m->flat, cfg->zdim, 1, 1);
        m->mub = add_tensor(m, "Bottleneck/dense_z/bias", 1, cfg->zdim, 1, 1, 1);
        m->sgw = m->sgb = -1;
    }
    m->dw = add_tensor(m, std::string(n_dec) + "/kernel", 2, cfg->zdim, m->flat, 1, 1);
    m->db = add_tensor(m, std::string(n_dec) + "/bias", 1, m->flat, 1, 1, 1);
    m->rw = add_tensor(m, "Bottleneck/conv2d_1/kernel", 4, 1, 1, m->cmid, m->cenc);
    m->rb = add_tensor(m, "Bottleneck/conv2d_1/bias", 1, m->cenc, 1, 1, 1);
    }
    m->seg_off[UAD_SEG_BOTTLENECK] = m->seg_cnt[UAD_SEG_ENCODER];
    m->seg_cnt[UAD_SEG_BOTTLENECK] = m->nparams - m->seg_off[UAD_SEG_BOTTLENECK];
    {
        const std::string bs = bn_scope(DEC, -1);
        m->dbn_g = add_tensor(m, bs + "/gamma", 1, m->cenc, 1, 1, 1);
        m->dbn_b = add_tensor(m, bs + "/beta", 1, m->cenc, 1, 1, 1);
    }
    cin = m->cenc; res = ir;
    for (int i = 0; i < npool; ++i) {
        const int f = (128 >> i) > 32 ? (128 >> i) : 32;
        ConvLayer L;
        L.d = UadConvDesc{1, res * 2, res * 2, f, res, res, cin, 5, 2, 1};   // big = output (f ch), small = input (cin ch)
        snprintf(nm, sizeof nm, "%sdec_Conv2DT_%d/kernel", DEC, i); L.w = add_tensor(m, nm, 4, 5, 5, f, cin);
        snprintf(nm, sizeof nm, "%sdec_Conv2DT_%d/bias", DEC, i); L.b = add_tensor(m, nm, 1, f, 1, 1, 1);
        const std::string bs = bn_scope(DEC, i + 1);
        L.gamma = add_tensor(m, bs + "/gamma", 1, f, 1, 1, 1);
        L.beta = add_tensor(m, bs + "/beta", 1, f, 1, 1, 1);
        L.c = nullptr;
        m->dec.push_back(L);
        cin = f; res *= 2;
    }
    m->fw = add_tensor(m, std::string(DEC) + "dec_Conv2D_final/kernel", 4, 1, 1, cin, cfg->channels);
    m->fb = add_tensor(m, std::string(DEC) + "dec_Conv2D_final/bias", 1, cfg->channels, 1, 1, 1);
    m->seg_off[UAD_SEG_DECODER] = m->seg_off[UAD_SEG_BOTTLENECK] + m->seg_cnt[UAD_SEG_BOTTLENECK];
    m->seg_cnt[UAD_SEG_DECODER] = m->nparams - m->seg_off[UAD_SEG_DECODER];
    if (cin > 64 || cin % 4) { delete m; return fail(UAD_ERR_UNSUPPORTED, "last decoder width %d unsupported", cin); }
    if (m->enc[0].d.CS % 8 || 256 % m->enc[0].d.CS) { delete m; return fail(UAD_ERR_UNSUPPORTED, "first conv width"); }

    // ---- device memory ----
    const size_t NB = (size_t)cfg->max_batch * m->nmul;
    int rc = UAD_OK;
#define ALLOC(ptr, n) if (rc == UAD_OK) rc = dev_alloc(m, &(ptr), (n))
    ALLOC(m->params, (size_t)m->nparams); ALLOC(m->grads, (size_t)m->nparams);
    ALLOC(m->adam_m, (size_t)m->nparams); ALLOC(m->adam_v, (size_t)m->nparams);
    ALLOC(m->wpack_f, (size_t)m->nparams); ALLOC(m->wpack_d, (size_t)m->nparams);
    ALLOC(m->wpack16_f, (size_t)m->nparams); ALLOC(m->wpack16_d, (size_t)m->nparams);
    m->math = UAD_MATH_F32;
    m->packed_valid = false; m->pack_inflight = false; m->ev_opt = m->ev_pack = m->ev_pack_head = nullptr; m->pack_head = false;
    size_t maxact = 0;
    for (auto& L : m->enc) { size_t n = NB * L.d.HS * L.d.WS * L.d.CS; ALLOC(L.c, n); if (n > maxact) maxact = n; }
    for (auto& L : m->dec) { size_t n = NB * L.d.HB * L.d.WB * L.d.CB; ALLOC(L.c, n); if (n > maxact) maxact = n; }
    const size_t nz = NB * ((gm || sp) ? 8 : cfg->zdim), nflat = NB * m->flat, ncb = NB * ir * ir * m->cenc;
    ALLOC(m->t, nflat); ALLOC(m->mu_raw, nz); ALLOC(m->ls_raw, nz); ALLOC(m->mu, nz); ALLOC(m->ls, nz);
    ALLOC(m->sigma, nz); ALLOC(m->z, nz); ALLOC(m->dvec, nflat); ALLOC(m->cb, ncb); ALLOC(m->kl, NB);
    ALLOC(m->xhat_own, NB * H * Wd * cfg->channels);
    m->wT_d = m->wT_mu = m->wT_sg = nullptr;
    if (!gm && !sp) { const size_t fz = (size_t)m->flat * cfg->zdim; ALLOC(m->wT_d, fz); ALLOC(m->wT_mu, fz); ALLOC(m->wT_sg, fz); }
    m->xcat = m->mdec_cat = m->l1_own = nullptr;
    if (cevae) { ALLOC(m->xcat, NB * H * Wd * cfg->channels); ALLOC(m->mdec_cat, nflat); ALLOC(m->l1_own, NB * H * Wd * cfg->channels); }
    m->gm_h = m->gm_loc_loss = m->gm_dheads = m->gm_da7 = m->gm_mid = m->gm_dM = m->gm_dLq = m->gm_ws = m->gm_partial = m->gm_dxhat = nullptr;
    if (gm) {
        const size_t L = NB * ir * ir, Q = (size_t)cfg->dim_z * cfg->dim_c, O = 2 * (size_t)cfg->dim_w + 2 * (size_t)cfg->dim_z;
        ALLOC(m->gm_h, L * m->cenc); ALLOC(m->gm_loc_loss, L * 3); ALLOC(m->gm_dheads, L * O); ALLOC(m->gm_da7, L * 64);
        ALLOC(m->gm_mid, L * 64); ALLOC(m->gm_dM, L * Q); ALLOC(m->gm_dLq, L * Q); ALLOC(m->gm_ws, L * cfg->dim_w);
        ALLOC(m->gm_partial, (size_t)64 * m->gm_total); ALLOC(m->gm_dxhat, NB * H * Wd * cfg->channels);
    }
    if (sp) ALLOC(m->gm_h, NB * ir * ir * m->cenc);           // the latent feature map z
    if (cfg->arch == UAD_ARCH_VAE) ALLOC(m->gm_dxhat, NB * H * Wd * cfg->channels);     // restoration mode (trainers/VAE_You.py)
    m->dec_in0 = (gm || sp) ? m->gm_h : m->cb;
    ALLOC(m->G0, maxact); ALLOC(m->G1, maxact);
    { float* fbw = nullptr; ALLOC(fbw, NB * H * Wd); m->fin_bits = reinterpret_cast<unsigned*>(fbw); ALLOC(m->fin_dxh, NB * H * Wd); }
    m->last_fin_bits = false;
    ALLOC(m->g_small[0], nflat); ALLOC(m->g_small[1], nz); ALLOC(m->g_small[2], nz); ALLOC(m->g_small[3], nz);
    ALLOC(m->g_small[4], nflat); ALLOC(m->g_small[5], nflat);
    ALLOC(m->dcb_keep, NB * ir * ir * m->cenc);
    ALLOC(m->bott_xch, NB * 4 * 2 * (size_t)cfg->zdim);
    { float* fl = nullptr; ALLOC(fl, NB * 4); m->bott_flags = reinterpret_cast<unsigned*>(fl); m->bott_epoch = 0; }
    m->bott_err_host = m->bott_err_dev = nullptr;
    { float* fw = nullptr; ALLOC(fw, 4); m->bott_fault = reinterpret_cast<unsigned*>(fw); m->opt_epochs.clear(); m->fault_deferred = false; }      // (ALLOC zero-fills)
    if (rc == UAD_OK && hipHostMalloc((void**)&m->bott_err_host, sizeof(unsigned), hipHostMallocMapped) == hipSuccess) {
        *m->bott_err_host = 0u;
        if (hipHostGetDevicePointer((void**)&m->bott_err_dev, m->bott_err_host, 0) != hipSuccess) m->bott_err_dev = nullptr;
    }
    ALLOC(m->bnfin_scratch, uad_bn_grad_finalize_scratch_floats(512));
    ALLOC(m->bott_wpart, NB * 4 * (2 * (size_t)m->cenc * m->cmid + m->cmid));
    // column-partial scratch: worst case 64-row tiles
    size_t cp = 0;
    auto cp_need = [&](size_t rows, int classes, int C) { size_t v = ((rows + 63) / 64) * classes * 2 * C; if (v > cp) cp = v; };
    for (auto& L : m->enc) cp_need(NB * L.d.HS * L.d.WS, 4, L.d.CB);
    for (auto& L : m->dec) cp_need(NB * L.d.HS * L.d.WS, 1, L.d.CS);
    cp_need(NB * ir * ir, 1, m->cenc);
    if (NB * 4 * 2 * m->cenc > cp) cp = NB * 4 * 2 * m->cenc;      // fused bottleneck backward: one row pair per workgroup, 4 per sample
    if (gm) { size_t v = NB * ir * ir * 2 * m->cenc; if (v > cp) cp = v; }
    if (sp) { size_t v = (size_t)512 * 2 * m->cenc; if (v > cp) cp = v; }
    m->colpart_cap = cp; ALLOC(m->colpart, cp);
    for (int k = 0; k < 16; ++k) { m->cp_slot[k] = nullptr; ALLOC(m->cp_slot[k], cp); }
    m->ev_next = 0; m->side = nullptr;
    if (rc == UAD_OK && hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking) != hipSuccess) rc = fail(UAD_ERR_HIP, "hipStreamCreate failed");
    size_t wp = 0;
    auto wp_need = [&](UadConvDesc d) { d.N = (int)NB; size_t v = uad_conv_w_partial_floats(d); if (v > wp) wp = v; };
    for (size_t i = 1; i < m->enc.size(); ++i) wp_need(m->enc[i].d);
    for (auto& L : m->dec) wp_need(L.d);
    if (!gm && !sp) {
        wp_need(conv1x1_desc(1, ir, ir, m->cenc, m->cmid)); wp_need(conv1x1_desc(1, ir, ir, m->cmid, m->cenc));
        wp_need(dense_desc(1, m->flat, cfg->zdim)); wp_need(dense_desc(1, cfg->zdim, m->flat));
    }
    { UadConvDesc d0 = m->enc[0].d; d0.N = (int)NB; size_t v = uad_conv_first_wgrad_partial_floats(d0); if (v > wp) wp = v; }
    m->wpartial_cap = wp; ALLOC(m->wpartial, wp);
    for (int k = 0; k < 16; ++k) { m->wp_slot[k] = nullptr; ALLOC(m->wp_slot[k], wp); }
    {
        size_t need = (size_t)4 << 20;
        auto want = [&](UadConvDesc d, bool f, bool pack) { d.N = (int)NB; size_t v = uad_conv_ws_floats(d, f, pack); if (v > need) need = v; };
        for (size_t i = 1; i < m->enc.size(); ++i) { want(m->enc[i].d, true, true); want(m->enc[i].d, false, true); }
        for (auto& L : m->dec) { want(L.d, true, true); want(L.d, false, true); }
        if (!gm && !sp) {
        want(conv1x1_desc(1, ir, ir, m->cenc, m->cmid), true, false); want(conv1x1_desc(1, ir, ir, m->cenc, m->cmid), false, false);
        want(conv1x1_desc(1, ir, ir, m->cmid, m->cenc), true, false); want(conv1x1_desc(1, ir, ir, m->cmid, m->cenc), false, false);
        want(dense_desc(1, m->flat, cfg->zdim), true, false); want(dense_desc(1, m->flat, cfg->zdim), false, false);
        want(dense_desc(1, cfg->zdim, m->flat), true, false); want(dense_desc(1, cfg->zdim, m->flat), false, false);
        }
        m->ws.floats = need; m->ws.ptr = nullptr;
        ALLOC(m->ws.ptr, need);
        // arrival counters of the in-kernel split-K reduction: one per (spatial tile, column block) of a split launch
        { float* c = nullptr; m->ws.ncounters = 16384; ALLOC(c, m->ws.ncounters); m->ws.counters = reinterpret_cast<unsigned*>(c); }
    }
    ALLOC(m->colscratch, 64 * 1024);
    const int bps = uad_final_blocks_per_sample(H, Wd);
    ALLOC(m->red_partial, NB * bps * (3 * cin + 1)); ALLOC(m->rec_partial, NB * bps);
    ALLOC(m->rec_ps, NB); ALLOC(m->scalars_own, 8);
#undef ALLOC
    if (rc != UAD_OK) { uad_destroy(m); return rc; }
    *out = m;
    return UAD_OK;
}

int uad_destroy(uad_model_t* m) {
    if (!m) return UAD_OK;
    for (void* p : m->allocs) hipFree(p);
    for (hipEvent_t e : m->sync_events) (void)hipEventDestroy(e);
    if (m->ev_opt) (void)hipEventDestroy(m->ev_opt);
    if (m->ev_pack) (void)hipEventDestroy(m->ev_pack);
    if (m->ev_pack_head) (void)hipEventDestroy(m->ev_pack_head);
    for (int i = 0; i < 4; ++i) if (m->ar_ev_in[i]) (void)hipEventDestroy(m->ar_ev_in[i]);
    if (m->ar_ev_out) (void)hipEventDestroy(m->ar_ev_out);
    if (m->ar_own_stream && m->ar_stream) (void)hipStreamDestroy(m->ar_stream);
    if (m->side) (void)hipStreamDestroy(m->side);
    if (m->bott_err_host) (void)hipHostFree(m->bott_err_host);
    delete m;
    return UAD_OK;
}

long long uad_param_count(const uad_model_t* m) { return m ? m->nparams : 0; }
int uad_num_tensors(const uad_model_t* m) { return m ? (int)m->tensors.size() : 0; }

int uad_tensor_info(const uad_model_t* m, int idx, char* name, int name_cap, long long* offset, int* rank, int* shape4) {
    if (!m || idx < 0 || idx >= (int)m->tensors.size()) return fail(UAD_ERR_INVALID, "tensor index out of range");
    const Tensor& t = m->tensors[idx];
    if (name && name_cap > 0) { strncpy(name, t.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (offset) *offset = t.off;
    if (rank) *rank = t.rank;
    if (shape4) for (int i = 0; i < 4; ++i) shape4[i] = t.shape[i];
    return UAD_OK;
}

static void invalidate_pack(uad_model* m);
static int check_fault(uad_model* m);
static float* buffer_ptr(uad_model* m, int which) {      // no side effect: the readers inside the library
    if (!m) return nullptr;
    switch (which) {
        case UAD_BUF_PARAMS: return m->params;
        case UAD_BUF_GRADS: return m->grads;
        case UAD_BUF_ADAM_M: return m->adam_m;
        case UAD_BUF_ADAM_V: return m->adam_v;
    }
    return nullptr;
}
float* uad_buffer(uad_model_t* m, int which) {
    // the caller may WRITE through the parameter pointer (DP broadcast, checkpoint restore): an in-flight repack is waited for and the packed
    // copies are marked stale.  Read-only users inside the library (uad_get_buffer: checkpoint save) go through buffer_ptr and trigger neither.
    if (m && which == UAD_BUF_PARAMS) invalidate_pack(m);
    return buffer_ptr(m, which);
}

int uad_grad_segment(const uad_model_t* m, int segment, long long* offset, long long* count) {
    if (!m || segment < 0 || segment > UAD_SEG_ENCODER_LO) return fail(UAD_ERR_INVALID, "bad segment");
    if (offset) *offset = m->seg_off[segment];
    if (count) *count = m->seg_cnt[segment];
    return UAD_OK;
}

int uad_set_buffer(uad_model_t* m, int which, const float* host, long long count) {
    float* p = buffer_ptr(m, which);
    if (!p || !host || count != m->nparams) return fail(UAD_ERR_INVALID, "set_buffer: bad arguments (count=%lld, expected %lld)", count, m ? m->nparams : -1);
    if (which == UAD_BUF_PARAMS) invalidate_pack(m);
    HIP_TRY(hipMemcpy(p, host, (size_t)count * sizeof(float), hipMemcpyHostToDevice));
    return UAD_OK;
}
int uad_get_buffer(uad_model_t* m, int which, float* host, long long count) {
    float* p = buffer_ptr(m, which);
    if (!p || !host || count != m->nparams) return fail(UAD_ERR_INVALID, "get_buffer: bad arguments");
    HIP_TRY(hipDeviceSynchronize());
    if (!m->fault_deferred) { const int frc = check_fault(m); if (frc != UAD_OK) return frc; }      // (a checkpoint must not be written across an unreported fault; deferred mode: the trainer has agreed on the word before it saves)
    HIP_TRY(hipMemcpy(host, p, (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
    return UAD_OK;
}
int uad_set_params(uad_model_t* m, const float* host, long long count) { return uad_set_buffer(m, UAD_BUF_PARAMS, host, count); }
int uad_get_params(uad_model_t* m, float* host, long long count) { return uad_get_buffer(m, UAD_BUF_PARAMS, host, count); }

int uad_reset_optimizer(uad_model_t* m) {
    if (!m) return fail(UAD_ERR_INVALID, "null model");
    HIP_TRY(hipMemset(m->adam_m, 0, (size_t)m->nparams * sizeof(float)));
    HIP_TRY(hipMemset(m->adam_v, 0, (size_t)m->nparams * sizeof(float)));
    m->step = 0;
    return UAD_OK;
}
long long uad_get_step(const uad_model_t* m) { return m ? m->step : 0; }
int uad_set_step(uad_model_t* m, long long t) { if (!m || t < 0) return fail(UAD_ERR_INVALID, "bad step"); m->step = t; return UAD_OK; }

// packed (bf16 hi|lo or fp32) copies of the 5x5 kernels + transposed dense kernels: the forms the conv / fused bottleneck kernels read
// head_done (optional): the second conv block's tensor -- the FIRST packed-weight consumer of a forward -- is packed by a launch of its own and the
// event recorded behind it; everything else follows.  Round 5: the step's timeline showed enc1.fwd waiting ~20 us at the head of every step for the
// whole repack (17 us of packing + the dense transposes + two event hops behind the optimizer step) although its own tensor is 3 % of the bytes.
// part: 0 = everything; 1 = the head tensor only (enc[1]); 2 = everything but the head tensor
static void pack_weights(uad_model* m, hipStream_t st, hipEvent_t head_done = nullptr, int part = 0) {
    const bool gm = m->cfg.arch == UAD_ARCH_GMVAE_SPATIAL, sp = m->cfg.arch == UAD_ARCH_AE_SPATIAL;
    long long offs[16]; int cbs[16], css[16], taps[16]; int np = 0;
    auto add = [&](const ConvLayer& L) { if (np < 16 && L.d.CB % 4 == 0 && L.d.CS % 4 == 0) { offs[np] = L.w; cbs[np] = L.d.CB; css[np] = L.d.CS; taps[np] = 25; ++np; } };
    auto flush = [&]() {
        if (np > 0) {
            if (m->math == UAD_MATH_BF16X3)
                uad_launch_pack_weights_bf16(m->params, (unsigned short*)m->wpack16_f, (unsigned short*)m->wpack16_d, offs, cbs, css, taps, np, st);
            else
                uad_launch_pack_weights(m->params, m->wpack_f, m->wpack_d, offs, cbs, css, taps, np, st);
            if (m->math == UAD_MATH_BF16X6)
                uad_launch_pack_weights_bf16_3p(m->params, (unsigned short*)m->wpack3_f, (unsigned short*)m->wpack3_d, offs, cbs, css, taps, np, st);
        }
        np = 0;
    };
    size_t first = 1;
    if (part == 1) { if (m->enc.size() > 1) { add(m->enc[1]); flush(); } return; }
    if (part == 2) first = 2;
    if (head_done) {
        if (m->enc.size() > 1) { add(m->enc[1]); flush(); first = 2; }
        (void)hipEventRecord(head_done, st);
    }
    for (size_t i = first; i < m->enc.size(); ++i) add(m->enc[i]);
    for (auto& L : m->dec) add(L);
    flush();
    if (!gm && !sp) {
        // transposed copies of the dense kernels for the fused bottleneck backward
        const float* tin[3] = {P(m, m->dw), P(m, m->muw), m->sgw >= 0 ? P(m, m->sgw) : nullptr};
        float* tout[3] = {m->wT_d, m->wT_mu, m->wT_sg};
        const int tr[3] = {m->cfg.zdim, m->flat, m->flat}, tc[3] = {m->flat, m->cfg.zdim, m->cfg.zdim};
        uad_launch_transpose(tin, tr, tc, tout, m->sgw >= 0 ? 3 : 2, st);
    }
}
// parameters changed (optimizer step): repack on SIDE, overlapped with whatever the caller enqueues before the next forward's first
// packed-weight consumer (noise draw, batch gather, the first layer)
static void repack_on_side(uad_model* m, hipStream_t st) {
    static const bool off = getenv("UAD_NO_SIDE_PACK") != nullptr;
    m->packed_valid = false;
    if (off) { m->pack_inflight = false; return; }
    if (!m->ev_opt) {
        if (hipEventCreateWithFlags(&m->ev_opt, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) (void)hipEventCreateWithFlags(&m->ev_opt, hipEventDisableTiming);
        if (hipEventCreateWithFlags(&m->ev_pack, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) (void)hipEventCreateWithFlags(&m->ev_pack, hipEventDisableTiming);
        if (hipEventCreateWithFlags(&m->ev_pack_head, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) (void)hipEventCreateWithFlags(&m->ev_pack_head, hipEventDisableTiming);
    }
    static const bool no_head = getenv("UAD_NO_PACK_HEAD") != nullptr;      // A/B: one event behind the whole repack, as before round 5
    (void)hipEventRecord(m->ev_opt, st);
    (void)hipStreamWaitEvent(m->side, m->ev_opt, 0);
    m->pack_head = !no_head;
    // Round 6: the head tensor -- the next forward's FIRST packed-weight consumer's, 3 % of the bytes -- is packed on the CALLER's stream right behind the
    // optimizer step, the rest on SIDE: a forward on the same stream then needs no cross-stream wait in front of enc1.fwd (the step timeline showed ~13 us
    // between conv_first and enc1.fwd although the head pack had finished long before: the price of the wait itself; same-box A/B -0.5 % per step,
    // profiles/r06_f_pack_head_main_ab.log).  A forward on ANOTHER stream waits for ev_pack_head as before.  UAD_NO_PACK_HEAD=1: one event behind the whole repack.
    m->pack_head_main = m->pack_head;
    if (m->pack_head_main) {
        pack_weights(m, st, nullptr, 1);
        (void)hipEventRecord(m->ev_pack_head, st);
        m->pack_head_stream = st;
        pack_weights(m, m->side, nullptr, 2);
    } else pack_weights(m, m->side, nullptr);
    (void)hipEventRecord(m->ev_pack, m->side);
    m->pack_inflight = true;
}
// parameters are about to change from the host side: an in-flight repack must not race with it or with the repack that follows
static void invalidate_pack(uad_model* m) {
    if (m->pack_inflight) (void)hipStreamSynchronize(m->side);
    m->packed_valid = false; m->pack_inflight = false;
}

// ------------------------------------------------------------------------------------------------ forward
static int sk_counters(const uad_model* m);
// A timed-out sibling exchange of the fused bottleneck (uad_bott.hip: group_exchange) leaves that step's activations / gradients invalid.  The
// kernels raise two words: a device one, which every optimizer launch reads -- the update of the faulted step, and of every step enqueued
// behind it before the host noticed, is SKIPPED on the device, so parameters and optimizer slots are never touched by invalid gradients -- and a
// pinned host one, which this check turns into an error at the next uad_forward / uad_get_buffer / uad_check_fault.  The step counter is
// rolled back by the number of skipped updates, so a caller that handles the error resumes from a consistent state.
static int check_fault(uad_model* m) {
    if (!m->bott_err_host) return UAD_OK;
    const unsigned e = *reinterpret_cast<volatile unsigned*>(m->bott_err_host);
    if (!e) {
        // no fault so far: calls older than the newest one can no longer be "behind a fault" the host has not seen only if the stream is drained,
        // which this function does not know -- keep the list, but bound it (a fault older than 65536 optimizer calls rolls back at most that many)
        if (m->opt_epochs.size() > 65536) m->opt_epochs.erase(m->opt_epochs.begin(), m->opt_epochs.end() - 32768);
        return UAD_OK;
    }
    (void)hipDeviceSynchronize();            // error path: everything enqueued behind the fault has run (and skipped its update)
    *m->bott_err_host = 0u;
    if (m->bott_fault) (void)hipMemset(m->bott_fault, 0, sizeof(unsigned));
    const unsigned ep = e & 0x7fffffffu;     // launch epoch of the FIRST fault (group_exchange: compare-and-swap from 0)
    int skipped = 0;
    for (unsigned oe : m->opt_epochs) if (oe >= ep) ++skipped;      // every optimizer launch since that epoch found the device word raised
    m->step -= skipped; if (m->step < 0) m->step = 0;
    m->opt_epochs.clear();
    invalidate_pack(m);
    return fail(UAD_ERR_HIP, "fused bottleneck: a workgroup gave up waiting for its sibling workgroups (launch epoch %u); the results of that "
                             "step are invalid and %d optimizer update(s) behind it were skipped on the device (parameters and slots are those "
                             "of the last good step). UAD_BOTT_Q1=1 selects the one-workgroup-per-sample form", ep, skipped);
}
int uad_check_fault(uad_model_t* m, int synchronize, void* stream) {
    if (!m) return fail(UAD_ERR_INVALID, "null model");
    if (synchronize) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    const int rc = check_fault(m);
    if (rc == UAD_OK && synchronize) m->opt_epochs.clear();       // drained and clean: nothing enqueued so far sits behind a fault
    return rc;
}
int uad_set_fault_deferred(uad_model_t* m, int on) {
    if (!m) return fail(UAD_ERR_INVALID, "null model");
    m->fault_deferred = on != 0;
    return UAD_OK;
}
// (Plane-group tensors -- every producer also writing its ACTIVATED output pre-split into bf16 hi | lo groups for the consumers -- were built and
// measured in round 3: parity-green, 2 % slower; removed in round 4, tools/experiments/r03_pruned_opt_in_paths.patch.)
int uad_forward(uad_model_t* m, const uad_io_t* io, int n, int want_backward, void* stream) {
    if (!m || !io) return fail(UAD_ERR_INVALID, "null argument");
    if (n <= 0 || n > m->cfg.max_batch) return fail(UAD_ERR_INVALID, "batch %d outside (0, max_batch=%d]", n, m->cfg.max_batch);
    if (!io->x) return fail(UAD_ERR_INVALID, "io.x is null");
    if (!m->fault_deferred) { const int frc = check_fault(m); if (frc != UAD_OK) return frc; }
    hipStream_t st = (hipStream_t)stream;
    const bool vae = m->cfg.arch == UAD_ARCH_VAE || m->cfg.arch == UAD_ARCH_CEVAE;
    const bool cevae = m->cfg.arch == UAD_ARCH_CEVAE;
    const bool gm = m->cfg.arch == UAD_ARCH_GMVAE_SPATIAL;
    const bool sp = m->cfg.arch == UAD_ARCH_AE_SPATIAL;
    const int ir = m->cfg.inter_res;
    const int nu = n;                 // samples the caller passed
    const float* xin = io->x;
    const float* xtgt = io->x;        // what the L1 term compares the reconstruction with
    const float* mask_dec = io->mask_dec;
    if (!cevae && io->x_ce) {
        // context-encoder training (trainers/CE.py:19-21,87-92): the network reads the masked batch, the loss compares with the clean one
        if (m->cfg.arch != UAD_ARCH_AE && m->cfg.arch != UAD_ARCH_AE_SPATIAL)
            return fail(UAD_ERR_INVALID, "io.x_ce is the ceVAE's second input or an AE handle's context-encoder input");
        xin = io->x_ce;
    }
    if (cevae) {
        // both branches as ONE pass over 2n samples through the shared layers: [x ; x_ce]
        if ((io->mask_mu == nullptr) != (io->mask_mu_ce == nullptr) || (io->mask_dec == nullptr) != (io->mask_dec_ce == nullptr))
            return fail(UAD_ERR_INVALID, "ceVAE: mask_mu/mask_mu_ce and mask_dec/mask_dec_ce must be given together");
        const size_t xb = (size_t)nu * m->cfg.height * m->cfg.width * m->cfg.channels * sizeof(float);
        HIP_TRY(hipMemcpyAsync(m->xcat, io->x, xb, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync((char*)m->xcat + xb, io->x_ce ? io->x_ce : io->x, xb, hipMemcpyDeviceToDevice, st));
        xin = m->xcat;
        if (io->mask_dec) {
            const size_t mb = (size_t)nu * m->flat * sizeof(float);
            HIP_TRY(hipMemcpyAsync(m->mdec_cat, io->mask_dec, mb, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync((char*)m->mdec_cat + mb, io->mask_dec_ce, mb, hipMemcpyDeviceToDevice, st));
            mask_dec = m->mdec_cat;
        }
        n = 2 * nu;
    }

    // refresh the packed 5x5 kernels if the parameters changed since the last pack; after an optimizer step the repack already runs
    // on SIDE (uad_adam_step) and only has to be waited for before the first kernel that reads packed weights
    bool wait_pack = false;
    if (!m->packed_valid) {
        if (m->pack_inflight) wait_pack = true;
        else { PROF("pack.weights"); pack_weights(m, st); }
        m->packed_valid = true; m->pack_inflight = false;
    }
    // encoder
    static const char* kEncF[] = {"enc0.fwd", "enc1.fwd", "enc2.fwd", "enc3.fwd", "enc4.fwd", "enc5.fwd", "enc6.fwd", "enc7.fwd"};
    static const char* kDecF[] = {"dec0.fwd", "dec1.fwd", "dec2.fwd", "dec3.fwd", "dec4.fwd", "dec5.fwd", "dec6.fwd", "dec7.fwd"};
    {
        PROF(kEncF[0]);
        UadConvDesc d = m->enc[0].d; d.N = n;
        uad_launch_conv_first_fwd(d, xin, P(m, m->enc[0].w), P(m, m->enc[0].b), m->enc[0].c, st);
    }
    bool wait_full = wait_pack;
    if (wait_pack && m->pack_head && m->pack_head_main && st == m->pack_head_stream) { }     // enc1's tensor was packed on this stream, in order
    else if (wait_pack && m->pack_head) (void)hipStreamWaitEvent(st, m->ev_pack_head, 0);       // enc1's tensor only; everything else is waited for one layer later
    else if (wait_pack) { (void)hipStreamWaitEvent(st, m->ev_pack, 0); wait_full = false; }
    for (size_t i = 1; i < m->enc.size(); ++i) {
        if (i >= 2 && wait_full) { (void)hipStreamWaitEvent(st, m->ev_pack, 0); wait_full = false; }
        PROF(kEncF[i & 7]);
        UadConvDesc d = m->enc[i].d; d.N = n;
        uad_launch_conv_f(d, m->enc[i - 1].c, bn_xform(m, m->enc[i - 1].gamma, m->enc[i - 1].beta, kLrelu),
                          P(m, m->enc[i].w), m->enc[i].c, epi_bias(P(m, m->enc[i].b)), st, PKF(m, m->enc[i].w), m->ws, PK16F(m, m->enc[i].w), PLANE(m->enc[i]), false, planes_of(m));
    }
    if (wait_full) { (void)hipStreamWaitEvent(st, m->ev_pack, 0); wait_full = false; }
    const ConvLayer& EL = m->enc.back();
    // bottleneck
    if (gm) {
        PROF("gm.heads.fwd");
        UadGmArgs ga = gm_args(m, io->eps_w, io->eps_z, 1.0f / (float)nu);
        ga.w_mu = io->w_mu; ga.w_ls = io->w_log_sigma; ga.z_mu = io->z_mu; ga.z_ls = io->z_log_sigma; ga.pc = io->pc;
        uad_launch_gm_heads_fwd(ga, n * ir * ir, st);
    } else if (sp) {
        PROF("spatial.z.fwd");
        uad_launch_spatial_z_fwd(EL.c, P(m, EL.gamma), P(m, EL.beta), 1.0f / sqrtf(1.0f + kBnEps), kLrelu, io->mask_mu, n * ir * ir, m->cenc,
                                 m->gm_h, st);
    } else if (uad_bottleneck_fused_ok(bott_args(m, *io, mask_dec, nu))) {
        PROF("bott.fwd");
        UadBottArgs ba = bott_args(m, *io, mask_dec, nu);
        ba.epoch = ++m->bott_epoch;
        uad_launch_bottleneck_fwd(ba, n, st);
    } else {
    PROF("bott.fwd");
    uad_launch_conv_f(conv1x1_desc(n, ir, ir, m->cenc, m->cmid), EL.c, bn_xform(m, EL.gamma, EL.beta, kLrelu),
                      P(m, m->bw), m->t, epi_bias(P(m, m->bb)), st, nullptr, m->ws);
    if (vae) {
        uad_launch_conv_f(dense_desc(n, m->flat, m->cfg.zdim), m->t, no_xform(), P(m, m->muw), m->mu_raw, epi_bias(P(m, m->mub)), st, nullptr, m->ws);
        uad_launch_conv_f(dense_desc(n, m->flat, m->cfg.zdim), m->t, no_xform(), P(m, m->sgw), m->ls_raw, epi_bias(P(m, m->sgb)), st, nullptr, m->ws);
        uad_launch_reparam_fwd(n, nu, m->cfg.zdim, m->mu_raw, m->ls_raw, io->mask_mu, io->mask_sigma,
                               cevae ? io->mask_mu_ce : nullptr, io->eps, m->mu, m->ls, m->sigma, m->z, m->kl, st);
        uad_launch_conv_f(dense_desc(n, m->cfg.zdim, m->flat), m->z, no_xform(), P(m, m->dw), m->dvec,
                          epi_bias(P(m, m->db), mask_dec), st, nullptr, m->ws);
    } else {
        uad_launch_conv_f(dense_desc(n, m->flat, m->cfg.zdim), m->t, no_xform(), P(m, m->muw), m->z,
                          epi_bias(P(m, m->mub), io->mask_mu), st, nullptr, m->ws);
        uad_launch_conv_f(dense_desc(n, m->cfg.zdim, m->flat), m->z, no_xform(), P(m, m->dw), m->dvec, epi_bias(P(m, m->db)), st, nullptr, m->ws);
    }
    uad_launch_conv_f(conv1x1_desc(n, ir, ir, m->cmid, m->cenc), m->dvec, no_xform(), P(m, m->rw), m->cb, epi_bias(P(m, m->rb)), st, nullptr, m->ws);
    }
    // decoder
    const ConvLayer& DL = m->dec.back();
    const int bps = uad_final_blocks_per_sample(m->cfg.height, m->cfg.width);
    bool fused_final = false, fin_bits_mode = false, restore_bits = false;
    for (size_t i = 0; i < m->dec.size(); ++i) {
        PROF(kDecF[i & 7]);
        UadConvDesc d = m->dec[i].d; d.N = n;
        const float* in = (i == 0) ? m->dec_in0 : m->dec[i - 1].c;
        UadXform xf = (i == 0) ? bn_xform(m, m->dbn_g, m->dbn_b, 0.0f)
                               : bn_xform(m, m->dec[i - 1].gamma, m->dec[i - 1].beta, kLrelu);
        UadEpilogue ep = epi_bias(P(m, m->dec[i].b));
        float* out = m->dec[i].c;
        const bool bfm = bf_mode(m);
        if (i + 1 == m->dec.size() && (bfm ? uad_conv_d_can_fuse_final(d, true, m->ws.floats) : uad_conv_d_can_fuse_final_f32(d, m->math == UAD_MATH_F32, m->ws.floats)) &&
            (d.HS / 8) * (d.WS / 16) == bps) {
            // last block: its BN + LeakyReLU, the final 1x1 conv, the L1 loss and (training) the loss gradient run in the
            // ConvT kernel's epilogue; the pre-BN output is only written when a later pass needs it (restoration: TV term)
            fused_final = true;
            const bool restore_bwd = m->restore && want_backward;
            ep.kind = UAD_EPI_FINAL;
            ep.escale = P(m, DL.gamma); ep.eshift = P(m, DL.beta); ep.ealpha = kLrelu; ep.emult = 1.0f / sqrtf(1.0f + kBnEps);
            ep.fin_wf = P(m, m->fw); ep.fin_bf = P(m, m->fb); ep.fin_x = cevae ? xin : xtgt;
            ep.fin_xhat = (io->x_hat && !cevae) ? io->x_hat : m->xhat_own;
            ep.fin_l1 = cevae ? ((io->l1_map || io->l1_map_ce) ? m->l1_own : nullptr) : io->l1_map;
            ep.fin_rec_partial = m->rec_partial; ep.fin_red_partial = m->red_partial;
            ep.fin_dc = (want_backward && !restore_bwd) ? m->G0 : nullptr;
            ep.fin_bits = nullptr; ep.fin_dxhat = nullptr;
            if (bfm && ep.fin_dc && d.CB <= 32 && uad_conv_f_supports_final_bwd(d, true, m->ws.floats) &&
                (want_backward == 2 || uad_conv_w_supports_fb_bits(d, true))) {
                // both consumers of d loss / d c (this layer's data- and filter-gradient kernels) can expand it from one pattern word +
                // one float per pixel: 8 B instead of 128 B per pixel written here and read twice in the backward
                ep.fin_dc = nullptr; ep.fin_bits = m->fin_bits; ep.fin_dxhat = m->fin_dxh;
                fin_bits_mode = true;
            }
            ep.fin_inv_batch = 1.0f / (float)nu;
            // Restoration (round 5): the data gradient of the last block needs, per output element, only (d objective / d x_hat of its pixel) and
            // whether its BN output was positive -- the pattern word this epilogue already knows how to write.  With it the block's 128 B / pixel
            // pre-BN output is neither written here nor re-read by the data-gradient kernel (16 slices at 256 x 256: 134 MB each way per
            // iteration); tv_dxhat supplies d / d x_hat from the finished reconstruction as before.  UAD_NO_RESTORE_BITS=1: the round-2 path
            // (pre-BN output written, gradient formed from it on load).
            static const bool no_rbits = getenv("UAD_NO_RESTORE_BITS") != nullptr;
            if (restore_bwd && bfm && !no_rbits && d.CB <= 32 && uad_conv_f_supports_final_bwd(d, true, m->ws.floats)) {
                ep.fin_bits = m->fin_bits; ep.fin_dxhat = m->fin_dxh;      // (fin_dxh receives the L1 sign term only: unused, tv_dxhat's output is what the backward reads)
                fin_bits_mode = true; restore_bits = true;
            }
            out = (restore_bwd && !restore_bits) ? DL.c : nullptr;
        }
        uad_launch_conv_d(d, in, xf, P(m, m->dec[i].w), out, ep, st, PKD(m, m->dec[i].w), m->ws, PK16D(m, m->dec[i].w), PLANE(m->dec[i]), false, planes_of(m));
    }
    // final 1x1 conv + L1 loss (+ start of the backward)
    UadFinalArgs fa;
    fa.N = n; fa.H = m->cfg.height; fa.W = m->cfg.width; fa.C = DL.d.CB;
    fa.c_last = DL.c; fa.scale = P(m, DL.gamma); fa.shift = P(m, DL.beta); fa.alpha = kLrelu;
    fa.mult = 1.0f / sqrtf(1.0f + kBnEps);
    fa.wf = P(m, m->fw); fa.bf = P(m, m->fb); fa.x = cevae ? xin : xtgt;
    fa.x_hat = (io->x_hat && !cevae) ? io->x_hat : m->xhat_own;
    fa.l1_map = cevae ? ((io->l1_map || io->l1_map_ce) ? m->l1_own : nullptr) : io->l1_map;
    fa.rec_partial = m->rec_partial;
    fa.d_c = want_backward ? m->G0 : nullptr;
    fa.red_partial = m->red_partial;
    fa.inv_batch = 1.0f / (float)nu;
    fa.dxhat_in = nullptr;
    if (fused_final && !(m->restore && want_backward)) {
        // everything already happened in the last ConvT's epilogue
    } else if (m->restore && want_backward) {
        // restoration objective: d / d x_hat needs the finished reconstruction's neighbours (TV), so two passes
        PROF("final.fwd+tv+bwd");
        float* dc = fa.d_c; fa.d_c = nullptr;
        if (!fused_final) uad_launch_final_fwd_bwd(fa, st);
        uad_launch_tv_dxhat(xin, fa.x_hat, n, fa.H, fa.W, m->restore_scale, m->restore_tv, m->gm_dxhat, st);
        fa.d_c = dc; fa.dxhat_in = m->gm_dxhat;
        m->fb_on_load = fused_final && !restore_bits && restore_fb_on_load(m, n);
        if (!m->fb_on_load && !restore_bits) uad_launch_final_fwd_bwd(fa, st);   // else: folded into dec.back()'s data gradient
    } else {
        PROF(want_backward ? "final.fwd+bwd" : "final.fwd"); uad_launch_final_fwd_bwd(fa, st);
    }
    if (m->restore) {
        // restoration iterations fetch only `grads` (trainers/GMVAE_spatial.py:186): no loss scalars
    } else if (gm) {
        PROF("loss.finalize");
        uad_launch_gm_loss_finalize(m->rec_partial, n, bps, m->gm_loc_loss, ir * ir, 1.0f / (float)nu,
                                    io->rec_per_sample ? io->rec_per_sample : m->rec_ps,
                                    io->scalars ? io->scalars : m->scalars_own, st);
    } else {
    PROF("loss.finalize");
    uad_launch_loss_finalize(m->rec_partial, n, nu, bps, vae ? m->kl : nullptr, 1.0f / (float)nu, cevae ? 0.5f : 1.0f,
                             io->rec_per_sample ? io->rec_per_sample : m->rec_ps,
                             io->scalars ? io->scalars : m->scalars_own, st);
    }
    if (cevae) {
        const size_t xe = (size_t)nu * m->cfg.height * m->cfg.width * m->cfg.channels, xb = xe * sizeof(float);
        if (io->x_hat) HIP_TRY(hipMemcpyAsync(io->x_hat, m->xhat_own, xb, hipMemcpyDeviceToDevice, st));
        if (io->x_hat_ce) HIP_TRY(hipMemcpyAsync(io->x_hat_ce, m->xhat_own + xe, xb, hipMemcpyDeviceToDevice, st));
        if (io->l1_map) HIP_TRY(hipMemcpyAsync(io->l1_map, m->l1_own, xb, hipMemcpyDeviceToDevice, st));
        if (io->l1_map_ce) HIP_TRY(hipMemcpyAsync(io->l1_map_ce, m->l1_own + xe, xb, hipMemcpyDeviceToDevice, st));
    }
    // optional latent outputs (VAE-branch samples)
    const size_t zb = (gm || sp) ? 0 : (size_t)nu * m->cfg.zdim * sizeof(float);
    if (sp && io->z_mu) hipMemcpyAsync(io->z_mu, m->gm_h, (size_t)nu * ir * ir * m->cenc * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (!gm && !sp && io->z_mu) hipMemcpyAsync(io->z_mu, vae ? m->mu : m->z, zb, hipMemcpyDeviceToDevice, st);
    if (vae && io->z_log_sigma) hipMemcpyAsync(io->z_log_sigma, m->ls, zb, hipMemcpyDeviceToDevice, st);
    if (vae && io->z_sigma) hipMemcpyAsync(io->z_sigma, m->sigma, zb, hipMemcpyDeviceToDevice, st);
    m->last_n = n; m->last_nuser = nu; m->last_io = *io; m->have_fwd = want_backward != 0;
    m->x_eff = xin; m->mask_dec_eff = mask_dec; m->data_only = want_backward == 2;
    m->last_fused_final = fused_final && !(m->restore && want_backward);
    m->last_fin_bits = fin_bits_mode;
    // the first filter gradient of the backward may then start beside that single-workgroup kernel instead of behind it (UAD_NO_ANYORDER=1: off)
    m->fwd_tail_is_loss = !m->restore && !gm && !cevae && !io->z_mu && !io->z_log_sigma && !io->z_sigma;      // (no copy was enqueued behind it)
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}

// ------------------------------------------------------------------------------------------------ backward
// Two streams.  MAIN (the caller's stream) runs every heavy kernel: the data-gradient chain and the k5 s2 filter
// gradients.  SIDE (handle-owned) runs the small kernels that only finish parameter gradients -- split-K slab
// reductions, BN/bias finalizes, the bottleneck's dense weight gradients and column sums -- so that they overlap with
// the next heavy kernel instead of each costing a serialized 4-10 us.  Edges are hipEvents; every segment ends with MAIN
// waiting for SIDE, so the caller (Adam, or the DP all-reduce of that gradient segment) sees complete gradients.
// Scratch touched by SIDE is per layer (column partials, split-K slabs), so MAIN never overwrites what SIDE still reads.
// counters the split launches of this handle may use for the in-kernel reduction (split-bf16 mode only: the fp32 kernels have no such path)
static int sk_counters(const uad_model* m) { return (bf_mode(m) && m->ws.counters) ? m->ws.ncounters : 0; }
static hipEvent_t next_event(uad_model* m) {
    if (m->ev_next == m->sync_events.size()) {
        hipEvent_t e;
        // same-device stream ordering only: no system-scope fence (cache writeback) at the record
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess)
            (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        m->sync_events.push_back(e);
    }
    return m->sync_events[m->ev_next++];
}
// Ordering contract of an edge recorded behind an ANY-ORDER launch (a layer's data gradient is launched without the AQL barrier bit, so it may
// finish before or after the filter gradient enqueued ahead of it; the side stream's slab reduction needs BOTH).  hipEventRecord on ROCm
// (ROCclr, 6.x / 7.x) always submits its own marker command -- an AQL barrier-AND packet with the BARRIER BIT SET -- and the packet processor
// does not consume a barrier-bit packet until every earlier packet of that queue has COMPLETED, whatever those packets' own barrier bits were.
// The event's signal is that marker's completion signal, not the last dispatch's, so a waiter on `to` starts after the filter gradient AND
// the data gradient.  Pinned by tests/test_gpu_knobs.py::test_any_order_edge_waits_for_the_slower_filter_gradient: with UAD_W_ABL=32 every
// filter-gradient workgroup sleeps ~100 us before writing its slab (it then outlasts the data gradient by far) and the gradients of a step on
// fresh inputs must equal, bit for bit, those of a UAD_NO_ANYORDER run -- stale or half-written slabs would show.
static void edge(uad_model* m, hipStream_t from, hipStream_t to) {
    hipEvent_t e = next_event(m);
    (void)hipEventRecord(e, from);
    (void)hipStreamWaitEvent(to, e, 0);
}
#define PROF_ON(tag, stream) ProfScope prof_scope_s_##__LINE__(m, tag, stream)
// the side stream joined into the caller's stream: everything the handle has enqueued so far is complete in the caller's stream order
static void join_side(uad_model* m, hipStream_t st) { edge(m, m->side, st); m->joined = true; }

static int backward_decoder(uad_model* m, hipStream_t st, bool join_now) {
    const int n = m->last_n;
    const float rstd = 1.0f / sqrtf(1.0f + kBnEps);
    hipStream_t sd = m->side;
    const bool bf = bf_mode(m);
    const ConvLayer& DL = m->dec.back();
    const int C = DL.d.CB;
    const int bps = uad_final_blocks_per_sample(m->cfg.height, m->cfg.width);
    const int T = n * bps, L = 3 * C + 1;
    static const char* kDecW[] = {"dec0.wgrad", "dec1.wgrad", "dec2.wgrad", "dec3.wgrad", "dec4.wgrad", "dec5.wgrad", "dec6.wgrad", "dec7.wgrad"};
    static const char* kDecD[] = {"dec0.dgrad", "dec1.dgrad", "dec2.dgrad", "dec3.dgrad", "dec4.dgrad", "dec5.dgrad", "dec6.dgrad", "dec7.dgrad"};
    const bool pg = !m->data_only;   // parameter gradients wanted
    m->ev_next = 0;
    float* g = m->G0;      // d loss / d c of dec[i]
    float* gn = m->G1;
    for (int i = (int)m->dec.size() - 1; i >= 0; --i) {
        UadConvDesc d = m->dec[i].d; d.N = n;
        const float* in = (i == 0) ? m->dec_in0 : m->dec[i - 1].c;
        const long long ig = (i == 0) ? m->dbn_g : m->dec[i - 1].gamma;
        const long long ib = (i == 0) ? m->dbn_b : m->dec[i - 1].beta;
        const float ia = (i == 0) ? 0.0f : kLrelu;
        const long long ibias = (i == 0) ? m->rb : m->dec[i - 1].b;
        float* cp = m->cp_slot[i];
        // filter gradient on MAIN (big = d c raw, small = layer input with activation on load); slab reduce on SIDE
        const bool last = i + 1 == (int)m->dec.size();
        const bool fbb = last && m->last_fin_bits;      // d loss / d c of the last block exists only as pattern bits + d objective / d x_hat
        UadXform gbits = no_xform();
        if (fbb) { gbits = bn_xform(m, DL.gamma, DL.beta, kLrelu); gbits.fb_dxhat = m->restore ? m->gm_dxhat : m->fin_dxh; gbits.fb_wf = P(m, m->fw); gbits.fb_bits = m->fin_bits; }      // (restoration: d objective / d x_hat incl. the TV term, from tv_dxhat)
        { static const bool anyo = getenv("UAD_NO_ANYORDER") == nullptr;
          if (anyo && pg && bf && last && m->fwd_tail_is_loss && !m->prof_on) uad_conv_w_any_order_next(true);
          m->fwd_tail_is_loss = false; }
        if (pg) { PROF(kDecW[i & 7]); uad_launch_conv_w(d, g, fbb ? gbits : no_xform(), in, bn_xform(m, ig, ib, ia), Gr(m, m->dec[i].w), m->wp_slot[i], st, bf, nullptr, nullptr, false, true, planes_of(m)); }
        uad_conv_w_any_order_next(false);
        // data gradient (F-type on the ConvT kernel) fused with the producer's activation backward
        { static const bool anyo = getenv("UAD_NO_ANYORDER") == nullptr; if (anyo && pg) uad_conv_any_order_next(true); }
        { PROF(kDecD[i & 7]); UadEpilogue e = epi_bwd(m, in, ig, ib, ia); e.colpart = cp;
          const bool fb = m->restore && m->fb_on_load && last;
          UadXform gx = fbb ? gbits : no_xform();
          if (fb) { gx = bn_xform(m, DL.gamma, DL.beta, kLrelu); gx.fb_dxhat = m->gm_dxhat; gx.fb_wf = P(m, m->fw); }
          uad_launch_conv_f(d, fb ? DL.c : g, gx, P(m, m->dec[i].w), gn, e, st, PKF(m, m->dec[i].w), m->ws, PK16F(m, m->dec[i].w), PLANE(m->dec[i]), false, planes_of(m));
          uad_conv_any_order_next(false); }    // (a launch that did not take a spatial kernel must not leave the request to a later one)
        // ONE edge per layer: its filter-gradient slabs and column partials are ready.  Without parameter gradients (restoration, anomaly maps) nothing runs on SIDE and the
        // event record -- a marker packet with the barrier bit on the MAIN queue -- is skipped: -2.3 % per restoration iteration.  (Releasing the side work of SEVERAL layers behind
        // one edge was measured too, round 6: 0.808 -> 0.817 / 0.839 / 0.848 ms per VAE step at 2 / 3 / 4 layers per edge -- the side stream falls behind; one edge per layer stays.)
        if (pg) edge(m, st, sd);
        if (pg && last) {
            // final conv kernel/bias grads + BN grads of the last block from the fused loss kernel's partials (forward results; riding on
            // this layer's edge instead of one of their own): red_partial[T][3C+1] = {dwf[C], S1[C], S2[C], dbf}
            PROF_ON("final.gradfin", sd);
            uad_launch_reduce_partials(m->red_partial, T, L, 1.0f, m->colscratch, sd);
            uad_launch_final_gradfin(m->colscratch, C, P(m, DL.gamma), rstd, Gr(m, m->fw), Gr(m, m->fb), Gr(m, DL.gamma), Gr(m, DL.beta), Gr(m, DL.b), sd);
        }
        if (pg) { PROF_ON("bn.gradfin", sd); uad_launch_conv_w_reduce(d, Gr(m, m->dec[i].w), m->wp_slot[i], sd, bf); uad_launch_bn_grad_finalize(cp, uad_conv_f_tiles(d, true, m->ws.floats, sk_counters(m), planes_of(m)), d.CS, P(m, ig), rstd, Gr(m, ig), Gr(m, ib), ibias >= 0 ? Gr(m, ibias) : nullptr, sd, m->bnfin_scratch); }
        float* tsw = g; g = gn; gn = tsw;
    }
    m->G0 = g; m->G1 = gn;   // G0 = d loss / d cb (pre-BN output of Bottleneck/conv2d_1)
    if (join_now) join_side(m, st);        // join: decoder gradients complete (inside UAD_SEG_ALL the join is the encoder segment's)
    return UAD_OK;
}

// join_now: the caller asked for this gradient segment on its own (DP all-reduce), so SIDE must be joined before returning;
// inside uad_backward(UAD_SEG_ALL) the join is left to the encoder segment's end.
static int backward_bottleneck(uad_model* m, hipStream_t st, bool join_now) {
    const int n = m->last_n, nu = m->last_nuser;
    const bool vae = m->cfg.arch != UAD_ARCH_AE;
    const bool cevae = m->cfg.arch == UAD_ARCH_CEVAE;
    const bool pg = !m->data_only;
    const int ir = m->cfg.inter_res, zd = m->cfg.zdim;
    const float rstd = 1.0f / sqrtf(1.0f + kBnEps);
    const uad_io_t& io = m->last_io;
    hipStream_t sd = m->side;
    float* dcb = m->G0;                         // [n,ir,ir,cenc]
    float* dd = m->g_small[0];                  // [n,flat]
    float* dz = m->g_small[1];
    float* dmu = m->g_small[2];
    float* dls = m->g_small[3];
    float* dflat = m->g_small[4];
    float* cp = m->cp_slot[15];
    float* wp = m->wp_slot[15];                 // SIDE-only slab scratch of the bottleneck weight gradients
    const UadConvDesc d_r = conv1x1_desc(n, ir, ir, m->cmid, m->cenc);
    const UadConvDesc d_dec = dense_desc(n, zd, m->flat);
    const UadConvDesc d_in = dense_desc(n, m->flat, zd);
    const UadConvDesc d_b = conv1x1_desc(n, ir, ir, m->cenc, m->cmid);
    const ConvLayer& EL = m->enc.back();
    {
        UadBottArgs ba = bott_args(m, io, m->mask_dec_eff, nu);
        if (m->restore) ba.inv_batch = m->restore_scale;          // VAE_You: d (rec_n + kl_n) / d x per sample, no 1/n
        if (uad_bottleneck_fused_ok(ba)) {
            // MAIN: one workgroup per sample does the whole data-gradient chain.  SIDE: the parameter-gradient GEMMs, from the vectors
            // it leaves behind (one edge, no wait of MAIN on SIDE).
            ba.dcb = dcb; ba.dd = dd; ba.dmu = vae ? dmu : dz; ba.dls = dls; ba.dflat = dflat; ba.g_out = m->G1; ba.colpart = cp;
            ba.wpart = pg ? m->bott_wpart : nullptr;      // shares of conv2d / conv2d_1's parameter gradients, summed by the SIDE kernel
            ba.dcb_copy = pg ? m->dcb_keep : nullptr;     // conv2d_1's kernel gradient reads this copy on SIDE: dcb's buffer becomes encoder scratch
            ba.epoch = ++m->bott_epoch;
            { PROF("bott.bwd"); uad_launch_bottleneck_bwd(ba, n, st); }
            if (pg) edge(m, st, sd);
            UadBottWgradArgs wa;
            memset(&wa, 0, sizeof wa);
            wa.n = n; wa.cenc = m->cenc; wa.cmid = m->cmid; wa.npos = ir * ir; wa.zdim = zd; wa.alpha = kLrelu; wa.mult = rstd;
            wa.z = m->z; wa.dd = dd; wa.t = m->t; wa.dmu = vae ? dmu : dz; wa.dls = vae ? dls : nullptr;
            wa.c_enc = EL.c; wa.scale = P(m, EL.gamma); wa.shift = P(m, EL.beta);
            wa.dflat = dflat; wa.dvec = m->dvec; wa.dcb = m->dcb_keep;
            wa.gWd = Gr(m, m->dw); wa.gbd = Gr(m, m->db); wa.gWmu = Gr(m, m->muw); wa.gbmu = Gr(m, m->mub);
            if (vae) { wa.gWsg = Gr(m, m->sgw); wa.gbsg = Gr(m, m->sgb); }
            wa.gWb = Gr(m, m->bw); wa.gbb = Gr(m, m->bb); wa.gWr = Gr(m, m->rw);
            wa.part = m->bott_wpart; wa.nparts = uad_bottleneck_colpart_rows(ba, n);
            if (pg && uad_bottleneck_wgrad_ok(wa)) {
                // SIDE: every parameter gradient of the segment in one launch (+ the last encoder block's BN finalize)
                PROF_ON("bott.wgrad", sd);
                uad_launch_bottleneck_wgrad(wa, sd);
                uad_launch_bn_grad_finalize(cp, uad_bottleneck_colpart_rows(ba, n), m->cenc, P(m, EL.gamma), rstd, Gr(m, EL.gamma), Gr(m, EL.beta), Gr(m, EL.b), sd);
            } else if (pg) {
                PROF_ON("bott.wgrad", sd);
                uad_launch_conv_w(d_r, m->dvec, no_xform(), m->dcb_keep, no_xform(), Gr(m, m->rw), wp, sd);
                uad_launch_conv_w(d_dec, m->z, no_xform(), dd, no_xform(), Gr(m, m->dw), wp, sd);
                uad_launch_colsum(dd, n, m->flat, Gr(m, m->db), m->colscratch, sd);
                uad_launch_conv_w(d_in, m->t, no_xform(), vae ? dmu : dz, no_xform(), Gr(m, m->muw), wp, sd);
                uad_launch_colsum(vae ? dmu : dz, n, zd, Gr(m, m->mub), m->colscratch, sd);
                if (vae) {
                    uad_launch_conv_w(d_in, m->t, no_xform(), dls, no_xform(), Gr(m, m->sgw), wp, sd);
                    uad_launch_colsum(dls, n, zd, Gr(m, m->sgb), m->colscratch, sd);
                }
                uad_launch_conv_w(d_b, EL.c, bn_xform(m, EL.gamma, EL.beta, kLrelu), dflat, no_xform(), Gr(m, m->bw), wp, sd);
                uad_launch_colsum(dflat, n * ir * ir, m->cmid, Gr(m, m->bb), m->colscratch, sd);
                uad_launch_bn_grad_finalize(cp, uad_bottleneck_colpart_rows(ba, n), m->cenc, P(m, EL.gamma), rstd, Gr(m, EL.gamma), Gr(m, EL.beta), Gr(m, EL.b), sd);
            }
            float* tsw = m->G0; m->G0 = m->G1; m->G1 = tsw;
            if (join_now) join_side(m, st);
            return UAD_OK;
        }
    }
    edge(m, st, sd);
    // SIDE: conv2d_1 weight gradient (its bias gradient came from the decoder's BN finalize)
    if (pg) { PROF_ON("bott.wgrad", sd); uad_launch_conv_w(d_r, m->dvec, no_xform(), dcb, no_xform(), Gr(m, m->rw), wp, sd); }
    {
        PROF("bott.bwd");
        uad_launch_conv_d(d_r, dcb, no_xform(), P(m, m->rw), dd, epi_bias(nullptr, vae ? m->mask_dec_eff : nullptr), st, nullptr, m->ws);
        edge(m, st, sd);
        if (pg) { PROF_ON("bott.wgrad", sd);
          uad_launch_conv_w(d_dec, m->z, no_xform(), dd, no_xform(), Gr(m, m->dw), wp, sd);
          uad_launch_colsum(dd, n, m->flat, Gr(m, m->db), m->colscratch, sd); }
        uad_launch_conv_d(d_dec, dd, no_xform(), P(m, m->dw), dz, epi_bias(nullptr, vae ? nullptr : io.mask_mu), st, nullptr, m->ws);
        if (vae) {
            uad_launch_reparam_bwd(n, nu, zd, dz, m->mu, m->sigma, io.eps, io.mask_mu, io.mask_sigma,
                                   cevae ? io.mask_mu_ce : nullptr, m->restore ? m->restore_scale : 1.0f / (float)nu, dmu, dls, st);
            edge(m, st, sd);
            if (pg) { PROF_ON("bott.wgrad", sd);
              uad_launch_conv_w(d_in, m->t, no_xform(), dmu, no_xform(), Gr(m, m->muw), wp, sd);
              uad_launch_colsum(dmu, n, zd, Gr(m, m->mub), m->colscratch, sd);
              uad_launch_conv_w(d_in, m->t, no_xform(), dls, no_xform(), Gr(m, m->sgw), wp, sd);
              uad_launch_colsum(dls, n, zd, Gr(m, m->sgb), m->colscratch, sd); }
            uad_launch_conv_d(d_in, dmu, no_xform(), P(m, m->muw), m->g_small[5], epi_bias(nullptr), st, nullptr, m->ws);
            uad_launch_conv_d(d_in, dls, no_xform(), P(m, m->sgw), dflat, epi_bias(nullptr, nullptr, m->g_small[5]), st, nullptr, m->ws);
        } else {
            edge(m, st, sd);
            if (pg) { PROF_ON("bott.wgrad", sd);
              uad_launch_conv_w(d_in, m->t, no_xform(), dz, no_xform(), Gr(m, m->muw), wp, sd);
              uad_launch_colsum(dz, n, zd, Gr(m, m->mub), m->colscratch, sd); }
            uad_launch_conv_d(d_in, dz, no_xform(), P(m, m->muw), dflat, epi_bias(nullptr), st, nullptr, m->ws);
        }
        edge(m, st, sd);
        if (pg) { PROF_ON("bott.wgrad", sd);
          uad_launch_conv_w(d_b, EL.c, bn_xform(m, EL.gamma, EL.beta, kLrelu), dflat, no_xform(), Gr(m, m->bw), wp, sd);
          uad_launch_colsum(dflat, n * ir * ir, m->cmid, Gr(m, m->bb), m->colscratch, sd); }
        UadEpilogue e = epi_bwd(m, EL.c, EL.gamma, EL.beta, kLrelu); e.colpart = cp;
        uad_launch_conv_d(d_b, dflat, no_xform(), P(m, m->bw), m->G1, e, st, nullptr, m->ws);
    }
    edge(m, st, sd);
    if (pg) { PROF_ON("bn.gradfin", sd);
      uad_launch_bn_grad_finalize(cp, uad_conv_d_tiles(d_b, false, m->ws.floats), m->cenc, P(m, EL.gamma), rstd, Gr(m, EL.gamma),
                                  Gr(m, EL.beta), Gr(m, EL.b), sd); }
    float* tsw = m->G0; m->G0 = m->G1; m->G1 = tsw;   // G0 = d loss / d c of the last encoder conv
    join_side(m, st);  // join: bottleneck gradients complete (SIDE no longer reads dcb, now G1)
    return UAD_OK;
}

// spatial GMVAE: the "bottleneck" segment is the latent heads.  One kernel per map location recomputes the head forward,
// back-propagates the three prior terms, adds the decoder's d loss / d h and applies the last encoder block's activation
// backward; the head weight gradients are outer-product sums over the locations, reduced on the side stream.
static int backward_gm_heads(uad_model* m, hipStream_t st) {
    const int n = m->last_n;
    const int ir = m->cfg.inter_res, L = n * ir * ir;
    const float rstd = 1.0f / sqrtf(1.0f + kBnEps);
    const bool pg = !m->data_only;
    hipStream_t sd = m->side;
    const ConvLayer& EL = m->enc.back();
    float* cp = m->cp_slot[15];
    UadGmArgs ga = gm_args(m, m->last_io.eps_w, m->last_io.eps_z, m->restore ? m->restore_scale : 1.0f / (float)m->last_nuser);
    ga.h_out = nullptr;
    ga.dh_dec = m->G0; ga.g_out = m->G1; ga.colpart = cp;
    ga.dvec_heads = m->gm_dheads; ga.dvec_a7 = m->gm_da7; ga.dvec_M = m->gm_dM; ga.dvec_Lq = m->gm_dLq;
    ga.ws_out = m->gm_ws; ga.mid_out = m->gm_mid;
    { PROF("gm.heads.bwd"); uad_launch_gm_heads_bwd(ga, L, st); }
    edge(m, st, sd);
    if (pg) {
        PROF_ON("gm.heads.wgrad", sd);
        uad_launch_bn_grad_finalize(cp, L, m->cenc, P(m, EL.gamma), rstd, Gr(m, EL.gamma), Gr(m, EL.beta), Gr(m, EL.b), sd);
        const int W = ga.W, Z = ga.Z, Q = ga.Z * ga.C, O = 2 * W + 2 * Z, CE = m->cenc;
        UadGmWgradArgs wa;
        memset(&wa, 0, sizeof wa);
        const long long base = m->gm_off[0];
        auto job = [&](int k, const float* A, int lda, const float* B, int ldb, int b) {
            wa.job[k] = UadGmWgradArgs::Job{A, lda, B, ldb, b, (int)(m->gm_off[k] - base)};
        };
        const float* dh = m->gm_dheads;
        job(0, m->gm_h, CE, dh, O, W);              job(1, nullptr, 0, dh, O, W);
        job(2, m->gm_h, CE, dh + W, O, W);          job(3, nullptr, 0, dh + W, O, W);
        job(4, m->gm_h, CE, dh + 2 * W, O, Z);      job(5, nullptr, 0, dh + 2 * W, O, Z);
        job(6, m->gm_h, CE, dh + 2 * W + Z, O, Z);  job(7, nullptr, 0, dh + 2 * W + Z, O, Z);
        job(8, m->gm_ws, W, m->gm_da7, 64, 64);     job(9, nullptr, 0, m->gm_da7, 64, 64);
        job(10, m->gm_mid, 64, m->gm_dM, Q, Q);     job(11, nullptr, 0, m->gm_dM, Q, Q);
        job(12, m->gm_mid, 64, m->gm_dLq, Q, Q);    job(13, nullptr, 0, m->gm_dLq, Q, Q);
        job(14, nullptr, 0, m->gm_dLq, Q, Q);
        wa.njobs = 15; wa.total = (int)m->gm_total; wa.L = L; wa.partial = m->gm_partial;
        uad_launch_gm_heads_wgrad(wa, Gr(m, base), sd);
    }
    float* tsw = m->G0; m->G0 = m->G1; m->G1 = tsw;   // G0 = d loss / d c of the last encoder conv
    join_side(m, st);
    return UAD_OK;
}

// spatial AE: the "bottleneck" segment is the dropout mask + the last encoder block's activation backward
static int backward_spatial_z(uad_model* m, hipStream_t st) {
    const int n = m->last_n, ir = m->cfg.inter_res, rows = n * ir * ir;
    const float rstd = 1.0f / sqrtf(1.0f + kBnEps);
    const ConvLayer& EL = m->enc.back();
    hipStream_t sd = m->side;
    float* cp = m->cp_slot[15];
    { PROF("spatial.z.bwd"); uad_launch_spatial_z_bwd(m->G0, EL.c, P(m, EL.gamma), P(m, EL.beta), rstd, kLrelu, m->last_io.mask_mu, rows, m->cenc, m->G1, cp, st); }
    edge(m, st, sd);
    if (!m->data_only) { PROF_ON("bn.gradfin", sd); uad_launch_bn_grad_finalize(cp, uad_spatial_z_bwd_blocks(rows), m->cenc, P(m, EL.gamma), rstd, Gr(m, EL.gamma), Gr(m, EL.beta), Gr(m, EL.b), sd); }
    float* tsw = m->G0; m->G0 = m->G1; m->G1 = tsw;   // G0 = d loss / d c of the last encoder conv
    join_side(m, st);
    return UAD_OK;
}

// part: 0 = the whole segment; 1 = ENCODER_HI (blocks >= 2, joined: their variables are complete); 2 = ENCODER_LO (the rest)
static int backward_encoder(uad_model* m, hipStream_t st, int part, bool defer = false) {
    const int n = m->last_n;
    const float rstd = 1.0f / sqrtf(1.0f + kBnEps);
    hipStream_t sd = m->side;
    const bool bf = bf_mode(m);
    const bool pg = !m->data_only;
    float* g = m->G0;
    float* gn = m->G1;
    static const char* kEncW[] = {"enc0.wgrad", "enc1.wgrad", "enc2.wgrad", "enc3.wgrad", "enc4.wgrad", "enc5.wgrad", "enc6.wgrad", "enc7.wgrad"};
    static const char* kEncD[] = {"enc0.dgrad", "enc1.dgrad", "enc2.dgrad", "enc3.dgrad", "enc4.dgrad", "enc5.dgrad", "enc6.dgrad", "enc7.dgrad"};
    const int hi_from = (int)m->enc.size() - 1, hi_to = m->enc.size() >= 3 ? 2 : hi_from + 1;      // ENCODER_HI runs blocks hi_from .. hi_to
    const int i_first = part == 2 ? hi_to - 1 : hi_from, i_last = part == 1 ? hi_to : 1;
    for (int i = i_first; i >= i_last; --i) {
        UadConvDesc d = m->enc[i].d; d.N = n;
        const ConvLayer& PL = m->enc[i - 1];
        float* cp = m->cp_slot[8 + (i & 7)];
        if (pg) { PROF(kEncW[i & 7]); uad_launch_conv_w(d, PL.c, bn_xform(m, PL.gamma, PL.beta, kLrelu), g, no_xform(), Gr(m, m->enc[i].w), m->wp_slot[8 + (i & 7)], st, bf, nullptr, nullptr, false, true, planes_of(m)); }
        // The step's tail is the side stream's chain behind the LAST block's edge (event latency, slab reduction -- 79 MB for enc1 --, BN finalize, join).  For that block
        // the slab reduction is released by an edge of its own right behind the filter gradient, so that it runs beside the data gradient and the first layer's
        // filter gradient instead of after them; the block gives up the any-order overlap of its two kernels for it.  UAD_NO_TAIL_SPLIT: one edge behind both.
        static const bool tail_split_on = getenv("UAD_NO_TAIL_SPLIT") == nullptr;
        const bool tail_split = tail_split_on && pg && part != 1 && i == i_last && !m->prof_on;
        if (tail_split) { edge(m, st, sd); PROF_ON("bn.gradfin", sd); uad_launch_conv_w_reduce(d, Gr(m, m->enc[i].w), m->wp_slot[8 + (i & 7)], sd, bf); }
        { static const bool anyo = getenv("UAD_NO_ANYORDER") == nullptr; if (anyo && pg) uad_conv_any_order_next(true); }
        { PROF(kEncD[i & 7]); UadEpilogue e = epi_bwd(m, PL.c, PL.gamma, PL.beta, kLrelu); e.colpart = cp;
          uad_launch_conv_d(d, g, no_xform(), P(m, m->enc[i].w), gn, e, st, PKD(m, m->enc[i].w), m->ws, PK16D(m, m->enc[i].w), PLANE(m->enc[i]), false, planes_of(m));
          uad_conv_any_order_next(false); }
        if (pg) edge(m, st, sd);
        if (pg) { PROF_ON("bn.gradfin", sd); if (!tail_split) uad_launch_conv_w_reduce(d, Gr(m, m->enc[i].w), m->wp_slot[8 + (i & 7)], sd, bf);
                  uad_launch_bn_grad_finalize(cp, uad_conv_d_tiles(d, true, m->ws.floats, sk_counters(m), planes_of(m)), d.CB, P(m, PL.gamma), rstd, Gr(m, PL.gamma),
                                    Gr(m, PL.beta), Gr(m, PL.b), sd, m->bnfin_scratch); }
        float* tsw = g; g = gn; gn = tsw;
    }
    if (part == 1) {
        m->G0 = g; m->G1 = gn;
        if (!defer) join_side(m, st);   // join: the deep blocks' gradients are complete (deferred: they are in the side stream's order)
        return UAD_OK;
    }
    UadConvDesc d0 = m->enc[0].d; d0.N = n;
    if (pg) { PROF("enc0.wgrad"); uad_launch_conv_first_wgrad(d0, m->x_eff, g, Gr(m, m->enc[0].w), m->wp_slot[8], st); }
    if (m->restore && (m->restore_x || m->restore_grads)) {
        // d (loss + TV restore) / d x, and the in-place restoration update (trainers/GMVAE_spatial.py:186-190)
        PROF("enc0.dgrad");
        uad_launch_conv_first_dgrad_restore(d0, g, P(m, m->enc[0].w), m->gm_dxhat, m->restore_grads, m->restore_x,
                                            m->restore_lr, st);
    }
    if (m->cfg.arch == UAD_ARCH_CEVAE && m->last_io.anomaly) {
        // d loss_vae / d x of the VAE-branch samples (the leading last_nuser rows) -> anomaly map (trainers/ceVAE.py:51)
        PROF("enc0.dgrad");
        UadConvDesc dv = d0; dv.N = m->last_nuser;
        uad_launch_conv_first_dgrad(dv, g, P(m, m->enc[0].w), m->x_eff, m->xhat_own, 1.0f / (float)m->last_nuser,
                                    m->last_io.anomaly, nullptr, st);
    }
    m->G0 = g; m->G1 = gn;
    join_side(m, st);  // join: all gradients complete
    return UAD_OK;
}

// defer: a segment whose gradient writes all sit on the side stream (DECODER, the fused BOTTLENECK, ENCODER_HI) ends WITHOUT the side -> caller join
static int backward_impl(uad_model_t* m, int segment, void* stream, bool defer) {
    if (!m) return fail(UAD_ERR_INVALID, "null model");
    if (!m->have_fwd) return fail(UAD_ERR_INVALID, "uad_backward without a preceding uad_forward(want_backward=1)");
    if (segment < UAD_SEG_ALL || segment > UAD_SEG_ENCODER_LO) return fail(UAD_ERR_INVALID, "bad segment %d", segment);
    hipStream_t st = (hipStream_t)stream;
    int rc = UAD_OK;
    m->joined = false;
    if (segment == UAD_SEG_ALL || segment == UAD_SEG_DECODER) rc = backward_decoder(m, st, segment == UAD_SEG_DECODER && !defer);
    if (rc == UAD_OK && (segment == UAD_SEG_ALL || segment == UAD_SEG_BOTTLENECK))
        rc = m->cfg.arch == UAD_ARCH_GMVAE_SPATIAL ? backward_gm_heads(m, st)
             : m->cfg.arch == UAD_ARCH_AE_SPATIAL ? backward_spatial_z(m, st) : backward_bottleneck(m, st, segment == UAD_SEG_BOTTLENECK && !defer);
    if (rc == UAD_OK && segment == UAD_SEG_ENCODER_HI) rc = backward_encoder(m, st, 1, defer);
    if (rc == UAD_OK && (segment == UAD_SEG_ALL || segment == UAD_SEG_ENCODER || segment == UAD_SEG_ENCODER_LO)) {
        rc = backward_encoder(m, st, segment == UAD_SEG_ENCODER_LO ? 2 : 0);
        m->have_fwd = false;
    }
    HIP_TRY(hipGetLastError());
    return rc;
}

int uad_backward(uad_model_t* m, int segment, void* stream) { return backward_impl(m, segment, stream, false); }

int uad_backward_deferred(uad_model_t* m, int segment, void* stream, void** ready_stream) {
    const int rc = backward_impl(m, segment, stream, true);
    if (rc == UAD_OK && ready_stream) *ready_stream = m->joined ? stream : (void*)m->side;
    return rc;
}

// ------------------------------------------------------------------------------------------------ library-issued RCCL all-reduce
// RCCL is bound at run time: librccl.so.1 as the process already has it (the copy PyTorch-ROCm loads), else from the loader path.  libuad_hip.so
// itself has no link-time dependency on it -- single-GPU users and the CPU-side symbol tests never touch it.
namespace {
struct RcclApi {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) getUniqueId = nullptr;
    decltype(&ncclCommInitRank) commInitRank = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclAllReduce) allReduce = nullptr;
    decltype(&ncclGetErrorString) errString = nullptr;
    bool ok = false;
};
RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.ok ? &api : nullptr;
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    if (const char* over = getenv("UAD_RCCL_LIB")) api.h = dlopen(over, RTLD_NOW | RTLD_LOCAL);      // explicit override wins (tests: a stub whose "all-reduce" doubles)
    for (const char* nm : names) {
        if (api.h) break;
        api.h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);       // the copy the process already uses (torch's), if any
    }
    for (const char* nm : names) {
        if (api.h) break;
        api.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!api.h) return nullptr;
    api.getUniqueId = reinterpret_cast<decltype(api.getUniqueId)>(dlsym(api.h, "ncclGetUniqueId"));
    api.commInitRank = reinterpret_cast<decltype(api.commInitRank)>(dlsym(api.h, "ncclCommInitRank"));
    api.commDestroy = reinterpret_cast<decltype(api.commDestroy)>(dlsym(api.h, "ncclCommDestroy"));
    api.allReduce = reinterpret_cast<decltype(api.allReduce)>(dlsym(api.h, "ncclAllReduce"));
    api.errString = reinterpret_cast<decltype(api.errString)>(dlsym(api.h, "ncclGetErrorString"));
    api.ok = api.getUniqueId && api.commInitRank && api.commDestroy && api.allReduce && api.errString;
    return api.ok ? &api : nullptr;
}
#define RCCL_API(A)                                                                                                         \
    RcclApi* A = rccl_api();                                                                                                \
    if (!A) return fail(UAD_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded or lacks the nccl* entry points (set UAD_RCCL_LIB to its path)")
#define RCCL_TRY(A, expr)                                                                              \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess) return fail(UAD_ERR_HIP, "%s failed: %s", #expr, (A)->errString(r_));   \
    } while (0)
}  // namespace

int uad_rccl_unique_id(void* id_out, int cap) {
    if (!id_out || cap < (int)sizeof(ncclUniqueId)) return fail(UAD_ERR_INVALID, "uad_rccl_unique_id: need a buffer of %d bytes", (int)sizeof(ncclUniqueId));
    RCCL_API(api);
    ncclUniqueId id;
    RCCL_TRY(api, api->getUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return UAD_OK;
}
int uad_rccl_comm_create(const void* id_bytes, int world, int rank, void** comm_out) {
    if (!id_bytes || !comm_out || world < 1 || rank < 0 || rank >= world) return fail(UAD_ERR_INVALID, "uad_rccl_comm_create: bad arguments");
    RCCL_API(api);
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    ncclComm_t c = nullptr;
    // (RCCL prints a start-up banner -- version / host / library path -- to STDOUT from rank 0's first communicator.  A caller whose stdout is a protocol
    // points its file descriptor 1 elsewhere itself, as bench.py's claim_stdout() does: the library does not touch process-wide descriptors -- ADVICE r5.)
    const ncclResult_t r = api->commInitRank(&c, world, id, rank);        // collective over the ranks: every rank calls it with rank 0's id, on its own device
    if (r != ncclSuccess) return fail(UAD_ERR_HIP, "ncclCommInitRank failed: %s", api->errString(r));
    *comm_out = (void*)c;
    return UAD_OK;
}
int uad_rccl_comm_destroy(void* comm) {
    if (!comm) return UAD_OK;
    RCCL_API(api);
    RCCL_TRY(api, api->commDestroy((ncclComm_t)comm));
    return UAD_OK;
}
int uad_rccl_allreduce(void* comm, float* buf, long long count, void* stream) {
    if (!comm || !buf || count <= 0) return fail(UAD_ERR_INVALID, "uad_rccl_allreduce: bad arguments");
    RCCL_API(api);
    RCCL_TRY(api, api->allReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
    return UAD_OK;
}

int uad_allreduce_attach(uad_model_t* m, void* comm, int world, int nbuckets, const int* after_segment, const long long* offset, const long long* count) {
    if (!m) return fail(UAD_ERR_INVALID, "null model");
    if (!comm) { m->ar_comm = nullptr; m->ar_world = 1; m->ar_nb = 0; return UAD_OK; }      // detach
    if (world < 1 || nbuckets < 1 || nbuckets > 4 || !after_segment || !offset || !count) return fail(UAD_ERR_INVALID, "uad_allreduce_attach: 1..4 buckets");
    RCCL_API(api);
    (void)api;
    for (int i = 0; i < nbuckets; ++i) {
        if (after_segment[i] < UAD_SEG_DECODER || after_segment[i] > UAD_SEG_ENCODER_LO || offset[i] < 0 || count[i] < 0 || offset[i] + count[i] > m->nparams)
            return fail(UAD_ERR_INVALID, "uad_allreduce_attach: bucket %d (after segment %d, [%lld, +%lld)) outside the gradient buffer", i, after_segment[i], offset[i], count[i]);
        m->ar_after[i] = after_segment[i]; m->ar_off[i] = offset[i]; m->ar_cnt[i] = count[i];
    }
    m->ar_nb = nbuckets; m->ar_comm = comm; m->ar_world = world;
    // Default: the collectives of the deferred segments are enqueued on the handle's SIDE stream, right behind the slab reductions that complete
    // their bucket -- no event at all; the last bucket goes onto the caller's stream in front of the optimizer step.  Measured on one rank under RCCL
    // (profiles/r05_d_rccl_one_rank.log): the step costs what the plain step costs (0.822 vs 0.822 ms; torch.distributed's path 0.858).
    // UAD_AR_STREAM=own gives them a stream of their own (one event per bucket; the side stream's later reductions do not queue behind a ring) --
    // not the default because a third stream per handle makes the step depend on how HIP maps streams to hardware queues: bench.py's own N > 1
    // path read 1.54 ms per step with it (0.87 on the side stream) where tools/host_time_dp.py, creating its streams in another order, read 0.84.
    static const bool on_side = !(getenv("UAD_AR_STREAM") && !strcmp(getenv("UAD_AR_STREAM"), "own"));
    if (on_side) { m->ar_stream = m->side; m->ar_own_stream = false; }
    else if (!m->ar_own_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&m->ar_stream, hipStreamNonBlocking));
        m->ar_own_stream = true;
    }
    for (int i = 0; i < 4; ++i)
        if (!m->ar_ev_in[i] && hipEventCreateWithFlags(&m->ar_ev_in[i], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess)
            HIP_TRY(hipEventCreateWithFlags(&m->ar_ev_in[i], hipEventDisableTiming));
    if (!m->ar_ev_out && hipEventCreateWithFlags(&m->ar_ev_out, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess)
        HIP_TRY(hipEventCreateWithFlags(&m->ar_ev_out, hipEventDisableTiming));
    return UAD_OK;
}

int uad_backward_allreduce(uad_model_t* m, int segment, void* stream) {
    if (!m) return fail(UAD_ERR_INVALID, "null model");
    if (!m->ar_comm) return fail(UAD_ERR_INVALID, "uad_backward_allreduce without uad_allreduce_attach");
    if (segment == UAD_SEG_ALL || segment == UAD_SEG_ENCODER) return fail(UAD_ERR_INVALID, "uad_backward_allreduce runs ONE of DECODER, BOTTLENECK, ENCODER_HI, ENCODER_LO per call, in that order");
    hipStream_t st = (hipStream_t)stream;
    const int rc = backward_impl(m, segment, stream, true);
    if (rc != UAD_OK) return rc;
    RCCL_API(api);
    static const bool skip = getenv("UAD_AR_SKIP") != nullptr;       // measurement: everything but the ncclAllReduce call itself
    for (int i = 0; i < m->ar_nb; ++i) {
        if (m->ar_after[i] != segment || m->ar_cnt[i] == 0) continue;
        float* g = m->grads + m->ar_off[i];
        if (m->joined) {
            // The segment ended with the side stream joined into the caller's (the last one always does): its gradients are complete in the
            // CALLER's stream order and the next thing on that stream is the optimizer step, which needs the reduced values anyway -- the
            // collective goes straight onto the caller's stream, no event and no round trip through another stream.
            if (m->ar_pending && m->ar_stream != m->side) {
                // earlier buckets run on the collective stream: the caller's stream waits for them once (the side-stream variant needs nothing: the
                // segment's own join already ordered the caller's stream behind everything the side stream had been given, collectives included)
                (void)hipEventRecord(m->ar_ev_out, m->ar_stream);
                (void)hipStreamWaitEvent(st, m->ar_ev_out, 0);
            }
            m->ar_pending = false;
            if (!skip) RCCL_TRY(api, api->allReduce(g, g, (size_t)m->ar_cnt[i], ncclFloat, ncclSum, (ncclComm_t)m->ar_comm, st));
            continue;
        }
        // gradients complete in the SIDE stream's order (deferred segment): one event orders the collective stream behind the slab reductions of
        // this bucket; the caller's stream is not touched and runs on with the next segment
        if (m->ar_stream != m->side) {
            (void)hipEventRecord(m->ar_ev_in[i], m->side);
            (void)hipStreamWaitEvent(m->ar_stream, m->ar_ev_in[i], 0);
        }
        if (!skip) RCCL_TRY(api, api->allReduce(g, g, (size_t)m->ar_cnt[i], ncclFloat, ncclSum, (ncclComm_t)m->ar_comm, m->ar_stream));
        m->ar_pending = true;
    }
    if (segment == UAD_SEG_ENCODER_LO && m->ar_pending) {
        // (a plan without a bucket behind the last segment: the caller's stream still has to see the earlier ones before the optimizer step)
        if (m->ar_stream != m->side) {
            (void)hipEventRecord(m->ar_ev_out, m->ar_stream);
            (void)hipStreamWaitEvent(st, m->ar_ev_out, 0);
        } else if (!m->joined) join_side(m, st);
        m->ar_pending = false;
    }
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}

int uad_adam_step(uad_model_t* m, float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
    if (!m) return fail(UAD_ERR_INVALID, "null model");
    m->step += 1;
    const double t = (double)m->step;
    const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
    hipStream_t st = (hipStream_t)stream;
    invalidate_pack(m);
    m->opt_epochs.push_back(m->bott_epoch);
    { PROF("adam"); uad_launch_adam(m->params, m->grads, m->adam_m, m->adam_v, (size_t)m->nparams, lr_t, beta1, beta2, eps, grad_scale, st, m->bott_fault); }
    repack_on_side(m, st);
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}

int uad_optimizer_step(uad_model_t* m, int kind, float lr, float momentum, float decay, float eps, float grad_scale, void* stream) {
    if (!m) return fail(UAD_ERR_INVALID, "null model");
    if (kind < UAD_OPT_SGD || kind > UAD_OPT_RMS) return fail(UAD_ERR_INVALID, "optimizer kind %d: UAD_OPT_SGD | UAD_OPT_MOMENTUM | UAD_OPT_RMS (Adam is uad_adam_step)", kind);
    m->step += 1;
    invalidate_pack(m);
    hipStream_t st = (hipStream_t)stream;
    m->opt_epochs.push_back(m->bott_epoch);
    { PROF("optim"); uad_launch_optim(kind, m->params, m->grads, m->adam_m, m->adam_v, (size_t)m->nparams, lr, momentum, decay, eps, grad_scale, st, m->bott_fault); }
    repack_on_side(m, st);
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}

int uad_train_step(uad_model_t* m, const uad_io_t* io, int n, float lr, float beta1, float beta2, float eps, void* stream) {
    int rc = uad_forward(m, io, n, 1, stream);
    if (rc == UAD_OK) rc = uad_backward(m, UAD_SEG_ALL, stream);
    if (rc == UAD_OK) rc = uad_adam_step(m, lr, beta1, beta2, eps, 1.0f, stream);
    return rc;
}

int uad_restore_step(uad_model_t* m, float* x_restored, const float* eps_w, const float* eps_z, int n, float tv_lambda,
                     float restore_lr, float* grads_out, void* stream) {
    if (!m || !x_restored) return fail(UAD_ERR_INVALID, "null argument");
    const bool vae = m->cfg.arch == UAD_ARCH_VAE;
    if (m->cfg.arch != UAD_ARCH_GMVAE_SPATIAL && !vae) return fail(UAD_ERR_INVALID, "uad_restore_step needs a spatial GMVAE or a VAE handle");
    if (n <= 0 || n > m->cfg.max_batch) return fail(UAD_ERR_INVALID, "batch %d outside (0, max_batch=%d]", n, m->cfg.max_batch);
    uad_io_t io;
    memset(&io, 0, sizeof io);
    io.x = x_restored; io.eps_w = eps_w; io.eps_z = eps_z;
    if (vae) io.eps = eps_z;                     // trainers/VAE_You.py:52-53: grads = d (rec_n + kl_n + tv * TV_n) / d x, per sample
    m->restore = true; m->restore_tv = tv_lambda; m->restore_lr = restore_lr;
    // both trainers: tf.gradients of an [n]-shaped objective (per-sample pixel_loss, or scalar loss + per-image restore broadcast to [n])
    // differentiates the SUM of its elements -> every sample's own loss terms carry weight 1
    m->restore_scale = 1.0f;
    m->restore_x = x_restored; m->restore_grads = grads_out;
    int rc = uad_forward(m, &io, n, 2, stream);
    if (rc == UAD_OK) rc = uad_backward(m, UAD_SEG_ALL, stream);
    m->restore = false; m->fb_on_load = false; m->restore_x = nullptr; m->restore_grads = nullptr;
    return rc;
}

int uad_set_math_mode(uad_model_t* m, int mode) {
    if (!m || (mode != UAD_MATH_F32 && mode != UAD_MATH_BF16X3 && mode != UAD_MATH_BF16X6)) return fail(UAD_ERR_INVALID, "bad math mode");
    if (mode == UAD_MATH_BF16X6 && !m->wpack3_f) {      // three bf16 planes of the 5x5 kernels (4 ushorts per parameter: uad_launch_pack_weights_bf16_3p)
        int rc = dev_alloc(m, &m->wpack3_f, (size_t)m->nparams * 2);      // (floats: 8 bytes per parameter; freed with the handle's other buffers)
        if (rc == UAD_OK) rc = dev_alloc(m, &m->wpack3_d, (size_t)m->nparams * 2);
        if (rc != UAD_OK) return rc;
    }
    m->math = mode;
    invalidate_pack(m);
    return UAD_OK;
}
int uad_get_math_mode(const uad_model_t* m) { return m ? m->math : -1; }

// tests: named intermediates of the last uad_forward (pre-BN conv outputs, the decoder's input, the gradient ping-pong buffers).
// Counts are for the sample count of the last forward (2n rows inside a ceVAE handle).
int uad_debug_buffer(uad_model_t* m, const char* name, float** ptr, long long* count) {
    if (!m || !name) return fail(UAD_ERR_INVALID, "null argument");
    const long long n = m->last_n;
    float* p = nullptr; long long c = 0;
    int idx = -1;
    if (sscanf(name, "enc_c%d", &idx) == 1 && idx >= 0 && idx < (int)m->enc.size()) {
        const UadConvDesc& d = m->enc[idx].d; p = m->enc[idx].c; c = n * d.HS * d.WS * d.CS;
    } else if (sscanf(name, "dec_c%d", &idx) == 1 && idx >= 0 && idx < (int)m->dec.size()) {
        const UadConvDesc& d = m->dec[idx].d; p = m->dec[idx].c; c = n * d.HB * d.WB * d.CB;
    } else if (!strcmp(name, "dec_in")) {
        p = (float*)m->dec_in0; c = n * m->cfg.inter_res * m->cfg.inter_res * m->cenc;
    } else if (!strcmp(name, "fin_bits")) {       // one word per output pixel; null unless the last forward used the compressed form
        p = m->last_fin_bits ? reinterpret_cast<float*>(m->fin_bits) : nullptr; c = m->last_fin_bits ? n * m->cfg.height * m->cfg.width : 0;
    } else if (!strcmp(name, "fin_dxhat")) {
        p = m->last_fin_bits ? m->fin_dxh : nullptr; c = m->last_fin_bits ? n * m->cfg.height * m->cfg.width : 0;
    } else if (!strcmp(name, "fused_final")) {
        p = nullptr; c = m->last_fused_final ? 1 : 0;
    } else if (!strcmp(name, "G0") || !strcmp(name, "G1")) {
        const UadConvDesc& d = m->dec.back().d; p = name[1] == '0' ? m->G0 : m->G1; c = n * d.HB * d.WB * d.CB;
    } else {
        return fail(UAD_ERR_INVALID, "no debug buffer named %s", name);
    }
    if (ptr) *ptr = p;
    if (count) *count = c;
    return UAD_OK;
}

int uad_profile_enable(uad_model_t* m, int on) {
    if (!m) return fail(UAD_ERR_INVALID, "null model");
    m->prof_on = on != 0;
    return UAD_OK;
}

int uad_profile_report(uad_model_t* m, char* buf, int cap) {
    if (!m || !buf || cap <= 0) return fail(UAD_ERR_INVALID, "bad arguments");
    HIP_TRY(hipDeviceSynchronize());
    struct Agg { const char* tag; int count; double ms; };
    std::vector<Agg> agg;
    for (auto& r : m->prof) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        size_t k = 0;
        for (; k < agg.size(); ++k) if (!strcmp(agg[k].tag, r.tag)) break;
        if (k == agg.size()) agg.push_back({r.tag, 0, 0.0});
        agg[k].count += 1; agg[k].ms += ms;
        m->ev_pool.push_back(r.a); m->ev_pool.push_back(r.b);
    }
    m->prof.clear();
    int off = 0;
    buf[0] = 0;
    for (auto& a : agg) {
        int w = snprintf(buf + off, cap - off, "%s %d %.6f\n", a.tag, a.count, a.ms);
        if (w < 0 || w >= cap - off) break;
        off += w;
    }
    return UAD_OK;
}

int uad_residual(const float* x, const float* xr, const float* mask, int n, int hw, int pos_only, float prior_thresh,
                 float* out, float* l1err, void* stream) {
    if (!x || !xr || !out || n <= 0 || hw <= 0) return fail(UAD_ERR_INVALID, "uad_residual: bad arguments");
    uad_launch_residual(x, xr, mask, n, hw, pos_only, prior_thresh, out, l1err, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}

int uad_rng_fill(const uad_rng_job_t* jobs, int njobs, int n, unsigned long long seed, unsigned long long step, long long sample0, void* stream) {
    if (!jobs || njobs <= 0 || njobs > 8 || n <= 0 || sample0 < 0) return fail(UAD_ERR_INVALID, "rng_fill: 1..8 jobs, n > 0, sample0 >= 0");
    UadRngJob js[8];
    for (int i = 0; i < njobs; ++i) {
        if (!jobs[i].out || jobs[i].per_sample <= 0 || (jobs[i].kind != UAD_RNG_NORMAL && jobs[i].kind != UAD_RNG_KEEP_MASK))
            return fail(UAD_ERR_INVALID, "rng_fill: job %d: null output, empty sample or unknown kind", i);
        if (jobs[i].kind == UAD_RNG_KEEP_MASK && !(jobs[i].rate >= 0.f && jobs[i].rate < 1.f)) return fail(UAD_ERR_INVALID, "rng_fill: dropout rate must be in [0, 1)");
        js[i] = UadRngJob{jobs[i].out, jobs[i].per_sample, jobs[i].kind, jobs[i].rate, jobs[i].stream};
    }
    uad_launch_rng_fill(js, njobs, n, seed, step, sample0, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}

int uad_clock_probe(unsigned long long* out2, unsigned long long ticks_100mhz, void* stream) {
    if (!out2 || ticks_100mhz == 0 || ticks_100mhz > 1000000000ull) return fail(UAD_ERR_INVALID, "clock_probe: null output or a duration outside (0, 10 s]");
    uad_launch_clock_probe(out2, ticks_100mhz, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}

int uad_gather_slices(const float* src, const int* idx, int n, long long slice_elems, float* out, void* stream) {
    if (!src || !idx || !out || n <= 0 || slice_elems <= 0 || slice_elems % 4) return fail(UAD_ERR_INVALID, "gather_slices: bad arguments (slice elements must be a multiple of 4)");
    if (n > 65535) return fail(UAD_ERR_UNSUPPORTED, "gather_slices: at most 65535 slices per call");
    uad_launch_gather_slices(src, idx, n, slice_elems, out, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}
int uad_gather_mask(const unsigned char* labels, const int* idx, int n, long long slice_px, const unsigned char* lut256, float* out, void* stream) {
    if (!labels || !idx || !out || n <= 0 || slice_px <= 0) return fail(UAD_ERR_INVALID, "gather_mask: bad arguments");
    if (n > 65535) return fail(UAD_ERR_UNSUPPORTED, "gather_mask: at most 65535 slices per call");
    uad_launch_gather_mask(labels, idx, n, slice_px, lut256, out, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}

// ------------------------------------------------------------------------------------------------ op-level entry points
static UadConvDesc to_desc(const uad_conv_desc_t* d) { return UadConvDesc{d->N, d->HB, d->WB, d->CB, d->HS, d->WS, d->CS, d->KS, d->S, d->P}; }
static UadXform to_xf(const uad_xform_t* x) {
    UadXform r; r.scale = x ? x->scale : nullptr; r.shift = x ? x->shift : nullptr; r.alpha = x ? x->alpha : 1.f; r.mult = 1.f;
    return r;
}
static int check_gemm_desc(const uad_conv_desc_t* d, bool f_type) {
    if (!d) return fail(UAD_ERR_INVALID, "null desc");
    const int ca = f_type ? d->CB : d->CS, nn = f_type ? d->CS : d->CB;
    if (ca % 8 || nn % 4) return fail(UAD_ERR_UNSUPPORTED, "contraction channels must be a multiple of 8 and output channels of 4");
    return UAD_OK;
}

// op-level helpers: the spatial kernels need the packed weight copy; build it on the fly (synchronous, tests only)
static int ws_for_op(const UadConvDesc& d, bool f_type, bool have_pack, UadGemmWs* ws) {
    ws->ptr = nullptr; ws->floats = uad_conv_ws_floats(d, f_type, have_pack);
    ws->counters = nullptr; ws->ncounters = 0;
    if (ws->floats) {
        // slabs + (behind them) the arrival counters of the in-kernel reduction, zeroed: the op-level entry points exercise the same path
        // as the model handle
        const int nc = 16384;
        HIP_TRY(hipMalloc((void**)&ws->ptr, (ws->floats + nc) * sizeof(float)));
        HIP_TRY(hipMemset(ws->ptr + ws->floats, 0, nc * sizeof(float)));
        ws->counters = reinterpret_cast<unsigned*>(ws->ptr + ws->floats); ws->ncounters = nc;
    }
    return UAD_OK;
}
// UAD_MATH of the op-level entry points: unset / "f32" = exact fp32, "bf16x3", or "bf16x6" = three-plane products on the k3 tap-list kernel where
// it takes the shape (everything else as bf16x3)
static bool op_bf16x6() { const char* e = getenv("UAD_MATH"); return e && !strcmp(e, "bf16x6"); }
static bool op_bf16x3() { const char* e = getenv("UAD_MATH"); return e && (!strcmp(e, "bf16x3") || !strcmp(e, "bf16x6")); }
// plain: identity activation on load, bias (+ addend) epilogue without `mul` -- what the k3 tap-list kernel takes besides the shape (uad_convk16.inc:
// convk16_takes); any other launch of a k3 shape runs the generic kernels, which understand TWO planes only
static int op_planes(const UadConvDesc& d, bool f_type, bool plain) { return (op_bf16x6() && plain && uad_conv_k3_takes(d, f_type)) ? 3 : 2; }
static int pack_for_op(const UadConvDesc& d, const float* W, bool f_type, float** pf, float** pd, hipStream_t st, bool plain = false) {
    *pf = *pd = nullptr;
    if (!uad_conv_spatial_ok(d, f_type)) return UAD_OK;
    const size_t n = (size_t)d.KS * d.KS * d.CB * d.CS;
    HIP_TRY(hipMalloc((void**)pf, 2 * n * sizeof(float)));
    HIP_TRY(hipMalloc((void**)pd, 2 * n * sizeof(float)));
    long long off = 0; int cb = d.CB, cs = d.CS, taps = d.KS * d.KS;
    if (op_planes(d, f_type, plain) == 3) uad_launch_pack_weights_bf16_3p(W, (unsigned short*)*pf, (unsigned short*)*pd, &off, &cb, &cs, &taps, 1, st);
    else if (op_bf16x3()) uad_launch_pack_weights_bf16(W, (unsigned short*)*pf, (unsigned short*)*pd, &off, &cb, &cs, &taps, 1, st);
    else uad_launch_pack_weights(W, *pf, *pd, &off, &cb, &cs, &taps, 1, st);
    return UAD_OK;
}
// launch arguments for the op-level entry points in the selected math mode
#define OP_PACK_ARGS(pk, d, f, plain) (op_bf16x3() ? nullptr : (pk)), ws, (op_bf16x3() ? (const unsigned short*)(pk) : nullptr), \
                               (long long)(d).KS * (d).KS * (d).CB * (d).CS, op_bf16x3(), op_planes(to_desc(&(d)), f, plain)
static int finish_op(float* pf, float* pd, hipStream_t st, float* wsp = nullptr) {
    if (pf || pd || wsp) { HIP_TRY(hipStreamSynchronize(st)); (void)hipFree(pf); (void)hipFree(pd); (void)hipFree(wsp); }
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}

int uad_op_conv_f(const uad_conv_desc_t* d, const float* big_in, const uad_xform_t* xf, const float* W, const float* bias,
                  const float* mul, const float* add, float* small_out, void* stream) {
    if (int rc = check_gemm_desc(d, true)) return rc;
    float *pf = nullptr, *pd = nullptr;
    const bool plain = !(xf && xf->scale) && !mul;
    if (int rc = pack_for_op(to_desc(d), W, true, &pf, &pd, (hipStream_t)stream, plain)) return rc;
    UadGemmWs ws;
    if (int rc = ws_for_op(to_desc(d), true, pf != nullptr, &ws)) return rc;
    uad_launch_conv_f(to_desc(d), big_in, to_xf(xf), W, small_out, epi_bias(bias, mul, add), (hipStream_t)stream, OP_PACK_ARGS(pf, *d, true, plain));
    return finish_op(pf, pd, (hipStream_t)stream, ws.ptr);
}
int uad_op_conv_d(const uad_conv_desc_t* d, const float* small_in, const uad_xform_t* xf, const float* W, const float* bias,
                  const float* mul, const float* add, float* big_out, void* stream) {
    if (int rc = check_gemm_desc(d, false)) return rc;
    float *pf = nullptr, *pd = nullptr;
    const bool plain = !(xf && xf->scale) && !mul;
    if (int rc = pack_for_op(to_desc(d), W, false, &pf, &pd, (hipStream_t)stream, plain)) return rc;
    UadGemmWs ws;
    if (int rc = ws_for_op(to_desc(d), false, pd != nullptr, &ws)) return rc;
    uad_launch_conv_d(to_desc(d), small_in, to_xf(xf), W, big_out, epi_bias(bias, mul, add), (hipStream_t)stream, OP_PACK_ARGS(pd, *d, false, plain));
    return finish_op(pf, pd, (hipStream_t)stream, ws.ptr);
}

static int bwdact_common(bool f_type, const uad_conv_desc_t* dd, const float* in, const float* W, const float* cprev,
                         const uad_xform_t* act, float* out, float* s1, float* s2, void* stream) {
    if (int rc = check_gemm_desc(dd, f_type)) return rc;
    if (!act || !act->scale) return fail(UAD_ERR_INVALID, "bwdact needs the activation scale/shift");
    UadConvDesc d = to_desc(dd);
    const int C = f_type ? d.CS : d.CB;
    const bool can_pack = uad_conv_spatial_ok(d, f_type);
    UadGemmWs ws;
    if (int rc = ws_for_op(d, f_type, can_pack, &ws)) return rc;
    const int nc = (op_bf16x3() && ws.counters) ? ws.ncounters : 0;
    const int T = f_type ? uad_conv_f_tiles(d, can_pack, ws.floats, nc) : uad_conv_d_tiles(d, can_pack, ws.floats, nc);
    float *colpart = nullptr, *tmp = nullptr;
    HIP_TRY(hipMalloc((void**)&colpart, (size_t)T * 2 * C * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&tmp, (size_t)3 * C * sizeof(float)));
    UadEpilogue e;
    memset(&e, 0, sizeof e);
    e.kind = UAD_EPI_BWD_ACT; e.cprev = cprev; e.escale = act->scale; e.eshift = act->shift; e.ealpha = act->alpha;
    e.emult = 1.f; e.colpart = colpart;
    hipStream_t st = (hipStream_t)stream;
    float *pf = nullptr, *pd = nullptr;
    if (int rc = pack_for_op(d, W, f_type, &pf, &pd, st)) return rc;
    const long long plane = (long long)d.KS * d.KS * d.CB * d.CS;       // (the activation-backward epilogue never runs the three-plane kernel)
    if (f_type) uad_launch_conv_f(d, in, no_xform(), W, out, e, st, op_bf16x3() ? nullptr : pf, ws, op_bf16x3() ? (const unsigned short*)pf : nullptr, plane, op_bf16x3());
    else uad_launch_conv_d(d, in, no_xform(), W, out, e, st, op_bf16x3() ? nullptr : pd, ws, op_bf16x3() ? (const unsigned short*)pd : nullptr, plane, op_bf16x3());
    // rstd = 1, gamma unused for dbias=null: dbeta -> s1, dgamma -> s2
    uad_launch_bn_grad_finalize(colpart, T, C, act->scale, 1.0f, s2, s1, nullptr, st);
    HIP_TRY(hipStreamSynchronize(st));
    hipFree(colpart); hipFree(tmp); hipFree(pf); hipFree(pd); hipFree(ws.ptr);
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}
int uad_op_conv_f_bwdact(const uad_conv_desc_t* d, const float* big_in, const float* W, const float* cprev,
                         const uad_xform_t* act, float* small_out, float* s1, float* s2, void* stream) {
    return bwdact_common(true, d, big_in, W, cprev, act, small_out, s1, s2, stream);
}
int uad_op_conv_d_bwdact(const uad_conv_desc_t* d, const float* small_in, const float* W, const float* cprev,
                         const uad_xform_t* act, float* big_out, float* s1, float* s2, void* stream) {
    return bwdact_common(false, d, small_in, W, cprev, act, big_out, s1, s2, stream);
}

int uad_op_conv_w(const uad_conv_desc_t* dd, const float* big, const uad_xform_t* xfb, const float* small_,
                  const uad_xform_t* xfs, float* dW, void* stream) {
    if (!dd) return fail(UAD_ERR_INVALID, "null desc");
    if (dd->CB % 4 || dd->CS % 4) return fail(UAD_ERR_UNSUPPORTED, "channels must be multiples of 4");
    UadConvDesc d = to_desc(dd);
    float* partial = nullptr;
    HIP_TRY(hipMalloc((void**)&partial, uad_conv_w_partial_floats(d) * sizeof(float)));
    hipStream_t st = (hipStream_t)stream;
    uad_launch_conv_w(d, big, to_xf(xfb), small_, to_xf(xfs), dW, partial, st, op_bf16x3(), nullptr, nullptr, op_bf16x3());
    HIP_TRY(hipStreamSynchronize(st));
    hipFree(partial);
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}

int uad_op_conv_first_fwd(const uad_conv_desc_t* d, const float* x, const float* W, const float* bias, float* out, void* stream) {
    if (!d || d->CS % 8 || 256 % (d->CS / 8)) return fail(UAD_ERR_UNSUPPORTED, "first conv: Cout must be a multiple of 8 dividing 2048");
    uad_launch_conv_first_fwd(to_desc(d), x, W, bias, out, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}
int uad_op_conv_first_wgrad(const uad_conv_desc_t* dd, const float* x, const float* g, float* dW, void* stream) {
    if (!dd || !((dd->KS == 5 && (dd->CB == 1 || dd->CB == 3)) || (dd->KS == 3 && dd->CB == 1)) || 256 % dd->CS)
        return fail(UAD_ERR_UNSUPPORTED, "first wgrad: (KS=5, Cin in {1,3}) or (KS=3, Cin=1), Cout | 256");
    UadConvDesc d = to_desc(dd);
    float* partial = nullptr;
    HIP_TRY(hipMalloc((void**)&partial, uad_conv_first_wgrad_partial_floats(d) * sizeof(float)));
    hipStream_t st = (hipStream_t)stream;
    uad_launch_conv_first_wgrad(d, x, g, dW, partial, st);
    HIP_TRY(hipStreamSynchronize(st));
    hipFree(partial);
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}
int uad_op_adam(float* p, const float* g, float* mm, float* v, long long n, float lr_t, float beta1, float beta2, float eps,
                float gscale, void* stream) {
    uad_launch_adam(p, g, mm, v, (size_t)n, lr_t, beta1, beta2, eps, gscale, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}

}  // extern "C"
