// f-AnoGAN (unified graph) on gfx950: LayerNorm-HW kernels (forward, backward, and the second-order backward the WGAN-GP
// penalty needs), the small heads / losses, and the orchestration of the three optimisation phases behind the C-ABI
// (include/uad_hip.h, uad_gan_*).  The k5 s2 convolutions, their data and filter gradients, the dense layers and Adam are
// the kernels of uad_gemm.hip / uad_misc.hip.
//
// Graph restated (reference): models/fanogan.py:11-84, models/customlayers.py:16-38 (use_batchnorm=False branches),
// losses / optimisers trainers/fAnoGAN.py:50-77, reconstruct :220-239.
//
// Critic phase = four passes over the critic (per-sample normalisation makes every pass batchable):
//   A  forward of [x_ ; x ; x_hat]                                   (3n samples)
//   B  input gradient of sum(d_hat) on the x_hat third               (tf.gradients(d_hat, x_hat), n samples)
//   C  adjoint of pass B, bottom-up (forward convs of the adjoint, second-order LayerNorm term)
//   D  ordinary backward of all 3n samples with pass C's d penalty / d c injected on the x_hat third; each filter gradient
//      runs once over 4n "samples": the 3n (input, d c) pairs plus pass C's (adjoint, pass-B d c) pairs of the same layer.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/uad_hip.h"
#include "uad_kernels.h"

int uad_fail(int code, const char* fmt, ...);
#define fail uad_fail

namespace {

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(UAD_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr float kBnEps = 1e-3f;     // tf.layers.BatchNormalization default epsilon (encoder, frozen statistics)
constexpr float kLnEps = 1e-3f;     // keras LayerNormalization default epsilon
constexpr float kLrelu = 0.3f;      // keras LeakyReLU() default

#include "uad_gan_kernels.inc"

// ================================================================================================ handle
struct Tensor {
    std::string name;
    long long off;
    int rank;
    int shape[4];
    long long count() const { return (long long)shape[0] * shape[1] * shape[2] * shape[3]; }
};
struct Block {            // conv / convT followed by a norm + (Leaky)ReLU
    UadConvDesc d;        // geometry at batch 1
    long long w, b, gamma, beta;
    int H, W, C;          // output map of the block
};

}  // namespace

struct uad_gan {
    uad_gan_config_t cfg;
    int npool, cenc, cmid, flat;
    std::vector<Tensor> tensors;
    long long nparams;
    long long grp_off[3], grp_cnt[3], step[3];
    float *params, *grads, *adam_m, *adam_v;
    float *adam_m2, *adam_v2;          // AnoVAE-GAN: the Generator's slots inside optim_vae (its optim_gen slots are adam_m / adam_v)
    float *wpack_f, *wpack_d, *wpack16_f, *wpack16_d;
    float* ln_xch = nullptr; unsigned* ln_flags = nullptr; size_t ln_xcap = 0; unsigned ln_epoch = 0;     // exchange scratch of the clustered LayerNorm kernels (grown on first use)
    float *wpack3_f, *wpack3_d;        // ResNet graph: THREE bf16 planes of the k3 kernels (bf16x6 products of the exact passes), 4 * nparams ushorts each; null: fp32 kernels
    int math;
    bool packed_valid;
    // library-issued, bucketed gradient all-reduce of a phase's trained group (uad_gan_allreduce_attach, round 6): the group's tensors are cut into up to four
    // contiguous buckets; a bucket's ncclAllReduce is enqueued on ar_stream as soon as the LAST kernel that writes one of its tensors is enqueued
    // (gan_grad_final marks a tensor; the pending count of its bucket reaching zero issues it), i.e. while the backward of the earlier layers still runs
    void* ar_comm; int ar_world;
    hipStream_t ar_stream; hipEvent_t ar_ev_in[4], ar_ev_out;
    struct ArBucket { long long off, cnt; int pending; bool issued; } ar_b[4];
    int ar_nb; bool ar_active; hipStream_t ar_st;
    std::vector<std::pair<long long, int>> ar_tb;      // (tensor offset, bucket) of the trained group's tensors, and whether it was marked
    std::vector<char> ar_marked;
    bool pack_all; long long dirty_lo, dirty_hi;      // which parameters changed since the last pack: everything, or [dirty_lo, dirty_hi) (an optimizer step touches ONE group)
    UadGemmWs ws;
    std::vector<Block> E, G, D;
    long long e_cw, e_cb, e_dw, e_db;                   // Encoder/conv2d, Encoder/dense
    long long e_sw, e_sb;                               // AnoVAE-GAN: Encoder/dense_1 (log-sigma head; Encoder/dense is mu)
    float *v_mu_raw, *v_ls_raw, *v_mu, *v_ls, *v_sigma, *v_kl, *v_dmu, *v_dls, *v_dflat2;
    long long g_dw, g_db, g_cw, g_cb, g_ln0g, g_ln0b;   // Generator/dense, conv2d_1, first LayerNorm
    long long g_fw, g_fb;                               // dec_Conv2D_final
    long long d_hw, d_hb;                               // Discriminator/dense
    // encoder activations
    std::vector<float*> ec, ea;        // ea[i] = input of block i (ea[0] unused: the caller's x)
    float *et, *zr, *z;
    // generator activations
    float *gdv, *xg;
    std::vector<float*> gc, ga, gstat; // gc[0] = conv2d_1 output, gc[i+1] = ConvT i output
    // critic (4n layout where noted)
    float* din;                        // [4n] input images; tail = pass C's adjoint of the input gradient
    std::vector<float*> Dc, Dstat;     // [3n]
    std::vector<float*> Da;            // Da[i] = input of block i (i >= 1), Da[L] = features; [4n], tail = pass C adjoints
    std::vector<float*> Dg;            // [4n] d loss / d c_i ; tail = pass B's d c_i
    std::vector<float*> V, inj;        // [n] pass B's d / d norm output ; pass C's d penalty / d c_i
    std::vector<float*> lnpart;        // per critic level [4n * CG][2][HW]
    float *Q, *Dd, *Gx, *slopes;
    float *lnpart_g;                   // generator LayerNorm parameter-gradient partials
    // gradient ping-pong, small vectors, scratch
    float *Ga, *Gb, *dxbuf, *dzbuf, *dzr, *dflat, *ddv;
    float *wpartial, *colscratch, *colpart, *redpart, *raw, *scalars_own, *finpart;
    // ---- ResNet variant (models/fanogan_schlegl.py) ----
    int variant, dim;
    // ---- AAE family (UAD_GAN_AAE): dense-bottleneck BN autoencoder + optional re-encoding constraint + optional latent critic ----
    int aae_kind;                      // 0 constrained AE, 1 AAE, 2 constrained AAE
    bool a_constrained, a_critic;
    int a_h1, a_h2;
    long long a_cw, a_cb, a_zw, a_zb, a_dw, a_db, a_rw, a_rb, a_dbng, a_dbnb;   // conv2d, dense (z), dense (dec), conv2d_1, decoder input BN
    long long a_w1, a_b1, a_w2, a_b2, a_w3, a_b3, a_nd;                          // critic tensors, their total size
    std::vector<float*> a_eg, a_cp;    // per encoder level: d loss / d c_i [2n] and BN column partials (both encoder passes)
    float *a_xcat, *a_zm, *a_dzm, *a_dflat, *a_slab, *a_crit[3];                 // [x ; x_hat], z_ | z_rec, masked d z, d flat, critic slabs, d_fake / d_real / pen
    // dense GMVAE (aae_kind 3): four Dense heads on the flattened bottleneck, p(z|w,c) = two Dense layers on w_sampled + the 0.1 Variable
    bool gmv;
    int gm_W, gm_Z, gm_C, gm_J;        // dim_w, dim_z, dim_c, 2W + 2Z (row of head values: w_mu | w_log_sigma | z_mu | z_log_sigma)
    long long gm_hw[4], gm_hb[4], gm_mw, gm_mb, gm_lw, gm_lb, gm_var;
    int gm_hd[4], gm_ho[4];            // head widths and column offsets inside a row
    float *gm_hv, *gm_hvm, *gm_ws, *gm_M, *gm_Lq, *gm_pc, *gm_loss3, *gm_dhv, *gm_dM, *gm_dLq, *gm_dxhat;
    // Zimmerer VAE (aae_kind 4): k4 convolutions + bias + leaky_relu(0.2), no normalisation; E / G hold the blocks (gamma = beta = -1)
    bool chen;                         // aae_kind 7: constrained adversarial autoencoder on residual blocks (encoder = DB, decoder = GB)
    bool zim, zim_ce;                  // zim_ce (aae_kind 5): the context-encoding VAE on the same stack, both branches as one 2n-sample pass
    float* z_l1;                       // [2n] L1 maps of both branches
    long long z_muw, z_mub, z_lsw, z_lsb, z_dw, z_db, z_fw, z_fb;
    UadConvDesc z_fd;                  // final k4 s1 conv as the image-side (1-channel "big") relation: S 1, P 2, taps reversed
    float *z_wflip, *z_dwflip;         // [16 taps][16] reversed final-conv kernel and its gradient
    // original-architecture spatial GMVAE (aae_kind 6, models/gaussian_mixture_variational_autoencoder_You.py): a program of k3 layers
    struct YOp { int kind; Block L; bool relu; float *c, *a; int Hout, Cout; };   // kind 0 conv, 1 transposed conv, 2 nearest-neighbour x2
    bool you;
    std::vector<YOp> y_enc, y_dec;
    long long y_off[15], y_total;      // the 15 latent-head tensors (contiguous), their total size
    UadConvDesc y_fd;                  // p_x_z/y_mu (k3, 64 -> 1) as the image-side relation, taps reversed
    float *y_ones, *y_zeros, *y_loc, *y_colpart, *y_dheads, *y_da7, *y_dM, *y_dLq, *y_ws, *y_mid, *y_partial, *y_dzdec, *y_h;
    bool generic16;                    // UAD_MATH_BF16X3_ALL: generic contractions in bf16x3 too (opt-in, not parity-rated)
    int exact_from;                    // ResNet graph, bf16x3 mode: samples from this index on take the exact-fp32 kernels in g_conv_f / g_conv_d (-1: none)
    struct RB {                        // pre-activation residual block: LN -> ReLU -> conv1 (k3 s1) -> LN -> ReLU -> conv2, + shortcut
        bool gen;                      // generator block: conv2 / shortcut are transposed convolutions
        int stride, Hin, Hout, Cin, Cout;
        long long ln1g, ln1b, w1, b1, ln2g, ln2b, w2, b2, ws, bs;   // ws < 0: identity shortcut
        UadConvDesc d1, d2, ds;
        float *X, *H1, *C1, *H2, *OUT, *ST1, *ST2;   // X = previous block's OUT; critic: X/H1/H2/OUT hold 4n samples (tail = pass C adjoints)
        float *DX, *DOUT, *G1, *DSC;                 // gradients w.r.t. X (= previous DOUT), OUT, C1, shortcut-conv output (critic: 4n, tail = pass B)
        float *V1, *V2, *INJX, *INJC1, *LP1, *LP2;   // critic: pass B's d / d(norm output), pass C's injections, LayerNorm parameter partials
        size_t sx, sc1, sout;                        // floats per sample of X, C1, OUT
    };
    std::vector<RB> GB, DB;
    long long s_glg, s_glb, s_d0w, s_d0b;            // generator's last LayerNorm; critic's first conv (k3, Cin = 1)
    UadConvDesc s_d0;
    float *s_g0, *s_dg0;                             // generator dense output map [n,r,r,8d] and its gradient
    float *s_hf, *s_stf;                             // relu(LN(last generator block)) + statistics
    float *s_out0, *s_dout0;                         // critic first conv output [4n] and its gradient [4n]
    float *s_ta, *s_tb, *s_sp, *s_sct;               // temporaries: conv2 / conv1 data gradients, pooled shortcut, shortcut conv output
    std::map<std::string, std::pair<float*, long long>> dbg;
    std::vector<void*> allocs;
};

namespace {

float* P(uad_gan* m, long long off) { return m->params + off; }
float* Gr(uad_gan* m, long long off) { return m->grads + off; }

// ---- bucketed all-reduce of the trained group's gradients (data parallelism, library-issued RCCL) --------------------------------------------
static void gan_ar_issue(uad_gan* m, int b) {
    uad_gan::ArBucket& B = m->ar_b[b];
    if (B.issued || B.cnt == 0) { B.issued = true; return; }
    B.issued = true;
    (void)hipEventRecord(m->ar_ev_in[b], m->ar_st);                 // behind the kernels that wrote the bucket's tensors ...
    (void)hipStreamWaitEvent(m->ar_stream, m->ar_ev_in[b], 0);
    static const bool skip = getenv("UAD_AR_SKIP") != nullptr;      // measurement: everything but the collective itself
    if (!skip) (void)uad_rccl_allreduce(m->ar_comm, m->grads + B.off, B.cnt, m->ar_stream);      // ... on the collective stream: the phase's stream runs on
}
// phase start: buckets over the tensors inside [off, off + cnt), by cumulative size, on tensor boundaries, in offset order
static void gan_ar_begin(uad_gan* m, long long off, long long cnt, hipStream_t st) {
    m->ar_active = false;
    if (!m->ar_comm || cnt <= 0) return;
    std::vector<const Tensor*> ts;
    for (const Tensor& t : m->tensors) if (t.off >= off && t.off + t.count() <= off + cnt) ts.push_back(&t);
    std::sort(ts.begin(), ts.end(), [](const Tensor* a, const Tensor* b) { return a->off < b->off; });
    constexpr int K = 4;
    m->ar_nb = K; m->ar_tb.clear(); m->ar_marked.assign(ts.size(), 0);
    for (int b = 0; b < K; ++b) m->ar_b[b] = {0, 0, 0, false};
    long long cum = 0;
    for (const Tensor* t : ts) {
        int b = (int)((cum + t->count() / 2) * K / cnt);
        if (b >= K) b = K - 1;
        cum += t->count();
        uad_gan::ArBucket& B = m->ar_b[b];
        if (B.pending == 0) B.off = t->off;
        B.cnt = t->off + t->count() - B.off;
        B.pending += 1;
        m->ar_tb.push_back({t->off, b});
    }
    // the buckets must tile the slice (gaps between tensors would go unreduced): stretch each bucket to its successor / the slice's ends
    int prev = -1;
    for (int b = 0; b < K; ++b) {
        if (m->ar_b[b].pending == 0) continue;
        if (prev < 0) { m->ar_b[b].cnt += m->ar_b[b].off - off; m->ar_b[b].off = off; }
        else m->ar_b[prev].cnt = m->ar_b[b].off - m->ar_b[prev].off;
        prev = b;
    }
    if (prev >= 0) m->ar_b[prev].cnt = off + cnt - m->ar_b[prev].off;
    m->ar_st = st;
    m->ar_active = prev >= 0;
}
// the kernels that write the gradient of the tensor at `off` are enqueued and nothing later in this phase writes it again
void gan_grad_final(uad_gan* m, long long off) {
    if (!m->ar_active || off < 0) return;
    for (size_t i = 0; i < m->ar_tb.size(); ++i) {
        if (m->ar_tb[i].first != off || m->ar_marked[i]) continue;
        m->ar_marked[i] = 1;
        uad_gan::ArBucket& B = m->ar_b[m->ar_tb[i].second];
        if (--B.pending == 0) gan_ar_issue(m, m->ar_tb[i].second);
        return;
    }
}
// phase end: whatever was not marked (zero gradients that stay at their initialisation, graphs without hooks) goes now; the phase's stream -- the
// optimizer step comes next on it -- waits for the collective stream once
static void gan_ar_end(uad_gan* m) {
    if (!m->ar_active) return;
    for (int b = 0; b < m->ar_nb; ++b) if (!m->ar_b[b].issued) gan_ar_issue(m, b);
    (void)hipEventRecord(m->ar_ev_out, m->ar_stream);
    (void)hipStreamWaitEvent(m->ar_st, m->ar_ev_out, 0);
    m->ar_active = false;
}
const float* PKF(uad_gan* m, long long off) { return m->math == UAD_MATH_F32 ? m->wpack_f + off : nullptr; }
const float* PKD(uad_gan* m, long long off) { return m->math == UAD_MATH_F32 ? m->wpack_d + off : nullptr; }
const unsigned short* PK16F(uad_gan* m, long long off) { return m->math == UAD_MATH_BF16X3 ? (const unsigned short*)m->wpack16_f + 2 * off : nullptr; }
const unsigned short* PK16D(uad_gan* m, long long off) { return m->math == UAD_MATH_BF16X3 ? (const unsigned short*)m->wpack16_d + 2 * off : nullptr; }
// k3 / k1 tensors of the ResNet graph: the first two planes of the three-plane pack (h = bf16(x), m = bf16(x - h)) ARE the hi | lo planes of the two-plane
// pack, at the same plane stride -- with the three-plane buffers present the bf16x3 launches of the tap-list kernel read those and the tensors are packed
// once per optimizer step instead of twice (round 6; `generic16` = bf16x3_all keeps the two-plane pack for the generic kernels).
static bool k3_one_pack(const uad_gan* m) { return m->wpack3_f && m->wpack3_d && !m->generic16; }
const unsigned short* K3F(uad_gan* m, long long off) { return k3_one_pack(m) ? (const unsigned short*)m->wpack3_f + 4 * off : PK16F(m, off); }
const unsigned short* K3D(uad_gan* m, long long off) { return k3_one_pack(m) ? (const unsigned short*)m->wpack3_d + 4 * off : PK16D(m, off); }
long long PLANE(const Block& L) { return (long long)L.d.KS * L.d.KS * L.d.CB * L.d.CS; }
UadXform no_xform() { UadXform x; x.scale = nullptr; x.shift = nullptr; x.alpha = 1.f; x.mult = 1.f; return x; }
UadEpilogue epi_bias(const float* bias, const float* mul = nullptr, const float* add = nullptr) {
    UadEpilogue e;
    memset(&e, 0, sizeof e);
    e.kind = UAD_EPI_BIAS; e.bias = bias; e.mul = mul; e.add = add;
    return e;
}
UadConvDesc dense_desc(int n, int in, int out) { return UadConvDesc{n, 1, 1, in, 1, 1, out, 1, 1, 0}; }
UadConvDesc conv1x1_desc(int n, int h, int w, int cin, int cout) { return UadConvDesc{n, h, w, cin, h, w, cout, 1, 1, 0}; }
int ilog2i(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
inline unsigned blocks256(size_t n) { return (unsigned)((n + 255) / 256); }

long long add_tensor(uad_gan* m, const std::string& name, int rank, int s0, int s1, int s2, int s3) {
    Tensor t;
    t.name = name; t.off = m->nparams; t.rank = rank;
    t.shape[0] = s0; t.shape[1] = s1; t.shape[2] = s2; t.shape[3] = s3;
    m->nparams += t.count();
    m->tensors.push_back(t);
    return t.off;
}
int dev_alloc(uad_gan* m, float** p, size_t floats, const char* name = nullptr) {
    void* q = nullptr;
    if (floats == 0) floats = 4;
    HIP_TRY(hipMalloc(&q, floats * sizeof(float)));
    HIP_TRY(hipMemset(q, 0, floats * sizeof(float)));
    m->allocs.push_back(q);
    *p = (float*)q;
    if (name) m->dbg[name] = std::make_pair((float*)q, (long long)floats);
    return UAD_OK;
}

// ---- launch helpers ----
template <int ACT>
void rowdot(const float* feat, const float* w, const float* b, int rows, int C, float* out, hipStream_t st) {
    int lpr = C / 4;
    if (lpr > 64) lpr = 64;
    hipLaunchKernelGGL((rowdot_kernel<ACT>), dim3(blocks256((size_t)rows * lpr)), dim3(256), 0, st, feat, w, b, rows, C, lpr, out);
}
// One-pass LayerNorm forms (uad_gan_kernels.inc: the slice in registers between the statistics and the apply sweep).  Maps of up to 1024 pixels: one
// workgroup per (sample, 32 channels), with the pixel-lane count the two-pass kernels use for that map (same bits).  Larger maps: Q workgroups per slice
// that exchange their partial sums once (clustered forms).  UAD_NO_LN1 keeps the two-pass kernels everywhere, UAD_NO_LNQ on the large maps.
inline bool ln_one_pass(int HW) {
    static const bool on = getenv("UAD_NO_LN1") == nullptr;
    return on && (HW <= 256 || (HW >= 512 && HW <= 1024));
}
constexpr int kLnqPL = 128;           // pixel lanes per workgroup of the clustered forms (x 8 pixels per lane = 1024 pixels per workgroup; 128 / 64 / 32 lanes measured within 0.3 % of each other)
inline int ln_cluster_pl() { return kLnqPL; }
inline bool ln_clustered(int HW) {
    static const bool on = getenv("UAD_NO_LN1") == nullptr && getenv("UAD_NO_LNQ") == nullptr;
    return on && HW > 1024;
}
// exchange scratch for a launch of `wgs` workgroups; null when it cannot be had (the caller then takes the two-pass kernel)
inline bool ln_xch_for(uad_gan* m, int HW, size_t wgs, LnXch* x, int px_per_wg = kLnqPL * 8, int nv = 2) {
    x->Q = (HW + px_per_wg - 1) / px_per_wg;
    wgs *= (size_t)x->Q;
    static const int fault = getenv("UAD_LNQ_FAULT") ? 1 : 0;      // tests/test_gpu_knobs.py: a sibling that never shows up costs time, not correctness
    x->fault = fault;
    x->xch = nullptr; x->flags = nullptr; x->epoch = 0;
    if (x->Q == 1) return true;                 // no exchange
    if (wgs * 32 * nv > m->ln_xcap) {
        (void)hipDeviceSynchronize();
        float* xp = nullptr; float* fp = nullptr;
        const size_t cap = (wgs + wgs / 2) * 160;
        if (dev_alloc(m, &xp, cap) != UAD_OK || dev_alloc(m, &fp, cap / 32) != UAD_OK) return false;
        (void)hipDeviceSynchronize();          // dev_alloc's memset runs on the null stream; the phases' streams do not wait for it
        m->ln_xch = xp; m->ln_flags = reinterpret_cast<unsigned*>(fp); m->ln_xcap = cap;
    }
    x->xch = m->ln_xch; x->flags = m->ln_flags; x->epoch = ++m->ln_epoch;
    if (x->epoch == 0) x->epoch = ++m->ln_epoch;          // 0 is the scratch's initial content
    return true;
}
void ln_fwd(uad_gan* m, const float* c, const float* gamma, const float* beta, float alpha, int N, int HW, int C, float* a, float* stats, hipStream_t st) {
    if (ln_one_pass(HW)) {
        if (HW >= 512) hipLaunchKernelGGL((ln_fwd1_kernel<8, 128, 8>), dim3(C / 32, N), dim3(1024), 0, st, c, gamma, beta, alpha, HW, C, a, stats);
        else hipLaunchKernelGGL((ln_fwd1_kernel<8, 32, 8>), dim3(C / 32, N), dim3(256), 0, st, c, gamma, beta, alpha, HW, C, a, stats);
        return;
    }
    LnXch x;
    if (ln_clustered(HW) && ln_xch_for(m, HW, (size_t)(C / 32) * N, &x)) {
        const dim3 g((C / 32) * x.Q, N);
        hipLaunchKernelGGL((ln_fwdq_kernel<kLnqPL, 8>), g, dim3(8 * kLnqPL), 0, st, c, gamma, beta, alpha, HW, C, a, stats, x);
        return;
    }
    if (HW >= 512) hipLaunchKernelGGL((ln_fwd_kernel<128>), dim3(C / 32, N), dim3(1024), 0, st, c, gamma, beta, alpha, HW, C, a, stats);
    else hipLaunchKernelGGL((ln_fwd_kernel<32>), dim3(C / 32, N), dim3(256), 0, st, c, gamma, beta, alpha, HW, C, a, stats);
}
void ln_bwd(uad_gan* m, const LnBwdArgs& a, int N, hipStream_t st) {
    if (ln_one_pass(a.HW)) {
        if (a.HW >= 512) hipLaunchKernelGGL((ln_bwd1_kernel<8, 128, 8>), dim3(a.C / 32, N), dim3(1024), 0, st, a);
        else hipLaunchKernelGGL((ln_bwd1_kernel<8, 32, 8>), dim3(a.C / 32, N), dim3(256), 0, st, a);
        return;
    }
    LnXch x;
    if (ln_clustered(a.HW) && ln_xch_for(m, a.HW, (size_t)(a.C / 32) * N, &x)) {
        const dim3 g((a.C / 32) * x.Q, N);
        hipLaunchKernelGGL((ln_bwdq_kernel<kLnqPL, 8>), g, dim3(8 * kLnqPL), 0, st, a, x);
        return;
    }
    if (a.HW >= 512) hipLaunchKernelGGL((ln_bwd_kernel<128>), dim3(a.C / 32, N), dim3(1024), 0, st, a);
    else hipLaunchKernelGGL((ln_bwd_kernel<32>), dim3(a.C / 32, N), dim3(256), 0, st, a);
}
void ln_bwd2(uad_gan* m, const LnBwd2Args& a, int N, hipStream_t st) {
    static const bool one_pass = getenv("UAD_NO_LN1") == nullptr && getenv("UAD_NO_LN2Q") == nullptr;
    LnXch x;
    if (one_pass && a.HW <= 256 && ln_xch_for(m, a.HW, (size_t)(a.C / 32) * N, &x, 256, 5)) {
        hipLaunchKernelGGL((ln_bwd2q_kernel<32, 8>), dim3(a.C / 32, N), dim3(256), 0, st, a, x);
        return;
    }
    if (one_pass && a.HW >= 512 && ln_xch_for(m, a.HW, (size_t)(a.C / 32) * N, &x, 512, 5)) {
        hipLaunchKernelGGL((ln_bwd2q_kernel<128, 4>), dim3((a.C / 32) * x.Q, N), dim3(1024), 0, st, a, x);
        return;
    }
    if (a.HW >= 512) hipLaunchKernelGGL((ln_bwd2_kernel<128>), dim3(a.C / 32, N), dim3(1024), 0, st, a);
    else hipLaunchKernelGGL((ln_bwd2_kernel<32>), dim3(a.C / 32, N), dim3(256), 0, st, a);
}
void bn_act_fwd(uad_gan* m, const Block& L, const float* c, int N, float* a, hipStream_t st) {
    const size_t total4 = (size_t)N * L.H * L.W * L.C / 4;
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(blocks256(total4)), dim3(256), 0, st, c, P(m, L.gamma), P(m, L.beta),
                       1.0f / sqrtf(1.0f + kBnEps), kLrelu, total4, L.C, a);
}
constexpr int kBnBwdBlocks = 512;
void bn_act_bwd(uad_gan* m, const Block& L, const float* da, const float* c, int N, float* dc, hipStream_t st) {
    const float rs0 = 1.0f / sqrtf(1.0f + kBnEps);
    const int rows = N * L.H * L.W;
    const int rpb = (rows + kBnBwdBlocks - 1) / kBnBwdBlocks;
    const int blocks = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(bn_act_bwd_kernel, dim3(blocks), dim3(256), 0, st, da, c, P(m, L.gamma), P(m, L.beta), rs0, kLrelu, rows,
                       rpb, L.C, dc, m->colpart);
    uad_launch_bn_grad_finalize(m->colpart, blocks, L.C, P(m, L.gamma), rs0, Gr(m, L.gamma), Gr(m, L.beta), Gr(m, L.b), st);
}
// raw[k] = scale * sum(...)
template <int MODE>
void reduce_to(uad_gan* m, int k, const float* a, const float* b, size_t n, float scale, float* map, hipStream_t st) {
    hipLaunchKernelGGL((sum_kernel<MODE>), dim3(256), dim3(256), 0, st, a, b, n, map, m->redpart);
    uad_launch_reduce_partials(m->redpart, 256, 1, scale, m->raw + k, st);
}

int refresh_packs(uad_gan* m, hipStream_t st) {
    if (m->packed_valid) return UAD_OK;
    if (m->zim || m->you || m->chen) { m->packed_valid = true; m->pack_all = false; m->dirty_lo = (1ll << 62); m->dirty_hi = -1; return UAD_OK; }       // no k5 layers: nothing is packed
    long long offs[16]; int cbs[16], css[16], taps[16]; int np = 0;
    // an optimizer step changes ONE group's variables (round 5: a WGAN-GP iteration of the ResNet graph repacked all 21 M parameters twelve times --
    // 2 x 80 launches of 72 / 115 us in the round-4 rocprofv3 summary, 4.5 % of the iteration): only tensors inside the dirty range are repacked
    static const bool always_all = getenv("UAD_GAN_PACK_ALL") != nullptr;      // A/B: repack every tensor after every optimizer step, as before round 5
    const bool all = m->pack_all || always_all;
    const long long dlo = m->dirty_lo, dhi = m->dirty_hi;
    auto stale = [&](long long w) { return all || (w >= dlo && w < dhi); };
    auto add = [&](const Block& L) { if (np < 16 && L.d.CB % 4 == 0 && L.d.CS % 4 == 0 && stale(L.w)) { offs[np] = L.w; cbs[np] = L.d.CB; css[np] = L.d.CS; taps[np] = 25; ++np; } };
    for (size_t i = 1; i < m->E.size(); ++i) add(m->E[i]);
    for (auto& L : m->G) add(L);
    for (size_t i = 1; i < m->D.size(); ++i) add(m->D[i]);
    if (np > 0) {
        if (m->math == UAD_MATH_BF16X3)
            uad_launch_pack_weights_bf16(m->params, (unsigned short*)m->wpack16_f, (unsigned short*)m->wpack16_d, offs, cbs, css, taps, np, st);
        else
            uad_launch_pack_weights(m->params, m->wpack_f, m->wpack_d, offs, cbs, css, taps, np, st);
    }
    if (m->variant == 1 && m->math == UAD_MATH_BF16X3) {
        // ResNet graph: bf16 hi | lo planes of the residual blocks' k3 kernels for the tap-list spatial kernel (uad_convk16.inc), 16 tensors per launch
        np = 0;
        auto flush = [&]() {
            if (np > 0) {
                if (!k3_one_pack(m)) uad_launch_pack_weights_bf16(m->params, (unsigned short*)m->wpack16_f, (unsigned short*)m->wpack16_d, offs, cbs, css, taps, np, st);
                if (m->wpack3_f) uad_launch_pack_weights_bf16_3p(m->params, (unsigned short*)m->wpack3_f, (unsigned short*)m->wpack3_d, offs, cbs, css, taps, np, st);
            }
            np = 0;
        };
        auto addk = [&](long long w, const UadConvDesc& d) {
            if (w < 0 || (d.KS != 3 && d.KS != 1) || d.CB % 8 || d.CS % 8 || !stale(w)) return;
            offs[np] = w; cbs[np] = d.CB; css[np] = d.CS; taps[np] = d.KS * d.KS;
            if (++np == 16) flush();
        };
        for (auto& B : m->GB) { addk(B.w1, B.d1); addk(B.w2, B.d2); addk(B.ws, B.ds); }
        for (auto& B : m->DB) { addk(B.w1, B.d1); addk(B.w2, B.d2); addk(B.ws, B.ds); }
        flush();
    }
    m->packed_valid = true; m->pack_all = false; m->dirty_lo = (1ll << 62); m->dirty_hi = -1;
    return UAD_OK;
}

// Conv2D block forward / data gradient / filter gradient (E and D), ConvT block likewise (G)
void conv_fwd(uad_gan* m, const Block& L, int N, const float* in, float* out, bool bias, hipStream_t st) {
    UadConvDesc d = L.d; d.N = N;
    if (d.CB % 4) uad_launch_conv_first_fwd(d, in, P(m, L.w), bias ? P(m, L.b) : nullptr, out, st);
    else uad_launch_conv_f(d, in, no_xform(), P(m, L.w), out, epi_bias(bias ? P(m, L.b) : nullptr), st, PKF(m, L.w), m->ws, PK16F(m, L.w), PLANE(L));
}
void conv_dgrad(uad_gan* m, const Block& L, int N, const float* g, float* out, hipStream_t st) {
    UadConvDesc d = L.d; d.N = N;
    if (d.CB % 4) uad_launch_conv_first_dgrad_plain(d, g, P(m, L.w), out, st);
    else uad_launch_conv_d(d, g, no_xform(), P(m, L.w), out, epi_bias(nullptr), st, PKD(m, L.w), m->ws, PK16D(m, L.w), PLANE(L));
}
void conv_wgrad(uad_gan* m, const Block& L, int N, const float* in, const float* g, hipStream_t st) {
    UadConvDesc d = L.d; d.N = N;
    if (d.CB % 4) uad_launch_conv_first_wgrad(d, in, g, Gr(m, L.w), m->wpartial, st);
    else uad_launch_conv_w(d, in, no_xform(), g, no_xform(), Gr(m, L.w), m->wpartial, st, m->math == UAD_MATH_BF16X3);
}
void convT_fwd(uad_gan* m, const Block& L, int N, const float* in, float* out, hipStream_t st) {
    UadConvDesc d = L.d; d.N = N;
    uad_launch_conv_d(d, in, no_xform(), P(m, L.w), out, epi_bias(P(m, L.b)), st, PKD(m, L.w), m->ws, PK16D(m, L.w), PLANE(L));
}
void convT_dgrad(uad_gan* m, const Block& L, int N, const float* g, float* out, hipStream_t st) {
    UadConvDesc d = L.d; d.N = N;
    uad_launch_conv_f(d, g, no_xform(), P(m, L.w), out, epi_bias(nullptr), st, PKF(m, L.w), m->ws, PK16F(m, L.w), PLANE(L));
}
void convT_wgrad(uad_gan* m, const Block& L, int N, const float* in, const float* g, hipStream_t st) {
    UadConvDesc d = L.d; d.N = N;
    uad_launch_conv_w(d, g, no_xform(), in, no_xform(), Gr(m, L.w), m->wpartial, st, m->math == UAD_MATH_BF16X3);
}
size_t asz(const Block& L) { return (size_t)L.H * L.W * L.C; }   // floats per sample of the block's output

// ------------------------------------------------------------------------------------------------ Encoder
void enc_forward(uad_gan* m, const float* x, const float* mask_z, int n, hipStream_t st) {
    const int r = m->cfg.inter_res;
    const float* in = x;
    for (size_t i = 0; i < m->E.size(); ++i) {
        conv_fwd(m, m->E[i], n, in, m->ec[i], true, st);
        bn_act_fwd(m, m->E[i], m->ec[i], n, m->ea[i + 1], st);
        in = m->ea[i + 1];
    }
    uad_launch_conv_f(conv1x1_desc(n, r, r, m->cenc, m->cmid), in, no_xform(), P(m, m->e_cw), m->et, epi_bias(P(m, m->e_cb)), st, nullptr, m->ws);
    uad_launch_conv_f(dense_desc(n, m->flat, m->cfg.zdim), m->et, no_xform(), P(m, m->e_dw), m->zr, epi_bias(P(m, m->e_db), mask_z), st, nullptr, m->ws);
    const size_t nz = (size_t)n * m->cfg.zdim;
    hipLaunchKernelGGL(tanh_kernel, dim3(blocks256(nz)), dim3(256), 0, st, m->zr, nz, m->z);
}
// dz = d loss / d z_enc in m->dzbuf; writes every Encoder gradient
void enc_backward(uad_gan* m, const float* x, const float* mask_z, int n, hipStream_t st) {
    const int r = m->cfg.inter_res, zd = m->cfg.zdim;
    const size_t nz = (size_t)n * zd;
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(blocks256(nz)), dim3(256), 0, st, m->dzbuf, m->z, mask_z, nz, m->dzr);
    const UadConvDesc dd = dense_desc(n, m->flat, zd), dc1 = conv1x1_desc(n, r, r, m->cenc, m->cmid);
    uad_launch_conv_w(dd, m->et, no_xform(), m->dzr, no_xform(), Gr(m, m->e_dw), m->wpartial, st);
    uad_launch_colsum(m->dzr, n, zd, Gr(m, m->e_db), m->colscratch, st);
    uad_launch_conv_d(dd, m->dzr, no_xform(), P(m, m->e_dw), m->dflat, epi_bias(nullptr), st, nullptr, m->ws);
    uad_launch_conv_w(dc1, m->ea[m->E.size()], no_xform(), m->dflat, no_xform(), Gr(m, m->e_cw), m->wpartial, st);
    uad_launch_colsum(m->dflat, n * r * r, m->cmid, Gr(m, m->e_cb), m->colscratch, st);
    for (long long o : {m->e_dw, m->e_db, m->e_cw, m->e_cb}) gan_grad_final(m, o);
    float* g = m->Ga; float* gn = m->Gb;
    uad_launch_conv_d(dc1, m->dflat, no_xform(), P(m, m->e_cw), g, epi_bias(nullptr), st, nullptr, m->ws);
    for (int i = (int)m->E.size() - 1; i >= 0; --i) {
        bn_act_bwd(m, m->E[i], g, m->ec[i], n, gn, st);          // gn = d loss / d c_i ; gamma, beta, bias gradients
        conv_wgrad(m, m->E[i], n, i == 0 ? x : m->ea[i], gn, st);
        for (long long o : {m->E[i].w, m->E[i].b, m->E[i].gamma, m->E[i].beta}) gan_grad_final(m, o);
        if (i > 0) conv_dgrad(m, m->E[i], n, gn, g, st);
    }
}

// AnoVAE-GAN encoder (models/anovaegan.py:14-35): mu / log-sigma heads with their own dropout masks, z = mu + eps * exp(log_sigma)
void v_enc_forward(uad_gan* m, const uad_gan_io_t* io, int n, hipStream_t st) {
    const int r = m->cfg.inter_res, zd = m->cfg.zdim;
    const float* in = io->x;
    for (size_t i = 0; i < m->E.size(); ++i) {
        conv_fwd(m, m->E[i], n, in, m->ec[i], true, st);
        bn_act_fwd(m, m->E[i], m->ec[i], n, m->ea[i + 1], st);
        in = m->ea[i + 1];
    }
    uad_launch_conv_f(conv1x1_desc(n, r, r, m->cenc, m->cmid), in, no_xform(), P(m, m->e_cw), m->et, epi_bias(P(m, m->e_cb)), st, nullptr, m->ws);
    const UadConvDesc dd = dense_desc(n, m->flat, zd);
    uad_launch_conv_f(dd, m->et, no_xform(), P(m, m->e_dw), m->v_mu_raw, epi_bias(P(m, m->e_db)), st, nullptr, m->ws);
    uad_launch_conv_f(dd, m->et, no_xform(), P(m, m->e_sw), m->v_ls_raw, epi_bias(P(m, m->e_sb)), st, nullptr, m->ws);
    uad_launch_reparam_fwd(n, n, zd, m->v_mu_raw, m->v_ls_raw, io->mask_z, io->mask_sigma, nullptr, io->eps, m->v_mu, m->v_ls, m->v_sigma,
                           m->z, m->v_kl, st);
}
// dz in m->dzbuf; klw = kl_weight / n; writes every Encoder gradient
void v_enc_backward(uad_gan* m, const uad_gan_io_t* io, int n, float klw, hipStream_t st) {
    const int r = m->cfg.inter_res, zd = m->cfg.zdim;
    uad_launch_reparam_bwd(n, n, zd, m->dzbuf, m->v_mu, m->v_sigma, io->eps, io->mask_z, io->mask_sigma, nullptr, klw, m->v_dmu, m->v_dls, st);
    const UadConvDesc dd = dense_desc(n, m->flat, zd), dc1 = conv1x1_desc(n, r, r, m->cenc, m->cmid);
    uad_launch_conv_w(dd, m->et, no_xform(), m->v_dmu, no_xform(), Gr(m, m->e_dw), m->wpartial, st);
    uad_launch_colsum(m->v_dmu, n, zd, Gr(m, m->e_db), m->colscratch, st);
    uad_launch_conv_w(dd, m->et, no_xform(), m->v_dls, no_xform(), Gr(m, m->e_sw), m->wpartial, st);
    uad_launch_colsum(m->v_dls, n, zd, Gr(m, m->e_sb), m->colscratch, st);
    uad_launch_conv_d(dd, m->v_dmu, no_xform(), P(m, m->e_dw), m->v_dflat2, epi_bias(nullptr), st, nullptr, m->ws);
    uad_launch_conv_d(dd, m->v_dls, no_xform(), P(m, m->e_sw), m->dflat, epi_bias(nullptr, nullptr, m->v_dflat2), st, nullptr, m->ws);
    uad_launch_conv_w(dc1, m->ea[m->E.size()], no_xform(), m->dflat, no_xform(), Gr(m, m->e_cw), m->wpartial, st);
    uad_launch_colsum(m->dflat, n * r * r, m->cmid, Gr(m, m->e_cb), m->colscratch, st);
    float* g = m->Ga; float* gn = m->Gb;
    uad_launch_conv_d(dc1, m->dflat, no_xform(), P(m, m->e_cw), g, epi_bias(nullptr), st, nullptr, m->ws);
    for (int i = (int)m->E.size() - 1; i >= 0; --i) {
        bn_act_bwd(m, m->E[i], g, m->ec[i], n, gn, st);
        conv_wgrad(m, m->E[i], n, i == 0 ? io->x : m->ea[i], gn, st);
        if (i > 0) conv_dgrad(m, m->E[i], n, gn, g, st);
    }
}

// ------------------------------------------------------------------------------------------------ Generator
void gen_forward(uad_gan* m, const float* z, const float* mask_g, int n, hipStream_t st) {
    const int r = m->cfg.inter_res;
    uad_launch_conv_f(dense_desc(n, m->cfg.zdim, m->flat), z, no_xform(), P(m, m->g_dw), m->gdv, epi_bias(P(m, m->g_db), mask_g), st, nullptr, m->ws);
    uad_launch_conv_f(conv1x1_desc(n, r, r, m->cmid, m->cenc), m->gdv, no_xform(), P(m, m->g_cw), m->gc[0], epi_bias(P(m, m->g_cb)), st, nullptr, m->ws);
    ln_fwd(m, m->gc[0], P(m, m->g_ln0g), P(m, m->g_ln0b), 0.0f, n, r * r, m->cenc, m->ga[0], m->gstat[0], st);
    for (size_t i = 0; i < m->G.size(); ++i) {
        const Block& L = m->G[i];
        convT_fwd(m, L, n, m->ga[i], m->gc[i + 1], st);
        ln_fwd(m, m->gc[i + 1], P(m, L.gamma), P(m, L.beta), kLrelu, n, L.H * L.W, L.C, m->ga[i + 1], m->gstat[i + 1], st);
    }
    const Block& LL = m->G.back();
    const int rows = n * LL.H * LL.W;
    if (m->variant == UAD_GAN_ANOVAEGAN) rowdot<0>(m->ga[m->G.size()], P(m, m->g_fw), P(m, m->g_fb), rows, LL.C, m->xg, st);   // linear output
    else rowdot<1>(m->ga[m->G.size()], P(m, m->g_fw), P(m, m->g_fb), rows, LL.C, m->xg, st);
}
// dx = d loss / d generator output (post-sigmoid); pg: Generator parameter gradients; dz_out: optional d loss / d z
void gen_backward(uad_gan* m, const float* z, const float* mask_g, const float* dx, int n, bool pg, float* dz_out, hipStream_t st) {
    const int r = m->cfg.inter_res;
    const Block& LL = m->G.back();
    const int rows = n * LL.H * LL.W;
    float* g = m->Ga; float* gn = m->Gb;
    {
        const int rpb = (rows + 1023) / 1024, blocks = (rows + rpb - 1) / rpb;
        hipLaunchKernelGGL(gfinal_bwd_kernel, dim3(blocks), dim3(256), 0, st, dx, m->xg, m->ga[m->G.size()], P(m, m->g_fw), rows, rpb,
                           LL.C, m->variant == UAD_GAN_ANOVAEGAN ? 2 : 0, g, pg ? m->finpart : nullptr);
        if (pg) {
            uad_launch_reduce_partials(m->finpart, blocks, LL.C + 1, 1.0f, m->colscratch, st);
            hipMemcpyAsync(Gr(m, m->g_fw), m->colscratch, LL.C * sizeof(float), hipMemcpyDeviceToDevice, st);
            hipMemcpyAsync(Gr(m, m->g_fb), m->colscratch + LL.C, sizeof(float), hipMemcpyDeviceToDevice, st);
        }
    }
    for (int i = (int)m->G.size() - 1; i >= 0; --i) {
        const Block& L = m->G[i];
        LnBwdArgs a;
        memset(&a, 0, sizeof a);
        a.da = g; a.c = m->gc[i + 1]; a.stats = m->gstat[i + 1]; a.gamma = P(m, L.gamma); a.beta = P(m, L.beta); a.alpha = kLrelu;
        a.HW = L.H * L.W; a.C = L.C; a.dc = gn; a.gpart = pg ? m->lnpart_g : nullptr;
        ln_bwd(m, a, n, st);
        if (pg) {
            uad_launch_reduce_partials(m->lnpart_g, n * (L.C / 32), 2 * a.HW, 1.0f, Gr(m, L.gamma), st);   // gamma | beta adjacent
            convT_wgrad(m, L, n, m->ga[i], gn, st);
            // a bias in front of a LayerNorm over (H, W) is removed by the mean subtraction: its gradient is identically zero.
            // The gradient buffer is zero-initialised and nothing writes those entries, so they stay exact zeros.
            for (long long o : {L.w, L.b, L.gamma, L.beta}) gan_grad_final(m, o);
        }
        convT_dgrad(m, L, n, gn, g, st);
    }
    LnBwdArgs a;
    memset(&a, 0, sizeof a);
    a.da = g; a.c = m->gc[0]; a.stats = m->gstat[0]; a.gamma = P(m, m->g_ln0g); a.beta = P(m, m->g_ln0b); a.alpha = 0.0f;
    a.HW = r * r; a.C = m->cenc; a.dc = gn; a.gpart = pg ? m->lnpart_g : nullptr;
    ln_bwd(m, a, n, st);
    const UadConvDesc dc1 = conv1x1_desc(n, r, r, m->cmid, m->cenc), dd = dense_desc(n, m->cfg.zdim, m->flat);
    if (pg) {
        uad_launch_reduce_partials(m->lnpart_g, n * (m->cenc / 32), 2 * r * r, 1.0f, Gr(m, m->g_ln0g), st);
        uad_launch_conv_w(dc1, m->gdv, no_xform(), gn, no_xform(), Gr(m, m->g_cw), m->wpartial, st);
    }
    uad_launch_conv_d(dc1, gn, no_xform(), P(m, m->g_cw), m->ddv, epi_bias(nullptr, mask_g), st, nullptr, m->ws);
    if (pg) {
        uad_launch_conv_w(dd, z, no_xform(), m->ddv, no_xform(), Gr(m, m->g_dw), m->wpartial, st);
        uad_launch_colsum(m->ddv, n, m->flat, Gr(m, m->g_db), m->colscratch, st);
    }
    if (dz_out) uad_launch_conv_d(dd, m->ddv, no_xform(), P(m, m->g_dw), dz_out, epi_bias(nullptr), st, nullptr, m->ws);
}

// ------------------------------------------------------------------------------------------------ Critic
// forward of N samples starting at din; head: also d = Dense(1)(features)
void disc_forward(uad_gan* m, int N, bool head, hipStream_t st) {
    const float* in = m->din;
    for (size_t i = 0; i < m->D.size(); ++i) {
        const Block& L = m->D[i];
        conv_fwd(m, L, N, in, m->Dc[i], true, st);
        ln_fwd(m, m->Dc[i], P(m, L.gamma), P(m, L.beta), kLrelu, N, L.H * L.W, L.C, m->Da[i + 1], m->Dstat[i], st);
        in = m->Da[i + 1];
    }
    if (head) {
        const Block& LL = m->D.back();
        const int rows = N * LL.H * LL.W;
        rowdot<0>(m->Da[m->D.size()], P(m, m->d_hw), P(m, m->d_hb), rows, LL.C, m->Dd, st);
    }
}
// ordinary backward of the first N samples; the top gradient (d loss / d features) is in m->Ga.
//   pg: parameter gradients, with pass C's pairs as an extra `ntail` samples of every filter gradient / LayerNorm partial set
//   inject_lo >= 0: add m->inj[i] to d c_i of samples [inject_lo, inject_lo + ntail)
//   dx_out: optional d loss / d input
void disc_backward(uad_gan* m, int N, bool pg, int ntail, int inject_lo, float* dx_out, hipStream_t st) {
    float* g = m->Ga; float* gn = m->Gb;
    for (int i = (int)m->D.size() - 1; i >= 0; --i) {
        const Block& L = m->D[i];
        LnBwdArgs a;
        memset(&a, 0, sizeof a);
        a.da = g; a.c = m->Dc[i]; a.stats = m->Dstat[i]; a.gamma = P(m, L.gamma); a.beta = P(m, L.beta); a.alpha = kLrelu;
        a.HW = L.H * L.W; a.C = L.C; a.dc = m->Dg[i]; a.gpart = pg ? m->lnpart[i] : nullptr;
        if (inject_lo >= 0) { a.add = m->inj[i]; a.add_lo = inject_lo; a.add_hi = inject_lo + ntail; }
        ln_bwd(m, a, N, st);
        if (pg) {
            uad_launch_reduce_partials(m->lnpart[i], (N + ntail) * (L.C / 32), 2 * a.HW, 1.0f, Gr(m, L.gamma), st);
            conv_wgrad(m, L, N + ntail, i == 0 ? m->din : m->Da[i], m->Dg[i], st);
            for (long long o : {L.w, L.b, L.gamma, L.beta}) gan_grad_final(m, o);
        }
        if (i > 0) { conv_dgrad(m, L, N, m->Dg[i], gn, st); float* t = g; g = gn; gn = t; }
        else if (dx_out) conv_dgrad(m, L, N, m->Dg[0], dx_out, st);
    }
}



// residual-block constrained AAE (aae_kind 7): the AAE-family phases on the ResNet-graph blocks (defined after them)
void c_encode(uad_gan* m, const float* x, int n, int off, hipStream_t st);
void c_encode_backward(uad_gan* m, const float* dz, int n, int off, float* dx_out, hipStream_t st);
void c_encode_wgrads(uad_gan* m, int nall, hipStream_t st);
void c_decode(uad_gan* m, const float* z, int n, hipStream_t st);
void c_decode_backward(uad_gan* m, const float* z, const float* dxh, int n, float* dz_out, bool pg, hipStream_t st);

// ================================================================================================ AAE family
// models/constrained_autoencoder.py, adversarial_autoencoder.py, constrained_adversarial_autoencoder.py on materialised activations.
void bn_fwd(uad_gan* m, long long gamma, long long beta, float alpha, const float* c, size_t rows, int C, float* a, hipStream_t st) {
    const size_t total4 = rows * C / 4;
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(blocks256(total4)), dim3(256), 0, st, c, P(m, gamma), P(m, beta), 1.0f / sqrtf(1.0f + kBnEps),
                       alpha, total4, C, a);
}
// dc = da * act' * gamma'; column partials into `colpart`; returns the number of partial rows written
int bn_bwd_partial(uad_gan* m, long long gamma, long long beta, float alpha, const float* da, const float* c, int rows, int C, float* dc,
                   float* colpart, hipStream_t st) {
    const int rpb = (rows + kBnBwdBlocks - 1) / kBnBwdBlocks;
    const int blocks = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(bn_act_bwd_kernel, dim3(blocks), dim3(256), 0, st, da, c, P(m, gamma), P(m, beta), 1.0f / sqrtf(1.0f + kBnEps), alpha, rows,
                       rpb, C, dc, colpart);
    return blocks;
}
void bn_finalize(uad_gan* m, const float* colpart, int T, int C, long long gamma, long long beta, long long bias, hipStream_t st) {
    uad_launch_bn_grad_finalize(colpart, T, C, P(m, gamma), 1.0f / sqrtf(1.0f + kBnEps), Gr(m, gamma), Gr(m, beta), bias >= 0 ? Gr(m, bias) : nullptr, st);
}
// encoder pass of n samples at sample offset `off` (0: x, n: x_hat); z (post dropout) lands in a_zm + off * zdim
void a_encode(uad_gan* m, const float* x, const float* mask, int n, int off, hipStream_t st) {
    if (m->chen) { c_encode(m, x, n, off, st); return; }
    const int r = m->cfg.inter_res, zd = m->cfg.zdim;
    const float* in = x;
    for (size_t i = 0; i < m->E.size(); ++i) {
        const Block& L = m->E[i];
        float* c = m->ec[i] + (size_t)off * asz(L);
        float* a = m->ea[i + 1] + (size_t)off * asz(L);
        conv_fwd(m, L, n, in, c, true, st);
        bn_fwd(m, L.gamma, L.beta, kLrelu, c, (size_t)n * L.H * L.W, L.C, a, st);
        in = a;
    }
    float* t = m->et + (size_t)off * m->flat;
    uad_launch_conv_f(conv1x1_desc(n, r, r, m->cenc, m->cmid), in, no_xform(), P(m, m->a_cw), t, epi_bias(P(m, m->a_cb)), st, nullptr, m->ws);
    if (m->gmv) {
        // the four Dense heads (model :27-42): raw values (bias added) into one row per sample; dropout is applied by the latent kernel
        for (int h = 0; h < 4; ++h) {
            UadSdArgs a;
            memset(&a, 0, sizeof a);
            a.a = t; a.lda = m->flat; a.B = P(m, m->gm_hw[h]); a.sBi = m->gm_hd[h]; a.sBo = 1; a.bias = P(m, m->gm_hb[h]);
            a.R = n; a.I = m->flat; a.O = m->gm_hd[h]; a.out = m->gm_hv + m->gm_ho[h]; a.ldo = m->gm_J;
            uad_launch_sd(a, st);
        }
        return;
    }
    uad_launch_conv_f(dense_desc(n, m->flat, zd), t, no_xform(), P(m, m->a_zw), m->a_zm + (size_t)off * zd, epi_bias(P(m, m->a_zb), mask), st, nullptr, m->ws);
}
// data-gradient chain of one encoder pass from d / d z (post dropout) in `dz`: leaves d c_i in a_eg[i] (+off) and the BN column
// partials in a_cp[i] (rows [cp_row0, ...)); returns the partial-row count (identical for every level is NOT assumed: see cp_rows)
void a_encode_backward(uad_gan* m, const float* dz, const float* mask, int n, int off, int* cp_rows, float* dx_out, hipStream_t st) {
    if (m->chen) { c_encode_backward(m, dz, n, off, dx_out, st); return; }
    const int r = m->cfg.inter_res, zd = m->cfg.zdim;
    float* dfl = m->a_dflat + (size_t)off * m->flat;
    if (m->gmv) {
        // dz = d / d raw head values [n, 2W+2Z]: d flat = sum over the four heads of dhead . kernel^T
        for (int h = 0; h < 4; ++h) {
            UadSdArgs a;
            memset(&a, 0, sizeof a);
            a.a = dz + m->gm_ho[h]; a.lda = m->gm_J; a.B = P(m, m->gm_hw[h]); a.sBi = 1; a.sBo = m->gm_hd[h];
            a.R = n; a.I = m->gm_hd[h]; a.O = m->flat; a.out = dfl; a.ldo = m->flat; a.accumulate = h > 0;
            uad_launch_sd(a, st);
        }
    } else {
        float* dzm = m->a_dzm + (size_t)off * zd;
        uad_launch_mul(dz, mask, dzm, (size_t)n * zd, st);
        uad_launch_conv_d(dense_desc(n, m->flat, zd), dzm, no_xform(), P(m, m->a_zw), dfl, epi_bias(nullptr), st, nullptr, m->ws);
    }
    float* g = m->Ga; float* gn = m->Gb;
    uad_launch_conv_d(conv1x1_desc(n, r, r, m->cenc, m->cmid), dfl, no_xform(), P(m, m->a_cw), g, epi_bias(nullptr), st, nullptr, m->ws);
    for (int i = (int)m->E.size() - 1; i >= 0; --i) {
        const Block& L = m->E[i];
        float* dc = m->a_eg[i] + (size_t)off * asz(L);
        const int T = bn_bwd_partial(m, L.gamma, L.beta, kLrelu, g, m->ec[i] + (size_t)off * asz(L), n * L.H * L.W, L.C, dc,
                                     m->a_cp[i] + (size_t)cp_rows[i] * 2 * L.C, st);
        cp_rows[i] += T;
        if (i > 0) conv_dgrad(m, L, n, dc, g, st);
        else if (dx_out) conv_dgrad(m, L, n, dc, dx_out, st);
    }
    (void)gn;
}
// parameter gradients of the encoder path over `rows_n` samples (both passes when constrained)
void a_encode_wgrads(uad_gan* m, const float* x_all, int nall, const int* cp_rows, hipStream_t st) {
    if (m->chen) { c_encode_wgrads(m, nall, st); return; }
    const int r = m->cfg.inter_res, zd = m->cfg.zdim;
    if (m->gmv) {
        for (int h = 0; h < 4; ++h)
            uad_launch_sd_wgrad(m->et, m->flat, m->gm_dhv + m->gm_ho[h], m->gm_J, nall, m->flat, m->gm_hd[h], Gr(m, m->gm_hw[h]), Gr(m, m->gm_hb[h]), st);
    } else {
        uad_launch_conv_w(dense_desc(nall, m->flat, zd), m->et, no_xform(), m->a_dzm, no_xform(), Gr(m, m->a_zw), m->wpartial, st);
        uad_launch_colsum(m->a_dzm, nall, zd, Gr(m, m->a_zb), m->colscratch, st);
    }
    uad_launch_conv_w(conv1x1_desc(nall, r, r, m->cenc, m->cmid), m->ea[m->E.size()], no_xform(), m->a_dflat, no_xform(), Gr(m, m->a_cw), m->wpartial, st);
    uad_launch_colsum(m->a_dflat, nall * r * r, m->cmid, Gr(m, m->a_cb), m->colscratch, st);
    for (size_t i = 0; i < m->E.size(); ++i) {
        const Block& L = m->E[i];
        bn_finalize(m, m->a_cp[i], cp_rows[i], L.C, L.gamma, L.beta, L.b, st);
        conv_wgrad(m, L, nall, i == 0 ? x_all : m->ea[i], m->a_eg[i], st);
    }
}
void a_decode(uad_gan* m, const float* z, const float* mask_dec, int n, hipStream_t st) {
    if (m->chen) { c_decode(m, z, n, st); return; }
    const int r = m->cfg.inter_res;
    if (m->gmv) {          // dec_dense on z_sampled [n, dim_z] (dim_z may be 1: a skinny product)
        UadSdArgs a;
        memset(&a, 0, sizeof a);
        a.a = z; a.lda = m->gm_Z; a.B = P(m, m->a_dw); a.sBi = m->flat; a.sBo = 1; a.bias = P(m, m->a_db); a.mask = mask_dec; a.ldm = m->flat;
        a.R = n; a.I = m->gm_Z; a.O = m->flat; a.out = m->gdv; a.ldo = m->flat;
        uad_launch_sd(a, st);
    } else {
        uad_launch_conv_f(dense_desc(n, m->cfg.zdim, m->flat), z, no_xform(), P(m, m->a_dw), m->gdv, epi_bias(P(m, m->a_db), mask_dec), st, nullptr, m->ws);
    }
    uad_launch_conv_f(conv1x1_desc(n, r, r, m->cmid, m->cenc), m->gdv, no_xform(), P(m, m->a_rw), m->gc[0], epi_bias(P(m, m->a_rb)), st, nullptr, m->ws);
    bn_fwd(m, m->a_dbng, m->a_dbnb, 0.0f, m->gc[0], (size_t)n * r * r, m->cenc, m->ga[0], st);
    for (size_t i = 0; i < m->G.size(); ++i) {
        const Block& L = m->G[i];
        convT_fwd(m, L, n, m->ga[i], m->gc[i + 1], st);
        bn_fwd(m, L.gamma, L.beta, kLrelu, m->gc[i + 1], (size_t)n * L.H * L.W, L.C, m->ga[i + 1], st);
    }
    const Block& LL = m->G.back();
    rowdot<0>(m->ga[m->G.size()], P(m, m->g_fw), P(m, m->g_fb), n * LL.H * LL.W, LL.C, m->xg, st);
}
// dxh = d loss / d x_hat; writes every decoder-path gradient and d / d z into dz_out
void a_decode_backward(uad_gan* m, const float* z, const float* mask_dec, const float* dxh, int n, float* dz_out, hipStream_t st, bool pg = true) {
    if (m->chen) { c_decode_backward(m, z, dxh, n, dz_out, pg, st); return; }
    const int r = m->cfg.inter_res;
    const Block& LL = m->G.back();
    const int rows = n * LL.H * LL.W;
    float* g = m->Ga; float* gn = m->Gb;
    {
        const int rpb = (rows + 1023) / 1024, blocks = (rows + rpb - 1) / rpb;
        hipLaunchKernelGGL(gfinal_bwd_kernel, dim3(blocks), dim3(256), 0, st, dxh, m->xg, m->ga[m->G.size()], P(m, m->g_fw), rows, rpb, LL.C, 2, g,
                           pg ? m->finpart : nullptr);
        if (pg) {
            uad_launch_reduce_partials(m->finpart, blocks, LL.C + 1, 1.0f, m->colscratch, st);
            hipMemcpyAsync(Gr(m, m->g_fw), m->colscratch, LL.C * sizeof(float), hipMemcpyDeviceToDevice, st);
            hipMemcpyAsync(Gr(m, m->g_fb), m->colscratch + LL.C, sizeof(float), hipMemcpyDeviceToDevice, st);
        }
    }
    for (int i = (int)m->G.size() - 1; i >= 0; --i) {
        const Block& L = m->G[i];
        const int T = bn_bwd_partial(m, L.gamma, L.beta, kLrelu, g, m->gc[i + 1], n * L.H * L.W, L.C, gn, m->colpart, st);
        if (pg) {
            bn_finalize(m, m->colpart, T, L.C, L.gamma, L.beta, L.b, st);
            convT_wgrad(m, L, n, m->ga[i], gn, st);
        }
        convT_dgrad(m, L, n, gn, g, st);
    }
    const int T = bn_bwd_partial(m, m->a_dbng, m->a_dbnb, 0.0f, g, m->gc[0], n * r * r, m->cenc, gn, m->colpart, st);
    if (pg) bn_finalize(m, m->colpart, T, m->cenc, m->a_dbng, m->a_dbnb, m->a_rb, st);
    const UadConvDesc dc1 = conv1x1_desc(n, r, r, m->cmid, m->cenc), dd = dense_desc(n, m->cfg.zdim, m->flat);
    if (pg) uad_launch_conv_w(dc1, m->gdv, no_xform(), gn, no_xform(), Gr(m, m->a_rw), m->wpartial, st);
    uad_launch_conv_d(dc1, gn, no_xform(), P(m, m->a_rw), m->ddv, epi_bias(nullptr, mask_dec), st, nullptr, m->ws);
    if (m->gmv) {
        if (pg) uad_launch_sd_wgrad(z, m->gm_Z, m->ddv, m->flat, n, m->gm_Z, m->flat, Gr(m, m->a_dw), Gr(m, m->a_db), st);
        UadSdArgs a;
        memset(&a, 0, sizeof a);
        a.a = m->ddv; a.lda = m->flat; a.B = P(m, m->a_dw); a.sBi = 1; a.sBo = m->flat; a.R = n; a.I = m->flat; a.O = m->gm_Z; a.out = dz_out; a.ldo = m->gm_Z;
        uad_launch_sd(a, st);
        return;
    }
    uad_launch_conv_w(dd, z, no_xform(), m->ddv, no_xform(), Gr(m, m->a_dw), m->wpartial, st);
    uad_launch_colsum(m->ddv, n, m->flat, Gr(m, m->a_db), m->colscratch, st);
    uad_launch_conv_d(dd, m->ddv, no_xform(), P(m, m->a_dw), dz_out, epi_bias(nullptr), st, nullptr, m->ws);
}
void a_critic(uad_gan* m, int mode, const float* zf, const float* zr, const float* eps, int n, hipStream_t st) {
    CriticArgs a;
    memset(&a, 0, sizeof a);
    a.zd = m->cfg.zdim; a.h1 = m->a_h1; a.h2 = m->a_h2; a.mode = mode; a.hat_mode = m->chen ? 1 : 0;
    a.W1 = P(m, m->a_w1); a.b1 = P(m, m->a_b1); a.W2 = P(m, m->a_w2); a.b2 = P(m, m->a_b2); a.W3 = P(m, m->a_w3); a.b3 = P(m, m->a_b3);
    a.zf = zf; a.zr = zr; a.eps = eps; a.inv_n = 1.0f / (float)n; a.scale = m->cfg.scale;
    a.d_fake = m->a_crit[0]; a.d_real = m->a_crit[1]; a.pen = m->a_crit[2]; a.slab = m->a_slab; a.dz_fake = m->dzbuf;
    hipLaunchKernelGGL(critic_kernel, dim3(n), dim3(128), 0, st, a);
}


int you_phase(uad_gan* m, const uad_gan_io_t* io, const float* x, int n, int want_backward, bool restore, float tv, float rlr, float* x_upd,
              float* grads_out, hipStream_t st);
int zim_phase(uad_gan* m, const uad_gan_io_t* io, int n, int want_backward, hipStream_t st);      // Zimmerer VAE, defined with the generic-conv helpers below

// ---- dense GMVAE (aae_kind 3) ----
UadGmdArgs gmd_args(uad_gan* m, const uad_gan_io_t* io, float inv) {
    UadGmdArgs a;
    memset(&a, 0, sizeof a);
    a.W = m->gm_W; a.Z = m->gm_Z; a.C = m->gm_C; a.nmax = m->cfg.max_batch; a.c_lambda = m->cfg.c_lambda; a.inv = inv;
    a.hv = m->gm_hv; a.mask_wmu = io->mask_w_mu; a.mask_wls = io->mask_w_ls; a.mask_zmu = io->mask_z; a.e_w = io->eps_w; a.e_z = io->eps;
    a.Wm = P(m, m->gm_mw); a.bm = P(m, m->gm_mb); a.Wl = P(m, m->gm_lw); a.bl = P(m, m->gm_lb); a.var = P(m, m->gm_var);
    a.hvm = m->gm_hvm; a.w_s = m->gm_ws; a.z_s = m->a_zm; a.M = m->gm_M; a.Lq = m->gm_Lq; a.pc = m->gm_pc; a.loss3 = m->gm_loss3;
    a.dhv = m->gm_dhv; a.dM = m->gm_dM; a.dLq = m->gm_dLq;
    return a;
}
void gmv_forward(uad_gan* m, const uad_gan_io_t* io, const float* x, int n, float inv, hipStream_t st) {
    a_encode(m, x, nullptr, n, 0, st);
    uad_launch_gmd_fwd(gmd_args(m, io, inv), n, st);
    a_decode(m, m->a_zm, io->mask_g, n, st);
}
// one sess.run of trainers/GMVAE.py:122-139 (train / validation) or, with restore, of :172-184 (the `grads` fetch + the update)
int gmv_phase(uad_gan* m, const uad_gan_io_t* io, const float* x, int n, int want_backward, bool restore, float tv, float rlr, float* x_upd,
              float* grads_out, hipStream_t st) {
    const int H = m->cfg.height, Q = m->gm_Z * m->gm_C;
    const size_t img = (size_t)n * H * H;
    const float inv = restore ? 1.0f : 1.0f / (float)n;
    refresh_packs(m, st);
    gmv_forward(m, io, x, n, inv, st);
    if (!restore) {
        float* scal = io->scalars ? io->scalars : m->scalars_own;
        reduce_to<2>(m, 0, x, m->xg, img, 1.0f / (float)n, io->l1_map, st);                         // mean_p_loss (:58-60)
        for (int k = 0; k < 3; ++k) reduce_to<0>(m, 1 + k, m->gm_loss3 + (size_t)k * m->cfg.max_batch, nullptr, (size_t)n, 1.0f / (float)n, nullptr, st);
        hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, st, 5, m->raw, 0.0f, scal);
        if (io->reconstruction) HIP_TRY(hipMemcpyAsync(io->reconstruction, m->xg, img * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (io->z_enc) HIP_TRY(hipMemcpyAsync(io->z_enc, m->a_zm, (size_t)n * m->gm_Z * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    if (!want_backward) return UAD_OK;
    const float* dxh;
    if (restore) {
        uad_launch_tv_dxhat(x, m->xg, n, H, H, 1.0f, tv, m->gm_dxhat, st);                          // sign(x_hat - x) - tv * dTV/dr, r = x - x_hat
        dxh = m->gm_dxhat;
    } else {
        hipLaunchKernelGGL(sign_scale_kernel, dim3(blocks256(img)), dim3(256), 0, st, m->xg, x, 1.0f / (float)n, img, m->dxbuf);
        dxh = m->dxbuf;
    }
    a_decode_backward(m, m->a_zm, io->mask_g, dxh, n, m->dzbuf, st, !restore);
    UadGmdArgs ga = gmd_args(m, io, inv);
    ga.dz_dec = m->dzbuf;
    uad_launch_gmd_bwd(ga, n, st);
    int cp_rows[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    a_encode_backward(m, m->gm_dhv, nullptr, n, 0, cp_rows, nullptr, st);
    if (restore) {
        UadConvDesc d0 = m->E[0].d; d0.N = n;
        uad_launch_conv_first_dgrad_restore(d0, m->a_eg[0], P(m, m->E[0].w), m->gm_dxhat, grads_out, x_upd, rlr, st);
        return UAD_OK;
    }
    // p(z|w,c): dense (mu), dense_1 (log sigma inv) on w_sampled; the Variable's gradient is dense_1's bias gradient
    uad_launch_sd_wgrad(m->gm_ws, m->gm_W, m->gm_dM, Q, n, m->gm_W, Q, Gr(m, m->gm_mw), Gr(m, m->gm_mb), st);
    uad_launch_sd_wgrad(m->gm_ws, m->gm_W, m->gm_dLq, Q, n, m->gm_W, Q, Gr(m, m->gm_lw), Gr(m, m->gm_lb), st);
    HIP_TRY(hipMemcpyAsync(Gr(m, m->gm_var), Gr(m, m->gm_lb), (size_t)Q * sizeof(float), hipMemcpyDeviceToDevice, st));
    a_encode_wgrads(m, x, n, cp_rows, st);
    return UAD_OK;
}

int aae_phase(uad_gan* m, int phase, const uad_gan_io_t* io, int n, int want_backward, hipStream_t st) {
    if (m->you) {
        if (phase != UAD_GAN_GENERATOR) return fail(UAD_ERR_INVALID, "the GMVAE has one phase (UAD_GAN_GENERATOR): its optimizer covers every variable");
        if (!io->x) return fail(UAD_ERR_INVALID, "GMVAE phase needs io.x");
        return you_phase(m, io, io->x, n, want_backward, false, 0.f, 0.f, nullptr, nullptr, st);
    }
    if (m->zim) {
        if (phase != UAD_GAN_GENERATOR) return fail(UAD_ERR_INVALID, "the Zimmerer VAE has one phase (UAD_GAN_GENERATOR): its optimizer covers every variable");
        return zim_phase(m, io, n, want_backward, st);
    }
    if (m->gmv) {
        if (phase != UAD_GAN_GENERATOR) return fail(UAD_ERR_INVALID, "the dense GMVAE has one phase (UAD_GAN_GENERATOR): its optimizer covers every variable");
        if (!io->x) return fail(UAD_ERR_INVALID, "GMVAE phase needs io.x");
        return gmv_phase(m, io, io->x, n, want_backward, false, 0.f, 0.f, nullptr, nullptr, st);
    }
    if (!io->x) return fail(UAD_ERR_INVALID, "AAE-family phases need io.x");
    const int H = m->cfg.height, zd = m->cfg.zdim;
    const size_t img = (size_t)n * H * H;
    float* scal = io->scalars ? io->scalars : m->scalars_own;
    refresh_packs(m, st);
    int cp_rows[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (phase == UAD_GAN_GENERATOR) {
        // optim_ae (trainers/ConstrainedAE.py:42-45, AAE.py:54-56, ConstrainedAAE.py:59-62): loss = mean_n(L2_n [+ rho * Rec_z_n]) over every AE variable
        a_encode(m, io->x, io->mask_z, n, 0, st);
        a_decode(m, m->a_zm, io->mask_g, n, st);
        reduce_to<1>(m, 0, io->x, m->xg, img, 1.0f / (float)img, nullptr, st);                   // mean_n L2_n
        reduce_to<2>(m, 2, io->x, m->xg, img, 1.0f / (float)n, io->l1_map, st);                  // reconstructionLoss
        if (m->a_constrained) {
            a_encode(m, m->xg, io->mask_sigma, n, n, st);                                         // z_rec (mask_sigma = its dropout mask)
            reduce_to<1>(m, 1, m->a_zm, m->a_zm + (size_t)n * zd, (size_t)n * zd, 1.0f / (float)((size_t)n * zd), nullptr, st);
        } else {
            HIP_TRY(hipMemsetAsync(m->raw + 1, 0, sizeof(float), st));
        }
        hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, st, 4, m->raw, m->a_constrained ? m->cfg.rho : 0.0f, scal);
        if (want_backward) {
            hipLaunchKernelGGL((diff_scale_kernel<false>), dim3(blocks256(img)), dim3(256), 0, st, m->xg, io->x, 2.0f / (float)img, img, m->dxbuf);
            const float* x_all = io->x;
            int nall = n;
            if (m->a_constrained) {
                // second encoder pass first: d Rec_z / d z_rec = -2 rho (z - z_rec) / (n zd), through the shared layers down to x_hat
                const size_t nz = (size_t)n * zd;
                const float cz = 2.0f * m->cfg.rho / (float)nz;
                hipLaunchKernelGGL((diff_scale_kernel<false>), dim3(blocks256(nz)), dim3(256), 0, st, m->a_zm + nz, m->a_zm, cz, nz, m->dzr);       // cz (z_rec - z)
                a_encode_backward(m, m->dzr, io->mask_sigma, n, n, cp_rows, m->Gx, st);
                hipLaunchKernelGGL(add_kernel, dim3(blocks256(img)), dim3(256), 0, st, m->dxbuf, m->Gx, img);      // + the path through z_rec
                HIP_TRY(hipMemcpyAsync(m->a_xcat, io->x, img * sizeof(float), hipMemcpyDeviceToDevice, st));
                HIP_TRY(hipMemcpyAsync(m->a_xcat + img, m->xg, img * sizeof(float), hipMemcpyDeviceToDevice, st));
                x_all = m->a_xcat; nall = 2 * n;
            }
            a_decode_backward(m, m->a_zm, io->mask_g, m->dxbuf, n, m->dzbuf, st);
            if (m->a_constrained) {
                const size_t nz = (size_t)n * zd;
                hipLaunchKernelGGL((diff_scale_kernel<true>), dim3(blocks256(nz)), dim3(256), 0, st, m->a_zm, m->a_zm + nz, 2.0f * m->cfg.rho / (float)nz, nz, m->dzbuf);   // + 2 rho (z - z_rec)/(n zd)
            }
            a_encode_backward(m, m->dzbuf, io->mask_z, n, 0, cp_rows, nullptr, st);
            a_encode_wgrads(m, x_all, nall, cp_rows, st);
        }
        if (io->reconstruction) HIP_TRY(hipMemcpyAsync(io->reconstruction, m->xg, img * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (io->z_enc) HIP_TRY(hipMemcpyAsync(io->z_enc, m->a_zm, (size_t)n * zd * sizeof(float), hipMemcpyDeviceToDevice, st));
        return UAD_OK;
    }
    if (!m->a_critic) return fail(UAD_ERR_INVALID, "this model has no latent critic: only the autoencoder phase exists");
    a_encode(m, io->x, io->mask_z, n, 0, st);
    if (phase == UAD_GAN_DISCRIMINATOR) {
        // optim_dis (trainers/AAE.py:41-49,64): mean d_ - mean d + mean((||d d_hat/d z_hat|| - 1)^2 scale) over the Discriminator variables
        if (!io->z || !io->alpha) return fail(UAD_ERR_INVALID, "critic phase needs io.z (prior sample) and io.alpha (eps)");
        a_critic(m, 0, m->a_zm, io->z, io->alpha, n, st);
        reduce_to<0>(m, 0, m->a_crit[0], nullptr, (size_t)n, 1.0f / (float)n, nullptr, st);
        reduce_to<0>(m, 1, m->a_crit[1], nullptr, (size_t)n, 1.0f / (float)n, nullptr, st);
        reduce_to<0>(m, 2, m->a_crit[2], nullptr, (size_t)n, 1.0f, nullptr, st);
        hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, st, UAD_GAN_DISCRIMINATOR, m->raw, 0.0f, scal);
        if (want_backward) uad_launch_reduce_partials(m->a_slab, n, (int)m->a_nd, 1.0f, Gr(m, m->a_w1), st);
    } else {
        // optim_gen (:43,65): -mean d_ over the variables whose name contains 'Encoder'
        a_critic(m, 1, m->a_zm, nullptr, nullptr, n, st);
        reduce_to<0>(m, 0, m->a_crit[0], nullptr, (size_t)n, 1.0f / (float)n, nullptr, st);
        hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, st, UAD_GAN_GENERATOR, m->raw, 0.0f, scal);
        if (want_backward) {
            a_encode_backward(m, m->dzbuf, io->mask_z, n, 0, cp_rows, nullptr, st);
            a_encode_wgrads(m, io->x, n, cp_rows, st);
        }
    }
    if (io->z_enc) HIP_TRY(hipMemcpyAsync(io->z_enc, m->a_zm, (size_t)n * zd * sizeof(float), hipMemcpyDeviceToDevice, st));
    return UAD_OK;
}

// ================================================================================================ ResNet variant
// models/fanogan_schlegl.py:119-161.  Every k3 / k1 contraction runs on the generic F / D / W kernels (any KS / S / P).
typedef uad_gan::RB RB;
// Math of the k3 / k1 contractions (round 4).  UAD_MATH_BF16X3: the k3 data-path contractions run on the tap-list bf16x3 spatial kernel
// (uad_convk16.inc) -- EXCEPT for the samples from m->exact_from on (pass A's x_hat third and all of pass B of a critic phase, set by the
// phase code): the penalty (||d D(x_hat) / d x_hat|| - 1)^2 is an ill-conditioned function of that gradient (a 1e-5 relative error of the norm
// is a 1e-4 error of the penalty when the norm is within 0.1 of 1), so the two passes that determine its VALUE stay on the exact-fp32 kernels;
// its gradient is linear in (norm - 1) and takes the bf16x3 passes C / D.  UAD_MATH_BF16X3_ALL: everything in bf16x3 (not parity-rated).
static bool k3_packs(const uad_gan* m, const UadConvDesc& d) { return m->variant == 1 && m->math == UAD_MATH_BF16X3 && (d.KS == 3 || d.KS == 1) && d.CB % 8 == 0 && d.CS % 8 == 0; }
void g_conv_f(uad_gan* m, UadConvDesc d, int N, const float* big_in, long long w, const float* bias, const float* add, float* small_out, hipStream_t st) {
    const bool pk = k3_packs(m, d);
    const long long plane = (long long)d.KS * d.KS * d.CB * d.CS;
    int nfast = (pk && !m->generic16 && m->exact_from >= 0) ? (m->exact_from < N ? m->exact_from : N) : N;      // samples [nfast, N) exact
    // the k1 shortcuts (1/9 of a k3 layer's work) take the fp32-grade products everywhere: with them in bf16x3 as well the critic phase's last
    // LayerNorm gammas measured 1.04e-4 off the oracle (round 4, profiles/README.md) -- at the bar instead of inside it
    if (pk && !m->generic16 && d.KS == 1 && m->wpack3_f) nfast = 0;
    if (nfast > 0) {
        d.N = nfast;
        uad_launch_conv_f(d, big_in, no_xform(), P(m, w), small_out, epi_bias(bias, nullptr, add), st, nullptr, m->ws, pk ? K3F(m, w) : nullptr, plane,
                          m->generic16);
    }
    if (nfast < N) {
        // the exact range: bf16x6 on the same spatial kernel (three planes, six products: fp32-grade) where it takes the shape, else the fp32 MFMA kernels
        const size_t ob = (size_t)nfast * d.HB * d.WB * d.CB, os = (size_t)nfast * d.HS * d.WS * d.CS;
        d.N = N - nfast;
        const bool x6 = m->wpack3_f && uad_conv_k3_takes(d, true);
        uad_launch_conv_f(d, big_in + ob, no_xform(), P(m, w), small_out + os, epi_bias(bias, nullptr, add ? add + os : nullptr), st, nullptr, m->ws,
                          x6 ? (const unsigned short*)m->wpack3_f + 4 * w : nullptr, x6 ? plane : 0, false, x6 ? 3 : 2);
    }
}
void g_conv_d(uad_gan* m, UadConvDesc d, int N, const float* small_in, long long w, const float* bias, const float* add, float* big_out, hipStream_t st) {
    const bool pk = k3_packs(m, d);
    const long long plane = (long long)d.KS * d.KS * d.CB * d.CS;
    int nfast = (pk && !m->generic16 && m->exact_from >= 0) ? (m->exact_from < N ? m->exact_from : N) : N;
    if (pk && !m->generic16 && d.KS == 1 && m->wpack3_d) nfast = 0;
    if (nfast > 0) {
        d.N = nfast;
        uad_launch_conv_d(d, small_in, no_xform(), P(m, w), big_out, epi_bias(bias, nullptr, add), st, nullptr, m->ws, pk ? K3D(m, w) : nullptr, plane,
                          m->generic16);
    }
    if (nfast < N) {
        const size_t ob = (size_t)nfast * d.HB * d.WB * d.CB, os = (size_t)nfast * d.HS * d.WS * d.CS;
        d.N = N - nfast;
        const bool x6 = m->wpack3_d && uad_conv_k3_takes(d, false);
        uad_launch_conv_d(d, small_in + os, no_xform(), P(m, w), big_out + ob, epi_bias(bias, nullptr, add ? add + ob : nullptr), st, nullptr, m->ws,
                          x6 ? (const unsigned short*)m->wpack3_d + 4 * w : nullptr, x6 ? plane : 0, false, x6 ? 3 : 2);
    }
}
void g_conv_w(uad_gan* m, UadConvDesc d, int N, const float* big, const float* small_, long long w, hipStream_t st) {
    d.N = N;
    // bf16x3 mode: the k3 filter gradients run on convk_w16_kernel (leaves of the graph: their round-off does not propagate)
    const bool fast = m->variant == 1 && m->math == UAD_MATH_BF16X3 && d.KS == 3;
    uad_launch_conv_w(d, big, no_xform(), small_, no_xform(), Gr(m, w), m->wpartial, st, false, nullptr, nullptr, m->generic16 || fast);
}
void avgpool_fwd(const float* x, int N, int H, int C, float* y, hipStream_t st) {
    const size_t t4 = (size_t)N * (H / 2) * (H / 2) * C / 4;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(blocks256(t4)), dim3(256), 0, st, x, H, H, C, t4, y);
}
void avgpool_bwd(const float* g, int N, int H, int C, float* dx, hipStream_t st) {
    const size_t t4 = (size_t)N * H * H * C / 4;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(blocks256(t4)), dim3(256), 0, st, g, H, H, C, t4, dx);
}

// forward of the first N samples
void rb_forward(uad_gan* m, RB& B, int N, hipStream_t st, size_t off = 0) {      // off: first sample (the buffers hold several passes)
    const int HW = B.Hin * B.Hin;
    const float* X = B.X + off * B.sx;
    float* H1 = B.H1 + off * B.sx; float* C1 = B.C1 + off * B.sc1; float* H2 = B.H2 + off * B.sc1; float* OUT = B.OUT + off * B.sout;
    ln_fwd(m, X, P(m, B.ln1g), P(m, B.ln1b), 0.0f, N, HW, B.Cin, H1, B.ST1 + off * 2 * B.Cin, st);
    g_conv_f(m, B.d1, N, H1, B.w1, P(m, B.b1), nullptr, C1, st);
    ln_fwd(m, C1, P(m, B.ln2g), P(m, B.ln2b), 0.0f, N, HW, B.Cout, H2, B.ST2 + off * 2 * B.Cout, st);
    const float* add = X;                                     // identity shortcut
    if (B.ws >= 0) {
        if (B.gen) g_conv_d(m, B.ds, N, X, B.ws, P(m, B.bs), nullptr, m->s_sp, st);
        else { g_conv_f(m, B.ds, N, X, B.ws, P(m, B.bs), nullptr, m->s_sct, st); avgpool_fwd(m->s_sct, N, B.Hin, B.Cout, m->s_sp, st); }
        add = m->s_sp;
    }
    if (B.gen) g_conv_d(m, B.d2, N, H2, B.w2, P(m, B.b2), add, OUT, st);
    else g_conv_f(m, B.d2, N, H2, B.w2, P(m, B.b2), add, OUT, st);
}

struct RbBwd {
    int N;
    size_t in_off, out_off;   // sample offsets of the forward tensors read / of the gradient regions read and written
    bool pg;                  // parameter gradients (over N + ntail samples from offset 0)
    int ntail;
    bool store_v;             // pass B: keep d / d(norm output) for the adjoint pass
    int inj_lo;               // >= 0: add the second-order injections to samples [inj_lo, inj_lo + ntail)
    bool lnpart;              // data chain of one pass at out_off that still leaves its LayerNorm parameter partials (slots from out_off)
};
void rb_param_grads(uad_gan* m, RB& B, int M, int Nb, hipStream_t st);
void rb_backward(uad_gan* m, RB& B, const RbBwd& a, hipStream_t st) {
    const int HW = B.Hin * B.Hin, N = a.N;
    const float* dout = B.DOUT + a.out_off * B.sout;
    float* g1 = B.G1 + a.out_off * B.sc1;
    float* dx = B.DX + a.out_off * B.sx;
    if (B.gen) g_conv_f(m, B.d2, N, dout, B.w2, nullptr, nullptr, m->s_ta, st);      // data gradient of the transposed conv
    else g_conv_d(m, B.d2, N, dout, B.w2, nullptr, nullptr, m->s_ta, st);
    LnBwdArgs l;
    memset(&l, 0, sizeof l);
    l.da = m->s_ta; l.c = B.C1 + a.in_off * B.sc1; l.stats = B.ST2 + a.in_off * 2 * B.Cout; l.gamma = P(m, B.ln2g); l.beta = P(m, B.ln2b);
    l.alpha = 0.0f; l.HW = HW; l.C = B.Cout; l.dc = g1; l.v_out = a.store_v ? B.V2 : nullptr; l.gpart = (a.pg || a.lnpart) ? B.LP2 : nullptr; l.slot0 = a.lnpart ? (int)a.out_off : 0;
    if (a.inj_lo >= 0) { l.add = B.INJC1; l.add_lo = a.inj_lo; l.add_hi = a.inj_lo + a.ntail; }
    ln_bwd(m, l, N, st);
    g_conv_d(m, B.d1, N, g1, B.w1, nullptr, nullptr, m->s_tb, st);
    const float* addp = dout;                                  // identity shortcut: d / d x gets d / d out
    if (B.ws >= 0) {
        if (B.gen) g_conv_f(m, B.ds, N, dout, B.ws, nullptr, nullptr, dx, st);
        else {
            float* dsc = B.DSC + a.out_off * B.sc1;
            avgpool_bwd(dout, N, B.Hin, B.Cout, dsc, st);
            g_conv_d(m, B.ds, N, dsc, B.ws, nullptr, nullptr, dx, st);
        }
        addp = dx;
    }
    memset(&l, 0, sizeof l);
    l.da = m->s_tb; l.c = B.X + a.in_off * B.sx; l.stats = B.ST1 + a.in_off * 2 * B.Cin; l.gamma = P(m, B.ln1g); l.beta = P(m, B.ln1b);
    l.alpha = 0.0f; l.HW = HW; l.C = B.Cin; l.dc = dx; l.v_out = a.store_v ? B.V1 : nullptr; l.gpart = (a.pg || a.lnpart) ? B.LP1 : nullptr; l.slot0 = a.lnpart ? (int)a.out_off : 0;
    l.add = addp; l.add_lo = 0; l.add_hi = N;
    if (a.inj_lo >= 0) { l.add2 = B.INJX; l.add2_lo = a.inj_lo; l.add2_hi = a.inj_lo + a.ntail; }
    ln_bwd(m, l, N, st);
    if (!a.pg) return;
    rb_param_grads(m, B, N + a.ntail, N, st);
}
// parameter gradients of a block over the first M samples of its buffers (bias column sums over the first Nb)
void rb_param_grads(uad_gan* m, RB& B, int M, int Nb, hipStream_t st) {
    const int HW = B.Hin * B.Hin, N = Nb;
    uad_launch_reduce_partials(B.LP2, M * (B.Cout / 32), 2 * HW, 1.0f, Gr(m, B.ln2g), st);
    uad_launch_reduce_partials(B.LP1, M * (B.Cin / 32), 2 * HW, 1.0f, Gr(m, B.ln1g), st);
    if (B.gen) g_conv_w(m, B.d2, M, B.DOUT, B.H2, B.w2, st); else g_conv_w(m, B.d2, M, B.H2, B.DOUT, B.w2, st);
    g_conv_w(m, B.d1, M, B.H1, B.G1, B.w1, st);
    uad_launch_colsum(B.DOUT, N * B.Hout * B.Hout, B.Cout, Gr(m, B.b2), m->colscratch, st);
    if (B.ws >= 0) {
        if (B.gen) g_conv_w(m, B.ds, M, B.DOUT, B.X, B.ws, st); else g_conv_w(m, B.ds, M, B.X, B.DSC, B.ws, st);
        // the shortcut's bias sees the same column sums as conv2's (the pooling / stride-2 scatter only redistributes d / d out)
        hipMemcpyAsync(Gr(m, B.bs), Gr(m, B.b2), B.Cout * sizeof(float), hipMemcpyDeviceToDevice, st);
    }
    // conv1's bias feeds a LayerNorm over (H, W): identically zero gradient (stays at its zero initialisation)
    for (long long o : {B.ln2g, B.ln2b, B.ln1g, B.ln1b, B.w2, B.w1, B.b2, B.b1, B.ws, B.bs}) gan_grad_final(m, o);      // (data parallelism: this block's gradients are final)
}
// adjoint of the data-gradient map of a critic block on the x_hat third (pass C): reads the adjoint of d / d x from X's tail,
// leaves the adjoint of d / d out in OUT's tail, the pass-C operands of the filter gradients in H1 / H2's tails, the injections
void rb_adjoint(uad_gan* m, RB& B, int n, hipStream_t st) {
    const int HW = B.Hin * B.Hin;
    const size_t hat = (size_t)2 * n, tail = (size_t)3 * n;
    const float* ubx = B.X + tail * B.sx;
    LnBwd2Args a;
    memset(&a, 0, sizeof a);
    a.q = ubx; a.v = B.V1; a.c = B.X + hat * B.sx; a.stats = B.ST1 + hat * 2 * B.Cin; a.gamma = P(m, B.ln1g); a.beta = P(m, B.ln1b);
    a.alpha = 0.0f; a.HW = HW; a.C = B.Cin; a.ubar = B.H1 + tail * B.sx; a.inj = B.INJX; a.gpart = B.LP1; a.slot0 = 3 * n;
    ln_bwd2(m, a, n, st);
    g_conv_f(m, B.d1, n, B.H1 + tail * B.sx, B.w1, nullptr, nullptr, m->Q, st);
    memset(&a, 0, sizeof a);
    a.q = m->Q; a.v = B.V2; a.c = B.C1 + hat * B.sc1; a.stats = B.ST2 + hat * 2 * B.Cout; a.gamma = P(m, B.ln2g); a.beta = P(m, B.ln2b);
    a.alpha = 0.0f; a.HW = HW; a.C = B.Cout; a.ubar = B.H2 + tail * B.sc1; a.inj = B.INJC1; a.gpart = B.LP2; a.slot0 = 3 * n;
    ln_bwd2(m, a, n, st);
    const float* add = ubx;
    if (B.ws >= 0) {
        g_conv_f(m, B.ds, n, ubx, B.ws, nullptr, nullptr, m->s_sct, st);
        avgpool_fwd(m->s_sct, n, B.Hin, B.Cout, m->s_sp, st);
        add = m->s_sp;
    }
    g_conv_f(m, B.d2, n, B.H2 + tail * B.sc1, B.w2, nullptr, add, B.OUT + tail * B.sout, st);
}

void s_enc_forward(uad_gan* m, const float* x, int n, hipStream_t st) {
    const float* in = x;
    for (size_t i = 0; i < m->E.size(); ++i) {
        conv_fwd(m, m->E[i], n, in, m->ec[i], true, st);
        bn_act_fwd(m, m->E[i], m->ec[i], n, m->ea[i + 1], st);
        in = m->ea[i + 1];
    }
    uad_launch_conv_f(dense_desc(n, m->flat, m->cfg.zdim), in, no_xform(), P(m, m->e_dw), m->zr, epi_bias(P(m, m->e_db)), st, nullptr, m->ws);
    const size_t nz = (size_t)n * m->cfg.zdim;
    hipLaunchKernelGGL(tanh_kernel, dim3(blocks256(nz)), dim3(256), 0, st, m->zr, nz, m->z);
}
void s_enc_backward(uad_gan* m, const float* x, int n, hipStream_t st) {
    const int zd = m->cfg.zdim;
    const size_t nz = (size_t)n * zd;
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(blocks256(nz)), dim3(256), 0, st, m->dzbuf, m->z, (const float*)nullptr, nz, m->dzr);
    const UadConvDesc dd = dense_desc(n, m->flat, zd);
    uad_launch_conv_w(dd, m->ea[m->E.size()], no_xform(), m->dzr, no_xform(), Gr(m, m->e_dw), m->wpartial, st);
    uad_launch_colsum(m->dzr, n, zd, Gr(m, m->e_db), m->colscratch, st);
    gan_grad_final(m, m->e_dw); gan_grad_final(m, m->e_db);
    float* g = m->Ga; float* gn = m->Gb;
    uad_launch_conv_d(dd, m->dzr, no_xform(), P(m, m->e_dw), g, epi_bias(nullptr), st, nullptr, m->ws);
    for (int i = (int)m->E.size() - 1; i >= 0; --i) {
        bn_act_bwd(m, m->E[i], g, m->ec[i], n, gn, st);
        conv_wgrad(m, m->E[i], n, i == 0 ? x : m->ea[i], gn, st);
        for (long long o : {m->E[i].w, m->E[i].b, m->E[i].gamma, m->E[i].beta}) gan_grad_final(m, o);
        if (i > 0) conv_dgrad(m, m->E[i], n, gn, g, st);
    }
}
void s_gen_forward(uad_gan* m, const float* z, int n, hipStream_t st, bool linear = false) {
    const int flatg = m->cfg.inter_res * m->cfg.inter_res * 8 * m->dim;
    uad_launch_conv_f(dense_desc(n, m->cfg.zdim, flatg), z, no_xform(), P(m, m->g_dw), m->s_g0, epi_bias(P(m, m->g_db)), st, nullptr, m->ws);
    for (auto& B : m->GB) rb_forward(m, B, n, st);
    const RB& L = m->GB.back();
    ln_fwd(m, L.OUT, P(m, m->s_glg), P(m, m->s_glb), 0.0f, n, L.Hout * L.Hout, L.Cout, m->s_hf, m->s_stf, st);
    if (linear) rowdot<0>(m->s_hf, P(m, m->g_fw), P(m, m->g_fb), n * L.Hout * L.Hout, L.Cout, m->xg, st);
    else rowdot<2>(m->s_hf, P(m, m->g_fw), P(m, m->g_fb), n * L.Hout * L.Hout, L.Cout, m->xg, st);
}
void s_gen_backward(uad_gan* m, const float* z, const float* dx, int n, bool pg, float* dz_out, hipStream_t st, bool linear = false) {
    RB& L = m->GB.back();
    const int rows = n * L.Hout * L.Hout, HW = L.Hout * L.Hout;
    {
        const int rpb = (rows + 1023) / 1024, blocks = (rows + rpb - 1) / rpb;
        hipLaunchKernelGGL(gfinal_bwd_kernel, dim3(blocks), dim3(256), 0, st, dx, m->xg, m->s_hf, P(m, m->g_fw), rows, rpb, L.Cout, linear ? 2 : 1, m->Ga,
                           pg ? m->finpart : nullptr);
        if (pg) {
            uad_launch_reduce_partials(m->finpart, blocks, L.Cout + 1, 1.0f, m->colscratch, st);
            hipMemcpyAsync(Gr(m, m->g_fw), m->colscratch, L.Cout * sizeof(float), hipMemcpyDeviceToDevice, st);
            hipMemcpyAsync(Gr(m, m->g_fb), m->colscratch + L.Cout, sizeof(float), hipMemcpyDeviceToDevice, st);
        }
    }
    LnBwdArgs l;
    memset(&l, 0, sizeof l);
    l.da = m->Ga; l.c = L.OUT; l.stats = m->s_stf; l.gamma = P(m, m->s_glg); l.beta = P(m, m->s_glb); l.alpha = 0.0f; l.HW = HW; l.C = L.Cout;
    l.dc = L.DOUT; l.gpart = pg ? m->lnpart_g : nullptr;
    ln_bwd(m, l, n, st);
    if (pg) {
        uad_launch_reduce_partials(m->lnpart_g, n * (L.Cout / 32), 2 * HW, 1.0f, Gr(m, m->s_glg), st);
        for (long long o : {m->g_fw, m->g_fb, m->s_glg, m->s_glb}) gan_grad_final(m, o);
    }
    RbBwd a{n, 0, 0, pg, 0, false, -1};
    for (int k = (int)m->GB.size() - 1; k >= 0; --k) rb_backward(m, m->GB[k], a, st);
    const int flatg = m->cfg.inter_res * m->cfg.inter_res * 8 * m->dim;
    const UadConvDesc dd = dense_desc(n, m->cfg.zdim, flatg);
    if (pg) {
        uad_launch_conv_w(dd, z, no_xform(), m->s_dg0, no_xform(), Gr(m, m->g_dw), m->wpartial, st);
        uad_launch_colsum(m->s_dg0, n, flatg, Gr(m, m->g_db), m->colscratch, st);
        gan_grad_final(m, m->g_dw); gan_grad_final(m, m->g_db);
    }
    if (dz_out) uad_launch_conv_d(dd, m->s_dg0, no_xform(), P(m, m->g_dw), dz_out, epi_bias(nullptr), st, nullptr, m->ws);
}
void s_disc_forward(uad_gan* m, int N, bool head, hipStream_t st) {
    UadConvDesc d0 = m->s_d0; d0.N = N;
    uad_launch_conv_first_fwd(d0, m->din, P(m, m->s_d0w), P(m, m->s_d0b), m->s_out0, st);
    for (auto& B : m->DB) rb_forward(m, B, N, st);
    if (head) {
        const RB& L = m->DB.back();
        rowdot<0>(L.OUT, P(m, m->d_hw), P(m, m->d_hb), N * L.Hout * L.Hout, L.Cout, m->Dd, st);
    }
}
// ordinary backward of the first N samples from DB.back().DOUT (see disc_backward)
void s_disc_backward(uad_gan* m, int N, bool pg, int ntail, int inject_lo, float* dx_out, hipStream_t st) {
    RbBwd a{N, 0, 0, pg, ntail, false, inject_lo};
    for (int k = (int)m->DB.size() - 1; k >= 0; --k) rb_backward(m, m->DB[k], a, st);
    UadConvDesc d0 = m->s_d0;
    if (pg) {
        d0.N = N + ntail;
        uad_launch_conv_first_wgrad(d0, m->din, m->s_dout0, Gr(m, m->s_d0w), m->wpartial, st);
        uad_launch_colsum(m->s_dout0, N * d0.HS * d0.WS, d0.CS, Gr(m, m->s_d0b), m->colscratch, st);
        gan_grad_final(m, m->s_d0w); gan_grad_final(m, m->s_d0b);
    }
    if (dx_out) { d0.N = N; uad_launch_conv_first_dgrad_plain(d0, m->s_dout0, P(m, m->s_d0w), dx_out, st); }
}


// ================================================================================================ constrained AAE on residual blocks (aae_kind 7)
// models/constrained_adversarial_autoencoder_Chen.py:11-162 under trainers/ConstrainedAAE.py:44-70.  Encoder = the ResNet graph's critic
// stack (k3 conv + four blocks, DB) + Dense(zDim); decoder = its generator stack (Dense + four blocks, GB) with a linear 1x1 output; the
// phases are the AAE family's (aae_phase): these five functions are its encoder / decoder hooks.  The encoder buffers hold both passes
// ([x ; x_hat], 2n samples); each pass runs its own data-gradient chain, the parameter gradients are ONE launch per tensor over both.
void c_encode(uad_gan* m, const float* x, int n, int off, hipStream_t st) {
    const size_t HW = (size_t)m->cfg.height * m->cfg.height;
    float* din = m->din + (size_t)off * HW;
    (void)hipMemcpyAsync(din, x, (size_t)n * HW * sizeof(float), hipMemcpyDeviceToDevice, st);
    UadConvDesc d0 = m->s_d0; d0.N = n;
    uad_launch_conv_first_fwd(d0, din, P(m, m->s_d0w), P(m, m->s_d0b), m->s_out0 + (size_t)off * HW * m->dim, st);
    for (auto& B : m->DB) rb_forward(m, B, n, st, (size_t)off);
    const RB& L = m->DB.back();
    uad_launch_conv_f(dense_desc(n, m->flat, m->cfg.zdim), L.OUT + (size_t)off * L.sout, no_xform(), P(m, m->e_dw), m->a_zm + (size_t)off * m->cfg.zdim,
                      epi_bias(P(m, m->e_db)), st, nullptr, m->ws);
}
void c_encode_backward(uad_gan* m, const float* dz, int n, int off, float* dx_out, hipStream_t st) {
    const int zd = m->cfg.zdim;
    const size_t HW = (size_t)m->cfg.height * m->cfg.height;
    (void)hipMemcpyAsync(m->a_dzm + (size_t)off * zd, dz, (size_t)n * zd * sizeof(float), hipMemcpyDeviceToDevice, st);
    RB& L = m->DB.back();
    uad_launch_conv_d(dense_desc(n, m->flat, zd), dz, no_xform(), P(m, m->e_dw), L.DOUT + (size_t)off * L.sout, epi_bias(nullptr), st, nullptr, m->ws);
    RbBwd a{n, (size_t)off, (size_t)off, false, 0, false, -1, true};
    for (int k = (int)m->DB.size() - 1; k >= 0; --k) rb_backward(m, m->DB[k], a, st);
    if (dx_out) {
        UadConvDesc d0 = m->s_d0; d0.N = n;
        uad_launch_conv_first_dgrad_plain(d0, m->s_dout0 + (size_t)off * HW * m->dim, P(m, m->s_d0w), dx_out, st);
    }
}
void c_encode_wgrads(uad_gan* m, int nall, hipStream_t st) {
    const int zd = m->cfg.zdim;
    RB& L = m->DB.back();
    uad_launch_conv_w(dense_desc(nall, m->flat, zd), L.OUT, no_xform(), m->a_dzm, no_xform(), Gr(m, m->e_dw), m->wpartial, st);
    uad_launch_colsum(m->a_dzm, nall, zd, Gr(m, m->e_db), m->colscratch, st);
    for (auto& B : m->DB) rb_param_grads(m, B, nall, nall, st);
    UadConvDesc d0 = m->s_d0; d0.N = nall;
    uad_launch_conv_first_wgrad(d0, m->din, m->s_dout0, Gr(m, m->s_d0w), m->wpartial, st);
    uad_launch_colsum(m->s_dout0, nall * d0.HS * d0.WS, d0.CS, Gr(m, m->s_d0b), m->colscratch, st);
}
void c_decode(uad_gan* m, const float* z, int n, hipStream_t st) { s_gen_forward(m, z, n, st, true); }
void c_decode_backward(uad_gan* m, const float* z, const float* dxh, int n, float* dz_out, bool pg, hipStream_t st) {
    s_gen_backward(m, z, dxh, n, pg, dz_out, st, true);
}

// ================================================================================================ Zimmerer VAE (aae_kind 4)
// models/variational_autoencoder_Zimmerer.py:7-32 under trainers/VAE.py:36-42.  Every k4 contraction with >= 4 channels on both sides runs
// on the generic F / D / W kernels; the two single-channel ends (first conv 1 -> 16, final conv 16 -> 1) run on the image-side kernels:
// the final k4 s1 convolution is their "data gradient" relation (big = the 1-channel output, S 1, P 2) with the taps reversed.
constexpr float kZimAlpha = 0.2f;          // tf.nn.leaky_relu default
void z_act_fwd(const float* c, size_t total, float* a, hipStream_t st, float alpha = kZimAlpha) {
    hipLaunchKernelGGL(lrelu_fwd_kernel, dim3(blocks256(total / 4)), dim3(256), 0, st, c, alpha, total / 4, a);
}
void z_act_bwd(const float* da, const float* c, size_t total, float* dc, hipStream_t st, float alpha = kZimAlpha) {
    hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(blocks256(total / 4)), dim3(256), 0, st, da, c, alpha, total / 4, dc);
}
// n samples, the first n_vae of them sampled (z = mu + eps sigma, KL), the rest decode z = mu (the ceVAE's context branch)
void zim_forward(uad_gan* m, const float* xin, const float* eps, int n, int n_vae, hipStream_t st) {
    const int zd = m->cfg.zdim, H = m->cfg.height;
    const float* in = xin;
    for (size_t i = 0; i < m->E.size(); ++i) {
        const Block& L = m->E[i];
        UadConvDesc d = L.d; d.N = n;
        if (i == 0) uad_launch_conv_first_fwd(d, in, P(m, L.w), P(m, L.b), m->ec[i], st);
        else g_conv_f(m, L.d, n, in, L.w, P(m, L.b), nullptr, m->ec[i], st);
        z_act_fwd(m->ec[i], (size_t)n * asz(L), m->ea[i + 1], st);
        in = m->ea[i + 1];
    }
    const UadConvDesc dd = dense_desc(n, m->flat, zd);
    uad_launch_conv_f(dd, in, no_xform(), P(m, m->z_muw), m->v_mu_raw, epi_bias(P(m, m->z_mub)), st, nullptr, m->ws);
    uad_launch_conv_f(dd, in, no_xform(), P(m, m->z_lsw), m->v_ls_raw, epi_bias(P(m, m->z_lsb)), st, nullptr, m->ws);
    uad_launch_reparam_fwd(n, n_vae, zd, m->v_mu_raw, m->v_ls_raw, nullptr, nullptr, nullptr, eps, m->v_mu, m->v_ls, m->v_sigma, m->z, m->v_kl, st);
    uad_launch_conv_f(dense_desc(n, zd, m->flat), m->z, no_xform(), P(m, m->z_dw), m->ga[0], epi_bias(P(m, m->z_db)), st, nullptr, m->ws);
    for (size_t i = 0; i < m->G.size(); ++i) {
        const Block& L = m->G[i];
        g_conv_d(m, L.d, n, m->ga[i], L.w, P(m, L.b), nullptr, m->gc[i + 1], st);
        z_act_fwd(m->gc[i + 1], (size_t)n * asz(L), m->ga[i + 1], st);
    }
    hipLaunchKernelGGL(flip_taps_kernel, dim3(1), dim3(256), 0, st, P(m, m->z_fw), 16, 16, m->z_wflip);
    UadConvDesc df = m->z_fd; df.N = n;
    uad_launch_conv_first_dgrad_plain(df, m->ga[m->G.size()], m->z_wflip, m->xg, st);
    const size_t img = (size_t)n * H * H;
    hipLaunchKernelGGL(add_scalar_kernel, dim3(blocks256(img)), dim3(256), 0, st, m->xg, P(m, m->z_fb), img);
}
// dxh = d loss / d x_hat in m->dxbuf; klw = weight of a sample's KL term; pg: also every parameter gradient; anomaly (optional, [n_vae]):
// |x - x_hat| * |d loss_vae / d x| of the sampled branch (trainers/ceVAE.py:51)
void zim_backward(uad_gan* m, const float* xin, const float* eps, int n, int n_vae, float klw, bool pg, float* anomaly, hipStream_t st) {
    const int zd = m->cfg.zdim, H = m->cfg.height;
    UadConvDesc df = m->z_fd; df.N = n;
    float* g = m->Ga; float* gn = m->Gb;
    // final conv: kernel gradient in the reversed-tap layout, then un-reversed; bias; data gradient = the "forward" of the image-side relation
    if (pg) {
        uad_launch_conv_first_wgrad(df, m->dxbuf, m->ga[m->G.size()], m->z_dwflip, m->wpartial, st);
        hipLaunchKernelGGL(flip_taps_kernel, dim3(1), dim3(256), 0, st, m->z_dwflip, 16, 16, Gr(m, m->z_fw));
        uad_launch_colsum(m->dxbuf, n * H * H, 1, Gr(m, m->z_fb), m->colscratch, st);
    }
    uad_launch_conv_first_fwd(df, m->dxbuf, m->z_wflip, nullptr, g, st);
    for (int i = (int)m->G.size() - 1; i >= 0; --i) {
        const Block& L = m->G[i];
        z_act_bwd(g, m->gc[i + 1], (size_t)n * asz(L), gn, st);                          // gn = d loss / d c
        if (pg) {
            uad_launch_colsum(gn, n * L.H * L.W, L.C, Gr(m, L.b), m->colscratch, st);
            g_conv_w(m, L.d, n, gn, m->ga[i], L.w, st);
        }
        g_conv_f(m, L.d, n, gn, L.w, nullptr, nullptr, g, st);                           // d / d (block input)
    }
    // dec_dense (no activation on its output)
    const UadConvDesc ddec = dense_desc(n, zd, m->flat), dd = dense_desc(n, m->flat, zd);
    if (pg) {
        uad_launch_conv_w(ddec, m->z, no_xform(), g, no_xform(), Gr(m, m->z_dw), m->wpartial, st);
        uad_launch_colsum(g, n, m->flat, Gr(m, m->z_db), m->colscratch, st);
    }
    uad_launch_conv_d(ddec, g, no_xform(), P(m, m->z_dw), m->dzbuf, epi_bias(nullptr), st, nullptr, m->ws);
    uad_launch_reparam_bwd(n, n_vae, zd, m->dzbuf, m->v_mu, m->v_sigma, eps, nullptr, nullptr, nullptr, klw, m->v_dmu, m->v_dls, st);
    const float* flat = m->ea[m->E.size()];
    if (pg) {
        uad_launch_conv_w(dd, flat, no_xform(), m->v_dmu, no_xform(), Gr(m, m->z_muw), m->wpartial, st);
        uad_launch_colsum(m->v_dmu, n, zd, Gr(m, m->z_mub), m->colscratch, st);
        uad_launch_conv_w(dd, flat, no_xform(), m->v_dls, no_xform(), Gr(m, m->z_lsw), m->wpartial, st);
        uad_launch_colsum(m->v_dls, n, zd, Gr(m, m->z_lsb), m->colscratch, st);
    }
    uad_launch_conv_d(dd, m->v_dmu, no_xform(), P(m, m->z_muw), gn, epi_bias(nullptr), st, nullptr, m->ws);
    uad_launch_conv_d(dd, m->v_dls, no_xform(), P(m, m->z_lsw), g, epi_bias(nullptr, nullptr, gn), st, nullptr, m->ws);   // g = d loss / d a_4
    for (int i = (int)m->E.size() - 1; i >= 0; --i) {
        const Block& L = m->E[i];
        z_act_bwd(g, m->ec[i], (size_t)n * asz(L), gn, st);
        if (pg) uad_launch_colsum(gn, n * L.H * L.W, L.C, Gr(m, L.b), m->colscratch, st);
        if (i == 0) {
            UadConvDesc d0 = L.d; d0.N = n;
            if (pg) uad_launch_conv_first_wgrad(d0, xin, gn, Gr(m, L.w), m->wpartial, st);
            if (anomaly) {
                d0.N = n_vae;
                uad_launch_conv_first_dgrad(d0, gn, P(m, L.w), xin, m->xg, 1.0f / (float)n_vae, anomaly, nullptr, st);
            }
        } else {
            if (pg) g_conv_w(m, L.d, n, m->ea[i], gn, L.w, st);
            g_conv_d(m, L.d, n, gn, L.w, nullptr, nullptr, g, st);
        }
    }
}
int zim_phase(uad_gan* m, const uad_gan_io_t* io, int n, int want_backward, hipStream_t st) {
    if (!io->x) return fail(UAD_ERR_INVALID, "Zimmerer VAE phase needs io.x");
    const int H = m->cfg.height;
    const size_t img = (size_t)n * H * H;
    float* scal = io->scalars ? io->scalars : m->scalars_own;
    if (m->zim_ce) {
        // trainers/ceVAE.py:38-51 on models/context_encoder_variational_autoencoder_Zimmerer.py: [x ; x_ce] as one 2n-sample pass through the
        // shared layers; the context rows decode z = mu and carry no KL term; each branch is scored against ITS OWN input (:39)
        HIP_TRY(hipMemcpyAsync(m->a_xcat, io->x, img * sizeof(float), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(m->a_xcat + img, io->x_ce ? io->x_ce : io->x, img * sizeof(float), hipMemcpyDeviceToDevice, st));
        zim_forward(m, m->a_xcat, io->eps, 2 * n, n, st);
        reduce_to<2>(m, 0, m->a_xcat, m->xg, img, 1.0f / (float)n, m->z_l1, st);                         // Rec_vae
        reduce_to<2>(m, 1, m->a_xcat + img, m->xg + img, img, 1.0f / (float)n, m->z_l1 + img, st);       // Rec_ce
        reduce_to<0>(m, 2, m->v_kl, nullptr, (size_t)n, 1.0f / (float)n, nullptr, st);                   // kl
        hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, st, 6, m->raw, 0.0f, scal);
        if (want_backward) {
            hipLaunchKernelGGL(sign_scale_kernel, dim3(blocks256(2 * img)), dim3(256), 0, st, m->xg, m->a_xcat, 1.0f / (float)n, 2 * img, m->dxbuf);
            zim_backward(m, m->a_xcat, io->eps, 2 * n, n, 1.0f / (float)n, want_backward == 1, io->anomaly, st);
        }
        if (io->reconstruction) HIP_TRY(hipMemcpyAsync(io->reconstruction, m->xg, img * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (io->generated) HIP_TRY(hipMemcpyAsync(io->generated, m->xg + img, img * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (io->l1_map) HIP_TRY(hipMemcpyAsync(io->l1_map, m->z_l1, img * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (io->l1_map_ce) HIP_TRY(hipMemcpyAsync(io->l1_map_ce, m->z_l1 + img, img * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (io->z_enc) HIP_TRY(hipMemcpyAsync(io->z_enc, m->z, (size_t)n * m->cfg.zdim * sizeof(float), hipMemcpyDeviceToDevice, st));
        return UAD_OK;
    }
    zim_forward(m, io->x, io->eps, n, n, st);
    reduce_to<2>(m, 0, io->x, m->xg, img, 1.0f / (float)n, io->l1_map, st);                       // reconstructionLoss (trainers/VAE.py:36-38)
    reduce_to<0>(m, 1, m->v_kl, nullptr, (size_t)n, 1.0f / (float)n, nullptr, st);                 // kl (:39-41)
    hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, st, 3, m->raw, 1.0f, scal);
    if (want_backward) {
        hipLaunchKernelGGL(sign_scale_kernel, dim3(blocks256(img)), dim3(256), 0, st, m->xg, io->x, 1.0f / (float)n, img, m->dxbuf);
        zim_backward(m, io->x, io->eps, n, n, 1.0f / (float)n, true, nullptr, st);
    }
    if (io->reconstruction) HIP_TRY(hipMemcpyAsync(io->reconstruction, m->xg, img * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (io->z_enc) HIP_TRY(hipMemcpyAsync(io->z_enc, m->z, (size_t)n * m->cfg.zdim * sizeof(float), hipMemcpyDeviceToDevice, st));
    return UAD_OK;
}

// ================================================================================================ spatial GMVAE, original architecture (aae_kind 6)
// models/gaussian_mixture_variational_autoencoder_You.py:8-85 under trainers/GMVAE_spatial.py:61-97,168-199.  k3 layers on the generic F / D / W
// kernels (SAME: P 0 at stride 2, P 1 at stride 1), the single-channel ends on the image-side kernels (y_mu as their data-gradient relation with
// reversed taps), the latent heads on the spatial GMVAE's per-location kernels (uad_gmvae.hip) with z_sampled handed to / from the decoder.
typedef uad_gan::YOp YOp;
size_t yop_out(const YOp& o) { return (size_t)o.Hout * o.Hout * o.Cout; }
UadGmArgs you_gm_args(uad_gan* m, const uad_gan_io_t* io, float inv) {
    UadGmArgs a;
    memset(&a, 0, sizeof a);
    a.cenc = 64; a.W = m->gm_W; a.Z = m->gm_Z; a.C = m->gm_C; a.c_lambda = m->cfg.c_lambda; a.inv_batch = inv;
    a.c_enc = m->y_enc.back().c; a.scale = m->y_ones; a.shift = m->y_zeros; a.alpha = 0.0f; a.mult = 1.0f;      // h = relu(conv + bias)
    const float** slots[15] = {&a.wmu_k, &a.wmu_b, &a.wls_k, &a.wls_b, &a.zmu_k, &a.zmu_b, &a.zls_k, &a.zls_b, &a.c7_k, &a.c7_b, &a.m_k, &a.m_b, &a.l_k,
                               &a.l_b, &a.var};
    for (int k = 0; k < 15; ++k) *slots[k] = P(m, m->y_off[k]);
    a.eps_w = io->eps_w; a.eps_z = io->eps;
    a.h_out = m->y_h; a.loc_loss = m->y_loc; a.zs_out = m->a_zm;
    return a;
}
void you_forward(uad_gan* m, const uad_gan_io_t* io, const float* x, int n, float inv, hipStream_t st) {
    const int H = m->cfg.height, r = H / 4;
    const float* in = x;
    for (size_t i = 0; i < m->y_enc.size(); ++i) {
        YOp& o = m->y_enc[i];
        UadConvDesc d = o.L.d; d.N = n;
        if (d.CB % 4) uad_launch_conv_first_fwd(d, in, P(m, o.L.w), P(m, o.L.b), o.c, st);
        else g_conv_f(m, o.L.d, n, in, o.L.w, P(m, o.L.b), nullptr, o.c, st);
        if (i + 1 < m->y_enc.size()) { z_act_fwd(o.c, (size_t)n * yop_out(o), o.a, st, 0.0f); in = o.a; }
    }
    uad_launch_gm_heads_fwd(you_gm_args(m, io, inv), n * r * r, st);
    in = m->a_zm;
    int hin = r, cin = m->gm_Z;
    for (size_t i = 0; i + 1 < m->y_dec.size(); ++i) {
        YOp& o = m->y_dec[i];
        if (o.kind == 2) {
            const size_t t4 = (size_t)n * o.Hout * o.Hout * cin / 4;
            hipLaunchKernelGGL(up2_fwd_kernel, dim3(blocks256(t4)), dim3(256), 0, st, in, hin, hin, cin, t4, o.a);
        } else {
            UadConvDesc d = o.L.d; d.N = n;
            if (o.kind == 0 && d.CB % 4) uad_launch_conv_first_fwd(d, in, P(m, o.L.w), P(m, o.L.b), o.c, st);
            else if (o.kind == 0) g_conv_f(m, o.L.d, n, in, o.L.w, P(m, o.L.b), nullptr, o.c, st);
            else g_conv_d(m, o.L.d, n, in, o.L.w, P(m, o.L.b), nullptr, o.c, st);
            if (o.relu) z_act_fwd(o.c, (size_t)n * yop_out(o), o.a, st, 0.0f);
        }
        in = o.a; hin = o.Hout; cin = o.Cout;
    }
    // y_mu
    const YOp& F = m->y_dec.back();
    hipLaunchKernelGGL(flip_taps_kernel, dim3(3), dim3(256), 0, st, P(m, F.L.w), 9, 64, m->z_wflip);
    UadConvDesc df = m->y_fd; df.N = n;
    uad_launch_conv_first_dgrad_plain(df, in, m->z_wflip, m->xg, st);
    const size_t img = (size_t)n * H * H;
    hipLaunchKernelGGL(add_scalar_kernel, dim3(blocks256(img)), dim3(256), 0, st, m->xg, P(m, F.L.b), img);
}
// dxh = d objective / d x_hat; leaves d objective / d c of the FIRST encoder conv in *g_first (for the caller's input gradient)
void you_backward(uad_gan* m, const uad_gan_io_t* io, const float* x, const float* dxh, int n, float inv, bool pg, float** g_first, hipStream_t st) {
    const int H = m->cfg.height, r = H / 4, L = n * r * r;
    float* g = m->Ga; float* gn = m->Gb;
    const YOp& F = m->y_dec.back();
    const size_t nd = m->y_dec.size();
    const float* fin = m->y_dec[nd - 2].a;
    UadConvDesc df = m->y_fd; df.N = n;
    if (pg) {
        uad_launch_conv_first_wgrad(df, dxh, fin, m->z_dwflip, m->wpartial, st);
        hipLaunchKernelGGL(flip_taps_kernel, dim3(3), dim3(256), 0, st, m->z_dwflip, 9, 64, Gr(m, F.L.w));
        uad_launch_colsum(dxh, n * H * H, 1, Gr(m, F.L.b), m->colscratch, st);
    }
    uad_launch_conv_first_fwd(df, dxh, m->z_wflip, nullptr, g, st);
    for (int i = (int)nd - 2; i >= 0; --i) {
        YOp& o = m->y_dec[i];
        const float* in = i == 0 ? m->a_zm : m->y_dec[i - 1].a;
        const int hin = i == 0 ? r : m->y_dec[i - 1].Hout, cin = i == 0 ? m->gm_Z : m->y_dec[i - 1].Cout;
        if (o.kind == 2) {
            const size_t t4 = (size_t)n * hin * hin * cin / 4;
            hipLaunchKernelGGL(up2_bwd_kernel, dim3(blocks256(t4)), dim3(256), 0, st, g, hin, hin, cin, t4, gn);
            float* t = g; g = gn; gn = t;
            continue;
        }
        const float* dc = g;
        if (o.relu) { z_act_bwd(g, o.c, (size_t)n * yop_out(o), gn, st, 0.0f); dc = gn; }
        if (pg) uad_launch_colsum(dc, n * o.Hout * o.Hout, o.Cout, Gr(m, o.L.b), m->colscratch, st);
        float* dst = (dc == g) ? gn : g;
        UadConvDesc d = o.L.d; d.N = n;
        if (o.kind == 0 && d.CB % 4) {            // first decoder conv on a 1-channel z_sampled
            if (pg) uad_launch_conv_first_wgrad(d, in, dc, Gr(m, o.L.w), m->wpartial, st);
            uad_launch_conv_first_dgrad_plain(d, dc, P(m, o.L.w), m->y_dzdec, st);
        } else if (o.kind == 0) {
            if (pg) g_conv_w(m, o.L.d, n, in, dc, o.L.w, st);
            g_conv_d(m, o.L.d, n, dc, o.L.w, nullptr, nullptr, i == 0 ? m->y_dzdec : dst, st);
        } else {
            if (pg) g_conv_w(m, o.L.d, n, dc, in, o.L.w, st);
            g_conv_f(m, o.L.d, n, dc, o.L.w, nullptr, nullptr, dst, st);
        }
        if (dst != g) { float* t = g; g = gn; gn = t; }
    }
    // latent heads: per-location backward (recomputes its forward), z_sampled's decoder gradient added in
    UadGmArgs ga = you_gm_args(m, io, inv);
    ga.h_out = nullptr; ga.zs_out = nullptr;
    ga.dz_dec = m->y_dzdec; ga.g_out = g; ga.colpart = m->y_colpart;
    ga.dvec_heads = m->y_dheads; ga.dvec_a7 = m->y_da7; ga.dvec_M = m->y_dM; ga.dvec_Lq = m->y_dLq; ga.ws_out = m->y_ws; ga.mid_out = m->y_mid;
    uad_launch_gm_heads_bwd(ga, L, st);
    if (pg) {
        const int W = ga.W, Z = ga.Z, Q = ga.Z * ga.C, O = 2 * W + 2 * Z, CE = 64;
        UadGmWgradArgs wa;
        memset(&wa, 0, sizeof wa);
        const long long base = m->y_off[0];
        auto job = [&](int k, const float* A, int lda, const float* B, int ldb, int b) {
            wa.job[k] = UadGmWgradArgs::Job{A, lda, B, ldb, b, (int)(m->y_off[k] - base)};
        };
        const float* dh = m->y_dheads;
        job(0, m->y_h, CE, dh, O, W);              job(1, nullptr, 0, dh, O, W);
        job(2, m->y_h, CE, dh + W, O, W);          job(3, nullptr, 0, dh + W, O, W);
        job(4, m->y_h, CE, dh + 2 * W, O, Z);      job(5, nullptr, 0, dh + 2 * W, O, Z);
        job(6, m->y_h, CE, dh + 2 * W + Z, O, Z);  job(7, nullptr, 0, dh + 2 * W + Z, O, Z);
        job(8, m->y_ws, W, m->y_da7, 64, 64);      job(9, nullptr, 0, m->y_da7, 64, 64);
        job(10, m->y_mid, 64, m->y_dM, Q, Q);      job(11, nullptr, 0, m->y_dM, Q, Q);
        job(12, m->y_mid, 64, m->y_dLq, Q, Q);     job(13, nullptr, 0, m->y_dLq, Q, Q);
        job(14, nullptr, 0, m->y_dLq, Q, Q);
        wa.njobs = 15; wa.total = (int)m->y_total; wa.L = L; wa.partial = m->y_partial;
        uad_launch_gm_heads_wgrad(wa, Gr(m, base), st);
    }
    // encoder: g = d objective / d c of the last conv
    for (int i = (int)m->y_enc.size() - 1; i >= 0; --i) {
        YOp& o = m->y_enc[i];
        const float* in = i == 0 ? x : m->y_enc[i - 1].a;
        if (pg) uad_launch_colsum(g, n * o.Hout * o.Hout, o.Cout, Gr(m, o.L.b), m->colscratch, st);
        UadConvDesc d = o.L.d; d.N = n;
        if (i == 0) {
            if (pg) uad_launch_conv_first_wgrad(d, in, g, Gr(m, o.L.w), m->wpartial, st);
            break;
        }
        if (pg) g_conv_w(m, o.L.d, n, in, g, o.L.w, st);
        g_conv_d(m, o.L.d, n, g, o.L.w, nullptr, nullptr, gn, st);
        z_act_bwd(gn, m->y_enc[i - 1].c, (size_t)n * yop_out(m->y_enc[i - 1]), g, st, 0.0f);
    }
    *g_first = g;
}
int you_phase(uad_gan* m, const uad_gan_io_t* io, const float* x, int n, int want_backward, bool restore, float tv, float rlr, float* x_upd,
              float* grads_out, hipStream_t st) {
    const int H = m->cfg.height, r = H / 4, L = n * r * r;
    const size_t img = (size_t)n * H * H;
    const float inv = restore ? 1.0f : 1.0f / (float)n;
    you_forward(m, io, x, n, inv, st);
    if (!restore) {
        float* scal = io->scalars ? io->scalars : m->scalars_own;
        reduce_to<2>(m, 0, x, m->xg, img, 1.0f / (float)n, io->l1_map, st);
        uad_launch_colsum(m->y_loc, L, 3, m->raw + 1, m->colscratch, st);            // sums over locations of con, w-prior, c-prior
        hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, st, 7, m->raw, 1.0f / (float)n, scal);
        if (io->reconstruction) HIP_TRY(hipMemcpyAsync(io->reconstruction, m->xg, img * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (io->z_enc) HIP_TRY(hipMemcpyAsync(io->z_enc, m->a_zm, (size_t)L * m->gm_Z * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    if (!want_backward) return UAD_OK;
    const float* dxh;
    if (restore) { uad_launch_tv_dxhat(x, m->xg, n, H, H, 1.0f, tv, m->gm_dxhat, st); dxh = m->gm_dxhat; }
    else { hipLaunchKernelGGL(sign_scale_kernel, dim3(blocks256(img)), dim3(256), 0, st, m->xg, x, 1.0f / (float)n, img, m->dxbuf); dxh = m->dxbuf; }
    float* g0 = nullptr;
    you_backward(m, io, x, dxh, n, inv, !restore, &g0, st);
    if (restore) {
        UadConvDesc d0 = m->y_enc[0].L.d; d0.N = n;
        uad_launch_conv_first_dgrad_restore(d0, g0, P(m, m->y_enc[0].L.w), m->gm_dxhat, grads_out, x_upd, rlr, st);
    }
    return UAD_OK;
}

}  // namespace

extern "C" {

#include "uad_gan_create.inc"

int uad_gan_destroy(uad_gan_t* m) {
    if (!m) return UAD_OK;
    for (void* p : m->allocs) hipFree(p);
    delete m;
    return UAD_OK;
}
long long uad_gan_param_count(const uad_gan_t* m) { return m ? m->nparams : 0; }
int uad_gan_num_tensors(const uad_gan_t* m) { return m ? (int)m->tensors.size() : 0; }
int uad_gan_tensor_info(const uad_gan_t* m, int idx, char* name, int name_cap, long long* offset, int* rank, int* shape4) {
    if (!m || idx < 0 || idx >= (int)m->tensors.size()) return fail(UAD_ERR_INVALID, "tensor index out of range");
    const Tensor& t = m->tensors[idx];
    if (name && name_cap > 0) { strncpy(name, t.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (offset) *offset = t.off;
    if (rank) *rank = t.rank;
    if (shape4) for (int i = 0; i < 4; ++i) shape4[i] = t.shape[i];
    return UAD_OK;
}
float* uad_gan_buffer(uad_gan_t* m, int which) {
    if (!m) return nullptr;
    switch (which) {
        case UAD_BUF_PARAMS: m->packed_valid = false; m->pack_all = true; return m->params;      // the caller may write through the pointer (DP broadcast): repack before the next phase
        case UAD_BUF_GRADS: return m->grads;
        case UAD_BUF_ADAM_M: return m->adam_m;
        case UAD_BUF_ADAM_V: return m->adam_v;
        case UAD_BUF_ADAM_M2: return m->adam_m2;
        case UAD_BUF_ADAM_V2: return m->adam_v2;
    }
    return nullptr;
}
int uad_gan_group(const uad_gan_t* m, int group, long long* offset, long long* count) {
    if (m && group == UAD_GAN_GROUP_VAE) {       // Encoder | Generator are adjacent in the flat buffers
        if (offset) *offset = 0;
        if (count) *count = m->grp_cnt[UAD_GAN_ENCODER] + m->grp_cnt[UAD_GAN_GENERATOR];
        return UAD_OK;
    }
    if (!m || group < 0 || group > 2) return fail(UAD_ERR_INVALID, "bad group");
    if (offset) *offset = m->grp_off[group];
    if (count) *count = m->grp_cnt[group];
    return UAD_OK;
}
int uad_gan_set_buffer(uad_gan_t* m, int which, const float* host, long long count) {
    float* p = uad_gan_buffer(m, which);
    if (!p || !host || count != m->nparams) return fail(UAD_ERR_INVALID, "gan set_buffer: bad arguments (count=%lld, expected %lld)", count, m ? m->nparams : -1);
    HIP_TRY(hipMemcpy(p, host, (size_t)count * sizeof(float), hipMemcpyHostToDevice));
    if (which == UAD_BUF_PARAMS) { m->packed_valid = false; m->pack_all = true; }
    return UAD_OK;
}
int uad_gan_get_buffer(uad_gan_t* m, int which, float* host, long long count) {
    float* p = uad_gan_buffer(m, which);
    if (!p || !host || count != m->nparams) return fail(UAD_ERR_INVALID, "gan get_buffer: bad arguments");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host, p, (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
    return UAD_OK;
}
int uad_gan_set_math_mode(uad_gan_t* m, int mode) {
    if (!m || (mode != UAD_MATH_F32 && mode != UAD_MATH_BF16X3 && mode != UAD_MATH_BF16X3_ALL)) return fail(UAD_ERR_INVALID, "bad math mode");
    m->math = mode == UAD_MATH_F32 ? UAD_MATH_F32 : UAD_MATH_BF16X3; m->generic16 = mode == UAD_MATH_BF16X3_ALL; m->packed_valid = false; m->pack_all = true;
    return UAD_OK;
}
long long uad_gan_get_step(const uad_gan_t* m, int group) { return (m && group >= 0 && group < 3) ? m->step[group] : 0; }
int uad_gan_set_step(uad_gan_t* m, int group, long long t) {
    if (!m || group < 0 || group > 2 || t < 0) return fail(UAD_ERR_INVALID, "bad group / step");
    m->step[group] = t;
    return UAD_OK;
}
int uad_gan_debug_buffer(uad_gan_t* m, const char* name, float** ptr, long long* count) {
    if (!m || !name) return fail(UAD_ERR_INVALID, "null argument");
    auto it = m->dbg.find(name);
    if (it == m->dbg.end()) return fail(UAD_ERR_INVALID, "no debug buffer named %s", name);
    if (ptr) *ptr = it->second.first;
    if (count) *count = it->second.second;
    return UAD_OK;
}

static int gan_phase_body(uad_gan_t* m, int phase, const uad_gan_io_t* io, int n, int want_backward, void* stream) {
    if (!m || !io) return fail(UAD_ERR_INVALID, "null argument");
    if (n <= 0 || n > m->cfg.max_batch) return fail(UAD_ERR_INVALID, "batch %d outside (0, max_batch=%d]", n, m->cfg.max_batch);
    if (phase < 0 || phase > 2) return fail(UAD_ERR_INVALID, "bad phase");
    hipStream_t st = (hipStream_t)stream;
    if (m->variant == UAD_GAN_AAE) {
        const int rc = aae_phase(m, phase, io, n, want_backward, st);
        if (rc != UAD_OK) return rc;
        HIP_TRY(hipGetLastError());
        return UAD_OK;
    }
    const bool rn = m->variant == UAD_GAN_RESNET;
    const bool av = m->variant == UAD_GAN_ANOVAEGAN;      // the generator's input is the encoder's z_vae, never io->z
    if (av && !io->x) return fail(UAD_ERR_INVALID, "AnoVAE-GAN phases need io.x");
    const int H = m->cfg.height, ir = m->cfg.inter_res, L = m->npool;
    const size_t HW = (size_t)H * H, img = (size_t)n * HW;
    const int P2 = ir * ir;                        // feature-map locations per sample
    const int FC = rn ? m->DB.back().Cout : m->D.back().C;                 // feature channels
    float* feat = rn ? m->DB.back().OUT : m->Da[L];                        // critic features, 4n layout
    float* top = rn ? m->DB.back().DOUT : m->Ga;                           // d loss / d features
    float* scal = io->scalars ? io->scalars : m->scalars_own;
    refresh_packs(m, st);
    auto gen_fwd = [&](const float* z) {
        if (rn) s_gen_forward(m, z, n, st);
        else if (av) { v_enc_forward(m, io, n, st); gen_forward(m, m->z, nullptr, n, st); }
        else gen_forward(m, z, io->mask_g, n, st);
    };
    auto gen_bwd = [&](const float* z, bool pg, float* dz) {
        if (rn) s_gen_backward(m, z, m->dxbuf, n, pg, dz, st);
        else gen_backward(m, av ? m->z : z, av ? nullptr : io->mask_g, m->dxbuf, n, pg, dz, st);
    };
    auto disc_fwd = [&](int N, bool head) { if (rn) s_disc_forward(m, N, head, st); else disc_forward(m, N, head, st); };
    auto disc_bwd = [&](int N, bool pg, int ntail, int inj_lo, float* dx) { if (rn) s_disc_backward(m, N, pg, ntail, inj_lo, dx, st); else disc_backward(m, N, pg, ntail, inj_lo, dx, st); };
    // ResNet graph, bf16x3 mode: the generator and encoder phases (7 of the ~69 n sample-passes of a WGAN iteration + the encoder stage) keep
    // their k3 DATA path on the exact-fp32 kernels (g_conv_f / _d).  Measured in round 4 (profiles/README.md): with the generator's 8 + 8
    // contractions in bf16x3 the last gradients of those chains (the generator's first LayerNorm gamma, the encoder's first kernel) sit at
    // 1.00-1.20e-4 of their tensor's max off the oracle -- at the parity bar, not inside it -- whether or not the critic between them is exact.
    // The critic phases (5 x 12 n sample-passes: pass A's x / x_ thirds, C, D and every filter gradient) hold 1e-4 in bf16x3 and carry the speed-up.
    // (scope guard: EVERY way out of this function -- an early `return fail(...)`, a HIP_TRY -- leaves exact_from at -1; a stale value would silently
    // route a later phase's or reconstruct()'s samples to the exact kernels.  The critic phase moves the mark by hand inside the guard's scope.)
    struct ExactPhase { uad_gan* m; ExactPhase(uad_gan* m_, bool on) : m(m_) { m->exact_from = on ? 0 : -1; } ~ExactPhase() { m->exact_from = -1; } };
    ExactPhase exact_phase(m, rn && phase != UAD_GAN_DISCRIMINATOR);

    if (phase == UAD_GAN_GENERATOR) {
        // trainers/fAnoGAN.py:52,75: gen_loss = -mean(D(G(z))), gradient w.r.t. the Generator variables
        if (!av && !io->z) return fail(UAD_ERR_INVALID, "generator phase needs io.z");
        gen_fwd(io->z);
        HIP_TRY(hipMemcpyAsync(m->din, m->xg, img * sizeof(float), hipMemcpyDeviceToDevice, st));
        disc_fwd(n, true);
        reduce_to<0>(m, 0, m->Dd, nullptr, (size_t)n * P2, 1.0f / (float)(n * P2), nullptr, st);
        hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, st, phase, m->raw, m->cfg.kappa, scal);
        if (want_backward) {
            Coef4 cf{{-1.0f / (float)(n * P2), 0.f, 0.f, 0.f}};
            const size_t t4 = (size_t)n * P2 * FC / 4;
            hipLaunchKernelGGL(topgrad_kernel, dim3(blocks256(t4)), dim3(256), 0, st, P(m, m->d_hw), n * P2, FC, t4, cf, top);
            disc_bwd(n, false, 0, -1, m->dxbuf);
            gen_bwd(io->z, true, nullptr);
        }
        if (io->generated) HIP_TRY(hipMemcpyAsync(io->generated, m->xg, img * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else if (phase == UAD_GAN_DISCRIMINATOR) {
        // trainers/fAnoGAN.py:50-58,74
        if ((!av && !io->z) || !io->x || !io->alpha) return fail(UAD_ERR_INVALID, "critic phase needs io.x, io.z and io.alpha");
        gen_fwd(io->z);
        hipLaunchKernelGGL(interp_kernel, dim3(blocks256(img)), dim3(256), 0, st, m->xg, io->x, io->alpha, (int)HW, img, m->din);
        m->exact_from = 2 * n;                     // the x_hat third of pass A and all of pass B fix the penalty's VALUE: exact fp32 (see g_conv_f)
        disc_fwd(3 * n, true);                                                                    // pass A
        m->exact_from = 0;
        reduce_to<0>(m, 0, m->Dd, nullptr, (size_t)n * P2, 1.0f / (float)(n * P2), nullptr, st);
        reduce_to<0>(m, 1, m->Dd + (size_t)n * P2, nullptr, (size_t)n * P2, 1.0f / (float)(n * P2), nullptr, st);
        // pass B: ddx = d sum(d_hat) / d x_hat on samples [2n, 3n)
        {
            Coef4 one{{1.f, 1.f, 1.f, 1.f}};
            const size_t t4 = (size_t)n * P2 * FC / 4;
            if (rn) {
                const RB& LB = m->DB.back();
                hipLaunchKernelGGL(topgrad_kernel, dim3(blocks256(t4)), dim3(256), 0, st, P(m, m->d_hw), n * P2, FC, t4, one, LB.DOUT + (size_t)3 * n * LB.sout);
                RbBwd a{n, (size_t)2 * n, (size_t)3 * n, false, 0, true, -1};
                for (int k = (int)m->DB.size() - 1; k >= 0; --k) rb_backward(m, m->DB[k], a, st);
                UadConvDesc d0 = m->s_d0; d0.N = n;
                uad_launch_conv_first_dgrad_plain(d0, m->s_dout0 + (size_t)3 * n * HW * m->dim, P(m, m->s_d0w), m->Gx, st);
            } else {
                float* u = m->Ga; float* un = m->Gb;
                hipLaunchKernelGGL(topgrad_kernel, dim3(blocks256(t4)), dim3(256), 0, st, P(m, m->d_hw), n * P2, FC, t4, one, u);
                for (int i = L - 1; i >= 0; --i) {
                    const Block& B = m->D[i];
                    const size_t per = asz(B);
                    LnBwdArgs a;
                    memset(&a, 0, sizeof a);
                    a.da = u; a.c = m->Dc[i] + 2 * n * per; a.stats = m->Dstat[i] + (size_t)2 * n * 2 * B.C;
                    a.gamma = P(m, B.gamma); a.beta = P(m, B.beta); a.alpha = kLrelu; a.HW = B.H * B.W; a.C = B.C;
                    a.dc = m->Dg[i] + 3 * n * per; a.v_out = m->V[i];
                    ln_bwd(m, a, n, st);
                    if (i > 0) { conv_dgrad(m, B, n, a.dc, un, st); float* t = u; u = un; un = t; }
                    else conv_dgrad(m, B, n, a.dc, m->Gx, st);
                }
            }
        }
        m->exact_from = -1;
        const int cols = n * H;
        hipLaunchKernelGGL(pen_col_kernel, dim3(blocks256(cols)), dim3(256), 0, st, m->Gx, n, H, H, m->slopes, m->redpart);
        uad_launch_reduce_partials(m->redpart, (int)blocks256(cols), 1, m->cfg.scale / (float)cols, m->raw + 2, st);
        hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, st, phase, m->raw, m->cfg.kappa, scal);
        if (want_backward) {
            // pass C, bottom-up: adjoint of pass B.  ubar_0 = d penalty / d ddx lives in din's tail.
            hipLaunchKernelGGL(pen_grad_kernel, dim3(blocks256(img)), dim3(256), 0, st, m->Gx, m->slopes,
                               m->cfg.scale * 2.0f / (float)cols, H, H, img, m->din + 3 * img);
            if (rn) {
                UadConvDesc d0 = m->s_d0; d0.N = n;
                uad_launch_conv_first_fwd(d0, m->din + 3 * img, P(m, m->s_d0w), nullptr, m->s_out0 + (size_t)3 * n * HW * m->dim, st);
                for (auto& B : m->DB) rb_adjoint(m, B, n, st);
            } else {
                for (int i = 0; i < L; ++i) {
                    const Block& B = m->D[i];
                    const size_t per = asz(B);
                    const float* ubar = i == 0 ? m->din + 3 * img : m->Da[i] + 3 * n * asz(m->D[i - 1]);
                    conv_fwd(m, B, n, ubar, m->Q, false, st);                 // adjoint of the data gradient w.r.t. its input
                    LnBwd2Args a;
                    memset(&a, 0, sizeof a);
                    a.q = m->Q; a.v = m->V[i]; a.c = m->Dc[i] + 2 * n * per; a.stats = m->Dstat[i] + (size_t)2 * n * 2 * B.C;
                    a.gamma = P(m, B.gamma); a.beta = P(m, B.beta); a.alpha = kLrelu; a.HW = B.H * B.W; a.C = B.C;
                    a.ubar = m->Da[i + 1] + 3 * n * per; a.inj = m->inj[i]; a.gpart = m->lnpart[i]; a.slot0 = 3 * n;
                    ln_bwd2(m, a, n, st);
                }
            }
            // pass D: ordinary backward of all 3n samples; top gradient +1/(n P) fake, -1/(n P) real, 0 for x_hat
            const float k = 1.0f / (float)(n * P2);
            Coef4 cf{{k, -k, 0.f, 1.f}};
            const size_t t4 = (size_t)3 * n * P2 * FC / 4;
            hipLaunchKernelGGL(topgrad_kernel, dim3(blocks256(t4)), dim3(256), 0, st, P(m, m->d_hw), n * P2, FC, t4, cf, top);
            {   // Dense(1): kernel gradient over the 3n feature rows (coefficient per third) + pass C's adjoint rows (coefficient 1)
                const int rows = 4 * n * P2, rpb = (rows + 255) / 256, blocks = (rows + rpb - 1) / rpb;
                hipLaunchKernelGGL(coef_colsum_kernel, dim3(blocks), dim3(256), 0, st, feat, rows, rpb, n * P2, FC, cf, m->finpart);
                uad_launch_reduce_partials(m->finpart, blocks, FC, 1.0f, Gr(m, m->d_hw), st);
                gan_grad_final(m, m->d_hw);
                // Dense(1) bias: +1/(nP) over the fake rows, -1/(nP) over the real rows = 0 (stays at its zero initialisation)
            }
            disc_bwd(3 * n, true, n, 2 * n, nullptr);
        }
        if (io->generated) HIP_TRY(hipMemcpyAsync(io->generated, m->xg, img * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else if (av) {
        // trainers/AnoVAEGAN.py:61-71,84: enc_loss = mean_n sum|x - out| + kl_weight * mean_n KL over Encoder + Generator variables
        gen_fwd(nullptr);
        reduce_to<2>(m, 0, io->x, m->xg, img, 1.0f / (float)n, io->l1_map, st);
        reduce_to<0>(m, 1, m->v_kl, nullptr, (size_t)n, 1.0f / (float)n, nullptr, st);
        hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, st, 3, m->raw, m->cfg.kl_weight, scal);
        if (want_backward) {
            hipLaunchKernelGGL(sign_scale_kernel, dim3(blocks256(img)), dim3(256), 0, st, m->xg, io->x, 1.0f / (float)n, img, m->dxbuf);
            gen_bwd(nullptr, true, m->dzbuf);
            v_enc_backward(m, io, n, m->cfg.kl_weight / (float)n, st);
        }
        if (io->reconstruction) HIP_TRY(hipMemcpyAsync(io->reconstruction, m->xg, img * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (io->z_enc) HIP_TRY(hipMemcpyAsync(io->z_enc, m->z, (size_t)n * m->cfg.zdim * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else {
        // trainers/fAnoGAN.py:60-66,76: enc_loss = MSE(x, x_enc) + kappa * MSE(features(x_enc), features(x)) w.r.t. the Encoder
        if (!io->x) return fail(UAD_ERR_INVALID, "encoder phase needs io.x");
        if (rn) s_enc_forward(m, io->x, n, st); else enc_forward(m, io->x, io->mask_z, n, st);
        gen_fwd(m->z);
        HIP_TRY(hipMemcpyAsync(m->din, m->xg, img * sizeof(float), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(m->din + img, io->x, img * sizeof(float), hipMemcpyDeviceToDevice, st));
        disc_fwd(2 * n, false);
        const size_t nf = (size_t)n * P2 * FC;
        const float* f_enc = feat;
        const float* f_real = feat + nf;
        reduce_to<1>(m, 0, io->x, m->xg, img, 1.0f / (float)img, nullptr, st);
        reduce_to<1>(m, 1, f_enc, f_real, nf, 1.0f / (float)nf, nullptr, st);
        reduce_to<2>(m, 2, io->x, m->xg, img, 1.0f / (float)n, io->l1_map, st);
        hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, st, phase, m->raw, m->cfg.kappa, scal);
        if (want_backward) {
            hipLaunchKernelGGL((diff_scale_kernel<false>), dim3(blocks256(nf)), dim3(256), 0, st, f_enc, f_real,
                               m->cfg.kappa * 2.0f / (float)nf, nf, top);
            disc_bwd(n, false, 0, -1, m->dxbuf);
            hipLaunchKernelGGL((diff_scale_kernel<true>), dim3(blocks256(img)), dim3(256), 0, st, m->xg, io->x, 2.0f / (float)img, img, m->dxbuf);
            gen_bwd(m->z, false, m->dzbuf);
            if (rn) s_enc_backward(m, io->x, n, st); else enc_backward(m, io->x, io->mask_z, n, st);
        }
        if (io->reconstruction) HIP_TRY(hipMemcpyAsync(io->reconstruction, m->xg, img * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (io->z_enc) HIP_TRY(hipMemcpyAsync(io->z_enc, m->z, (size_t)n * m->cfg.zdim * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}

static int gan_reconstruct_body(uad_gan_t* m, const uad_gan_io_t* io, int n, void* stream) {
    if (!m || !io || !io->x) return fail(UAD_ERR_INVALID, "null argument");
    if (n <= 0 || n > m->cfg.max_batch) return fail(UAD_ERR_INVALID, "batch %d outside (0, max_batch=%d]", n, m->cfg.max_batch);
    hipStream_t st = (hipStream_t)stream;
    const size_t img = (size_t)n * m->cfg.height * m->cfg.width;
    refresh_packs(m, st);
    if (m->variant == UAD_GAN_AAE && m->you) you_forward(m, io, io->x, n, 1.0f / (float)n, st);
    else if (m->variant == UAD_GAN_AAE && m->zim) zim_forward(m, io->x, io->eps, n, n, st);
    else if (m->variant == UAD_GAN_AAE && m->gmv) gmv_forward(m, io, io->x, n, 1.0f / (float)n, st);
    else if (m->variant == UAD_GAN_AAE) { a_encode(m, io->x, io->mask_z, n, 0, st); a_decode(m, m->a_zm, io->mask_g, n, st); }
    else if (m->variant == UAD_GAN_RESNET) { s_enc_forward(m, io->x, n, st); s_gen_forward(m, m->z, n, st); }
    else if (m->variant == UAD_GAN_ANOVAEGAN) { v_enc_forward(m, io, n, st); gen_forward(m, m->z, nullptr, n, st); }
    else { enc_forward(m, io->x, io->mask_z, n, st); gen_forward(m, m->z, io->mask_g, n, st); }
    if (io->reconstruction) HIP_TRY(hipMemcpyAsync(io->reconstruction, m->xg, img * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (io->z_enc) HIP_TRY(hipMemcpyAsync(io->z_enc, m->variant == UAD_GAN_AAE ? m->a_zm : m->z, (size_t)n * m->cfg.zdim * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (io->l1_map) hipLaunchKernelGGL((sum_kernel<2>), dim3(256), dim3(256), 0, st, io->x, m->xg, img, io->l1_map, m->redpart);
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}

int uad_gan_restore_step(uad_gan_t* m, float* x_restored, const uad_gan_io_t* io, int n, float tv_lambda, float restore_lr, float* grads_out,
                         void* stream) {
    if (!m || !io || !x_restored) return fail(UAD_ERR_INVALID, "null argument");
    if (m->variant != UAD_GAN_AAE || !(m->gmv || m->you)) return fail(UAD_ERR_INVALID, "uad_gan_restore_step needs a GMVAE handle (aae_kind 3 or 6)");
    if (n <= 0 || n > m->cfg.max_batch) return fail(UAD_ERR_INVALID, "batch %d outside (0, max_batch=%d]", n, m->cfg.max_batch);
    auto body = [&](hipStream_t st) {
        const int rc = m->you ? you_phase(m, io, x_restored, n, 1, true, tv_lambda, restore_lr, x_restored, grads_out, st)
                              : gmv_phase(m, io, x_restored, n, 1, true, tv_lambda, restore_lr, x_restored, grads_out, st);
        if (rc != UAD_OK) return rc;
        HIP_TRY(hipGetLastError());
        return (int)UAD_OK;
    };
    return body((hipStream_t)stream);
}

int uad_gan_phase(uad_gan_t* m, int phase, const uad_gan_io_t* io, int n, int want_backward, void* stream) {
    if (!m || !io) return fail(UAD_ERR_INVALID, "null argument");
    if (m->ar_comm && want_backward && phase >= 0 && phase <= 2 && m->variant != UAD_GAN_AAE) {
        // data parallelism, library-issued: the trained group's slice (AnoVAE-GAN's Encoder phase trains Encoder + Generator, one contiguous slice)
        long long off = m->grp_off[phase], cnt = m->grp_cnt[phase];
        if (m->variant == UAD_GAN_ANOVAEGAN && phase == UAD_GAN_ENCODER) { off = 0; cnt = m->grp_cnt[UAD_GAN_ENCODER] + m->grp_cnt[UAD_GAN_GENERATOR]; }
        gan_ar_begin(m, off, cnt, (hipStream_t)stream);
    }
    const int rc = gan_phase_body(m, phase, io, n, want_backward, stream);
    if (rc == UAD_OK) gan_ar_end(m); else m->ar_active = false;
    return rc;
}
// Attaches an RCCL communicator (uad_rccl_comm_create) to the handle: from then on uad_gan_phase(..., want_backward) all-reduces the trained group's
// gradient slice itself, in up to four buckets issued on a collective stream while the phase's backward still runs; the phase's stream waits for the last
// one before it returns to the caller (whose next launch is uad_gan_adam with grad_scale = 1 / world).  comm = NULL detaches.  The AAE-family graphs have no
// bucket hooks: UAD_ERR_UNSUPPORTED, the caller all-reduces their slice itself (uad_rccl_allreduce).
int uad_gan_allreduce_attach(uad_gan_t* m, void* comm, int world) {
    if (!m) return fail(UAD_ERR_INVALID, "null handle");
    if (!comm) { m->ar_comm = nullptr; m->ar_world = 1; m->ar_active = false; return UAD_OK; }
    if (world < 1) return fail(UAD_ERR_INVALID, "uad_gan_allreduce_attach: world %d", world);
    if (m->variant == UAD_GAN_AAE) return fail(UAD_ERR_UNSUPPORTED, "uad_gan_allreduce_attach: the AAE-family phases are all-reduced by the caller");
    if (!m->ar_stream) {
        // HIGH-priority, non-blocking.  Measured on one rank under RCCL with GPU_MAX_HW_QUEUES=8 (the package's multi-process default): a normal-priority
        // stream here made the WGAN iteration 1.62x the plain one (83 vs 51 ms) WITH OR WITHOUT the collectives being issued -- the extra stream's hardware
        // queue, not RCCL; a blocking stream 2.2x; the high-priority stream 1.016x, as do 4-6 hardware queues (profiles/r06_e_gan_dp_one_rank.md).
        { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); HIP_TRY(hipStreamCreateWithPriority(&m->ar_stream, hipStreamNonBlocking, hi)); }
        for (int i = 0; i < 4; ++i)
            if (hipEventCreateWithFlags(&m->ar_ev_in[i], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) HIP_TRY(hipEventCreateWithFlags(&m->ar_ev_in[i], hipEventDisableTiming));
        if (hipEventCreateWithFlags(&m->ar_ev_out, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) HIP_TRY(hipEventCreateWithFlags(&m->ar_ev_out, hipEventDisableTiming));
    }
    m->ar_comm = comm; m->ar_world = world;
    return UAD_OK;
}
int uad_gan_reconstruct(uad_gan_t* m, const uad_gan_io_t* io, int n, void* stream) {
    if (!m || !io) return fail(UAD_ERR_INVALID, "null argument");
    return gan_reconstruct_body(m, io, n, stream);
}
int uad_k3_profile_enable(int on) { uad_k3_prof_enable(on != 0); return UAD_OK; }
int uad_k3_profile_read(char* buf, int cap) { return uad_k3_prof_read(buf, cap); }

int uad_gan_adam(uad_gan_t* m, int group, float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
    if (!m || group < 0 || group > 2) return fail(UAD_ERR_INVALID, "bad group");
    m->step[group] += 1;
    const double t = (double)m->step[group];
    const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
    const long long off = m->grp_off[group];
    m->packed_valid = false;
    auto dirty = [&](long long lo, long long hi) { if (lo < m->dirty_lo) m->dirty_lo = lo; if (hi > m->dirty_hi) m->dirty_hi = hi; };
    dirty(off, off + m->grp_cnt[group]);
    // AAE family: optim_gen (group 0) shares its variables with optim_ae (group 1) and keeps slots of its own
    float* am = (m->variant == UAD_GAN_AAE && group == 0) ? m->adam_m2 : m->adam_m;
    float* avv = (m->variant == UAD_GAN_AAE && group == 0) ? m->adam_v2 : m->adam_v;
    uad_launch_adam(m->params + off, m->grads + off, am + off, avv + off, (size_t)m->grp_cnt[group], lr_t, beta1, beta2,
                    eps, grad_scale, (hipStream_t)stream);
    if (m->variant == UAD_GAN_ANOVAEGAN && group == UAD_GAN_ENCODER) {
        // optim_vae also owns the Generator variables (trainers/AnoVAEGAN.py:84), with slots of its own and the same step count
        const long long go = m->grp_off[UAD_GAN_GENERATOR];
        dirty(go, go + m->grp_cnt[UAD_GAN_GENERATOR]);
        uad_launch_adam(m->params + go, m->grads + go, m->adam_m2 + go, m->adam_v2 + go, (size_t)m->grp_cnt[UAD_GAN_GENERATOR], lr_t, beta1,
                        beta2, eps, grad_scale, (hipStream_t)stream);
    }
    HIP_TRY(hipGetLastError());
    return UAD_OK;
}

}  // extern "C"
