// Internal C++ launch interface for the gfx950 kernels (not the public C-ABI;
// that is include/uad_hip.h).  All pointers are device pointers, all tensors are
// fp32 NHWC, all launches are asynchronous on the given stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// One strided-conv relation between a "big" (finer grid) and a "small" image:
//   big pixel  (S*i - P + ky, S*j - P + kx)  <->  small pixel (i, j),  tap (ky,kx)
// Weight tensor is always W[tap][cb][cs] (big-side channel first):
//   Conv2D          kernel [kh,kw,Cin ,Cout] : big = input , small = output
//   Conv2DTranspose kernel [kh,kw,Cout,Cin ] : big = output, small = input
// Dense [in,out] and 1x1 convs are the KS=1,S=1,P=0 case.
struct UadConvDesc {
    int N;
    int HB, WB, CB;
    int HS, WS, CS;
    int KS, S, P;
};

// Activation-on-load: v -> lrelu_alpha(scale[c]*v + shift[c]) (frozen-stats BN +
// LeakyReLU/ReLU of the producer layer, applied by the consumer).  scale == nullptr
// means identity.
struct UadXform {
    const float* scale = nullptr;   // gamma (or a ready-made scale when mult == 1)
    const float* shift = nullptr;   // beta
    float alpha = 1.f;
    float mult = 1.f;     // scale multiplier: 1/sqrt(1+eps) of the frozen-stats BN
    // F-kind bf16x3 kernel only ("final-backward on load", GMVAE restoration): the staged value becomes
    //   d loss / d c = dxhat[pixel] * wf[ch] * lrelu'(scale*c + shift) * scale     instead of lrelu(scale*c + shift),
    // i.e. the big operand is the last block's PRE-BN output and the loss gradient is never materialised.
    const float* fb_dxhat = nullptr;   // [N, HB, WB]
    const float* fb_wf = nullptr;      // [CB]
    // ... and the same without reading c at all (training step): fb_bits[N, HB, WB] holds one word per pixel whose bit ch is
    // (scale*c + shift > 0) as the last block's fused epilogue saw it (UadEpilogue::fin_bits).  The big operand pointer is then unused:
    // the 128 B / pixel of d loss / d c shrink to 8 B / pixel (bits + dxhat).  Understood by the F-kind bf16x3 kernel and by the k5 s2
    // bf16x3 filter-gradient kernel (there as the transform of `big`); CB <= 32.
    const unsigned* fb_bits = nullptr;
};

enum { UAD_EPI_BIAS = 0, UAD_EPI_BWD_ACT = 1, UAD_EPI_FINAL = 2 };

// optional split-K workspace of the generic kernels (slabs of raw partial outputs)
// split-K workspace: slabs + (optional) zero-initialised arrival counters, one per output tile of a split launch: with counters the last
// workgroup to arrive at a tile sums the slabs and applies the epilogue inside the conv kernel (no splitk_epilogue launch)
struct UadGemmWs { float* ptr; size_t floats; unsigned* counters; int ncounters; };

struct UadEpilogue {
    int kind;             // UAD_EPI_*
    const float* bias;    // EPI_BIAS: per output channel, may be null
    const float* mul;     // EPI_BIAS: optional elementwise multiplier, same layout as out
    const float* add;     // EPI_BIAS: optional elementwise addend (applied after mul), same layout as out
    // EPI_BWD_ACT: out = acc * lrelu'(escale*cprev+eshift) * escale ; column sums of
    // d_bn and d_bn*cprev are written to colpart[tile][2][Cout]
    const float* cprev;
    const float* escale;
    const float* eshift;
    float ealpha;
    float emult;          // escale multiplier (see UadXform::mult)
    float* colpart;
    // EPI_FINAL (last decoder ConvT, all output channels in one workgroup): the layer's own BN + LeakyReLU
    // (escale/eshift/ealpha/emult), the final 1x1 conv C -> 1 and the L1 loss are applied to the accumulator tile, with the
    // loss gradient when fin_dc is given -- the pre-BN output never goes to HBM unless Out is non-null.
    const float* fin_wf;        // [C] final conv kernel
    const float* fin_bf;        // [1]
    const float* fin_x;         // [N,H,W,1] target
    float* fin_xhat;            // [N,H,W,1]
    float* fin_l1;              // [N,H,W,1] or null
    float* fin_rec_partial;     // [tiles]               (tile = blockIdx.y * gridDim.x + blockIdx.x)
    float* fin_red_partial;     // [tiles][3C+1]: dwf[C], S1[C], S2[C], dbf   (backward only)
    float* fin_dc;              // [N,H,W,C] d loss / d c, or null
    unsigned* fin_bits;         // [N,H,W] activation-pattern word per pixel (bit ch = BN output > 0) + fin_dxhat[N,H,W] = sign(x_hat - x) *
    float* fin_dxhat;           // fin_inv_batch: the compressed form of d loss / d c (UadXform::fb_bits); both or neither
    float fin_inv_batch;
};

// F-type: small_out[n,i,j,cs] = sum_{tap,cb} xf(big_in)[n,S*i-P+ky,S*j-P+kx,cb] * W[tap][cb][cs]
// Wpacked (optional): this tensor inside the F-pack buffer written by uad_launch_pack_weights; enables the k5 s2
// spatial kernel.  The tile count of EPI_BWD_ACT (uad_conv_*_tiles) assumes Wpacked is given whenever it can be used.
void uad_launch_conv_f(const UadConvDesc& d, const float* big_in, UadXform xf, const float* W,
                       float* small_out, UadEpilogue ep, hipStream_t st, const float* Wpacked = nullptr,
                       UadGemmWs ws = UadGemmWs{nullptr, 0}, const unsigned short* Wp16 = nullptr, long long w16_plane = 0,
                       bool generic_bf16x3 = false, int planes16 = 2);
// D-type: big_out[n,S*i-P+ky,S*j-P+kx,cb] += xf(small_in)[n,i,j,cs] * W[tap][cb][cs]
void uad_launch_conv_d(const UadConvDesc& d, const float* small_in, UadXform xf, const float* W,
                       float* big_out, UadEpilogue ep, hipStream_t st, const float* Wpacked = nullptr,
                       UadGemmWs ws = UadGemmWs{nullptr, 0}, const unsigned short* Wp16 = nullptr, long long w16_plane = 0,
                       bool generic_bf16x3 = false, int planes16 = 2);
// generic_bf16x3: when no specialised kernel applies, let the generic kernel use bf16x3 products (identity xf, BK = 32 tiles)
// bf16x3 math mode: Wp16 = this tensor's hi plane inside the bf16 pack buffer (ushort index 2*offset), w16_plane = its
// element count (the lo plane follows the hi plane)
void uad_launch_pack_weights_bf16(const float* params, unsigned short* w16_f, unsigned short* w16_d, const long long* offs,
                                  const int* cbs, const int* css, const int* taps, int n, hipStream_t st);
// planes16 = 3 ("bf16x6"): Wp16 = the tensor's first plane inside a THREE-plane pack buffer (ushort index 4 * offset, planes w16_plane elements apart)
// written by this function; products are fp32-grade (six bf16 MFMAs per K = 16).  Understood by the k3 tap-list kernel (uad_conv_k3_takes) and, since
// round 6, by the k5 s2 spatial kernels (F-kind, lane = pixel D-kind, filter gradient): a k5 launch those do not take (uad_conv_x6_takes) runs on the
// exact-fp32 kernels and needs Wpacked beside the planes.
void uad_launch_pack_weights_bf16_3p(const float* params, unsigned short* w3_f, unsigned short* w3_d, const long long* offs,
                                     const int* cbs, const int* css, const int* taps, int n, hipStream_t st);
// true when uad_launch_conv_f / _d would run the k3 tap-list kernel for this shape when bf16 planes are given (identity activation, bias / addend epilogue)
bool uad_conv_k3_takes(const UadConvDesc& d, bool f_type);
// HIP-event profiler of the k3 launches (uad_convk16.inc): per launch shape, calls and summed duration as text lines
void uad_k3_prof_enable(bool on);
int uad_k3_prof_read(char* buf, int cap);
// re-layout of n (<= 8) weight tensors W[tap][cb][cs] living at params+offs[i] into the F-pack / D-pack buffers
void uad_launch_pack_weights(const float* params, float* wpack_f, float* wpack_d, const long long* offs, const int* cbs,
                             const int* css, const int* taps, int n, hipStream_t st);
bool uad_conv_spatial_ok(const UadConvDesc& d, bool f_type);
// number of colpart tiles the above launches write for EPI_BWD_ACT
// (the same have_pack / workspace capacity as the launch must be passed: they select the kernel path)
int uad_conv_f_tiles(const UadConvDesc& d, bool have_pack = true, size_t ws_floats = 0, int ncounters = 0, int planes16 = 2);      // ncounters: UadGemmWs::ncounters when the launch will run in a split-bf16 mode, else 0; planes16: as the launch
bool uad_conv_x6_takes(const UadConvDesc& d, bool f_type, size_t ws_floats, int ncounters);      // a planes16 = 3 launch of this k5 s2 shape runs on the three-plane spatial kernels
// true when uad_launch_conv_d(d, ..., UAD_EPI_FINAL) is available: bf16x3 planes given, class-sequential spatial kernel, all
// output channels (32) in one workgroup, no split
bool uad_conv_d_can_fuse_final(const UadConvDesc& d, bool have_pack16, size_t ws_floats);
bool uad_conv_d_can_fuse_final_f32(const UadConvDesc& d, bool have_pack, size_t ws_floats);
// true when uad_launch_conv_f would run the bf16x3 spatial kernel that understands UadXform::fb_* (see there)
bool uad_conv_f_supports_final_bwd(const UadConvDesc& d, bool have_pack16, size_t ws_floats);
int uad_conv_d_tiles(const UadConvDesc& d, bool have_pack = true, size_t ws_floats = 0, int ncounters = 0, int planes16 = 2);
// workspace floats the split-K path would like for this op (0 = it would not split)
size_t uad_conv_ws_floats(const UadConvDesc& d, bool f_type, bool have_pack);
// W-type: dW[tap][cb][cs] = sum_{n,i,j} xfb(big)[..tap..,cb] * xfs(small)[n,i,j,cs]
// `partial` must hold uad_conv_w_partial_floats(d) floats.
size_t uad_conv_w_partial_floats(const UadConvDesc& d);
// true when uad_launch_conv_w(d, ..., math_bf16x3) runs the kernel that understands xfb.fb_bits (compressed d loss / d c as `big`)
bool uad_conv_w_supports_fb_bits(const UadConvDesc& d, bool math_bf16x3);
void uad_launch_conv_w(const UadConvDesc& d, const float* big, UadXform xfb, const float* small, UadXform xfs,
                       float* dW, float* partial, hipStream_t st, bool math_bf16x3 = false,
                       hipStream_t reduce_st = nullptr, hipEvent_t ev = nullptr, bool generic_bf16x3 = false, bool defer_reduce = false, int planes16 = 2);
// the split-K slab reduction of a uad_launch_conv_w(..., math_bf16x3, ..., defer_reduce = true) call (no-op when that launch did not split); the
// math mode has to be the launch's: the k5 kernels split differently in the two modes
void uad_launch_conv_w_reduce(const UadConvDesc& d, float* dW, float* partial, hipStream_t st, bool math_bf16x3);

// out[j] = scale * sum_{s<S} partial[s*L + j]
void uad_launch_reduce_partials(const float* partial, int S, int L, float scale, float* out, hipStream_t st);
// dbeta = S1, dgamma = rstd*S2, dbias = gamma*rstd*S1 with S1,S2 = sum over T tiles of colpart[t][0/1][c]
// scratch (optional, uad_bn_grad_finalize_scratch_floats(C) zero-initialised floats, one per stream): tile ranges in parallel
void uad_launch_bn_grad_finalize(const float* colpart, int T, int C, const float* gamma, float rstd,
                                 float* dgamma, float* dbeta, float* dbias, hipStream_t st, float* scratch = nullptr);
size_t uad_bn_grad_finalize_scratch_floats(int C);
// red[3C+1] = {dwf[C], S1[C], S2[C], dbf} (already summed over the tiles) -> final conv kernel / bias gradients + the last block's BN / bias gradients
void uad_launch_final_gradfin(const float* red, int C, const float* gamma, float rstd, float* dwf, float* dbf, float* dgamma, float* dbeta,
                              float* dbias, hipStream_t st);
// out[c] = sum_rows g[row][c]
void uad_launch_colsum(const float* g, int rows, int C, float* out, float* scratch, hipStream_t st);
size_t uad_colsum_scratch_floats(int rows, int C);

// first layer (tiny Cin, raw image input): direct conv + bias, and its weight gradient
void uad_launch_conv_first_fwd(const UadConvDesc& d, const float* x, const float* W, const float* bias,
                               float* out, hipStream_t st);
size_t uad_conv_first_wgrad_partial_floats(const UadConvDesc& d);
void uad_launch_conv_first_wgrad(const UadConvDesc& d, const float* x, const float* g, float* dW,
                                 float* partial, hipStream_t st);

// final 1x1 conv (C -> 1) fused with the L1 loss and its backward
struct UadFinalArgs {
    int N, H, W, C;          // c_last is [N,H,W,C]
    const float* c_last;     // pre-BN output of the last ConvT
    const float* scale;      // BN scale/shift of the last block (activation on load)
    const float* shift;
    float alpha;
    float mult;              // scale multiplier (1/sqrt(1+eps))
    const float* wf;         // [C] final conv kernel
    const float* bf;         // [1]
    const float* x;          // [N,H,W,1] target
    float* x_hat;            // [N,H,W,1]
    float* l1_map;           // [N,H,W,1] or null
    float* rec_partial;      // [N][blocks_per_sample]
    // backward (null d_c => forward only)
    float* d_c;              // [N,H,W,C]
    float* red_partial;      // [N*blocks_per_sample][3*C+1]: dwf[C], S1[C], S2[C], dbf
    float inv_batch;         // 1/global_batch
    const float* dxhat_in;   // optional [N,H,W,1]: d objective / d x_hat given by the caller (GMVAE restore term);
                             // null => sign(x_hat - x) * inv_batch
};
int uad_final_blocks_per_sample(int H, int W);
void uad_launch_final_fwd_bwd(const UadFinalArgs& a, hipStream_t st);

// reparameterisation + KL (VAE bottleneck).  Samples [n_vae, n) are the ceVAE context branch: z = mu (masked by
// mask_mu_ce, indexed from 0), no noise, no KL, zero log-sigma gradient.
void uad_launch_reparam_fwd(int n, int n_vae, int zdim, const float* mu_raw, const float* ls_raw, const float* mask_mu,
                            const float* mask_ls, const float* mask_mu_ce, const float* eps, float* mu, float* ls,
                            float* sigma, float* z, float* kl_per_sample, hipStream_t st);
void uad_launch_reparam_bwd(int n, int n_vae, int zdim, const float* dz, const float* mu, const float* sigma,
                            const float* eps, const float* mask_mu, const float* mask_ls, const float* mask_mu_ce,
                            float inv_batch, float* dmu_raw, float* dls_raw, hipStream_t st);
// ceVAE: data gradient of the first conv (d.N samples) + direct L1-label term -> anomaly = |x-x_hat| * |d loss_vae/dx|
void uad_launch_conv_first_dgrad(const UadConvDesc& d, const float* g, const float* W, const float* x,
                                 const float* x_hat, float inv_batch, float* anomaly, float* dx, hipStream_t st);
// GMVAE form: gx = (data gradient) - dxhat[pix] (x's direct term is minus the reconstruction's), dx (optional) = gx,
// and, if x_upd is given, the restoration update x_upd[pix] -= restore_lr * gx (trainers/GMVAE_spatial.py:189-190)
void uad_launch_conv_first_dgrad_restore(const UadConvDesc& d, const float* g, const float* W, const float* dxhat,
                                         float* dx, float* x_upd, float restore_lr, hipStream_t st);

// ---- dense bottleneck, one workgroup per sample (uad_bott.hip) ----
struct UadBottArgs {
    int cenc, cmid, npos, zdim;          // encoder width, conv2d width, r*r positions, zDim
    int n_vae;                           // samples >= n_vae are the ceVAE context branch
    float alpha, mult, inv_batch;
    const float* c_enc; const float* scale; const float* shift;      // last encoder block (pre-BN) + its BN
    const float *Wb, *bb, *Wmu, *bmu, *Wsg, *bsg, *Wd, *bd, *Wr, *br; // Wsg == null: AE (Wmu = dense_z)
    const float *WdT, *WmuT, *WsgT;      // transposed copies ([F][Z], [Z][F], [Z][F]) for the backward's coalesced GEMVs
    const float *eps, *mask_mu, *mask_ls, *mask_mu_ce, *mask_dec;
    // forward outputs (all kept for the backward / the parameter-gradient GEMMs)
    float *t, *mu, *ls, *sigma, *z, *kl, *dvec, *cb;
    // backward
    const float* dcb;                    // [n, npos, cenc] d loss / d cb
    float* dcb_copy;                     // optional: the backward kernel leaves a copy of dcb here
    float* wpart;                        // optional: per-workgroup shares of conv2d / conv2d_1's parameter gradients, [wg][2*C*M + M]
    // exchange between the workgroups of one sample (uad_bott.hip): partial vectors, completion flags, this launch's epoch
    float* xch; unsigned* flags; unsigned epoch; int xw;
    unsigned* err;                       // optional, host-visible (pinned): a workgroup that gives up waiting for its siblings stores (epoch | 1 << 31) here
    unsigned* err_dev;                   // optional, device memory: the same word for the optimizer kernels (uad_launch_adam / _optim `fault`): they skip their update
    int fault;                           // tests (UAD_BOTT_FAULT=1): workgroup 1 of sample 0 never publishes its flag
    unsigned long long* stamps;          // debug (UAD_BOTT_DBG): phase clocks of workgroup 0
    float *dd, *dmu, *dls, *dflat, *g_out, *colpart;   // colpart [n][2][cenc]
};
size_t uad_bottleneck_lds_bytes(const UadBottArgs& a, bool bwd);
bool uad_bottleneck_fused_ok(const UadBottArgs& a);
int uad_bottleneck_group(const UadBottArgs& a);                  // workgroups per sample (1 or 4)
int uad_bottleneck_colpart_rows(const UadBottArgs& a, int n);    // rows of colpart [rows][2][cenc] the backward writes
void uad_launch_bottleneck_fwd(const UadBottArgs& a, int n, hipStream_t st);
void uad_launch_bottleneck_bwd(const UadBottArgs& a, int n, hipStream_t st);
// every parameter gradient of the dense bottleneck in one launch (uad_bott.hip); dls / gWsg / gbsg null: AE
struct UadBottWgradArgs {
    int n, cenc, cmid, npos, zdim;
    float alpha, mult;
    const float *z, *dd, *t, *dmu, *dls;                 // [n][Z], [n][F], [n][F], [n][Z], [n][Z]
    const float *c_enc, *scale, *shift;                  // last encoder block (pre-BN) + its BN: h = lrelu(bn(c_enc)) on load
    const float *dflat, *dvec, *dcb;                     // [n][F], [n][F], [n][P][C]
    float *gWd, *gbd, *gWmu, *gbmu, *gWsg, *gbsg, *gWb, *gbb, *gWr;
    const float* part; int nparts;                       // [nparts][2*C*M + M] shares of dWb | dWr | db_b left by the backward kernel
};
bool uad_bottleneck_wgrad_ok(const UadBottWgradArgs& a);
size_t uad_bottleneck_wgrad_part_floats(const UadBottWgradArgs& a);
void uad_launch_bottleneck_wgrad(const UadBottWgradArgs& a, hipStream_t st);
// out_i[c][r] = in_i[r][c] for up to 3 matrices in one launch
void uad_launch_transpose(const float* const* in, const int* R, const int* C, float* const* out, int njobs, hipStream_t st);

// ---- spatial GMVAE latent heads (uad_gmvae.hip) ----
struct UadGmArgs {
    int cenc, W, Z, C;                  // encoder width, dim_w, dim_z, dim_c
    float c_lambda, inv_batch;
    // last encoder block: pre-BN output + its BN (activation on load)
    const float* c_enc; const float* scale; const float* shift; float alpha, mult;
    // head parameters (flat-buffer pointers)
    const float *wmu_k, *wmu_b, *wls_k, *wls_b, *zmu_k, *zmu_b, *zls_k, *zls_b, *c7_k, *c7_b, *m_k, *m_b, *l_k, *l_b, *var;
    const float *eps_w, *eps_z;         // [L,W], [L,Z] or null
    // forward outputs
    float* h_out;                       // [L,cenc] activated encoder feature map (decoder input, wgrad operand)
    float* loc_loss;                    // [L,3] con, w-prior, c-prior of each location
    float *w_mu, *w_ls, *z_mu, *z_ls, *pc;   // optional maps
    // backward
    const float* dh_dec;                // [L,cenc] d loss / d h through the decoder
    float* g_out;                       // [L,cenc] d loss / d c_enc
    float* colpart;                     // [L][2][cenc]
    float *dvec_heads, *dvec_a7, *dvec_M, *dvec_Lq, *ws_out, *mid_out;   // per-location vectors for the weight gradients
    // decoders that consume z_sampled instead of h (models/gaussian_mixture_variational_autoencoder_You.py:54-68)
    float* zs_out;                      // forward, optional: [L,Z] z_sampled
    const float* dz_dec;                // backward, optional: [L,Z] d loss / d z_sampled through the decoder
};
struct UadGmWgradArgs {
    struct Job { const float* A; int lda; const float* B; int ldb; int b; int off; };
    Job job[16];
    int njobs, total, L, chunk;
    float* partial;                     // [chunks][total]
};
size_t uad_gm_lds_bytes(const UadGmArgs& a, bool bwd);
void uad_launch_gm_heads_fwd(const UadGmArgs& a, int locations, hipStream_t st);
void uad_launch_gm_heads_bwd(const UadGmArgs& a, int locations, hipStream_t st);
int uad_gm_wgrad_chunks(int L);
void uad_launch_gm_heads_wgrad(UadGmWgradArgs a, float* out, hipStream_t st);
void uad_launch_tv_dxhat(const float* x, const float* xh, int N, int H, int W, float inv_batch, float tv_lambda,
                         float* dxhat, hipStream_t st);
void uad_launch_gm_loss_finalize(const float* rec_partial, int n, int bps, const float* loc_loss, int lps,
                                 float inv_batch, float* rec_per_sample, float* scalars, hipStream_t st);
// ---- dense GMVAE latent (uad_gmd.hip) ----
// skinny dense contraction: out[r*ldo + o] (=|+=) mask[r*ldm + o] * (bias[o] + sum_{i<I} a[r*lda + i] * B[i*sBi + o*sBo])
struct UadSdArgs {
    const float* a; int lda;
    const float* B; long long sBi, sBo;
    const float* bias;          // [O] or null
    const float* mask; int ldm; // [R, ldm] or null
    int R, I, O;
    float* out; int ldo;
    int accumulate;
};
void uad_launch_sd(const UadSdArgs& a, hipStream_t st);
// dW[k*J + j] = sum_r a[r*lda + k] * g[r*ldg + j], db[j] = sum_r g[r*ldg + j] (db may be null); fixed summation order
void uad_launch_sd_wgrad(const float* a, int lda, const float* g, int ldg, int R, int K, int J, float* dW, float* db, hipStream_t st);
struct UadGmdArgs {
    int W, Z, C, nmax;
    float c_lambda, inv;                                  // inv: weight of a sample's prior terms in the objective (1/n; 1 when restoring)
    const float* hv;                                      // [n, 2W+2Z] raw head outputs (bias added, before dropout)
    const float *mask_wmu, *mask_wls, *mask_zmu;          // [n,W], [n,W], [n,Z] keep/(1-rate) or null
    const float *e_w, *e_z;                               // [n,W], [n,Z] N(0,1) or null (= 0)
    const float *Wm, *bm, *Wl, *bl, *var;                 // p(z|w,c): dense [W,Q], [Q], dense_1 [W,Q], [Q], Variable [Q]; Q = Z*C
    float *hvm, *w_s, *z_s, *M, *Lq, *pc;                 // saved forward state ([n,2W+2Z], [n,W], [n,Z], [n,Q], [n,Q], [n,C])
    float* loss3;                                         // [3][nmax]: per-sample conditional-prior, w-prior, c-prior terms
    const float* dz_dec;                                  // [n,Z] decoder-side d / d z_sampled (backward)
    float *dhv, *dM, *dLq;                                // [n,2W+2Z] d / d raw heads, [n,Q], [n,Q]
};
size_t uad_gmd_lds_bytes(int W, int Z, int C);
void uad_launch_gmd_fwd(const UadGmdArgs& a, int n, hipStream_t st);
void uad_launch_gmd_bwd(const UadGmdArgs& a, int n, hipStream_t st);
// dx = data gradient of the first conv, nothing else (f-AnoGAN critic input gradient)
void uad_launch_conv_first_dgrad_plain(const UadConvDesc& d, const float* g, const float* W, float* dx, hipStream_t st);
// spatial autoencoder latent: z = mask * lrelu(gamma * rs0 * c + beta) and its backward (colpart [blocks][2][C], blocks as returned)
void uad_launch_spatial_z_fwd(const float* c, const float* gamma, const float* beta, float rs0, float alpha, const float* mask, int rows,
                              int C, float* z, hipStream_t st);
int uad_spatial_z_bwd_blocks(int rows);
void uad_launch_spatial_z_bwd(const float* dz, const float* c, const float* gamma, const float* beta, float rs0, float alpha,
                              const float* mask, int rows, int C, float* dc, float* colpart, hipStream_t st);
// batch assembly from the HBM-resident slice cache: out[b] = src[idx[b]]; mask[b][p] = lut[labels[idx[b]][p]] (lut null: the label value)
// counter-based noise of one step (uad_misc.hip: rng_fill_kernel); at most 8 jobs
struct UadRngJob { float* out; int per_sample; int kind; float rate; int stream; };
void uad_launch_rng_fill(const UadRngJob* jobs, int njobs, int n, unsigned long long seed, unsigned long long step, long long sample0,
                         hipStream_t st);
// shader-clock probe: one wave samples s_memtime / s_memrealtime for `ticks` 100 MHz ticks (uad_misc.hip: clock_probe_kernel)
void uad_launch_clock_probe(unsigned long long* out, unsigned long long ticks, hipStream_t st);
void uad_launch_gather_slices(const float* src, const int* idx, int n, long long slice_elems, float* out, hipStream_t st);
void uad_launch_gather_mask(const unsigned char* labels, const int* idx, int n, long long slice_px, const unsigned char* lut,
                            float* out, hipStream_t st);
// y = x * mask (mask may be null -> copy)
void uad_launch_mul(const float* x, const float* mask, float* y, size_t n, hipStream_t st);
// scalars[8] = {reconstructionLoss, kl, loss, 0, Rec_vae, Rec_ce, loss_vae, 0}; samples [n_vae, n) = ceVAE context branch
void uad_launch_loss_finalize(const float* rec_partial, int n, int n_vae, int bps, const float* kl_per_sample,
                              float inv_batch, float rec_scale, float* rec_per_sample, float* scalars, hipStream_t st);

// TF-form Adam over a flat parameter buffer: g is multiplied by gscale first
// fault (optional, device word): non-zero = the gradients of this step are invalid (fused bottleneck: UadBottArgs::err_dev) -> the kernel leaves
// parameters and slots untouched
void uad_launch_optim(int kind, float* p, const float* g, float* s1, float* s2, size_t n, float lr, float momentum, float decay, float eps,
                      float gscale, hipStream_t st, const unsigned* fault = nullptr);
void uad_launch_adam(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float beta1, float beta2,
                     float eps, float gscale, hipStream_t st, const unsigned* fault = nullptr);

// residual anomaly map (utils/Evaluation.py:282-289): out = mask * (pos_only ? max(x-xr,0) : |x-xr|),
// zeroed where x < prior_thresh (pass -inf to disable); per-sample sum|x-xr| into l1err[n] (may be null)
void uad_launch_residual(const float* x, const float* xr, const float* mask, int n, int hw, int pos_only,
                         float prior_thresh, float* out, float* l1err, hipStream_t st);
