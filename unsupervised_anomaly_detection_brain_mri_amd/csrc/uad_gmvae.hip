// Spatial Gaussian-mixture VAE pieces that sit between the shared conv trunk and the loss
// (models/gaussian_mixture_variational_autoencoder_spatial.py:14-63, trainers/GMVAE_spatial.py:61-92):
//   * the latent heads on the inter_res x inter_res map (1x1 convs q(w|x), q(z|x), p(z|w,c), the mixture posterior pc and
//     the three prior terms), forward and backward, one workgroup per map location;
//   * deterministic reduction of the head weight gradients (outer products over all locations);
//   * the total-variation restore term's gradient w.r.t. the reconstruction.
// These are tiny next to the trunk (<= 0.1 % of the FLOPs): plain fp32 VALU code, LDS for the per-location vectors,
// no atomics.  Everything heavy (encoder / decoder k5 s2 convs) runs in uad_gemm.hip.
#include "uad_kernels.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

constexpr float kLogPi = 1.1447298858494002f;
constexpr int kMid = 64;   // p_z_wc/1x1convlayer width (gaussian_mixture_variational_autoencoder_spatial.py:36)

// LDS layout of one location (floats)
struct GmLds {
    int h, heads, ws, zs, a7, mid, M, Lq, logit, pc, red, total;
    // backward extras
    int dM, dLq, dheads, da7, dpc, dzs;
};
__host__ __device__ inline GmLds gm_lds(const UadGmArgs& a, bool bwd) {
    GmLds l;
    const int O = 2 * a.W + 2 * a.Z, Q = a.Z * a.C;
    int o = 0;
    l.h = o; o += a.cenc;
    l.heads = o; o += O;
    l.ws = o; o += a.W;
    l.zs = o; o += a.Z;
    l.a7 = o; o += kMid;
    l.mid = o; o += kMid;
    l.M = o; o += Q;
    l.Lq = o; o += Q;
    l.logit = o; o += a.C;
    l.pc = o; o += a.C;
    l.red = o; o += 16;
    l.dM = l.dLq = l.dheads = l.da7 = l.dpc = l.dzs = 0;
    if (bwd) {
        l.dM = o; o += Q;
        l.dLq = o; o += Q;
        l.dheads = o; o += O;
        l.da7 = o; o += kMid;
        l.dpc = o; o += a.C;
        l.dzs = o; o += a.Z;
    }
    l.total = o;
    return l;
}

// Forward of one location into LDS.  Returns (in s[red..red+2]) con, w-prior, c-prior terms of this location and in
// s[red+3] the raw closs1 (for the max gate of the backward).
__device__ void gm_location_forward(const UadGmArgs& a, const GmLds& L, float* s, int loc) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const int W = a.W, Z = a.Z, C = a.C, O = 2 * W + 2 * Z, Q = Z * C, CE = a.cenc;
    // h = LeakyReLU(BN(c_enc))   (customlayers.py:22-23 of the last encoder block)
    for (int c = tid; c < CE; c += nt) {
        const float bn = fmaf(a.c_enc[(size_t)loc * CE + c], a.scale[c] * a.mult, a.shift[c]);
        const float hv = bn > 0.f ? bn : bn * a.alpha;
        s[L.h + c] = hv;
        if (a.h_out) a.h_out[(size_t)loc * CE + c] = hv;
    }
    __syncthreads();
    // 1x1 heads: o in [0,W) w_mu, [W,2W) w_log_sigma, [2W,2W+Z) z_mu, [2W+Z,O) z_log_sigma
    for (int o = wave; o < O; o += nw) {
        const float* k; const float* b; int j, ld;
        if (o < W) { k = a.wmu_k; b = a.wmu_b; j = o; ld = W; }
        else if (o < 2 * W) { k = a.wls_k; b = a.wls_b; j = o - W; ld = W; }
        else if (o < 2 * W + Z) { k = a.zmu_k; b = a.zmu_b; j = o - 2 * W; ld = Z; }
        else { k = a.zls_k; b = a.zls_b; j = o - 2 * W - Z; ld = Z; }
        float acc = 0.f;
        for (int c = lane; c < CE; c += 64) acc = fmaf(s[L.h + c], k[(size_t)c * ld + j], acc);
        acc = wave_sum(acc);
        if (lane == 0) s[L.heads + o] = acc + b[j];
    }
    __syncthreads();
    // reparameterisations with log-VARIANCE heads (:27,32)
    for (int w = tid; w < W; w += nt)
        s[L.ws + w] = fmaf(a.eps_w ? a.eps_w[(size_t)loc * W + w] : 0.f, expf(0.5f * s[L.heads + W + w]), s[L.heads + w]);
    for (int z = tid; z < Z; z += nt)
        s[L.zs + z] = fmaf(a.eps_z ? a.eps_z[(size_t)loc * Z + z] : 0.f, expf(0.5f * s[L.heads + 2 * W + Z + z]), s[L.heads + 2 * W + z]);
    __syncthreads();
    for (int k = tid; k < kMid; k += nt) {
        float acc = a.c7_b[k];
        for (int w = 0; w < W; ++w) acc = fmaf(s[L.ws + w], a.c7_k[w * kMid + k], acc);
        s[L.a7 + k] = acc;
        s[L.mid + k] = acc > 0.f ? acc : 0.f;
    }
    __syncthreads();
    for (int q = tid; q < Q; q += nt) {
        float m = a.m_b[q], l = a.l_b[q] + a.var[q];
#pragma unroll 8
        for (int k = 0; k < kMid; ++k) {
            const float mv = s[L.mid + k];
            m = fmaf(mv, a.m_k[(size_t)k * Q + q], m);
            l = fmaf(mv, a.l_k[(size_t)k * Q + q], l);
        }
        s[L.M + q] = m;
        s[L.Lq + q] = l;
    }
    __syncthreads();
    // pc_logit[c] = sum_z loglh[z,c]  (:59-62)
    for (int c = tid; c < C; c += nt) {
        float acc = 0.f;
        for (int z = 0; z < Z; ++z) {
            const float D = s[L.zs + z] - s[L.M + z * C + c], lq = s[L.Lq + z * C + c];
            acc += -0.5f * (D * D * expf(lq)) - lq + kLogPi;
        }
        s[L.logit + c] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        float mx = s[L.logit];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, s[L.logit + c]);
        float sum = 0.f;
        for (int c = 0; c < C; ++c) { const float e = expf(s[L.logit + c] - mx); s[L.pc + c] = e; sum += e; }
        const float inv = 1.f / sum;
        float cl1 = 0.f;
        for (int c = 0; c < C; ++c) {
            const float p = s[L.pc + c] * inv;
            s[L.pc + c] = p;
            cl1 += p * logf(p * (float)C + 1e-8f);
        }
        float wl = 0.f;
        for (int w = 0; w < W; ++w) {
            const float mu = s[L.heads + w], ls = s[L.heads + W + w];
            wl += mu * mu + expf(ls) - ls - 1.f;
        }
        s[L.red + 1] = 0.5f * wl;
        s[L.red + 2] = fmaxf(cl1, a.c_lambda);
        s[L.red + 3] = cl1;
    }
    __syncthreads();
    // conditional prior: sum_{z,c} kl[z,c] * pc[c]   (trainers/GMVAE_spatial.py:69-75)
    float con = 0.f;
    for (int q = tid; q < Q; q += nt) {
        const int z = q / C, c = q - z * C;
        const float zls = s[L.heads + 2 * W + Z + z];
        const float D2 = s[L.heads + 2 * W + z] - s[L.M + q], lq = s[L.Lq + q];
        const float kl = 0.5f * ((expf(zls) + D2 * D2) * (expf(lq) + 1e-6f) - (lq + zls) - 1.f);
        con = fmaf(kl, s[L.pc + c], con);
    }
    con = wave_sum(con);
    if (lane == 0) s[L.red + 8 + wave] = con;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < nw; ++w) t += s[L.red + 8 + w];
        s[L.red + 0] = t;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) gm_heads_fwd_kernel(const UadGmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s[];
    const GmLds L = gm_lds(a, false);
    const int loc = blockIdx.x;
    gm_location_forward(a, L, s, loc);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int W = a.W, Z = a.Z, C = a.C;
    if (tid < 3) a.loc_loss[(size_t)loc * 3 + tid] = s[L.red + tid];
    if (a.w_mu) for (int w = tid; w < W; w += nt) a.w_mu[(size_t)loc * W + w] = s[L.heads + w];
    if (a.w_ls) for (int w = tid; w < W; w += nt) a.w_ls[(size_t)loc * W + w] = s[L.heads + W + w];
    if (a.z_mu) for (int z = tid; z < Z; z += nt) a.z_mu[(size_t)loc * Z + z] = s[L.heads + 2 * W + z];
    if (a.z_ls) for (int z = tid; z < Z; z += nt) a.z_ls[(size_t)loc * Z + z] = s[L.heads + 2 * W + Z + z];
    if (a.pc) for (int c = tid; c < C; c += nt) a.pc[(size_t)loc * C + c] = s[L.pc + c];
    if (a.zs_out) for (int z = tid; z < Z; z += nt) a.zs_out[(size_t)loc * Z + z] = s[L.zs + z];
}

// Backward of one location (recomputes its forward), fused with the activation backward of the last encoder block:
//   G[loc,:]   = (d_h_dec + d_h_heads) * lrelu'(bn) * gamma'          (d loss / d c of the last encoder conv)
//   colpart    = per-workgroup S1 = sum d_bn, S2 = sum d_bn * c       (BN gamma/beta/bias gradients, finalized later)
//   dvec_*     = per-location gradient vectors consumed by gm_heads_wgrad_kernel
__global__ void __launch_bounds__(256) gm_heads_bwd_kernel(const UadGmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s[];
    const GmLds L = gm_lds(a, true);
    const int loc = blockIdx.x;
    gm_location_forward(a, L, s, loc);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int W = a.W, Z = a.Z, C = a.C, O = 2 * W + 2 * Z, Q = Z * C, CE = a.cenc;
    const float g = a.inv_batch;
    // d pc
    for (int c = tid; c < C; c += nt) {
        float acc = 0.f;
        for (int z = 0; z < Z; ++z) {
            const float zls = s[L.heads + 2 * W + Z + z];
            const float D2 = s[L.heads + 2 * W + z] - s[L.M + z * C + c], lq = s[L.Lq + z * C + c];
            acc += 0.5f * ((expf(zls) + D2 * D2) * (expf(lq) + 1e-6f) - (lq + zls) - 1.f);
        }
        float d = g * acc;
        if (s[L.red + 3] >= a.c_lambda) {          // tf.maximum routes the gradient to closs1 where closs1 >= c_lambda
            const float pC = s[L.pc + c] * (float)C;
            d += g * (logf(pC + 1e-8f) + pC / (pC + 1e-8f));
        }
        s[L.dpc + c] = d;
    }
    __syncthreads();
    if (tid == 0) {
        float dot = 0.f;
        for (int c = 0; c < C; ++c) dot = fmaf(s[L.dpc + c], s[L.pc + c], dot);
        for (int c = 0; c < C; ++c) s[L.dpc + c] = s[L.pc + c] * (s[L.dpc + c] - dot);   // now d logit
    }
    __syncthreads();
    for (int q = tid; q < Q; q += nt) {
        const int z = q / C, c = q - z * C;
        const float zls = s[L.heads + 2 * W + Z + z], V = expf(zls);
        const float lq = s[L.Lq + q], E = expf(lq), E6 = E + 1e-6f;
        const float D = s[L.zs + z] - s[L.M + q], D2 = s[L.heads + 2 * W + z] - s[L.M + q];
        const float dll = s[L.dpc + c], dkl = g * s[L.pc + c];
        const float dM = dll * (D * E) - dkl * D2 * E6;
        const float dLq = dll * (-0.5f * D * D * E - 1.f) + dkl * 0.5f * ((V + D2 * D2) * E - 1.f);
        s[L.dM + q] = dM;
        s[L.dLq + q] = dLq;
        a.dvec_M[(size_t)loc * Q + q] = dM;
        a.dvec_Lq[(size_t)loc * Q + q] = dLq;
    }
    // d z_mu / d z_log_sigma (heads slots 2W.., 2W+Z..)
    for (int z = tid; z < Z; z += nt) {
        const float zls = s[L.heads + 2 * W + Z + z], V = expf(zls);
        float dzs = a.dz_dec ? a.dz_dec[(size_t)loc * Z + z] : 0.f, dzmu = 0.f, dzls = 0.f;
        for (int c = 0; c < C; ++c) {
            const int q = z * C + c;
            const float lq = s[L.Lq + q], E = expf(lq), E6 = E + 1e-6f;
            const float D = s[L.zs + z] - s[L.M + q], D2 = s[L.heads + 2 * W + z] - s[L.M + q];
            const float dll = s[L.dpc + c], dkl = g * s[L.pc + c];
            dzs += dll * (-D * E);
            dzmu += dkl * D2 * E6;
            dzls += dkl * 0.5f * (V * E6 - 1.f);
        }
        const float ez = a.eps_z ? a.eps_z[(size_t)loc * Z + z] : 0.f;
        s[L.dheads + 2 * W + z] = dzmu + dzs;
        s[L.dheads + 2 * W + Z + z] = dzls + dzs * ez * 0.5f * expf(0.5f * zls);
    }
    __syncthreads();
    // d mid -> d a7
    for (int k = tid; k < kMid; k += nt) {
        float acc = 0.f;
        for (int q = 0; q < Q; ++q) {
            acc = fmaf(a.m_k[(size_t)k * Q + q], s[L.dM + q], acc);
            acc = fmaf(a.l_k[(size_t)k * Q + q], s[L.dLq + q], acc);
        }
        const float d = s[L.a7 + k] > 0.f ? acc : 0.f;
        s[L.da7 + k] = d;
        a.dvec_a7[(size_t)loc * kMid + k] = d;
        a.mid_out[(size_t)loc * kMid + k] = s[L.mid + k];
    }
    __syncthreads();
    for (int w = tid; w < W; w += nt) {
        float dws = 0.f;
        for (int k = 0; k < kMid; ++k) dws = fmaf(a.c7_k[w * kMid + k], s[L.da7 + k], dws);
        const float mu = s[L.heads + w], ls = s[L.heads + W + w];
        const float ew = a.eps_w ? a.eps_w[(size_t)loc * W + w] : 0.f;
        s[L.dheads + w] = g * mu + dws;
        s[L.dheads + W + w] = g * 0.5f * (expf(ls) - 1.f) + dws * ew * 0.5f * expf(0.5f * ls);
        a.ws_out[(size_t)loc * W + w] = s[L.ws + w];
    }
    __syncthreads();
    for (int o = tid; o < O; o += nt) a.dvec_heads[(size_t)loc * O + o] = s[L.dheads + o];
    // d h (heads) + d h (decoder) -> activation backward of the last encoder block
    for (int c = tid; c < CE; c += nt) {
        float dh = a.dh_dec ? a.dh_dec[(size_t)loc * CE + c] : 0.f;
        for (int w = 0; w < W; ++w) {
            dh = fmaf(a.wmu_k[(size_t)c * W + w], s[L.dheads + w], dh);
            dh = fmaf(a.wls_k[(size_t)c * W + w], s[L.dheads + W + w], dh);
        }
        for (int z = 0; z < Z; ++z) {
            dh = fmaf(a.zmu_k[(size_t)c * Z + z], s[L.dheads + 2 * W + z], dh);
            dh = fmaf(a.zls_k[(size_t)c * Z + z], s[L.dheads + 2 * W + Z + z], dh);
        }
        const float cv = a.c_enc[(size_t)loc * CE + c];
        const float sc = a.scale[c] * a.mult;
        const float bn = fmaf(cv, sc, a.shift[c]);
        const float dbn = bn > 0.f ? dh : dh * a.alpha;
        a.g_out[(size_t)loc * CE + c] = dbn * sc;
        // per-location BN partials; reduced over locations by bn_grad_finalize (T = number of locations)
        a.colpart[(size_t)loc * 2 * CE + c] = dbn;
        a.colpart[(size_t)loc * 2 * CE + CE + c] = dbn * cv;
    }
}

// Head weight gradients: every head tensor is an outer-product sum over the L locations,
//   dK[i,j] = sum_l A[l,i] * B[l,j],  db[j] = sum_l B[l,j].
// The 15 jobs are laid out exactly like the flat parameter segment, so the reduced vector IS the gradient segment.
// grid = (ceil(HP/256), nchunks); partial[chunk][HP]; then reduce_partials.
__global__ void __launch_bounds__(256) gm_heads_wgrad_kernel(const UadGmWgradArgs a) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= a.total) return;
    int j = 0;
    while (j + 1 < a.njobs && e >= a.job[j + 1].off) ++j;
    const UadGmWgradArgs::Job jb = a.job[j];
    const int r = e - jb.off;
    const int i = r / jb.b, c = r - i * jb.b;
    const int l0 = blockIdx.y * a.chunk, l1 = min(l0 + a.chunk, a.L);
    float acc0 = 0.f, acc1 = 0.f;
    if (jb.A) {
        int l = l0;
        for (; l + 1 < l1; l += 2) {
            acc0 = fmaf(jb.A[(size_t)l * jb.lda + i], jb.B[(size_t)l * jb.ldb + c], acc0);
            acc1 = fmaf(jb.A[(size_t)(l + 1) * jb.lda + i], jb.B[(size_t)(l + 1) * jb.ldb + c], acc1);
        }
        if (l < l1) acc0 = fmaf(jb.A[(size_t)l * jb.lda + i], jb.B[(size_t)l * jb.ldb + c], acc0);
    } else {
        int l = l0;
        for (; l + 1 < l1; l += 2) { acc0 += jb.B[(size_t)l * jb.ldb + c]; acc1 += jb.B[(size_t)(l + 1) * jb.ldb + c]; }
        if (l < l1) acc0 += jb.B[(size_t)l * jb.ldb + c];
    }
    a.partial[(size_t)blockIdx.y * a.total + e] = acc0 + acc1;
}

// d xz_mu of  loss + sum_n tv_lambda * TV_n(x - xz_mu)  (trainers/GMVAE_spatial.py:57-58,90-92):
//   dxhat = sign(xz_mu - x) / N  -  tv_lambda * dTV/dr,   r = x - xz_mu
// (x's own direct gradient is exactly -dxhat; the input-gradient kernel subtracts it.)  Single-channel images.
__global__ void __launch_bounds__(256) tv_dxhat_kernel(const float* __restrict__ x, const float* __restrict__ xh, int N,
                                                       int H, int Wd, float inv_batch, float tv_lambda,
                                                       float* __restrict__ dxhat) {
    const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)N * H * Wd;
    if (pix >= total) return;
    const int j = (int)(pix % Wd), i = (int)((pix / Wd) % H);
    const float r = x[pix] - xh[pix];
    auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
    float tv = 0.f;
    if (i > 0) tv += sgn(r - (x[pix - Wd] - xh[pix - Wd]));
    if (i < H - 1) tv -= sgn((x[pix + Wd] - xh[pix + Wd]) - r);
    if (j > 0) tv += sgn(r - (x[pix - 1] - xh[pix - 1]));
    if (j < Wd - 1) tv -= sgn((x[pix + 1] - xh[pix + 1]) - r);
    dxhat[pix] = -sgn(r) * inv_batch - tv_lambda * tv;
}

// scalars = {mean_p_loss, conditional_prior_loss, loss, w_prior_loss, c_prior_loss, 0, 0, 0}
__global__ void __launch_bounds__(256) gm_loss_finalize_kernel(const float* __restrict__ rec_partial, int n, int bps,
                                                               const float* __restrict__ loc_loss, int lps,
                                                               float inv_batch, float* __restrict__ rec_per_sample,
                                                               float* __restrict__ scalars) {
    __shared__ float sh[4][256];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        float r = 0.f;
        for (int b = 0; b < bps; ++b) r += rec_partial[(size_t)i * bps + b];
        rec_per_sample[i] = r;
        a0 += r;
        for (int l = 0; l < lps; ++l) {
            const float* p = loc_loss + ((size_t)i * lps + l) * 3;
            a1 += p[0]; a2 += p[1]; a3 += p[2];
        }
    }
    sh[0][threadIdx.x] = a0; sh[1][threadIdx.x] = a1; sh[2][threadIdx.x] = a2; sh[3][threadIdx.x] = a3;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < 4; ++k) for (int i = 0; i < 256; ++i) t[k] += sh[k][i];
        scalars[0] = t[0] * inv_batch;
        scalars[1] = t[1] * inv_batch;
        scalars[2] = (t[0] + t[1] + t[2] + t[3]) * inv_batch;
        scalars[3] = t[2] * inv_batch;
        scalars[4] = t[3] * inv_batch;
        scalars[5] = scalars[6] = scalars[7] = 0.f;
    }
}

}  // namespace

size_t uad_gm_lds_bytes(const UadGmArgs& a, bool bwd) { return (size_t)gm_lds(a, bwd).total * sizeof(float); }

void uad_launch_gm_heads_fwd(const UadGmArgs& a, int locations, hipStream_t st) {
    hipLaunchKernelGGL(gm_heads_fwd_kernel, dim3(locations), dim3(256), uad_gm_lds_bytes(a, false), st, a);
}
void uad_launch_gm_heads_bwd(const UadGmArgs& a, int locations, hipStream_t st) {
    hipLaunchKernelGGL(gm_heads_bwd_kernel, dim3(locations), dim3(256), uad_gm_lds_bytes(a, true), st, a);
}
int uad_gm_wgrad_chunks(int L) { int c = (L + 63) / 64; return c < 1 ? 1 : (c > 64 ? 64 : c); }
void uad_launch_gm_heads_wgrad(UadGmWgradArgs a, float* out, hipStream_t st) {
    const int chunks = uad_gm_wgrad_chunks(a.L);
    a.chunk = (a.L + chunks - 1) / chunks;
    hipLaunchKernelGGL(gm_heads_wgrad_kernel, dim3((a.total + 255) / 256, chunks), dim3(256), 0, st, a);
    uad_launch_reduce_partials(a.partial, chunks, a.total, 1.0f, out, st);
}
void uad_launch_tv_dxhat(const float* x, const float* xh, int N, int H, int W, float inv_batch, float tv_lambda,
                         float* dxhat, hipStream_t st) {
    const size_t total = (size_t)N * H * W;
    hipLaunchKernelGGL(tv_dxhat_kernel, dim3((total + 255) / 256), dim3(256), 0, st, x, xh, N, H, W, inv_batch,
                       tv_lambda, dxhat);
}
void uad_launch_gm_loss_finalize(const float* rec_partial, int n, int bps, const float* loc_loss, int lps,
                                 float inv_batch, float* rec_per_sample, float* scalars, hipStream_t st) {
    hipLaunchKernelGGL(gm_loss_finalize_kernel, dim3(1), dim3(256), 0, st, rec_partial, n, bps, loc_loss, lps, inv_batch,
                       rec_per_sample, scalars);
}
