// uad_gmd.hip — dense GMVAE (models/gaussian_mixture_variational_autoencoder.py:11-76, trainers/GMVAE.py:56-94): the latent part.
// The latent sizes are tiny (defaults dim_w = dim_z = 1, dim_c = 6): every contraction that touches them is a skinny matrix product
// with one dimension of 1..a few hundred, far below an MFMA tile; they are HBM/latency-bound row reductions, so they run as
//   sd_tall   one wavefront per output element, lanes striding the long reduction axis (coalesced when that axis is contiguous)
//   sd_wide   one thread per output element, looping the short reduction axis
//   sd_wgrad  one thread per weight element, summing over the batch in a fixed order (bit-reproducible)
// and the per-sample mixture maths (reparameterisation, p(z|w,c), p(c|z), the three prior terms and their backward) is one
// workgroup per sample.
#include <hip/hip_runtime.h>

#include "uad_kernels.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float block_sum256(float v, float* red) {      // red: 4 floats of LDS
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(256) sd_tall_kernel(const UadSdArgs A) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long total = (long long)A.R * A.O;
    for (long long e = (long long)blockIdx.x * 4 + wave; e < total; e += (long long)gridDim.x * 4) {
        const int r = (int)(e / A.O), o = (int)(e % A.O);
        const float* a = A.a + (size_t)r * A.lda;
        const float* B = A.B + (size_t)o * A.sBo;
        float s = 0.f;
        for (int i = lane; i < A.I; i += 64) s = fmaf(a[i], B[(size_t)i * A.sBi], s);
        s = wave_sum(s);
        if (lane == 0) {
            if (A.bias) s += A.bias[o];
            if (A.mask) s *= A.mask[(size_t)r * A.ldm + o];
            float* dst = A.out + (size_t)r * A.ldo + o;
            *dst = A.accumulate ? *dst + s : s;
        }
    }
}
__global__ void __launch_bounds__(256) sd_wide_kernel(const UadSdArgs A) {
    const long long total = (long long)A.R * A.O;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int r = (int)(e / A.O), o = (int)(e % A.O);
    const float* a = A.a + (size_t)r * A.lda;
    const float* B = A.B + (size_t)o * A.sBo;
    float s = 0.f;
    for (int i = 0; i < A.I; ++i) s = fmaf(a[i], B[(size_t)i * A.sBi], s);
    if (A.bias) s += A.bias[o];
    if (A.mask) s *= A.mask[(size_t)r * A.ldm + o];
    float* dst = A.out + (size_t)r * A.ldo + o;
    *dst = A.accumulate ? *dst + s : s;
}
// dW[k*J + j] = sum_r a[r*lda + k] * g[r*ldg + j];  db[j] = sum_r g[r*ldg + j]  (threads K*J .. K*J+J-1)
__global__ void __launch_bounds__(256) sd_wgrad_kernel(const float* __restrict__ a, int lda, const float* __restrict__ g, int ldg, int R,
                                                       int K, int J, float* __restrict__ dW, float* __restrict__ db) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long nW = (long long)K * J;
    if (e >= nW + J) return;
    float s = 0.f;
    if (e < nW) {
        const int k = (int)(e / J), j = (int)(e % J);
        for (int r = 0; r < R; ++r) s = fmaf(a[(size_t)r * lda + k], g[(size_t)r * ldg + j], s);
        dW[e] = s;
    } else if (db) {
        const int j = (int)(e - nW);
        for (int r = 0; r < R; ++r) s += g[(size_t)r * ldg + j];
        db[j] = s;
    }
}

// ------------------------------------------------------------------------------------------------ per-sample mixture maths
// hv row layout: [w_mu W | w_log_sigma W | z_mu Z | z_log_sigma Z]
__global__ void __launch_bounds__(256) gmd_fwd_kernel(const UadGmdArgs A) {
    extern __shared__ float lds[];
    const int W = A.W, Z = A.Z, C = A.C, Q = Z * C, J = 2 * W + 2 * Z;
    float* ws = lds;              // [W]
    float* zs = ws + W;           // [Z]
    float* zm = zs + Z;           // [Z]
    float* zl = zm + Z;           // [Z]
    float* Ms = zl + Z;           // [Q]
    float* Ls = Ms + Q;           // [Q]
    float* pcs = Ls + Q;          // [C]
    float* red = pcs + C;         // [4]
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* hv = A.hv + (size_t)n * J;
    float* hvm = A.hvm + (size_t)n * J;
    float wl = 0.f;
    for (int w = tid; w < W; w += 256) {
        float mu = hv[w], ls = hv[W + w];
        if (A.mask_wmu) mu *= A.mask_wmu[(size_t)n * W + w];
        if (A.mask_wls) ls *= A.mask_wls[(size_t)n * W + w];
        hvm[w] = mu; hvm[W + w] = ls;
        const float e = A.e_w ? A.e_w[(size_t)n * W + w] : 0.f;
        const float s = mu + e * expf(0.5f * ls);
        ws[w] = s; A.w_s[(size_t)n * W + w] = s;
        wl += 0.5f * (mu * mu + expf(ls) - ls - 1.0f);                     // trainers/GMVAE.py:78
    }
    for (int z = tid; z < Z; z += 256) {
        float mu = hv[2 * W + z];
        const float ls = hv[2 * W + Z + z];                                 // no dropout on z_log_sigma (model :42)
        if (A.mask_zmu) mu *= A.mask_zmu[(size_t)n * Z + z];
        hvm[2 * W + z] = mu; hvm[2 * W + Z + z] = ls;
        const float e = A.e_z ? A.e_z[(size_t)n * Z + z] : 0.f;
        const float s = mu + e * expf(0.5f * ls);
        zs[z] = s; zm[z] = mu; zl[z] = ls; A.z_s[(size_t)n * Z + z] = s;
    }
    __syncthreads();
    for (int q = tid; q < Q; q += 256) {
        float mq = A.bm[q], lq = A.bl[q] + A.var[q];
        for (int w = 0; w < W; ++w) { mq = fmaf(ws[w], A.Wm[(size_t)w * Q + q], mq); lq = fmaf(ws[w], A.Wl[(size_t)w * Q + q], lq); }
        Ms[q] = mq; Ls[q] = lq;
        A.M[(size_t)n * Q + q] = mq; A.Lq[(size_t)n * Q + q] = lq;
    }
    __syncthreads();
    // p(c|z) (model :68-73): logit_c = sum_z -0.5 (z_s - M)^2 exp(Lq) - Lq + log(pi)
    for (int c = tid; c < C; c += 256) {
        float lg = 0.f;
        for (int z = 0; z < Z; ++z) {
            const float d = zs[z] - Ms[z * C + c], l = Ls[z * C + c];
            lg += -0.5f * (d * d * expf(l)) - l + 1.1447298858494002f;
        }
        pcs[c] = lg;
    }
    __syncthreads();
    if (tid == 0) {
        float mx = pcs[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, pcs[c]);
        float sum = 0.f;
        for (int c = 0; c < C; ++c) { pcs[c] = expf(pcs[c] - mx); sum += pcs[c]; }
        float cl = 0.f;
        for (int c = 0; c < C; ++c) {
            pcs[c] /= sum;
            cl += pcs[c] * logf(pcs[c] * (float)C + 1e-8f);                 // :85
        }
        A.loss3[2 * (size_t)A.nmax + n] = fmaxf(cl, A.c_lambda);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) A.pc[(size_t)n * C + c] = pcs[c];
    float con = 0.f;
    for (int q = tid; q < Q; q += 256) {
        const int z = q / C, c = q % C;
        const float d2 = zm[z] - Ms[q], l = Ls[q];
        con += 0.5f * ((expf(zl[z]) + d2 * d2) * (expf(l) + 1e-6f) - (l + zl[z]) - 1.0f) * pcs[c];      // :66-72
    }
    con = block_sum256(con, red);
    wl = block_sum256(wl, red);
    if (tid == 0) { A.loss3[n] = con; A.loss3[(size_t)A.nmax + n] = wl; }
}

__global__ void __launch_bounds__(256) gmd_bwd_kernel(const UadGmdArgs A) {
    extern __shared__ float lds[];
    const int W = A.W, Z = A.Z, C = A.C, Q = Z * C, J = 2 * W + 2 * Z;
    float* dMs = lds;             // [Q]
    float* dLs = dMs + Q;         // [Q]
    float* pcs = dLs + Q;         // [C]
    float* dlg = pcs + C;         // [C]
    const int n = blockIdx.x, tid = threadIdx.x;
    const float inv = A.inv;
    const float* hvm = A.hvm + (size_t)n * J;
    const float* M = A.M + (size_t)n * Q;
    const float* Lq = A.Lq + (size_t)n * Q;
    const float* zmu = hvm + 2 * W;
    const float* zls = hvm + 2 * W + Z;
    for (int c = tid; c < C; c += 256) {
        pcs[c] = A.pc[(size_t)n * C + c];
        float ks = 0.f;
        for (int z = 0; z < Z; ++z) {
            const float d2 = zmu[z] - M[z * C + c], l = Lq[z * C + c];
            ks += 0.5f * ((expf(zls[z]) + d2 * d2) * (expf(l) + 1e-6f) - (l + zls[z]) - 1.0f);
        }
        dlg[c] = inv * ks;                                                   // d con / d pc
    }
    __syncthreads();
    if (tid == 0) {
        float cl = 0.f;
        for (int c = 0; c < C; ++c) cl += pcs[c] * logf(pcs[c] * (float)C + 1e-8f);
        const bool act = cl >= A.c_lambda;                                   // tf.maximum routes the gradient to x where x >= y
        float dot = 0.f;
        for (int c = 0; c < C; ++c) {
            const float pC = pcs[c] * (float)C;
            if (act) dlg[c] += inv * (logf(pC + 1e-8f) + pC / (pC + 1e-8f));
            dot += dlg[c] * pcs[c];
        }
        for (int c = 0; c < C; ++c) dlg[c] = pcs[c] * (dlg[c] - dot);       // softmax backward -> d / d logit
    }
    __syncthreads();
    float* dhv = A.dhv + (size_t)n * J;
    for (int z = tid; z < Z; z += 256) {
        const float zs = A.z_s[(size_t)n * Z + z], mu = zmu[z], ls = zls[z], V = expf(ls);
        float dzs = A.dz_dec ? A.dz_dec[(size_t)n * Z + z] : 0.f, dmu = 0.f, dls = 0.f;
        for (int c = 0; c < C; ++c) {
            const int q = z * C + c;
            const float l = Lq[q], E = expf(l), E6 = E + 1e-6f, D = zs - M[q], D2 = mu - M[q], dkl = inv * pcs[c], dl = dlg[c];
            dzs += dl * (-D * E);
            const float dm = dl * (D * E) - dkl * D2 * E6;
            const float dq = dl * (-0.5f * D * D * E - 1.0f) + dkl * 0.5f * ((V + D2 * D2) * E - 1.0f);
            dMs[q] = dm; dLs[q] = dq;
            A.dM[(size_t)n * Q + q] = dm; A.dLq[(size_t)n * Q + q] = dq;
            dmu += dkl * D2 * E6;
            dls += dkl * 0.5f * (V * E6 - 1.0f);
        }
        const float e = A.e_z ? A.e_z[(size_t)n * Z + z] : 0.f;
        float gmu = dmu + dzs;
        if (A.mask_zmu) gmu *= A.mask_zmu[(size_t)n * Z + z];
        dhv[2 * W + z] = gmu;
        dhv[2 * W + Z + z] = dls + dzs * e * 0.5f * expf(0.5f * ls);
    }
    __syncthreads();
    for (int w = tid; w < W; w += 256) {
        float dws = 0.f;
        for (int q = 0; q < Q; ++q) dws += dMs[q] * A.Wm[(size_t)w * Q + q] + dLs[q] * A.Wl[(size_t)w * Q + q];
        const float mu = hvm[w], ls = hvm[W + w];
        const float e = A.e_w ? A.e_w[(size_t)n * W + w] : 0.f;
        float gmu = inv * mu + dws;
        float gls = inv * 0.5f * (expf(ls) - 1.0f) + dws * e * 0.5f * expf(0.5f * ls);
        if (A.mask_wmu) gmu *= A.mask_wmu[(size_t)n * W + w];
        if (A.mask_wls) gls *= A.mask_wls[(size_t)n * W + w];
        dhv[w] = gmu; dhv[W + w] = gls;
    }
}

}  // namespace

void uad_launch_sd(const UadSdArgs& a, hipStream_t st) {
    const long long total = (long long)a.R * a.O;
    if (total <= 0) return;
    if (a.I >= 64) {
        long long blocks = (total + 3) / 4;
        if (blocks > 65535) blocks = 65535;
        hipLaunchKernelGGL(sd_tall_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL(sd_wide_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    }
}
void uad_launch_sd_wgrad(const float* a, int lda, const float* g, int ldg, int R, int K, int J, float* dW, float* db, hipStream_t st) {
    const long long total = (long long)K * J + J;
    hipLaunchKernelGGL(sd_wgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, lda, g, ldg, R, K, J, dW, db);
}
size_t uad_gmd_lds_bytes(int W, int Z, int C) {
    const size_t fwd = (size_t)W + 3 * Z + 2 * (size_t)Z * C + C + 4, bwd = 2 * (size_t)Z * C + 2 * C;
    return (fwd > bwd ? fwd : bwd) * sizeof(float);
}
void uad_launch_gmd_fwd(const UadGmdArgs& a, int n, hipStream_t st) {
    hipLaunchKernelGGL(gmd_fwd_kernel, dim3(n), dim3(256), uad_gmd_lds_bytes(a.W, a.Z, a.C), st, a);
}
void uad_launch_gmd_bwd(const UadGmdArgs& a, int n, hipStream_t st) {
    hipLaunchKernelGGL(gmd_bwd_kernel, dim3(n), dim3(256), uad_gmd_lds_bytes(a.W, a.Z, a.C), st, a);
}
