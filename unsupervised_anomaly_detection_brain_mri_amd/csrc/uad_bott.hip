// Dense bottleneck of the AE / VAE / ceVAE graphs as ONE workgroup per sample (1024 threads), forward and data-gradient backward
// (models/autoencoder.py:20-33, variational_autoencoder.py:20-40, context_encoder_variational_autoencoder.py:23-47).
// As a chain of batched GEMMs this part is ~16 dependent launches of tiny kernels (M = batch = 64 rows): 79 + 87 us of a 1.3 ms
// step with almost no arithmetic in it.  Per sample everything fits in LDS; the only real traffic is streaming the three
// dense kernels (1 MB + 0.5 MB, L2 resident) once per sample, done with 16 independent row loads in flight per thread.
// The parameter gradients of these layers stay batched GEMMs (uad_model.hip runs them on the side stream from the vectors
// this kernel leaves in global memory).
#include "uad_kernels.h"

namespace {

constexpr int NT = 1024;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// 1x1 conv with register tiling: out[p][o] = sum_k in[p][k] * w[k][o]  (in [P][K] and w [K][O] in LDS, O % 4 == 0).
// A thread owns 4 consecutive outputs of one position for a quarter of the K range (the 4 lanes of a quad split K and are
// combined with two shuffles): one scalar + one 16-byte LDS read per 4 FMAs instead of two scalar reads per FMA.
template <typename Epi>
__device__ __forceinline__ void conv1x1_tiled(const float* s_in, const float* s_w, int P, int K, int O, int tid, Epi epi) {
    const int kq = tid & 3;
    const int OQ = O / 4;
    for (int it = tid >> 2; it < P * OQ; it += NT / 4) {
        const int p = it / OQ, oq = it % OQ;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int kper = K / 4, k0 = kq * kper;
        for (int k = k0; k < k0 + kper; ++k) {
            const float x = s_in[p * K + k];
            const float4 w = *reinterpret_cast<const float4*>(s_w + k * O + oq * 4);
            acc.x = fmaf(x, w.x, acc.x); acc.y = fmaf(x, w.y, acc.y); acc.z = fmaf(x, w.z, acc.z); acc.w = fmaf(x, w.w, acc.w);
        }
        acc.x += __shfl_xor(acc.x, 1); acc.y += __shfl_xor(acc.y, 1); acc.z += __shfl_xor(acc.z, 1); acc.w += __shfl_xor(acc.w, 1);
        acc.x += __shfl_xor(acc.x, 2); acc.y += __shfl_xor(acc.y, 2); acc.z += __shfl_xor(acc.z, 2); acc.w += __shfl_xor(acc.w, 2);
        if (kq == 0) epi(p, oq * 4, acc);
    }
}

// out[o] (o < NO) = sum_k x[k] * W[k][o]  with W row-major [K][NO] in global memory (coalesced over o), x in LDS; the NT threads
// split into NT/NO k-groups, 16 independent row loads in flight per thread; partial sums through s_part [NT]; the caller
// finishes with the per-output reduction  sum_g s_part[g*NO + o].
__device__ __forceinline__ void gemv_cols_partial(const float* __restrict__ W, const float* x_lds, int K, int NO, float* s_part, int tid) {
    const int G = NT / NO;
    const int o = tid % NO, kg = tid / NO;
    const int kper = (K + G - 1) / G, k0 = kg * kper, k1 = min(k0 + kper, K);
    float acc = 0.f;
    int k = k0;
    for (; k + 32 <= k1; k += 32) {
        float w[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) w[u] = W[(size_t)(k + u) * NO + o];
#pragma unroll
        for (int u = 0; u < 32; ++u) acc = fmaf(x_lds[k + u], w[u], acc);
    }
    for (; k < k1; ++k) acc = fmaf(x_lds[k], W[(size_t)k * NO + o], acc);
    s_part[tid] = acc;
    __syncthreads();
}

__global__ void __launch_bounds__(NT) bottleneck_fwd_kernel(const UadBottArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, n = blockIdx.x;
    const int C = a.cenc, M = a.cmid, P = a.npos, F = a.npos * a.cmid, Z = a.zdim;
    float* s_h = sm;                 // [P][C]   activated encoder features
    float* s_t = s_h + P * C;        // [F]      flattened conv2d output
    float* s_part = s_t + F;         // [NT]
    float* s_z = s_part + NT;        // [Z]
    float* s_d = s_z + Z;            // [F]      dec_dense output (after dropout)
    float* s_w = s_d + F;            // [C*M]    1x1 kernels (conv2d, then conv2d_1)
    // stage h = lrelu(bn(c_enc)) and the conv2d kernel
    for (int i = tid; i < P * C; i += NT) {
        const int c = i % C;
        const float bn = fmaf(a.c_enc[(size_t)n * P * C + i], a.scale[c] * a.mult, a.shift[c]);
        s_h[i] = bn > 0.f ? bn : bn * a.alpha;
    }
    for (int i = tid; i < C * M; i += NT) s_w[i] = a.Wb[i];
    __syncthreads();
    // conv2d 1x1: t[p*M + j] = sum_c h[p][c] * Wb[c][j] + bb[j]
    conv1x1_tiled(s_h, s_w, P, C, M, tid, [&](int p, int o, const float4& v) {
        const float4 b = *reinterpret_cast<const float4*>(a.bb + o);
        const float4 r = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
        *reinterpret_cast<float4*>(s_t + p * M + o) = r;
        *reinterpret_cast<float4*>(a.t + (size_t)n * F + p * M + o) = r;
    });
    __syncthreads();
    const bool ctx = n >= a.n_vae;          // ceVAE context branch: z = z_mu_ce, no sampling / KL
    // dense heads: NO columns (2Z: mu | log-sigma from two [F][Z] kernels; AE: Z columns of dense_z), K = F split over NT/NO groups
    {
        const int NO = a.Wsg ? 2 * Z : Z;
        const int G = NT / NO;
        const int o2 = tid % NO, kg = tid / NO;
        const float* W = (o2 < Z) ? a.Wmu : a.Wsg;
        const int o = (o2 < Z) ? o2 : o2 - Z;
        const int kper = (F + G - 1) / G, k0 = kg * kper, k1 = min(k0 + kper, F);
        float acc = 0.f;
        int k = k0;
        for (; k + 32 <= k1; k += 32) {
            float w[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) w[u] = W[(size_t)(k + u) * Z + o];
#pragma unroll
            for (int u = 0; u < 32; ++u) acc = fmaf(s_t[k + u], w[u], acc);
        }
        for (; k < k1; ++k) acc = fmaf(s_t[k], W[(size_t)k * Z + o], acc);
        s_part[tid] = acc;
        __syncthreads();
        float klv = 0.f;
        if (tid < Z) {
            const size_t i = (size_t)n * Z + tid;
            float mr = a.bmu[tid];
            for (int g = 0; g < G; ++g) mr += s_part[g * NO + tid];
            if (a.Wsg) {
                float lr = a.bsg[tid];
                for (int g = 0; g < G; ++g) lr += s_part[g * NO + Z + tid];
                float mval = mr, l = 0.f, s = 1.f, zv;
                if (ctx) {
                    if (a.mask_mu_ce) mval *= a.mask_mu_ce[(size_t)(n - a.n_vae) * Z + tid];
                    zv = mval;
                } else {
                    l = lr;
                    if (a.mask_mu) mval *= a.mask_mu[i];
                    if (a.mask_ls) l *= a.mask_ls[i];
                    s = expf(l);
                    const float e = a.eps ? a.eps[i] : 0.f;
                    zv = fmaf(e, s, mval);
                    klv = mval * mval + s * s - 2.f * l - 1.f;
                }
                a.mu[i] = mval; a.ls[i] = l; a.sigma[i] = s; a.z[i] = zv;
                s_z[tid] = zv;
            } else {
                float zv = mr;
                if (a.mask_mu) zv *= a.mask_mu[i];
                a.z[i] = zv;
                s_z[tid] = zv;
            }
        }
        if (a.Wsg) {
            // KL of the sample: the first Z threads hold the terms
            klv = wave_sum(klv);
            __syncthreads();
            if ((tid & 63) == 0 && tid < Z) s_part[tid >> 6] = klv;
            __syncthreads();
            if (tid == 0) { float t = 0.f; for (int w = 0; w < (Z + 63) / 64; ++w) t += s_part[w]; a.kl[n] = 0.5f * t; }
        }
        __syncthreads();
    }
    // dec_dense: d[f] = (sum_k z[k] * Wd[k][f] + bd[f]) * mask_dec[f]      (F columns, K = Z)
    for (int f0 = 0; f0 < F; f0 += NT) {
        const int f = f0 + tid;
        if (f < F) {
            float acc = a.bd[f];
            for (int k = 0; k < Z; k += 16) {
                float w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = a.Wd[(size_t)(k + u) * F + f];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc = fmaf(s_z[k + u], w[u], acc);
            }
            if (a.mask_dec) acc *= a.mask_dec[(size_t)n * F + f];
            s_d[f] = acc;
            a.dvec[(size_t)n * F + f] = acc;
        }
    }
    for (int i = tid; i < C * M; i += NT) s_w[i] = a.Wr[i];      // conv2d_1 kernel [M][C]
    __syncthreads();
    // conv2d_1 1x1: cb[p][c] = sum_j d[p*M + j] * Wr[j][c] + br[c]
    conv1x1_tiled(s_d, s_w, P, M, C, tid, [&](int p, int c0, const float4& v) {
        const float4 b = *reinterpret_cast<const float4*>(a.br + c0);
        *reinterpret_cast<float4*>(a.cb + (size_t)n * P * C + p * C + c0) = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
    });
}

__global__ void __launch_bounds__(NT) bottleneck_bwd_kernel(const UadBottArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, n = blockIdx.x;
    const int C = a.cenc, M = a.cmid, P = a.npos, F = a.npos * a.cmid, Z = a.zdim;
    float* s_g = sm;                 // [P][C]   d loss / d cb, later d_bn of the last encoder block
    float* s_dd = s_g + P * C;       // [F]
    float* s_dz = s_dd + F;          // [2Z]     dmu | dls
    float* s_df = s_dz + 2 * Z;      // [F]      dflat
    float* s_w = s_df + F;           // [C*M]
    float* s_part = s_w + C * M;     // [NT]
    for (int i = tid; i < P * C; i += NT) {
        const float v = a.dcb[(size_t)n * P * C + i];
        s_g[i] = v;
        if (a.dcb_copy) a.dcb_copy[(size_t)n * P * C + i] = v;
    }
    // conv2d_1 kernel [M][C] staged transposed ([C][M]) so that it is the [K][O] operand of the data gradient
    for (int i = tid; i < C * M; i += NT) s_w[(i % C) * M + i / C] = a.Wr[i];
    __syncthreads();
    // d dec_dense output: dd[p*M + j] = (sum_c dcb[p][c] * Wr[j][c]) * mask_dec
    conv1x1_tiled(s_g, s_w, P, C, M, tid, [&](int p, int o, const float4& v) {
        const int f = p * M + o;
        float4 r = v;
        if (a.mask_dec) {
            const float4 mk = *reinterpret_cast<const float4*>(a.mask_dec + (size_t)n * F + f);
            r.x *= mk.x; r.y *= mk.y; r.z *= mk.z; r.w *= mk.w;
        }
        *reinterpret_cast<float4*>(s_dd + f) = r;
        *reinterpret_cast<float4*>(a.dd + (size_t)n * F + f) = r;
    });
    __syncthreads();
    // dz[k] = sum_f dd[f] * Wd^T[f][k]   (transposed copy: coalesced over k)
    gemv_cols_partial(a.WdT, s_dd, F, Z, s_part, tid);
    const bool ctx = n >= a.n_vae;
    if (tid < Z) {
        const size_t i = (size_t)n * Z + tid;
        float g = 0.f;
        for (int q = 0; q < NT / Z; ++q) g += s_part[q * Z + tid];
        float dm, dl = 0.f;
        if (a.Wsg) {
            if (ctx) {
                dm = a.mask_mu_ce ? g * a.mask_mu_ce[(size_t)(n - a.n_vae) * Z + tid] : g;
            } else {
                const float s = a.sigma[i], e = a.eps ? a.eps[i] : 0.f;
                dm = g + a.mu[i] * a.inv_batch;
                dl = g * e * s + (s * s - 1.f) * a.inv_batch;
                if (a.mask_mu) dm *= a.mask_mu[i];
                if (a.mask_ls) dl *= a.mask_ls[i];
            }
            a.dmu[i] = dm; a.dls[i] = dl;
        } else {
            dm = a.mask_mu ? g * a.mask_mu[i] : g;
            a.dmu[i] = dm;
        }
        s_dz[tid] = dm; s_dz[Z + tid] = dl;
    }
    __syncthreads();
    // dflat[r] = sum_o dmu[o] * Wmu^T[o][r] (+ dls[o] * Wsg^T[o][r])   (transposed copies: coalesced over r)
    for (int r0 = 0; r0 < F; r0 += NT) {
        const int r = r0 + tid;
        if (r < F) {
            float acc = 0.f;
            for (int o = 0; o < Z; o += 16) {
                float w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = a.WmuT[(size_t)(o + u) * F + r];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc = fmaf(s_dz[o + u], w[u], acc);
            }
            if (a.Wsg)
                for (int o = 0; o < Z; o += 16) {
                    float w[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) w[u] = a.WsgT[(size_t)(o + u) * F + r];
#pragma unroll
                    for (int u = 0; u < 16; ++u) acc = fmaf(s_dz[Z + o + u], w[u], acc);
                }
            s_df[r] = acc;
            a.dflat[(size_t)n * F + r] = acc;
        }
    }
    for (int i = tid; i < C * M; i += NT) s_w[(i % M) * C + i / M] = a.Wb[i];      // conv2d kernel [C][M] staged as [M][C] = [K][O]
    __syncthreads();
    // d h[p][c] = sum_j dflat[p*M + j] * Wb[c][j]; activation backward of the last encoder block; per-sample BN partials
    conv1x1_tiled(s_df, s_w, P, M, C, tid, [&](int p, int c0, const float4& v) {
        const size_t gi = (size_t)n * P * C + p * C + c0;
        const float4 cv = *reinterpret_cast<const float4*>(a.c_enc + gi);
        const float dh[4] = {v.x, v.y, v.z, v.w}, cc[4] = {cv.x, cv.y, cv.z, cv.w};
        float dbn[4], out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sc = a.scale[c0 + e] * a.mult;
            const float bn = fmaf(cc[e], sc, a.shift[c0 + e]);
            dbn[e] = bn > 0.f ? dh[e] : dh[e] * a.alpha;
            out[e] = dbn[e] * sc;
        }
        *reinterpret_cast<float4*>(a.g_out + gi) = make_float4(out[0], out[1], out[2], out[3]);
        *reinterpret_cast<float4*>(s_g + p * C + c0) = make_float4(dbn[0], dbn[1], dbn[2], dbn[3]);   // keep d_bn for the column sums
    });
    __syncthreads();
    if (tid < C) {
        float t1 = 0.f, t2 = 0.f;
        for (int p = 0; p < P; ++p) {
            const float dbn = s_g[p * C + tid];
            t1 += dbn;
            t2 = fmaf(dbn, a.c_enc[(size_t)n * P * C + p * C + tid], t2);
        }
        a.colpart[((size_t)n * 2 + 0) * C + tid] = t1;
        a.colpart[((size_t)n * 2 + 1) * C + tid] = t2;
    }
}

// out[c][r] = in[r][c]  (32x32 LDS tiles): transposed copies of the dense kernels for the backward's coalesced GEMVs
struct TransposeJobs { const float* in[3]; float* out[3]; int R[3], C[3]; };
__global__ void __launch_bounds__(256) transpose_kernel(const TransposeJobs jb) {
    __shared__ float tile[32][33];
    const float* __restrict__ in = jb.in[blockIdx.z];
    float* __restrict__ out = jb.out[blockIdx.z];
    const int R = jb.R[blockIdx.z], Cc = jb.C[blockIdx.z];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int y = ty; y < 32; y += 8)
        if (r0 + y < R && c0 + tx < Cc) tile[y][tx] = in[(size_t)(r0 + y) * Cc + c0 + tx];
    __syncthreads();
    for (int y = ty; y < 32; y += 8)
        if (c0 + y < Cc && r0 + tx < R) out[(size_t)(c0 + y) * R + r0 + tx] = tile[tx][y];
}

}  // namespace

size_t uad_bottleneck_lds_bytes(const UadBottArgs& a, bool bwd) {
    const size_t PC = (size_t)a.npos * a.cenc, F = (size_t)a.npos * a.cmid, CM = (size_t)a.cenc * a.cmid;
    return (bwd ? PC + F + 2 * a.zdim + F + CM + NT : PC + F + NT + a.zdim + F + CM) * sizeof(float);
}
bool uad_bottleneck_fused_ok(const UadBottArgs& a) {
    if (getenv("UAD_NO_FUSED_BOTT")) return false;
    const int NO = a.Wsg ? 2 * a.zdim : a.zdim;
    if (a.zdim % 16 || NO > NT || NT % NO || NT % a.zdim || a.cmid % 4 || a.cenc % 16 || (a.npos * a.cmid) % 4) return false;
    return uad_bottleneck_lds_bytes(a, false) <= 150 * 1024 && uad_bottleneck_lds_bytes(a, true) <= 150 * 1024;
}
static void bott_attrs() {
    static bool attr = false;
    if (attr) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bottleneck_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bottleneck_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
}
void uad_launch_bottleneck_fwd(const UadBottArgs& a, int n, hipStream_t st) {
    bott_attrs();
    hipLaunchKernelGGL(bottleneck_fwd_kernel, dim3(n), dim3(NT), uad_bottleneck_lds_bytes(a, false), st, a);
}
void uad_launch_bottleneck_bwd(const UadBottArgs& a, int n, hipStream_t st) {
    bott_attrs();
    hipLaunchKernelGGL(bottleneck_bwd_kernel, dim3(n), dim3(NT), uad_bottleneck_lds_bytes(a, true), st, a);
}
void uad_launch_transpose(const float* const* in, const int* R, const int* C, float* const* out, int njobs, hipStream_t st) {
    TransposeJobs jb;
    int gx = 1, gy = 1;
    for (int i = 0; i < 3; ++i) {
        const int k = i < njobs ? i : 0;
        jb.in[i] = in[k]; jb.out[i] = out[k]; jb.R[i] = R[k]; jb.C[i] = C[k];
        if ((C[k] + 31) / 32 > gx) gx = (C[k] + 31) / 32;
        if ((R[k] + 31) / 32 > gy) gy = (R[k] + 31) / 32;
    }
    hipLaunchKernelGGL(transpose_kernel, dim3(gx, gy, njobs), dim3(256), 0, st, jb);
}
