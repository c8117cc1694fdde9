// Dense bottleneck of the AE / VAE / ceVAE graphs as ONE group of workgroups per sample (4 x 1024 threads, or one workgroup when the
// shapes do not split), forward and data-gradient backward
// (models/autoencoder.py:20-33, variational_autoencoder.py:20-40, context_encoder_variational_autoencoder.py:23-47).
// As a chain of batched GEMMs this part is ~16 dependent launches of tiny kernels (M = batch = 64 rows): 79 + 87 us of a 1.3 ms
// step with almost no arithmetic in it.  Per sample everything fits in LDS; the only real traffic is streaming the three
// dense kernels (1 MB + 0.5 MB, L2 resident) once per sample, done with 16 independent row loads in flight per thread.
// The parameter gradients of these layers stay batched GEMMs (uad_model.hip runs them on the side stream from the vectors
// this kernel leaves in global memory).
#include "uad_kernels.h"

namespace {

constexpr int NT = 1024;

// UAD_BOTT_DBG=1: phase stamps (100 MHz wall clock) of workgroup 0, printed by the launcher for its first calls
#define STAMP(i) do { if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[i] = wall_clock64(); } while (0)

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
        // the 4 lanes of a quad take k = kq, kq + 4, ...: their s_in words sit in adjacent banks and their s_w rows in different ones
        // (contiguous K quarters put all 64 lanes of a wave on one bank: a 16-way conflict on every read)
        for (int k = kq; k < K; k += 4) {
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

// ---- exchange between the Q workgroups that share one sample ----
// Every workgroup owns 1/Q of the positions (and so of the flattened features); the two reductions that run over ALL features
// (dense heads forward, dec_dense data gradient backward) are computed as Q partial vectors and combined in a fixed order
// (deterministic) after ONE exchange through global memory: xch[(n*Q + q) * xw + i], flags[n*Q + q] = epoch of this launch.
// The Q workgroups of a sample have adjacent block ids, so whenever one of them is resident its siblings are the next to be
// dispatched: the spin cannot starve them (resident complete groups finish on their own and free their slots).
template <int Q>
__device__ __forceinline__ void group_exchange(const UadBottArgs& a, int n, int q, int tid, int count, float mine, float* s_tot) {
    if (Q == 1) {
        if (tid < count) s_tot[tid] = mine;
        __syncthreads();
        return;
    }
    // Agent-scope RELAXED atomics move the data and the flag: they are performed at the device's coherence point (the workgroups of a
    // group sit on different XCDs, i.e. behind different L2s), and "data before flag" is enforced by draining the stores (s_waitcnt)
    // before the barrier that precedes the flag store.  A release fence instead writes back the whole dirty L2 of the XCD: 20 us here.
    if (tid < count) {
        __hip_atomic_store(a.xch + ((size_t)n * Q + q) * a.xw + tid, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0 && !(a.fault && n == 0 && q == 1)) __hip_atomic_store(a.flags + n * Q + q, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < Q) {
        // Bounded: the argument above needs in-order dispatch per XCD; if that ever fails (a serialising tool, a masked queue) the step must not
        // hang the device.  ~2^22 polls of ~0.5-1 us: seconds, far beyond any preemption of a healthy run.  The workgroup then carries on with
        // whatever the slots hold and reports through the host-visible word; the next uad_forward() fails with that report.
        unsigned polls = 0;
        while (__hip_atomic_load(a.flags + n * Q + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (++polls > (1u << 22)) {
                // the FIRST fault wins (compare-and-swap from 0): the host counts the optimizer calls since THAT launch epoch, every one of which the
                // device word has made a no-op; a later timeout must not move the epoch forward
                unsigned zero = 0u;
                if (a.err) __hip_atomic_compare_exchange_strong(a.err, &zero, a.epoch | 0x80000000u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                zero = 0u;
                if (a.err_dev) __hip_atomic_compare_exchange_strong(a.err_dev, &zero, a.epoch | 0x80000000u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // what the optimizer kernels read
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __syncthreads();
    if (tid < count) {
        float t = 0.f;
#pragma unroll
        for (int qq = 0; qq < Q; ++qq) t += __hip_atomic_load(a.xch + ((size_t)n * Q + qq) * a.xw + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_tot[tid] = t;
    }
    __syncthreads();
}

// out[col] (col in this workgroup's slice [c0, c0 + NC) of a row-major [K][ld] matrix, coalesced over col) = sum_k x[k] * W[k][c0 + col]:
// NT threads = G k-groups x CP columns per pass (G > 1 when the slice is narrower than the workgroup), 16 independent row loads in
// flight per thread, group partials through s_part.  fin(col, sum) runs on the k-group-0 thread of each column.
template <typename Fin>
__device__ __forceinline__ void gemv_slice(const float* __restrict__ W, size_t ld, int c0, int NC, const float* x, int K, float* s_part, int tid, Fin fin) {
    const int G = NC < NT ? NT / NC : 1;
    const int CP = NT / G;
    const int kg = tid / CP, cl = tid % CP;
    const int kper = K / G;
    for (int cb = 0; cb < NC; cb += CP) {
        const int col = cb + cl;
        float acc = 0.f;
        if (col < NC)
            for (int k = kg * kper; k < (kg + 1) * kper; k += 16) {
                float w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = W[(size_t)(k + u) * ld + c0 + col];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc = fmaf(x[k + u], w[u], acc);
            }
        if (G > 1) {
            s_part[tid] = acc;
            __syncthreads();
            if (kg == 0 && col < NC)
                for (int g = 1; g < G; ++g) acc += s_part[g * CP + cl];
        }
        if (kg == 0 && col < NC) fin(col, acc);
        if (G > 1) __syncthreads();
    }
}

template <int Q>
__global__ void __launch_bounds__(NT) bottleneck_fwd_kernel(const UadBottArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, n = blockIdx.x / Q, q = blockIdx.x % Q;
    const int C = a.cenc, M = a.cmid, P = a.npos, F = a.npos * a.cmid, Z = a.zdim;
    const int Pq = P / Q, Fq = Pq * M, p0 = q * Pq, f0 = q * Fq;     // this workgroup's positions / flattened features
    float* s_h = sm;                 // [Pq][C]  activated encoder features
    float* s_t = s_h + Pq * C;       // [Fq]     flattened conv2d output
    float* s_part = s_t + Fq;        // [NT]
    float* s_z = s_part + NT;        // [Z]
    float* s_d = s_z + Z;            // [Fq]     dec_dense output (after dropout)
    float* s_w = s_d + Fq;           // [C*M]    1x1 kernels (conv2d, then conv2d_1)
    STAMP(0);
    // stage h = lrelu(bn(c_enc)) and the conv2d kernel
    const size_t hb = ((size_t)n * P + p0) * C;
    for (int i = tid; i < Pq * C; i += NT) {
        const int c = i % C;
        const float bn = fmaf(a.c_enc[hb + i], a.scale[c] * a.mult, a.shift[c]);
        s_h[i] = bn > 0.f ? bn : bn * a.alpha;
    }
    for (int i = tid; i < C * M; i += NT) s_w[i] = a.Wb[i];
    __syncthreads();
    STAMP(1);
    // conv2d 1x1: t[p*M + j] = sum_c h[p][c] * Wb[c][j] + bb[j]
    conv1x1_tiled(s_h, s_w, Pq, C, M, tid, [&](int p, int o, const float4& v) {
        const float4 b = *reinterpret_cast<const float4*>(a.bb + o);
        const float4 r = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
        *reinterpret_cast<float4*>(s_t + p * M + o) = r;
        *reinterpret_cast<float4*>(a.t + (size_t)n * F + f0 + p * M + o) = r;
    });
    __syncthreads();
    STAMP(2);
    const bool ctx = n >= a.n_vae;          // ceVAE context branch: z = z_mu_ce, no sampling / KL
    // dense heads: NO columns (2Z: mu | log-sigma from two [F][Z] kernels; AE: Z columns of dense_z); this workgroup contracts its
    // Fq features (split over NT/NO groups), the group's partial vectors are exchanged
    {
        const int NO = a.Wsg ? 2 * Z : Z;
        const int G = NT / NO;
        const int o2 = tid % NO, kg = tid / NO;
        const float* W = (o2 < Z) ? a.Wmu : a.Wsg;
        const int o = (o2 < Z) ? o2 : o2 - Z;
        const int kper = (Fq + G - 1) / G, k0 = kg * kper, k1 = min(k0 + kper, Fq);
        float acc = 0.f;
        int k = k0;
        for (; k + 32 <= k1; k += 32) {
            float w[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) w[u] = W[(size_t)(f0 + k + u) * Z + o];
#pragma unroll
            for (int u = 0; u < 32; ++u) acc = fmaf(s_t[k + u], w[u], acc);
        }
        for (; k < k1; ++k) acc = fmaf(s_t[k], W[(size_t)(f0 + k) * Z + o], acc);
        s_part[tid] = acc;
        __syncthreads();
        float mine = 0.f;
        if (tid < NO)
            for (int g = 0; g < G; ++g) mine += s_part[g * NO + tid];
        __syncthreads();
        STAMP(3);
        group_exchange<Q>(a, n, q, tid, NO, mine, s_part);
        STAMP(4);      // s_part[0 .. NO) = head pre-activations without bias
        float klv = 0.f;
        if (tid < Z) {
            const size_t i = (size_t)n * Z + tid;
            const float mr = a.bmu[tid] + s_part[tid];
            if (a.Wsg) {
                const float lr = a.bsg[tid] + s_part[Z + tid];
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
                if (q == 0) { a.mu[i] = mval; a.ls[i] = l; a.sigma[i] = s; a.z[i] = zv; }
                s_z[tid] = zv;
            } else {
                float zv = mr;
                if (a.mask_mu) zv *= a.mask_mu[i];
                if (q == 0) a.z[i] = zv;
                s_z[tid] = zv;
            }
        }
        if (a.Wsg) {
            // KL of the sample: the first Z threads hold the terms
            klv = wave_sum(klv);
            __syncthreads();
            if ((tid & 63) == 0 && tid < Z) s_part[tid >> 6] = klv;
            __syncthreads();
            if (tid == 0 && q == 0) { float t = 0.f; for (int w = 0; w < (Z + 63) / 64; ++w) t += s_part[w]; a.kl[n] = 0.5f * t; }
        }
        __syncthreads();
    }
    STAMP(5);
    // dec_dense: d[f] = (sum_k z[k] * Wd[k][f] + bd[f]) * mask_dec[f]      (this workgroup's Fq columns, K = Z)
    gemv_slice(a.Wd, (size_t)F, f0, Fq, s_z, Z, s_part, tid, [&](int fl, float acc) {
        acc += a.bd[f0 + fl];
        if (a.mask_dec) acc *= a.mask_dec[(size_t)n * F + f0 + fl];
        s_d[fl] = acc;
        a.dvec[(size_t)n * F + f0 + fl] = acc;
    });
    STAMP(6);
    for (int i = tid; i < C * M; i += NT) s_w[i] = a.Wr[i];      // conv2d_1 kernel [M][C]
    __syncthreads();
    STAMP(7);
    // conv2d_1 1x1: cb[p][c] = sum_j d[p*M + j] * Wr[j][c] + br[c]
    conv1x1_tiled(s_d, s_w, Pq, M, C, tid, [&](int p, int c0, const float4& v) {
        const float4 b = *reinterpret_cast<const float4*>(a.br + c0);
        *reinterpret_cast<float4*>(a.cb + ((size_t)n * P + p0 + p) * C + c0) = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
    });
    __syncthreads();
    STAMP(8);
}

template <int Q>
__global__ void __launch_bounds__(NT) bottleneck_bwd_kernel(const UadBottArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, n = blockIdx.x / Q, q = blockIdx.x % Q;
    const int C = a.cenc, M = a.cmid, P = a.npos, F = a.npos * a.cmid, Z = a.zdim;
    const int Pq = P / Q, Fq = Pq * M, p0 = q * Pq, f0 = q * Fq;
    float* s_g = sm;                 // [Pq][C]  d loss / d cb, later d_bn of the last encoder block
    float* s_dd = s_g + Pq * C;      // [Fq]
    float* s_dz = s_dd + Fq;         // [2Z]     dmu | dls
    float* s_df = s_dz + 2 * Z;      // [Fq]     dflat
    float* s_w = s_df + Fq;          // [max(C*M, Pq*C)]
    float* s_part = s_w + max(C * M, Pq * C);     // [NT]
    const size_t gb = ((size_t)n * P + p0) * C;
    for (int i = tid; i < Pq * C; i += NT) {
        const float v = a.dcb[gb + i];
        s_g[i] = v;
        if (a.dcb_copy) a.dcb_copy[gb + i] = v;
    }
    // conv2d_1 kernel [M][C] staged transposed ([C][M]) so that it is the [K][O] operand of the data gradient
    for (int i = tid; i < C * M; i += NT) s_w[(i % C) * M + i / C] = a.Wr[i];
    float* wp = a.wpart ? a.wpart + (size_t)blockIdx.x * (2 * C * M + M) : nullptr;      // [dWb (C x M) | dWr (M x C) | db_b (M)]
    if (wp) for (int i = tid; i < Fq; i += NT) s_df[i] = a.dvec[(size_t)n * F + f0 + i];
    __syncthreads();
    if (wp) {
        // this workgroup's share of conv2d_1's kernel gradient: dWr[j][c] = sum_p dvec[p*M + j] * dcb[p][c] over its positions
        for (int o = tid; o < M * C; o += NT) {
            const int j = o / C, c = o % C;
            float acc = 0.f;
            for (int p = 0; p < Pq; ++p) acc = fmaf(s_df[p * M + j], s_g[p * C + c], acc);
            wp[C * M + o] = acc;
        }
        __syncthreads();      // s_df is rewritten below
    }
    // d dec_dense output: dd[p*M + j] = (sum_c dcb[p][c] * Wr[j][c]) * mask_dec
    conv1x1_tiled(s_g, s_w, Pq, C, M, tid, [&](int p, int o, const float4& v) {
        const int fl = p * M + o;
        float4 r = v;
        if (a.mask_dec) {
            const float4 mk = *reinterpret_cast<const float4*>(a.mask_dec + (size_t)n * F + f0 + fl);
            r.x *= mk.x; r.y *= mk.y; r.z *= mk.z; r.w *= mk.w;
        }
        *reinterpret_cast<float4*>(s_dd + fl) = r;
        *reinterpret_cast<float4*>(a.dd + (size_t)n * F + f0 + fl) = r;
    });
    __syncthreads();
    // dz[k] = sum_f dd[f] * Wd^T[f][k]   (transposed copy: coalesced over k); this workgroup's Fq rows, then the exchange
    gemv_cols_partial(a.WdT + (size_t)f0 * Z, s_dd, Fq, Z, s_part, tid);
    float mine = 0.f;
    if (tid < Z)
        for (int g = 0; g < NT / Z; ++g) mine += s_part[g * Z + tid];
    __syncthreads();
    group_exchange<Q>(a, n, q, tid, Z, mine, s_part);
    const bool ctx = n >= a.n_vae;
    if (tid < Z) {
        const size_t i = (size_t)n * Z + tid;
        const float g = s_part[tid];
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
            if (q == 0) { a.dmu[i] = dm; a.dls[i] = dl; }
        } else {
            dm = a.mask_mu ? g * a.mask_mu[i] : g;
            if (q == 0) a.dmu[i] = dm;
        }
        s_dz[tid] = dm; s_dz[Z + tid] = dl;
    }
    __syncthreads();
    // dflat[r] = sum_o dmu[o] * Wmu^T[o][r] (+ dls[o] * Wsg^T[o][r])   (transposed copies: coalesced over r); this workgroup's Fq columns
    gemv_slice(a.WmuT, (size_t)F, f0, Fq, s_dz, Z, s_part, tid, [&](int fl, float acc) { s_df[fl] = acc; });
    if (a.Wsg) {
        __syncthreads();
        gemv_slice(a.WsgT, (size_t)F, f0, Fq, s_dz + Z, Z, s_part, tid, [&](int fl, float acc) { s_df[fl] += acc; });
    }
    __syncthreads();
    for (int fl = tid; fl < Fq; fl += NT) a.dflat[(size_t)n * F + f0 + fl] = s_df[fl];
    for (int i = tid; i < C * M; i += NT) s_w[(i % M) * C + i / M] = a.Wb[i];      // conv2d kernel [C][M] staged as [M][C] = [K][O]
    __syncthreads();
    // d h[p][c] = sum_j dflat[p*M + j] * Wb[c][j]; activation backward of the last encoder block; per-workgroup BN partials
    conv1x1_tiled(s_df, s_w, Pq, M, C, tid, [&](int p, int c0, const float4& v) {
        const size_t gi = gb + p * C + c0;
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
        for (int p = 0; p < Pq; ++p) {
            const float dbn = s_g[p * C + tid];
            t1 += dbn;
            t2 = fmaf(dbn, a.c_enc[gb + p * C + tid], t2);
        }
        a.colpart[((size_t)blockIdx.x * 2 + 0) * C + tid] = t1;      // rows: one per workgroup (uad_bottleneck_colpart_rows)
        a.colpart[((size_t)blockIdx.x * 2 + 1) * C + tid] = t2;
    }
    if (wp) {
        // this workgroup's share of conv2d's kernel / bias gradient: dWb[c][j] = sum_p h[p][c] * dflat[p*M + j], db_b[j] = sum_p dflat[p*M + j]
        __syncthreads();
        for (int i = tid; i < Pq * C; i += NT) {
            const int c = i % C;
            const float bn = fmaf(a.c_enc[gb + i], a.scale[c] * a.mult, a.shift[c]);
            s_w[i] = bn > 0.f ? bn : bn * a.alpha;      // h (s_w holds max(C*M, Pq*C) floats: uad_bottleneck_lds_bytes)
        }
        __syncthreads();
        for (int o = tid; o < C * M; o += NT) {
            const int c = o / M, j = o % M;
            float acc = 0.f;
            for (int p = 0; p < Pq; ++p) acc = fmaf(s_w[p * C + c], s_df[p * M + j], acc);
            wp[o] = acc;
        }
        if (tid < M) {
            float acc = 0.f;
            for (int p = 0; p < Pq; ++p) acc += s_df[p * M + tid];
            wp[2 * C * M + tid] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// All parameter gradients of the dense bottleneck in ONE launch (they used to be 5 split-K GEMM launches + 4 column sums + their
// reductions on the side stream: ~175 us there while it overlaps the encoder backward, and the critical path of that phase).
//   part A  (K = batch):      dWd [Z][F] = z^T dd,  dWmu [F][Z] = t^T dmu,  dWsg [F][Z] = t^T dls,  bias gradients = column sums of
//                             dd / dmu / dls; 32 x 64 output tiles, the whole batch staged in LDS in chunks of 64 samples
//   part B  (K = batch * P):  dWb [C][M] = h^T dflat,  dWr [M][C] = dvec^T dcb,  db_b = column sums of dflat: every workgroup of the
//                             backward kernel leaves its positions' share (it has dcb, dflat and the c_enc rows at hand), this kernel
//                             sums the shares.
// Deterministic: every sum runs in a fixed order.
constexpr int WG_NT = 256, WG_TI = 32, WG_TJ = 64;

__device__ __forceinline__ void wgrad_dense_tile(const float* __restrict__ A, int I, const float* __restrict__ B, int J, int n, int ti, int tj,
                                                 float* __restrict__ out, float* __restrict__ dbias, float* sm) {
    // out[i][j] = sum_s A[s][i] * B[s][j] for the tile (ti, tj); dbias[j] = sum_s B[s][j] (written by the ti == 0 tiles)
    float* sA = sm;                      // [64][WG_TI]
    float* sB = sm + 64 * WG_TI;         // [64][WG_TJ]
    const int tid = threadIdx.x;
    const int i0 = ti * WG_TI, j0 = tj * WG_TJ;
    const int ip = tid / 16, jq = tid % 16;            // this thread: rows 2*ip, 2*ip+1; columns 4*jq .. 4*jq+3
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float bsum = 0.f;
    for (int s0 = 0; s0 < n; s0 += 64) {
        const int ns = min(64, n - s0);
        __syncthreads();
        for (int e = tid; e < 64 * WG_TI / 4; e += WG_NT) {
            const int r = e / (WG_TI / 4), c4 = e % (WG_TI / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < ns) v = *reinterpret_cast<const float4*>(A + (size_t)(s0 + r) * I + i0 + c4 * 4);
            *reinterpret_cast<float4*>(sA + r * WG_TI + c4 * 4) = v;
        }
        for (int e = tid; e < 64 * WG_TJ / 4; e += WG_NT) {
            const int r = e / (WG_TJ / 4), c4 = e % (WG_TJ / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < ns) v = *reinterpret_cast<const float4*>(B + (size_t)(s0 + r) * J + j0 + c4 * 4);
            *reinterpret_cast<float4*>(sB + r * WG_TJ + c4 * 4) = v;
        }
        __syncthreads();
#pragma unroll 8
        for (int r = 0; r < 64; ++r) {
            const float2 a2 = *reinterpret_cast<const float2*>(sA + r * WG_TI + 2 * ip);
            const float4 b4 = *reinterpret_cast<const float4*>(sB + r * WG_TJ + 4 * jq);
            acc[0][0] = fmaf(a2.x, b4.x, acc[0][0]); acc[0][1] = fmaf(a2.x, b4.y, acc[0][1]);
            acc[0][2] = fmaf(a2.x, b4.z, acc[0][2]); acc[0][3] = fmaf(a2.x, b4.w, acc[0][3]);
            acc[1][0] = fmaf(a2.y, b4.x, acc[1][0]); acc[1][1] = fmaf(a2.y, b4.y, acc[1][1]);
            acc[1][2] = fmaf(a2.y, b4.z, acc[1][2]); acc[1][3] = fmaf(a2.y, b4.w, acc[1][3]);
        }
        if (ti == 0 && dbias && tid < WG_TJ)
            for (int r = 0; r < 64; ++r) bsum += sB[r * WG_TJ + tid];
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
        *reinterpret_cast<float4*>(out + (size_t)(i0 + 2 * ip + e) * J + j0 + 4 * jq) = make_float4(acc[e][0], acc[e][1], acc[e][2], acc[e][3]);
    if (ti == 0 && dbias && tid < WG_TJ) dbias[j0 + tid] = bsum;
}

__global__ void __launch_bounds__(WG_NT) bottleneck_wgrad_kernel(const UadBottWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
    const int Z = a.zdim, F = a.npos * a.cmid, C = a.cenc, M = a.cmid;
    int b = blockIdx.x;
    // ---- part A ----
    const int tiles_d = (Z / WG_TI) * (F / WG_TJ), tiles_h = (F / WG_TI) * (Z / WG_TJ);
    if (b < tiles_d) { wgrad_dense_tile(a.z, Z, a.dd, F, a.n, b / (F / WG_TJ), b % (F / WG_TJ), a.gWd, a.gbd, sm); return; }
    b -= tiles_d;
    if (b < tiles_h) { wgrad_dense_tile(a.t, F, a.dmu, Z, a.n, b / (Z / WG_TJ), b % (Z / WG_TJ), a.gWmu, a.gbmu, sm); return; }
    b -= tiles_h;
    if (a.dls) {
        if (b < tiles_h) { wgrad_dense_tile(a.t, F, a.dls, Z, a.n, b / (Z / WG_TJ), b % (Z / WG_TJ), a.gWsg, a.gbsg, sm); return; }
        b -= tiles_h;
    }
    // ---- part B: out[o] = sum_w wpart[w][o] over the W partials the backward kernel left (fixed order): 64 outputs x 4 w-groups ----
    const int PW = 2 * C * M + M;
    const int ol = tid % 64, wg = tid / 64;
    const int o = b * 64 + ol;
    const int wper = (a.nparts + 3) / 4, w0 = wg * wper, w1 = min(w0 + wper, a.nparts);
    float acc = 0.f;
    if (o < PW) {
        int w = w0;
        for (; w + 16 <= w1; w += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = a.part[(size_t)(w + u) * PW + o];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += v[u];
        }
        for (; w < w1; ++w) acc += a.part[(size_t)w * PW + o];
    }
    sm[tid] = acc;
    __syncthreads();
    if (wg == 0 && o < PW) {
        const float t = (sm[ol] + sm[64 + ol]) + (sm[128 + ol] + sm[192 + ol]);
        const int NOUT = C * M;
        if (o < NOUT) a.gWb[o] = t;
        else if (o < 2 * NOUT) a.gWr[o - NOUT] = t;
        else a.gbb[o - 2 * NOUT] = t;
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

// workgroups per sample: 4 when the shapes split evenly (one exchange per kernel, weights streamed by 4x as many CUs), else 1
int uad_bottleneck_group(const UadBottArgs& a) {
    static const bool q1 = getenv("UAD_BOTT_Q1") != nullptr;
    const int Q = 4;
    if (q1 || !a.xch || !a.flags || a.npos % Q) return 1;
    const int Fq = a.npos / Q * a.cmid;
    const bool slice_ok = (Fq >= NT ? Fq % NT == 0 : NT % Fq == 0) && a.zdim % (16 * (Fq < NT ? NT / Fq : 1)) == 0;
    return slice_ok ? Q : 1;
}
int uad_bottleneck_colpart_rows(const UadBottArgs& a, int n) { return n * uad_bottleneck_group(a); }
size_t uad_bottleneck_lds_bytes(const UadBottArgs& a, bool bwd) {
    const int Q = uad_bottleneck_group(a);
    const size_t PC = (size_t)a.npos / Q * a.cenc, F = (size_t)a.npos / Q * a.cmid, CM = (size_t)a.cenc * a.cmid;
    return (bwd ? PC + F + 2 * a.zdim + F + (CM > PC ? CM : PC) + NT : PC + F + NT + a.zdim + F + CM) * sizeof(float);
}
bool uad_bottleneck_fused_ok(const UadBottArgs& a) {
    if (getenv("UAD_NO_FUSED_BOTT")) return false;
    const int NO = a.Wsg ? 2 * a.zdim : a.zdim;
    const int F = a.npos * a.cmid;
    if (a.zdim % 16 || NO > NT || NT % NO || NT % a.zdim || a.cmid % 4 || a.cenc % 16 || F % 4) return false;
    if (uad_bottleneck_group(a) == 1 && !((F >= NT ? F % NT == 0 : NT % F == 0) && a.zdim % (16 * (F < NT ? NT / F : 1)) == 0)) return false;
    return uad_bottleneck_lds_bytes(a, false) <= 150 * 1024 && uad_bottleneck_lds_bytes(a, true) <= 150 * 1024;
}
static void bott_attrs() {
    static bool attr = false;
    if (attr) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bottleneck_fwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bottleneck_bwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bottleneck_fwd_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bottleneck_bwd_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
}
void uad_launch_bottleneck_fwd(const UadBottArgs& a, int n, hipStream_t st) {
    bott_attrs();
    static const bool dbg = getenv("UAD_BOTT_DBG") != nullptr;
    static unsigned long long* sbuf = nullptr;
    static int calls = 0;
    UadBottArgs b = a;
    static const int fault = getenv("UAD_BOTT_FAULT") ? 1 : 0;      // tests/test_gpu_knobs.py: the bounded sibling exchange reports instead of hanging
    b.fault = fault;
    const bool on = dbg && ++calls > 20 && calls <= 24;
    if (on) { if (!sbuf) (void)hipMalloc((void**)&sbuf, 16 * 8); b.stamps = sbuf; }
    if (uad_bottleneck_group(a) == 4) hipLaunchKernelGGL(bottleneck_fwd_kernel<4>, dim3(4 * n), dim3(NT), uad_bottleneck_lds_bytes(a, false), st, b);
    else hipLaunchKernelGGL(bottleneck_fwd_kernel<1>, dim3(n), dim3(NT), uad_bottleneck_lds_bytes(a, false), st, b);
    if (on) {
        unsigned long long h[16];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, sbuf, sizeof h, hipMemcpyDeviceToHost);
        fprintf(stderr, "[bott.fwd Q=%d] phases (x10 ns):", uad_bottleneck_group(a));
        for (int i = 1; i <= 8; ++i) fprintf(stderr, " %llu", h[i] - h[i - 1]);
        fprintf(stderr, " | total %llu\n", h[8] - h[0]);
    }
}
void uad_launch_bottleneck_bwd(const UadBottArgs& a0, int n, hipStream_t st) {
    bott_attrs();
    UadBottArgs a = a0;
    a.fault = 0;
    if (uad_bottleneck_group(a) == 4) hipLaunchKernelGGL(bottleneck_bwd_kernel<4>, dim3(4 * n), dim3(NT), uad_bottleneck_lds_bytes(a, true), st, a);
    else hipLaunchKernelGGL(bottleneck_bwd_kernel<1>, dim3(n), dim3(NT), uad_bottleneck_lds_bytes(a, true), st, a);
}
bool uad_bottleneck_wgrad_ok(const UadBottWgradArgs& a) {
    if (getenv("UAD_NO_FUSED_BOTT_WGRAD")) return false;
    const int F = a.npos * a.cmid;
    return a.part && a.nparts > 0 && a.zdim % WG_TJ == 0 && F % WG_TJ == 0;
}
size_t uad_bottleneck_wgrad_part_floats(const UadBottWgradArgs& a) { return (size_t)2 * a.cenc * a.cmid + a.cmid; }
void uad_launch_bottleneck_wgrad(const UadBottWgradArgs& a, hipStream_t st) {
    const int Z = a.zdim, F = a.npos * a.cmid;
    const int tiles_d = (Z / WG_TI) * (F / WG_TJ), tiles_h = (F / WG_TI) * (Z / WG_TJ);
    const int PW = (int)uad_bottleneck_wgrad_part_floats(a);
    const int grid = tiles_d + tiles_h * (a.dls ? 2 : 1) + (PW + 63) / 64;
    hipLaunchKernelGGL(bottleneck_wgrad_kernel, dim3(grid), dim3(WG_NT), (size_t)64 * (WG_TI + WG_TJ) * sizeof(float), st, a);
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
