// Residual-map scoring on the device (reference utils/Evaluation.py:84-127, trainers/Metrics.py):
//   * binary erosion of the brain masks (cross structuring element, `iterations` passes, zero border),
//   * 5x5x5 median filter of a residual volume (scipy default boundary: reflect),
//   * exact AUROC / AUPRC / Dice-at-threshold over tens of millions of voxels from ONE descending sort
//     (sklearn's definitions: distinct-score thresholds; the reference sorts on the host for AUPRC and re-thresholds the
//     whole array ~170 times for the Dice search).
// All of it is HBM-bound integer / order-statistic work: coalesced loads, LDS tiles, no matrix cores.
// Round 6: the sort, the prefix sums and the compaction are this file's own kernels (an 8-bit-digit LSD radix sort with wave-ballot ranking,
// a three-phase scan, a flag compaction) -- no library primitives.
#include <cstring>

#include "uad_kernels.h"
#include "../../include/uad_hip.h"

int uad_fail(int code, const char* fmt, ...);   // uad_model.hip
#define fail uad_fail

#define EV_TRY(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(UAD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

namespace {

// ------------------------------------------------------------------------------------------------
// Erosion: one workgroup per slice, the mask lives in LDS as bytes (two buffers), `iterations` passes.
// out = 1.0f where the pixel survives.  scipy.ndimage.binary_erosion(..., structure=cross, border_value=0).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) erode_cross_kernel(const float* __restrict__ mask, int H, int W, int iterations,
                                                           float* __restrict__ out) {
    extern __shared__ unsigned char sm[];
    const int hw = H * W;
    unsigned char* a = sm;
    unsigned char* b = sm + hw;
    const float* src = mask + (size_t)blockIdx.x * hw;
    for (int i = threadIdx.x; i < hw; i += blockDim.x) a[i] = src[i] != 0.f;
    __syncthreads();
    for (int it = 0; it < iterations; ++it) {
        for (int i = threadIdx.x; i < hw; i += blockDim.x) {
            const int y = i / W, x = i - y * W;
            unsigned char v = a[i];
            v &= (y > 0) ? a[i - W] : 0;
            v &= (y < H - 1) ? a[i + W] : 0;
            v &= (x > 0) ? a[i - 1] : 0;
            v &= (x < W - 1) ? a[i + 1] : 0;
            b[i] = v;
        }
        __syncthreads();
        unsigned char* t = a; a = b; b = t;
    }
    float* dst = out + (size_t)blockIdx.x * hw;
    for (int i = threadIdx.x; i < hw; i += blockDim.x) dst[i] = a[i] ? 1.f : 0.f;
}

// ------------------------------------------------------------------------------------------------
// 5x5x5 median (rank 62 of 125), reflect boundary.  One thread per voxel; an 8x8x8 output tile + 2-voxel halo is staged in
// LDS as order-preserving unsigned keys; each thread copies its 125 keys to registers and runs a 32-step bitwise
// selection (largest prefix p with count(key < p) <= 62), no sorting, no divergence.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ int reflect_idx(int i, int n) {   // numpy 'symmetric' == scipy 'reflect': (d c b a | a b c d | d c b a)
    if (n == 1) return 0;
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}
__global__ void __launch_bounds__(512) median5_kernel(const float* __restrict__ vol, int D, int H, int W,
                                                      float* __restrict__ out) {
    constexpr int T = 8, R = 2, E = T + 2 * R;   // 12
    __shared__ unsigned tile[E * E * E];
    const int x0 = blockIdx.x * T, y0 = blockIdx.y * T, z0 = blockIdx.z * T;
    for (int i = threadIdx.x; i < E * E * E; i += 512) {
        const int lx = i % E, ly = (i / E) % E, lz = i / (E * E);
        const int gz = reflect_idx(z0 + lz - R, D), gy = reflect_idx(y0 + ly - R, H), gx = reflect_idx(x0 + lx - R, W);
        tile[i] = f2key(vol[((size_t)gz * H + gy) * W + gx]);
    }
    __syncthreads();
    const int lx = threadIdx.x % T, ly = (threadIdx.x / T) % T, lz = threadIdx.x / (T * T);
    const int gx = x0 + lx, gy = y0 + ly, gz = z0 + lz;
    unsigned k[125];
#pragma unroll
    for (int dz = 0; dz < 5; ++dz)
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) k[(dz * 5 + dy) * 5 + dx] = tile[((lz + dz) * E + ly + dy) * E + lx + dx];
    unsigned prefix = 0;
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = prefix | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 125; ++i) cnt += (k[i] < cand) ? 1 : 0;
        if (cnt <= 62) prefix = cand;
    }
    if (gx < W && gy < H && gz < D) out[((size_t)gz * H + gy) * W + gx] = key2f(prefix);
}

// ------------------------------------------------------------------------------------------------
// Metrics from one descending sort.  After the sort: score[i] (desc), lab[i] in {0,1}; tp[i] = inclusive prefix sum of lab.
// A "distinct" position closes a run of equal scores (sklearn's thresholds).  For the m-th distinct position i:
//   tps = tp[i], fps = i + 1 - tps;  AP += (tps - tps_prev) / P * tps / (i + 1);  AUC += (fpr - fpr_prev) * (tpr + tpr_prev) / 2
// ------------------------------------------------------------------------------------------------
// ---- descending sort of (score, label) pairs: LSD radix sort, four 8-bit passes --------------------------------------------------------------
// Key = ~(order-preserving bits of the float): ascending unsigned order of the key = descending order of the score.  The label rides as one byte.
// A pass = (1) per-tile digit histogram, (2) exclusive scan of the [digit][tile] table (digit-major: the scan value IS the first output slot of
// that tile's keys with that digit), (3) stable scatter.  A tile is 4096 consecutive keys, read striped (element i * 256 + t by thread t in round i:
// coalesced); inside a round the 256 keys are ranked per digit with wave ballots (eight ballots give the mask of lanes holding the same digit;
// rank = popcount below the lane) plus a [wave][digit] count table, so equal digits keep their input order -- rounds, waves and lanes all ascend
// with the input index.  ~0.2 GB of traffic per pass for 21.6 M voxels: microseconds at HBM rates, the launches dominate.
constexpr int RS_THREADS = 256, RS_ROUNDS = 16, RS_TILE = RS_THREADS * RS_ROUNDS;
__device__ __forceinline__ unsigned desc_key(float f) {
    const unsigned u = __float_as_uint(f);
    const unsigned asc = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);       // ascending order of asc = ascending order of f
    return ~asc;
}
__device__ __forceinline__ float desc_key_to_float(unsigned k) {
    const unsigned asc = ~k;
    return __uint_as_float(asc ^ ((asc >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
__global__ void __launch_bounds__(256) rs_make_keys_kernel(const float* __restrict__ pred, const float* __restrict__ lab, unsigned* __restrict__ key,
                                                           unsigned char* __restrict__ lab8, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    key[i] = desc_key(pred[i]);
    lab8[i] = lab[i] != 0.f ? 1 : 0;
}
__global__ void __launch_bounds__(RS_THREADS) rs_hist_kernel(const unsigned* __restrict__ key, size_t n, int shift, unsigned* __restrict__ counts,
                                                             unsigned ntiles) {
    __shared__ unsigned h[256];
    const int t = threadIdx.x;
    h[t] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int i = 0; i < RS_ROUNDS; ++i) {
        const size_t idx = base + (size_t)i * RS_THREADS + t;
        if (idx < n) atomicAdd(&h[(key[idx] >> shift) & 255u], 1u);
    }
    __syncthreads();
    counts[(size_t)t * ntiles + blockIdx.x] = h[t];
}
__global__ void __launch_bounds__(RS_THREADS) rs_scatter_kernel(const unsigned* __restrict__ key_in, const unsigned char* __restrict__ lab_in,
                                                                unsigned* __restrict__ key_out, unsigned char* __restrict__ lab_out, size_t n, int shift,
                                                                const unsigned* __restrict__ offs, unsigned ntiles) {
    __shared__ unsigned s_run[256];               // next free output slot of this tile's keys, per digit
    __shared__ unsigned s_cnt[RS_THREADS / 64][256];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    s_run[t] = offs[(size_t)t * ntiles + blockIdx.x];
#pragma unroll
    for (int w = 0; w < RS_THREADS / 64; ++w) s_cnt[w][t] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RS_TILE;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int i = 0; i < RS_ROUNDS; ++i) {
        const size_t idx = base + (size_t)i * RS_THREADS + t;
        const bool valid = idx < n;
        const unsigned k = valid ? key_in[idx] : 0u;
        const unsigned char l = valid ? lab_in[idx] : (unsigned char)0;
        const unsigned dgt = (k >> shift) & 255u;
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (dgt >> b) & 1u;
            const unsigned long long bb = __ballot(bit);
            same &= bit ? bb : ~bb;
        }
        const unsigned rank_w = (unsigned)__popcll(same & below);
        if (valid && rank_w == 0) s_cnt[wave][dgt] = (unsigned)__popcll(same);
        __syncthreads();
        if (valid) {
            unsigned pos = s_run[dgt] + rank_w;
            for (int w = 0; w < wave; ++w) pos += s_cnt[w][dgt];
            key_out[pos] = k;
            lab_out[pos] = l;
        }
        __syncthreads();
        {
            unsigned tot = 0;
#pragma unroll
            for (int w = 0; w < RS_THREADS / 64; ++w) { tot += s_cnt[w][t]; s_cnt[w][t] = 0; }
            s_run[t] += tot;
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) rs_keys_to_scores_kernel(const unsigned* __restrict__ key, float* __restrict__ score, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) score[i] = desc_key_to_float(key[i]);
}

// ---- prefix sums: three phases (block sums | one block scans the block sums | rescan with the block's offset).  Fixed order: deterministic. ----
constexpr int SC_THREADS = 256, SC_ITEMS = 8, SC_BLOCK = SC_THREADS * SC_ITEMS;
// exclusive scan of one value per thread over the workgroup (Hillis-Steele in LDS, any arithmetic type); returns the exclusive prefix, *total = the sum
template <class T, int NT>
__device__ __forceinline__ T block_exclusive_scan(T v, T* buf, T* total) {
    const int t = threadIdx.x;
    buf[t] = v;
    __syncthreads();
    for (int o = 1; o < NT; o <<= 1) {
        const T add = t >= o ? buf[t - o] : T(0);
        __syncthreads();
        buf[t] += add;
        __syncthreads();
    }
    const T incl = buf[t];
    *total = buf[NT - 1];
    __syncthreads();
    return incl - v;
}
template <class Tin, class Tout>
__global__ void __launch_bounds__(SC_THREADS) scan_block_sums_kernel(const Tin* __restrict__ in, size_t n, Tout* __restrict__ bsum) {
    __shared__ Tout buf[SC_THREADS];
    const size_t base = (size_t)blockIdx.x * SC_BLOCK + (size_t)threadIdx.x * SC_ITEMS;
    Tout s = 0;
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i) if (base + i < n) s += (Tout)in[base + i];
    Tout tot;
    (void)block_exclusive_scan<Tout, SC_THREADS>(s, buf, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
template <class T>
__global__ void __launch_bounds__(1024) scan_single_block_kernel(T* __restrict__ v, size_t nb, T* __restrict__ grand_total) {
    __shared__ T buf[1024];
    T carry = 0;
    for (size_t o = 0; o < nb; o += 1024) {
        const size_t i = o + threadIdx.x;
        const T x = i < nb ? v[i] : T(0);
        T tot;
        const T ex = block_exclusive_scan<T, 1024>(x, buf, &tot);
        if (i < nb) v[i] = carry + ex;
        carry += tot;
    }
    if (grand_total && threadIdx.x == 0) *grand_total = carry;
}
template <class Tin, class Tout, bool INCLUSIVE>
__global__ void __launch_bounds__(SC_THREADS) scan_apply_kernel(const Tin* in, size_t n, const Tout* __restrict__ bsum, Tout* out) {      // (in == out is allowed: a thread reads its items before it writes them)
    __shared__ Tout buf[SC_THREADS];
    const size_t base = (size_t)blockIdx.x * SC_BLOCK + (size_t)threadIdx.x * SC_ITEMS;
    Tout x[SC_ITEMS];
    Tout s = 0;
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i) { x[i] = base + i < n ? (Tout)in[base + i] : Tout(0); s += x[i]; }
    Tout tot;
    Tout run = bsum[blockIdx.x] + block_exclusive_scan<Tout, SC_THREADS>(s, buf, &tot);
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i) {
        if (INCLUSIVE) run += x[i];
        if (base + i < n) out[base + i] = run;
        if (!INCLUSIVE) run += x[i];
    }
}
// out[i] = prefix sum of in[0..i] (inclusive) or in[0..i-1] (exclusive); bsum: scratch of ceil(n / SC_BLOCK) Tout; total (optional, device): the sum
template <class Tin, class Tout, bool INCLUSIVE>
void launch_scan(const Tin* in, size_t n, Tout* out, Tout* bsum, Tout* total, hipStream_t st) {
    const unsigned nb = (unsigned)((n + SC_BLOCK - 1) / SC_BLOCK);
    hipLaunchKernelGGL((scan_block_sums_kernel<Tin, Tout>), dim3(nb), dim3(SC_THREADS), 0, st, in, n, bsum);
    hipLaunchKernelGGL((scan_single_block_kernel<Tout>), dim3(1), dim3(1024), 0, st, bsum, (size_t)nb, total);
    hipLaunchKernelGGL((scan_apply_kernel<Tin, Tout, INCLUSIVE>), dim3(nb), dim3(SC_THREADS), 0, st, in, n, (const Tout*)bsum, out);
}

// ---- distinct-score positions (sklearn's thresholds): flag the last element of every run of equal scores, compact their indices ----
__global__ void __launch_bounds__(256) distinct_flag_kernel(const float* __restrict__ score, unsigned char* __restrict__ flag, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i + 1 == n) || (score[i] != score[i + 1]);
}
__global__ void __launch_bounds__(256) compact_kernel(const unsigned char* __restrict__ flag, const unsigned* __restrict__ pos, unsigned* __restrict__ didx, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && flag[i]) didx[pos[i]] = (unsigned)i;
}
// one block: fixed-order double accumulation over the compacted distinct positions (deterministic)
__global__ void __launch_bounds__(1024) auc_ap_kernel(const unsigned* __restrict__ didx, unsigned nd,
                                                      const unsigned long long* __restrict__ tp, unsigned long long n,
                                                      double* __restrict__ out) {
    __shared__ double s_ap[1024], s_auc[1024];
    const double P = (double)tp[n - 1], Nn = (double)n - P;
    double ap = 0.0, auc = 0.0;
    for (unsigned m = threadIdx.x; m < nd; m += 1024) {
        const unsigned i = didx[m];
        const double tps = (double)tp[i], fps = (double)(i + 1) - tps;
        double tps_p = 0.0, fps_p = 0.0;
        if (m > 0) { const unsigned j = didx[m - 1]; tps_p = (double)tp[j]; fps_p = (double)(j + 1) - tps_p; }
        if (P > 0) ap += (tps - tps_p) / P * (tps / (tps + fps));
        if (P > 0 && Nn > 0) auc += (fps - fps_p) / Nn * (tps + tps_p) / P * 0.5;
    }
    s_ap[threadIdx.x] = ap;
    s_auc[threadIdx.x] = auc;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { s_ap[threadIdx.x] += s_ap[threadIdx.x + o]; s_auc[threadIdx.x] += s_auc[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = s_auc[0]; out[1] = s_ap[0]; out[2] = P; }
}
// Dice of (score > t) against the labels for a batch of thresholds: count = first index with score <= t (descending order)
__global__ void dice_at_kernel(const float* __restrict__ score, const unsigned long long* __restrict__ tp, unsigned long long n,
                               const double* __restrict__ thr, int k, double* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= k) return;
    const double t = thr[q];
    unsigned long long lo = 0, hi = n;           // first index whose score is NOT > t
    while (lo < hi) {
        const unsigned long long mid = (lo + hi) >> 1;
        if ((double)score[mid] > t) lo = mid + 1; else hi = mid;
    }
    const double cnt = (double)lo;
    const double tps = lo ? (double)tp[lo - 1] : 0.0;
    const double P = (double)tp[n - 1];
    out[q] = (2.0 * tps) / (cnt + P);            // 0/0 -> nan like the reference's numpy division
}

// ------------------------------------------------------------------------------------------------
// Small-component filter (utils/Evaluation.py:113-127: label(volume, connectivity=3) + regionprops, components with
// filled_area <= 7 are zeroed).  No labelling pass is needed for that rule: a 26-connected component of at most `maxv` voxels is
// found completely by a flood fill that stops at maxv + 1 voxels.  One thread per voxel runs that bounded fill from its own voxel
// (list of visited linear indices in LDS, <= maxv + 1 entries, linear membership test) and keeps the voxel iff the fill reaches
// maxv + 1.  (skimage fills holes with the full 3x3x3 structure, so a component this small has filled_area == area.)
// Integer work on L2-resident reads; exact.
// ------------------------------------------------------------------------------------------------
// Monte-Carlo dropout statistics (utils/Evaluation.py:238-266, Metrics.combined_predictive_uncertainty :170-173): per pixel, over the K
// brain-masked reconstructions p_k = mask * rec_k:  mean = E[p],  var = E[p^2] - E[p]^2  (the epistemic term; no aleatoric input here).
__global__ void __launch_bounds__(256) mc_stats_kernel(const float* __restrict__ recs, const float* __restrict__ mask, int K, size_t total,
                                                       float* __restrict__ mean, float* __restrict__ var) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float m = mask ? mask[i] : 1.0f;
    float s = 0.f, q = 0.f;
    for (int k = 0; k < K; ++k) {
        const float p = recs[(size_t)k * total + i] * m;
        s += p; q = fmaf(p, p, q);
    }
    const float mu = s / (float)K;
    mean[i] = mu;
    if (var) var[i] = q / (float)K - mu * mu;
}

constexpr int CC_CAP = 16;
__global__ void __launch_bounds__(256) cc_filter_kernel(const float* __restrict__ vol, int D, int H, int W, int maxv,
                                                        float* __restrict__ out) {
    __shared__ int s_list[CC_CAP][256];
    const size_t total = (size_t)D * H * W;
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= total) return;
    const float val = vol[v];
    if (val == 0.f) { out[v] = 0.f; return; }
    const int t = threadIdx.x;
    s_list[0][t] = (int)v;
    int count = 1, head = 0;
    const int HW = H * W;
    while (head < count && count <= maxv) {
        const int cur = s_list[head++][t];
        const int z = cur / HW, y = (cur - z * HW) / W, x = cur - z * HW - y * W;
        for (int dz = -1; dz <= 1 && count <= maxv; ++dz) {
            const int zz = z + dz;
            if ((unsigned)zz >= (unsigned)D) continue;
            for (int dy = -1; dy <= 1 && count <= maxv; ++dy) {
                const int yy = y + dy;
                if ((unsigned)yy >= (unsigned)H) continue;
                for (int dx = -1; dx <= 1; ++dx) {
                    const int xx = x + dx;
                    if ((unsigned)xx >= (unsigned)W || (dz == 0 && dy == 0 && dx == 0)) continue;
                    const int nb = (zz * H + yy) * W + xx;
                    if (vol[nb] == 0.f) continue;
                    bool seen = false;
                    for (int k = 0; k < count; ++k) seen |= (s_list[k][t] == nb);
                    if (!seen) {
                        s_list[count++][t] = nb;
                        if (count > maxv) break;
                    }
                }
            }
        }
    }
    out[v] = count > maxv ? val : 0.f;
}

}  // namespace

struct uad_scores {
    unsigned long long n;
    float* score;               // descending
    unsigned long long* tp;     // inclusive prefix sum of the labels in that order
    double auc, ap, npos;
    double* dthr;               // device scratch for threshold batches
    double* dout;
    int cap;
};

extern "C" {

int uad_erode_cross(const float* mask, int n, int H, int W, int iterations, float* out, void* stream) {
    if (!mask || !out || n <= 0 || H <= 0 || W <= 0 || iterations < 0) return fail(UAD_ERR_INVALID, "erode: bad arguments");
    const size_t lds = (size_t)2 * H * W;
    if (lds > 160 * 1024) return fail(UAD_ERR_UNSUPPORTED, "erode: slice %dx%d does not fit in LDS", H, W);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(erode_cross_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL(erode_cross_kernel, dim3(n), dim3(1024), lds, (hipStream_t)stream, mask, H, W, iterations, out);
    EV_TRY(hipGetLastError());
    return UAD_OK;
}

int uad_median3d(const float* vol, int D, int H, int W, int ksize, float* out, void* stream) {
    if (!vol || !out || D <= 0 || H <= 0 || W <= 0) return fail(UAD_ERR_INVALID, "median3d: bad arguments");
    if (ksize != 5) return fail(UAD_ERR_UNSUPPORTED, "median3d: only the reference's 5x5x5 window is implemented");
    if (vol == out) return fail(UAD_ERR_INVALID, "median3d: in-place is not supported");
    dim3 grid((W + 7) / 8, (H + 7) / 8, (D + 7) / 8);
    hipLaunchKernelGGL(median5_kernel, grid, dim3(512), 0, (hipStream_t)stream, vol, D, H, W, out);
    EV_TRY(hipGetLastError());
    return UAD_OK;
}

int uad_mc_stats(const float* recs, const float* mask, int K, long long total, float* mean, float* var, void* stream) {
    if (!recs || !mean || K <= 0 || total <= 0) return fail(UAD_ERR_INVALID, "mc_stats: bad arguments");
    hipLaunchKernelGGL(mc_stats_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, recs, mask, K, (size_t)total, mean, var);
    EV_TRY(hipGetLastError());
    return UAD_OK;
}

int uad_cc_filter(const float* vol, int D, int H, int W, int max_voxels, float* out, void* stream) {
    if (!vol || !out || D <= 0 || H <= 0 || W <= 0) return fail(UAD_ERR_INVALID, "cc_filter: bad arguments");
    if (max_voxels < 0 || max_voxels >= CC_CAP) return fail(UAD_ERR_UNSUPPORTED, "cc_filter: max_voxels must be in [0, %d)", CC_CAP);
    if (vol == out) return fail(UAD_ERR_INVALID, "cc_filter: in-place is not supported");
    const size_t total = (size_t)D * H * W;
    if (total > 0x7fffffffULL) return fail(UAD_ERR_UNSUPPORTED, "cc_filter: volume too large");
    hipLaunchKernelGGL(cc_filter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vol, D, H, W, max_voxels, out);
    EV_TRY(hipGetLastError());
    return UAD_OK;
}

int uad_scores_create(const float* pred, const float* label, long long n, uad_scores_t** out, void* stream) {
    if (!pred || !label || !out || n <= 0 || n > 0x7fffffffLL) return fail(UAD_ERR_INVALID, "scores: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    uad_scores* s = new uad_scores();
    memset(s, 0, sizeof *s);
    s->n = (unsigned long long)n;
    unsigned *key_a = nullptr, *key_b = nullptr, *counts = nullptr, *pos = nullptr, *didx = nullptr, *d_nd = nullptr, *bsum32 = nullptr;
    unsigned char *lab_a = nullptr, *lab_b = nullptr, *flag = nullptr;
    unsigned long long* bsum64 = nullptr;
    double* d_res = nullptr;
    int rc = UAD_OK;
    auto cleanup = [&]() { hipFree(key_a); hipFree(key_b); hipFree(counts); hipFree(pos); hipFree(didx); hipFree(d_nd); hipFree(bsum32); hipFree(lab_a); hipFree(lab_b);
                           hipFree(flag); hipFree(bsum64); hipFree(d_res); };
#define EV_TRY2(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(UAD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); cleanup(); uad_scores_destroy(s); return rc; } } while (0)
    const size_t N = (size_t)n;
    const unsigned blocks = (unsigned)((N + 255) / 256);
    const unsigned ntiles = (unsigned)((N + RS_TILE - 1) / RS_TILE);
    const size_t ncounts = (size_t)256 * ntiles;
    const size_t nb_max = ((N > ncounts ? N : ncounts) + SC_BLOCK - 1) / SC_BLOCK;
    EV_TRY2(hipMalloc((void**)&s->score, N * sizeof(float)));
    EV_TRY2(hipMalloc((void**)&s->tp, N * sizeof(unsigned long long)));
    EV_TRY2(hipMalloc((void**)&key_a, N * sizeof(unsigned)));
    EV_TRY2(hipMalloc((void**)&key_b, N * sizeof(unsigned)));
    EV_TRY2(hipMalloc((void**)&lab_a, N));
    EV_TRY2(hipMalloc((void**)&lab_b, N));
    EV_TRY2(hipMalloc((void**)&counts, ncounts * sizeof(unsigned)));
    EV_TRY2(hipMalloc((void**)&bsum32, nb_max * sizeof(unsigned)));
    EV_TRY2(hipMalloc((void**)&bsum64, nb_max * sizeof(unsigned long long)));
    // (score, label) -> sort keys; four stable passes over the key's bytes, least significant first
    hipLaunchKernelGGL(rs_make_keys_kernel, dim3(blocks), dim3(256), 0, st, pred, label, key_a, lab_a, N);
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 8 * pass;
        hipLaunchKernelGGL(rs_hist_kernel, dim3(ntiles), dim3(RS_THREADS), 0, st, (const unsigned*)key_a, N, shift, counts, ntiles);
        launch_scan<unsigned, unsigned, false>(counts, ncounts, counts, bsum32, (unsigned*)nullptr, st);
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(ntiles), dim3(RS_THREADS), 0, st, (const unsigned*)key_a, (const unsigned char*)lab_a, key_b, lab_b, N, shift,
                           (const unsigned*)counts, ntiles);
        unsigned* tk = key_a; key_a = key_b; key_b = tk;
        unsigned char* tl = lab_a; lab_a = lab_b; lab_b = tl;
    }
    EV_TRY2(hipGetLastError());
    hipLaunchKernelGGL(rs_keys_to_scores_kernel, dim3(blocks), dim3(256), 0, st, (const unsigned*)key_a, s->score, N);
    // tp = inclusive prefix sum of the labels in that order (64-bit)
    launch_scan<unsigned char, unsigned long long, true>(lab_a, N, s->tp, bsum64, (unsigned long long*)nullptr, st);
    // distinct-threshold positions, compacted
    EV_TRY2(hipMalloc((void**)&flag, N));
    EV_TRY2(hipMalloc((void**)&d_nd, sizeof(unsigned)));
    pos = key_b; key_b = nullptr;            // (the spare key buffer is free now)
    EV_TRY2(hipMalloc((void**)&didx, N * sizeof(unsigned)));
    hipLaunchKernelGGL(distinct_flag_kernel, dim3(blocks), dim3(256), 0, st, (const float*)s->score, flag, N);
    launch_scan<unsigned char, unsigned, false>(flag, N, pos, bsum32, d_nd, st);
    hipLaunchKernelGGL(compact_kernel, dim3(blocks), dim3(256), 0, st, (const unsigned char*)flag, (const unsigned*)pos, didx, N);
    EV_TRY2(hipGetLastError());
    unsigned nd = 0;
    EV_TRY2(hipMemcpyAsync(&nd, d_nd, sizeof nd, hipMemcpyDeviceToHost, st));
    EV_TRY2(hipStreamSynchronize(st));
    EV_TRY2(hipMalloc((void**)&d_res, 3 * sizeof(double)));
    hipLaunchKernelGGL(auc_ap_kernel, dim3(1), dim3(1024), 0, st, didx, nd, s->tp, s->n, d_res);
    double res[3];
    EV_TRY2(hipMemcpyAsync(res, d_res, sizeof res, hipMemcpyDeviceToHost, st));
    EV_TRY2(hipStreamSynchronize(st));
    s->auc = res[0]; s->ap = res[1]; s->npos = res[2];
    s->cap = 64;
    EV_TRY2(hipMalloc((void**)&s->dthr, s->cap * sizeof(double)));
    EV_TRY2(hipMalloc((void**)&s->dout, s->cap * sizeof(double)));
    cleanup();
#undef EV_TRY2
    *out = s;
    return UAD_OK;
}

int uad_scores_auc(const uad_scores_t* s, double* auroc, double* auprc, double* positives) {
    if (!s) return fail(UAD_ERR_INVALID, "scores: null handle");
    if (auroc) *auroc = s->auc;
    if (auprc) *auprc = s->ap;
    if (positives) *positives = s->npos;
    return UAD_OK;
}

int uad_scores_dice(uad_scores_t* s, const double* thresholds, int k, double* dice, void* stream) {
    if (!s || !thresholds || !dice || k <= 0) return fail(UAD_ERR_INVALID, "scores_dice: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    for (int o = 0; o < k; o += s->cap) {
        const int c = (k - o < s->cap) ? k - o : s->cap;
        EV_TRY(hipMemcpyAsync(s->dthr, thresholds + o, c * sizeof(double), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(dice_at_kernel, dim3(1), dim3(64), 0, st, s->score, s->tp, s->n, s->dthr, c, s->dout);
        EV_TRY(hipMemcpyAsync(dice + o, s->dout, c * sizeof(double), hipMemcpyDeviceToHost, st));
        EV_TRY(hipStreamSynchronize(st));
    }
    return UAD_OK;
}

int uad_scores_destroy(uad_scores_t* s) {
    if (!s) return UAD_OK;
    hipFree(s->score); hipFree(s->tp); hipFree(s->dthr); hipFree(s->dout);
    delete s;
    return UAD_OK;
}

}  // extern "C"
