// Issue-port micro-benchmark for gfx950 (round 5, verdict item 1 / step A).
//
// Question: do the VALU instructions that surround the bf16 MFMAs of the k5 kernels (13.7 VALU per MFMA in dec3.fwd, 6-11 in the filter
// gradients) overlap with the matrix pipe, (a) inside one wave, (b) between the two waves of a SIMD when one does only MFMAs and the other only
// VALU work (the producer / consumer wave split the round-4 verdict asks for)?  Every variant issues the SAME instruction multiset per SIMD:
//   per iteration and SIMD: 8 v_mfma_f32_32x32x16_bf16 (4 independent accumulators per wave) + 8 V v_fma_f32 (8 independent chains per wave)
//     same : 8 waves per workgroup, every wave 4 MFMAs + 4 V FMAs per iteration (V FMAs behind each MFMA)
//     split: waves 0-3 (one per SIMD) 8 MFMAs per iteration, waves 4-7 8 V FMAs per iteration -- consumer / producer halves
//     mfma : the MFMAs alone;  valu : the FMAs alone
// and reports wall time, the shader clock measured inside the kernel (s_memtime ticks per 100 MHz s_memrealtime tick) and shader cycles per
// iteration and SIMD.  If VALU issue hides under the matrix pipe the mixed variants take max(mfma, valu); if not, their sum.
//
//   hipcc -O3 --offload-arch=gfx950 tools/issue_ubench.hip -o tools/issue_ubench && tools/issue_ubench
//   hipcc -O3 --offload-arch=gfx950 -DF32_MFMA tools/issue_ubench.hip -o tools/issue_ubench_f32 && tools/issue_ubench_f32      (the exact-fp32 MFMA)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#ifdef F32_MFMA      // -DF32_MFMA: the exact-fp32 matrix instruction (64 cycles of matrix pipe per SIMD) instead of the bf16 one (32)
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"((a)[0]), "v"((b)[0]))
#else
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#endif
#define FMA(x, a, b) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b))
#define PKFMA(x, a, b) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b))

enum { SAME = 0, SPLIT = 1, MFMA_ONLY = 2, VALU_ONLY = 3, PK_ONLY = 4, SPLIT_PRIO = 5 };

template <int V>
__device__ __forceinline__ void fmas(float (&x)[8], float a, float b) {
#pragma unroll
    for (int i = 0; i < V; ++i) FMA(x[i & 7], a, b);
}

template <int MODE, int V>
__global__ void __launch_bounds__(512) k(float* out, const int* in, int iters, unsigned long long* clk) {
    const int tid = threadIdx.x, wave = tid >> 6;
    v16f acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    v4i a, b;
    for (int i = 0; i < 4; ++i) { a[i] = in[(tid + i * 64) & 4095]; b[i] = in[(tid * 3 + i) & 4095]; }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)in[(tid + i) & 4095] * 1e-9f;
    v2f xp[8];
    for (int i = 0; i < 8; ++i) { xp[i][0] = x[i]; xp[i][1] = x[i] + 1.f; }
    const float fa = 0.999f, fb = 1e-6f;
    const v2f pa = {0.999f, 0.998f}, pb = {1e-6f, 2e-6f};
    const bool stamp = blockIdx.x == 0 && tid == 0;
    unsigned long long c0 = 0, r0 = 0;
    if (MODE == SPLIT_PRIO && wave < 4) __builtin_amdgcn_s_setprio(1);
    __syncthreads();
    if (stamp) { c0 = clock64(); r0 = wall_clock64(); }
    if (MODE == SAME) {
        for (int it = 0; it < iters; ++it) {
            MFMA(acc[0], a, b); fmas<V>(x, fa, fb);
            MFMA(acc[1], a, b); fmas<V>(x, fa, fb);
            MFMA(acc[2], a, b); fmas<V>(x, fa, fb);
            MFMA(acc[3], a, b); fmas<V>(x, fa, fb);
        }
    } else if (MODE == SPLIT || MODE == SPLIT_PRIO) {
        if (wave < 4) {
            for (int it = 0; it < iters; ++it) {
                MFMA(acc[0], a, b); MFMA(acc[1], a, b); MFMA(acc[2], a, b); MFMA(acc[3], a, b);
                MFMA(acc[0], a, b); MFMA(acc[1], a, b); MFMA(acc[2], a, b); MFMA(acc[3], a, b);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
                fmas<V>(x, fa, fb); fmas<V>(x, fa, fb); fmas<V>(x, fa, fb); fmas<V>(x, fa, fb);
                fmas<V>(x, fa, fb); fmas<V>(x, fa, fb); fmas<V>(x, fa, fb); fmas<V>(x, fa, fb);
            }
        }
    } else if (MODE == MFMA_ONLY) {
        for (int it = 0; it < iters; ++it) { MFMA(acc[0], a, b); MFMA(acc[1], a, b); MFMA(acc[2], a, b); MFMA(acc[3], a, b); }
    } else if (MODE == VALU_ONLY) {
        for (int it = 0; it < iters; ++it) { fmas<V>(x, fa, fb); fmas<V>(x, fa, fb); fmas<V>(x, fa, fb); fmas<V>(x, fa, fb); }
    } else if (MODE == PK_ONLY) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4 * V; ++i) PKFMA(xp[i & 7], pa, pb);
        }
    }
    if (stamp) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    for (int i = 0; i < 8; ++i) s += x[i] + xp[i][0] + xp[i][1];
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE, int V>
void run(const char* name, float* out, int* in, unsigned long long* clk, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, V><<<blocks, 512>>>(out, in, 100, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, V><<<blocks, 512>>>(out, in, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / (double)h[1] * 0.1;     // s_memrealtime ticks at 100 MHz
    const double cyc = (double)h[0] / iters;                  // shader cycles per iteration (of wave 0 of workgroup 0)
    printf("%-10s V=%2d  %8.3f ms  clock %.2f GHz  %7.1f cycles / iteration / SIMD\n", name, V, ms, ghz, cyc);
}

template <int V>
void sweep(float* out, int* in, unsigned long long* clk, int blocks) {
    run<MFMA_ONLY, V>("mfma", out, in, clk, blocks);
    run<VALU_ONLY, V>("valu", out, in, clk, blocks);
    run<SAME, V>("same", out, in, clk, blocks);
    run<SPLIT, V>("split", out, in, clk, blocks);
    run<SPLIT_PRIO, V>("split+prio", out, in, clk, blocks);
    printf("\n");
}

int main() {
    int* in; float* out; unsigned long long* clk;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&clk, 16);
    int h[4096]; for (int i = 0; i < 4096; ++i) h[i] = rand();
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    const int blocks = 256;        // one 8-wave workgroup per CU: two waves per SIMD
#ifdef F32_MFMA
    printf("# per iteration and SIMD: 8 MFMA (32x32x2 f32, 64 cycles of matrix pipe each = 512 cycles) + 8 V v_fma_f32; 'mfma' / 'valu' = one kind alone\n");
#else
    printf("# per iteration and SIMD: 8 MFMA (32x32x16 bf16, 32 cycles of matrix pipe each = 256 cycles) + 8 V v_fma_f32; 'mfma' / 'valu' = one kind alone\n");
#endif
    sweep<2>(out, in, clk, blocks);
    sweep<4>(out, in, clk, blocks);
    sweep<6>(out, in, clk, blocks);
    sweep<8>(out, in, clk, blocks);
    sweep<12>(out, in, clk, blocks);
    sweep<16>(out, in, clk, blocks);
    run<PK_ONLY, 8>("pk_fma", out, in, clk, blocks);
    run<VALU_ONLY, 8>("fma", out, in, clk, blocks);
    return 0;
}
