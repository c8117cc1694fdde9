// Calibration microbench: fp32 MFMA (32x32x2) ceiling on this box with non-trivial data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ void __launch_bounds__(256) k(float* out, const float* in, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[64 * 36 * 4];
    const int tid = threadIdx.x;
    for (int i = tid; i < 64 * 36 * 4; i += 256) sm[i] = in[i % 4096];
    __syncthreads();
    v16f acc[NACC];
    for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float a0 = in[tid], b0 = in[tid + 256], b1 = in[tid + 512], b2 = in[tid + 768], b3 = in[tid + 1024];
    const int aoff = (tid & 31) * 36 + 4 * ((tid & 63) >> 5);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float4 av;
            if (LDS) av = *reinterpret_cast<const float4*>(sm + aoff + ((it + u) & 3) * 8 + (u & 1) * 36 * 32);
            else av = make_float4(a0, a0 + 1.f, a0 + 2.f, a0 + 3.f);
            acc[(4 * u + 0) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0, acc[(4 * u + 0) % NACC], 0, 0, 0);
            acc[(4 * u + 1) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1, acc[(4 * u + 1) % NACC], 0, 0, 0);
            acc[(4 * u + 2) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b2, acc[(4 * u + 2) % NACC], 0, 0, 0);
            acc[(4 * u + 3) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b3, acc[(4 * u + 3) % NACC], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int NACC, bool LDS>
void run(const char* name, float* out, float* in, int blocks) {
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<NACC, LDS><<<blocks, 256>>>(out, in, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<NACC, LDS><<<blocks, 256>>>(out, in, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double flops = (double)blocks * 4 * iters * 32 * 4096.0;
    printf("%-28s blocks=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flops / ms * 1e-9);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 8192 * 4); hipMalloc(&out, 4096 * 256 * 4);
    float h[8192]; for (int i = 0; i < 8192; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    for (int blocks : {256, 512, 768}) {
        run<1, false>("1 acc, regs", out, in, blocks);
        run<4, false>("4 acc, regs", out, in, blocks);
        run<1, true>("1 acc, ds_read_b128/4mfma", out, in, blocks);
        run<4, true>("4 acc, ds_read_b128/4mfma", out, in, blocks);
    }
    return 0;
}
