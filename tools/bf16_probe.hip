// probe: v_cvt_pk_bf16_f32 operand order / rounding, and the 32x32x16 bf16 MFMA fragment layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
__global__ void cvt(unsigned* o, float x, float y) { unsigned pk; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(x), "v"(y)); o[0] = pk; }
// A[32][16], B[16][32] row-major floats exactly representable in bf16; lane l supplies A[l&31][8*(l>>5)+e], B[8*(l>>5)+e][l&31]
__global__ void mm(float* C, const float* A, const float* B) {
    const int l = threadIdx.x;
    unsigned short a[8], b[8];
    for (int e = 0; e < 8; ++e) {
        a[e] = __float_as_uint(A[(l & 31) * 16 + 8 * (l >> 5) + e]) >> 16;
        b[e] = __float_as_uint(B[(8 * (l >> 5) + e) * 32 + (l & 31)]) >> 16;
    }
    uint4 ua = make_uint4(a[0] | (a[1] << 16), a[2] | (a[3] << 16), a[4] | (a[5] << 16), a[6] | (a[7] << 16));
    uint4 ub = make_uint4(b[0] | (b[1] << 16), b[2] | (b[3] << 16), b[4] | (b[5] << 16), b[6] | (b[7] << 16));
    v16f acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, ua), __builtin_bit_cast(v8bf, ub), acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
int main() {
    unsigned* o; hipMalloc(&o, 4); unsigned h;
    cvt<<<1, 1>>>(o, 1.0f, 2.0f); hipMemcpy(&h, o, 4, hipMemcpyDeviceToHost); printf("cvt_pk(1,2) = %08x (expect lo=3f80 hi=4000)\n", h);
    cvt<<<1, 1>>>(o, 1.00390625f, 1.01171875f); hipMemcpy(&h, o, 4, hipMemcpyDeviceToHost); printf("cvt_pk(1+2^-8 [tie], 1+3*2^-8 [tie]) = %08x (RNE: 3f80, 3f82)\n", h);
    float A[32 * 16], B[16 * 32], C[32 * 32], *dA, *dB, *dC;
    for (int i = 0; i < 512; ++i) { A[i] = (float)((i * 7) % 13 - 6); B[i] = (float)((i * 5) % 11 - 5); }
    hipMalloc(&dA, sizeof A); hipMalloc(&dB, sizeof B); hipMalloc(&dC, sizeof C);
    hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
    mm<<<1, 64>>>(dC, dA, dB); hipMemcpy(C, dC, sizeof C, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j]; err = fmax(err, fabs(s - C[i * 32 + j])); }
    printf("mfma 32x32x16 bf16 layout check: max err = %g\n", err);
    return 0;
}
