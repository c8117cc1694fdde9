// Probe of gfx950's ds_read_b64_tr_b16 (LDS transpose read): which element does lane l, register element e receive for a given set of per-lane
// addresses?  hipcc --offload-arch=gfx950 tools/tr_probe.hip -o tools/tr_probe && tools/tr_probe   (round 4: groundwork for a pixel-major filter-
// gradient tile; profiles/r04_g_tr_probe.log)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out, const int* addr) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
    __syncthreads();
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + addr[threadIdx.x]));
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = (unsigned short)r[e];
}
int main() {
    unsigned short hin[4096], hout[256];
    int haddr[64];
    for (int i = 0; i < 4096; ++i) hin[i] = (unsigned short)i;
    unsigned short *din, *dout; int* daddr;
    hipMalloc(&din, sizeof hin); hipMalloc(&dout, sizeof hout); hipMalloc(&daddr, sizeof haddr);
    hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice);
    for (int pat = 0; pat < 3; ++pat) {
        for (int l = 0; l < 64; ++l) {
            const int i = l & 15, g = l >> 4;
            if (pat == 0) haddr[l] = l * 4;                                        // contiguous 8 bytes per lane
            else if (pat == 1) haddr[l] = (i / 4) * 40 + (i % 4) * 4 + g * 16;     // 4 rows of 16 elements, row stride 40 elements; groups 16 columns apart
            else haddr[l] = (i / 4) * 80 + (i % 4) * 4 + (g & 1) * 16 + (g >> 1) * 1000;   // row stride 80, groups 0/1 adjacent columns, 2/3 a far block
        }
        hipMemcpy(daddr, haddr, sizeof haddr, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, daddr);
        hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) printf("  lane %2d addr %4d -> %4d %4d %4d %4d\n", l, haddr[l], hout[4 * l], hout[4 * l + 1], hout[4 * l + 2], hout[4 * l + 3]);
    }
    return 0;
}
