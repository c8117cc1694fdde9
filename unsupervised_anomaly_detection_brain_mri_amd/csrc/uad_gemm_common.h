// Shared by the translation units of the contraction kernels (uad_gemm.hip, uad_gemm_d16s.hip): argument block, vector typedefs and the device helpers of
// the split-bf16 kernels.  Everything lives in an anonymous namespace: each unit gets its own copy (round 6: the lane = pixel family -- 28 fully unrolled
// instances -- moved into a unit of its own so that the two halves of the k5 kernels compile side by side).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "uad_kernels.h"
#include <stdio.h>
#include <stdlib.h>

typedef float v16f __attribute__((ext_vector_type(16)));

#define KIND_F 0
#define KIND_D 1

namespace {

struct ConvGemmArgs {
    const float* A;
    const float* W;
    const float* Wp;   // k-quad-interleaved copy of W for the spatial kernels (uad_launch_pack_weights)
    const unsigned short* Wp16;   // bf16 hi|lo planes, k-octet-interleaved (bf16x3 math mode)
    long long w16_plane;          // elements per plane
    float* Out;
    UadXform xf;
    UadEpilogue ep;
    UadConvDesc d;
    int M;         // N*HS*WS rows (small-image positions)
    int CA;        // channels of the A operand (contraction per tap)
    int Nn;        // output channels
    int lws, lhs;  // log2(WS), log2(HS) or -1 when not a power of two
    int nsplit;    // split-K factor (>1: raw partial tiles go to Out + split*out_elems, see splitk_epilogue_kernel)
    long long out_elems;
    unsigned* sk_counter;   // split-K with in-kernel reduction: arrival counters, one per (spatial tile, column block); null = splitk_epilogue_kernel
    float* out_final;       // ... and the real output (Out points at the slabs)
    unsigned long long* dbgbuf;   // per-phase clock stamps of a few workgroups (UAD_DBG & 8)
    int dbg;       // ablation switches for kernel tuning (UAD_DBG): 1 = no epilogue stores, 2 = no MFMA loop, 4 = no staging
    int math16;    // generic kernel: bf16x3 products (conv_gemm16_kernel) where the tile shape allows
    int npl = 2;   // bf16 planes behind Wp16 (2: bf16x3, 3: bf16x6 -- the F-kind and lane = pixel spatial kernels only)
};

__device__ __forceinline__ void decode_pos(int m, int HS, int WS, int lhs, int lws, int& n, int& i, int& j) {
    if (lws >= 0 && lhs >= 0) {
        j = m & (WS - 1);
        int t = m >> lws;
        i = t & (HS - 1);
        n = t >> lhs;
    } else {
        j = m % WS;
        int t = m / WS;
        i = t % HS;
        n = t / HS;
    }
}

// component-wise select (a float4 `c ? v : zero` is lowered through scratch memory by the compiler)
__device__ __forceinline__ float4 keep4(bool c, float4 v) {
    return make_float4(c ? v.x : 0.f, c ? v.y : 0.f, c ? v.z : 0.f, c ? v.w : 0.f);
}

__device__ __forceinline__ float4 xform4(float4 v, float4 sc, float4 sh, float alpha) {
    float4 r;
    r.x = fmaf(v.x, sc.x, sh.x); r.x = r.x > 0.f ? r.x : r.x * alpha;
    r.y = fmaf(v.y, sc.y, sh.y); r.y = r.y > 0.f ? r.y : r.y * alpha;
    r.z = fmaf(v.z, sc.z, sh.z); r.z = r.z > 0.f ? r.z : r.z * alpha;
    r.w = fmaf(v.w, sc.w, sh.w); r.w = r.w > 0.f ? r.w : r.w * alpha;
    return r;
}

// ================================================================================================
// bf16x3 math mode ("split-bf16"): every fp32 operand x is written x = hi + lo with hi = bf16(x), lo = bf16(x - hi);
// a product is computed as hi*hi + hi*lo + lo*hi on the bf16 matrix cores (products exact, fp32 accumulate), i.e. with
// ~2^-17 relative error per product -- inside the 1e-4 parity bar -- at 3 x 32-cycle v_mfma_f32_32x32x16_bf16 per K=16
// instead of 8 x 64-cycle fp32 MFMAs.  Activations are split while they are staged into LDS (two bf16 planes, same
// bytes as fp32), weights are pre-split by pack_weights_bf16_kernel.
// ================================================================================================
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float x, float y) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));   // lo16 = bf16_rne(x), hi16 = bf16_rne(y)
    return r;
}
__device__ __forceinline__ void split_bf16(float4 v, uint2& hi, uint2& lo) {
    hi.x = cvt_pk_bf16(v.x, v.y);
    hi.y = cvt_pk_bf16(v.z, v.w);
    const float rx = v.x - __uint_as_float(hi.x << 16), ry = v.y - __uint_as_float(hi.x & 0xFFFF0000u);
    const float rz = v.z - __uint_as_float(hi.y << 16), rw = v.w - __uint_as_float(hi.y & 0xFFFF0000u);
    lo.x = cvt_pk_bf16(rx, ry);
    lo.y = cvt_pk_bf16(rz, rw);
}
// 16-byte load through a buffer descriptor: wave-uniform base (descriptor) + wave-uniform byte offset (SGPR) + 32-bit per-lane byte
// offset.  The flat form spends a 64-bit VALU add per load on the same address.
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t rs, unsigned lane_bytes, unsigned uniform_bytes) {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_bytes, (int)uniform_bytes, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Butterfly over 8 consecutive lanes with DPP (one VALU op per step instead of a ds_bpermute round trip through the LDS unit): swap
// inside pairs, swap pairs inside quads, mirror the half row.  Every lane ends with the same value as the xor-shuffle butterfly (the
// operands of each add / or are the same pair, the operations commute).
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ float sum8_dpp(float v) {
    v += __uint_as_float(dpp_u32<0xB1>(__float_as_uint(v)));       // quad_perm [1,0,3,2]
    v += __uint_as_float(dpp_u32<0x4E>(__float_as_uint(v)));       // quad_perm [2,3,0,1]
    v += __uint_as_float(dpp_u32<0x141>(__float_as_uint(v)));      // row_half_mirror
    return v;
}
__device__ __forceinline__ unsigned or8_dpp(unsigned v) {
    v |= dpp_u32<0xB1>(v);
    v |= dpp_u32<0x4E>(v);
    v |= dpp_u32<0x141>(v);
    return v;
}

// Split-K with in-kernel reduction (guide: cross-workgroup hand-off, counter form).  The slabs are written with sc1 (write-through) 16-byte
// buffer stores and read back by the reducer with sc1 loads: performed at the device's coherence point, whichever XCD the contributors of a
// tile ran on, without a release fence (= write-back of the XCD's whole L2).  Order: slab stores -> every wave drains them (s_waitcnt) ->
// barrier -> one relaxed agent-scope ticket per workgroup; the workgroup that draws nsplit - 1 sums the slabs in split order (the order
// splitk_epilogue_kernel uses) and runs the normal epilogue.  The counter is reset by the reducer (zero-initialised at allocation).
__device__ __forceinline__ void sk_store16(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, float4 v) {
    const v4u u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, rs, (int)byte_off, 0, 16);
}
__device__ __forceinline__ float4 sk_load16(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    const v4u u = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}
// true in exactly one of the nsplit workgroups of a tile: the last to arrive.  flag: one int of LDS nobody else touches around the call.
__device__ __forceinline__ bool sk_last_arriver(unsigned* counter, int nsplit, int* flag, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const bool last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nsplit - 1);
        if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last ? 1 : 0;
    }
    __syncthreads();
    const bool r = *flag != 0;
    __syncthreads();          // the flag's LDS word may be reused right away
    return r;
}

// Compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N)
template <int I0, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I0 < N) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, N>(f);
    }
}

__device__ __forceinline__ v16f mfma_bf16(uint4 a, uint4 b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}
template <int NKS>
struct BFrag16 { uint4 hi[NKS], lo[NKS]; };
// NPL bf16 planes of an operand fragment (2: bf16x3 products, 3: bf16x6 -- x = h + m + l, six products, fp32-grade: uad_convk16.inc)
template <int NKS, int NPL>
struct KFrag { uint4 p[NPL][NKS]; };
// x = h + m (+ l): bf16 pairs of a float4
template <int NPL>
__device__ __forceinline__ void split_planes(float4 v, uint2 (&pl)[NPL]) {
    pl[0].x = cvt_pk_bf16(v.x, v.y);
    pl[0].y = cvt_pk_bf16(v.z, v.w);
    float rx = v.x - __uint_as_float(pl[0].x << 16), ry = v.y - __uint_as_float(pl[0].x & 0xFFFF0000u);
    float rz = v.z - __uint_as_float(pl[0].y << 16), rw = v.w - __uint_as_float(pl[0].y & 0xFFFF0000u);
    pl[1].x = cvt_pk_bf16(rx, ry);
    pl[1].y = cvt_pk_bf16(rz, rw);
    if constexpr (NPL == 3) {
        rx -= __uint_as_float(pl[1].x << 16); ry -= __uint_as_float(pl[1].x & 0xFFFF0000u);
        rz -= __uint_as_float(pl[1].y << 16); rw -= __uint_as_float(pl[1].y & 0xFFFF0000u);
        pl[2].x = cvt_pk_bf16(rx, ry);
        pl[2].y = cvt_pk_bf16(rz, rw);
    }
}




// tap order of the D-kind (transposed) kernels: the 25 taps grouped by output-parity class
struct TapOrderD { int tap[25]; int start[5]; };
constexpr TapOrderD make_tap_order_d() {
    TapOrderD o{};
    int n = 0;
    for (int cls = 0; cls < 4; ++cls) {
        o.start[cls] = n;
        for (int tap = 0; tap < 25; ++tap) {
            const int ky = tap / 5, kx = tap % 5;
            const int py = (ky + 1) & 1, px = (kx + 1) & 1;
            if (py * 2 + px == cls) o.tap[n++] = tap;
        }
    }
    o.start[4] = n;
    return o;
}
constexpr TapOrderD kTapOrderD = make_tap_order_d();

// Launches without the AQL barrier bit (hipExtAnyOrderLaunch): see uad_gemm.hip.  One flag per host thread and translation unit, consumed by the next launch.
static thread_local bool g_any_order_next = false;      // (per host thread: another thread's launch must not consume it)
static thread_local bool g_any_order_w_next = false;      // ... the same for the next channel-major filter-gradient launch (the first one of a backward, behind loss.finalize)
#define UAD_W_LAUNCH(kern, grid, block, lds, st, ...)                                                                       \
    do {                                                                                                                    \
        if (g_any_order_w_next) { g_any_order_w_next = false; hipExtLaunchKernelGGL(kern, grid, block, lds, st, nullptr, nullptr, 1u, __VA_ARGS__); } \
        else hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                                   \
    } while (0)
#define UAD_SPATIAL_LAUNCH(kern, grid, block, lds, st, arg)                                                                 \
    do {                                                                                                                    \
        if (g_any_order_next) { g_any_order_next = false; hipExtLaunchKernelGGL(kern, grid, block, lds, st, nullptr, nullptr, 1u, arg); } \
        else hipLaunchKernelGGL(kern, grid, block, lds, st, arg);                                                           \
    } while (0)

}  // namespace
