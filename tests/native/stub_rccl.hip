// TEST INFRASTRUCTURE (tests/test_gpu_dp_stub_collective.py): a stand-in for librccl.so.1 with the five entry points libuad_hip.so binds at run time
// (uad_model.hip: rccl_api).  Its "all-reduce" is a kernel on the caller's stream that DOUBLES the buffer -- what a sum over two ranks holding identical
// gradients produces -- so a one-GPU box can check that the library enqueues each bucket's collective where the data really is final: a collective that ran
// before a slab reduction had written its gradients would leave them un-doubled.  (RCCL itself refuses two ranks on one device, and over ONE rank an
// in-place all-reduce is a no-op that hides every ordering mistake.)
#include <hip/hip_runtime.h>
#include <string.h>
extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef void* ncclComm_t;
typedef int ncclResult_t;
static long long g_calls = 0, g_elems = 0;
__global__ void stub_twice(const float* s, float* r, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) r[i] = s[i] + s[i];
}
__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { memset(id, 7, sizeof *id); return 0; }
__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t* c, int, ncclUniqueId, int) { *c = (ncclComm_t)&g_calls; return 0; }
__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t) { return 0; }
__attribute__((visibility("default"))) const char* ncclGetErrorString(ncclResult_t) { return "stub_rccl"; }
__attribute__((visibility("default"))) ncclResult_t ncclAllReduce(const void* s, void* r, size_t count, int, int, ncclComm_t, hipStream_t st) {
    ++g_calls; g_elems += (long long)count;
    hipLaunchKernelGGL(stub_twice, dim3(256), dim3(256), 0, st, (const float*)s, (float*)r, count);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
__attribute__((visibility("default"))) long long stub_rccl_calls(void) { return g_calls; }
__attribute__((visibility("default"))) long long stub_rccl_elems(void) { return g_elems; }
}
