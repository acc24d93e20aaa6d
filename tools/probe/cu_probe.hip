// Measurement helper (NOT part of the product library): which compute units does a stream run on, and what bandwidth does a plain streaming copy get there?
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probe/cu_probe.hip -o tools/probe/cu_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

// one word per workgroup: XCC_ID (hwreg 20) in bits 31:24, HW_ID (hwreg 4: cu_id 11:8, sh_id 12, se_id 15:13) in bits 15:0
__global__ void cu_where_kernel(unsigned* out, int spin) {
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    // hold the CU for a moment so that the dispatcher has to spread the grid over everything the stream may use
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 24) | (hw & 0xFFFF);
}

// grid-stride float4 copy with UNROLL independent 16-byte loads in flight per lane
typedef float f4 __attribute__((ext_vector_type(4)));
template <int UNROLL>
__global__ void __launch_bounds__(256) copy_kernel(const f4* __restrict__ x, f4* __restrict__ y, size_t n4) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(x + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) __builtin_nontemporal_store(v[u], y + i + u * stride);
    }
    for (; i < n4; i += stride) y[i] = x[i];
}

extern "C" int cu_where(unsigned* out, int nblocks, int spin, hipStream_t s) {
    hipLaunchKernelGGL(cu_where_kernel, dim3(nblocks), dim3(64), 0, s, out, spin);
    return (int)hipGetLastError();
}

extern "C" int cu_copy(const void* x, void* y, size_t nbytes, int nblocks, int unroll, hipStream_t s) {
    size_t n4 = nbytes / 16;
    if (unroll >= 8) hipLaunchKernelGGL(copy_kernel<8>, dim3(nblocks), dim3(256), 0, s, (const f4*)x, (f4*)y, n4);
    else if (unroll >= 4) hipLaunchKernelGGL(copy_kernel<4>, dim3(nblocks), dim3(256), 0, s, (const f4*)x, (f4*)y, n4);
    else hipLaunchKernelGGL(copy_kernel<1>, dim3(nblocks), dim3(256), 0, s, (const f4*)x, (f4*)y, n4);
    return (int)hipGetLastError();
}
