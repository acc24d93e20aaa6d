// fin_mode (aid_kernels.h): the fold of a sample's epilogue partials by the LAST tile / block of the sample to finish -- shared by the row-shared
// Winograd kernels (aid_conv_wino.hip) and the output pass of the 2-D form (aid_wino2d.hip).  Same summation order as group_stats_final /
// norm_bwd_coef (aid_norm.hip): lane l adds partials l, l + 64, ..., then the xor tree -- hence the same bits as the launches it replaces.
// The partials are published with agent-scope stores before the arrival counter is bumped and read back past the XCD's L2 the same way.
//
// Ordering contract (ADVICE r4 asked for release / acquire here; this is why it is not that): an agent-scope release fence on gfx950 writes back, and the
// matching acquire invalidates, the WHOLE L2 of the XCD -- once per tile, with that L2 full of the tile's own output.  What the protocol needs is less:
//   publisher: every partial is an agent-scope (sc1) atomic store, i.e. written through to the coherent level; `s_waitcnt vmcnt(0)` (stores count in
//              vmcnt) then a workgroup barrier, then ONE relaxed agent-scope fetch_add by thread 0 -- the counter cannot be seen before the stores landed;
//   consumer:  the workgroup that reads fin_total - 1 passes a barrier and reads the partials with agent-scope (sc1) atomic loads, which bypass the
//              possibly stale lines of its own L2.
// The compiler cannot reorder across either side: `asm volatile("s_waitcnt ..." ::: "memory")` is a compiler-level fence, `__syncthreads()` carries
// workgroup-scope release / acquire fences, and atomics on the same object are never reordered with each other.  The counters / flags are left zero by
// every launch.  Host side (plan.py): a launcher that returns an error never ran its kernel, so nothing is half-way then -- the plan still re-zeroes after
// synchronising (Plan._fail); a run that is ABANDONED half-way for any other reason (exception, interrupt) marks the plan dirty and the next run re-zeroes
// first (Plan.run / zero_scratch); an asynchronous device fault kills the process on this stack and needs no recovery.
#pragma once
#include "aid_common.h"

static __device__ __forceinline__ double aid_ld_agent(const double* q) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
static __device__ __forceinline__ void aid_st_agent(double* q, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(q), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __attribute__((noinline)) void aid_wino_fin(int mode, const double* ws, int nk, int b, int C, double n, const float* gamma, const float* mod, int64_t mod_ld,
                                                       float eps, float* scale, float* stats, double* sh, int tid, int nthr) {
    const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    if (mode == 1) {
        for (int g = wave; g < 8; g += nw) {
            const double* w = ws + ((int64_t)(b * 8 + g) * nk) * 2;
            double s = 0.0, ss = 0.0;
            for (int i = lane; i < nk; i += 64) { s += aid_ld_agent(w + 2 * i); ss += aid_ld_agent(w + 2 * i + 1); }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { s += __shfl_xor(s, off, 64); ss += __shfl_xor(ss, off, 64); }
            if (lane == 0) { sh[2 * g] = s; sh[2 * g + 1] = ss; }
        }
        __syncthreads();
        const int cg = C >> 3;
        for (int c = tid; c < C; c += nthr) {
            const int g = c / cg;
            const double s = sh[2 * g], ss = sh[2 * g + 1];
            const double mean = s / n;
            double var = (ss - n * mean * mean) / (n - 1.0);
            if (var < 0.0) var = 0.0;
            const double inv = 1.0 / (sqrt(var) + (double)eps);
            const double m = mod ? (1.0 + (double)mod[(int64_t)b * mod_ld + c]) : 1.0;
            scale[(int64_t)b * C + c] = (float)((double)gamma[c] * m * inv);
            if (stats && c == g * cg) {
                stats[((int64_t)b * 8 + g) * 2 + 0] = (float)mean;
                stats[((int64_t)b * 8 + g) * 2 + 1] = (float)inv;
            }
        }
    } else {
        for (int g = wave; g < 8; g += nw) {
            const int i = b * 8 + g;
            const double* w = ws + (int64_t)i * nk;
            double d = 0.0;
            for (int k = lane; k < nk; k += 64) d += aid_ld_agent(w + k);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) d += __shfl_xor(d, off, 64);
            if (lane == 0) {
                const double inv = (double)stats[2 * i + 1];
                const double sd = 1.0 / inv - (double)eps;
                scale[i] = (sd > 0.0) ? (float)(d * inv / ((n - 1.0) * sd)) : 0.f;
            }
        }
    }
}
