// common.h -- shared helpers for liblidbox_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "lidbox_hip.h"

void lidbox_set_error(const char* fmt, ...);

#define LBX_ARG(cond, msg)                                              \
    do {                                                                \
        if (!(cond)) {                                                  \
            lidbox_set_error("%s: invalid argument: %s", __func__, msg); \
            return LIDBOX_E_INVALID;                                    \
        }                                                               \
    } while (0)

#define LBX_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess) {                                                        \
            lidbox_set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e__)); \
            return LIDBOX_E_LAUNCH;                                                     \
        }                                                                               \
    } while (0)

#define LBX_LAUNCH_OK()                                                                 \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            lidbox_set_error("%s: kernel launch failed: %s", __func__, hipGetErrorString(e__)); \
            return LIDBOX_E_LAUNCH;                                                     \
        }                                                                               \
    } while (0)

static inline long lbx_cdiv(long a, long b) { return (a + b - 1) / b; }

// LDS hand-off between lanes of ONE wave: the DS instructions of a wave are issued and executed in
// order, so a later ds_read observes an earlier ds_write of any lane without a hardware wait.
// Only the COMPILER must keep program order -- a wavefront-scope fence would do that too, but it
// lowers to s_waitcnt vmcnt(0) lgkmcnt(0) and serialises every hand-off behind all outstanding
// global loads/stores.
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware remap of a linear workgroup id: the dispatcher places block b on XCD b % 8; give
// each XCD one contiguous chunk of the work so neighbours share that XCD's L2.  Bijective for
// any nwg (speed only -- never correctness).
__device__ __forceinline__ unsigned xcd_chunk_id(unsigned bid, unsigned nwg) {
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    const unsigned q = nwg >> 3, r = nwg & 7u;
    const unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}
