// lds_dma.h -- LDS-DMA primitives of gfx950 shared by the fp32 (gemm_sk.h, gemm_dma.h) and bf16 (gemm16_dma.h) GEMM kernels:
// global_load_lds_dwordx4 in the saddr form (hipcc picks the flat 64-bit form for the builtin: twice the address payload,
// measured 0.83 vs 0.86-0.89 of the MFMA peak in the fp32 loop, profiles/r03_skgemm_ablation_*.txt), counted vmcnt waits.
#pragma once

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(16))) float g_sk_zero[4];        // source of the chunks past K of a tail step

// LDS-DMA of 16 bytes per lane: lane i of the wave lands at lds_dst + 16 i (M0 = wave-uniform destination).
__device__ __forceinline__ void sk_dma_s(const float* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
                 : "memory");
}
__device__ __forceinline__ void sk_dma_f(const float* p, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(p), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
                 : "memory");
}
// a pointer every lane holds the same value of, provably so for the compiler (SGPR pair): the "s" operands of the DMA
// statements otherwise cost a waterfall loop each
__device__ __forceinline__ const float* sk_uniform(const float* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const float*)(((unsigned long long)hi << 32) | lo);
}
template <int N>
__device__ __forceinline__ void sk_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace
