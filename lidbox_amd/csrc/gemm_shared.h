// gemm_shared.h -- pieces common to the fp32 (gemm.hip) and bf16 (gemm_bf16.hip) MFMA GEMM families:
// implicit-row descriptors, the fused epilogue of one workgroup tile, the fixed-order reduce kernels
// of the split decompositions, and argument validation.  Both families produce the accumulator
// layout of a 32x32 MFMA block (lane -> column lane&31, rows (r&3) + 8*(r>>2) + 4*(lane>>5)).
#pragma once
#include <stdint.h>
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NUM_CU = 256;

struct RowsD {
    const float* base;
    long bs, rs;
    int batch, rpb;
};

struct RowsOutD {
    float* base;
    long bs, rs;
    int batch, rpb;
};

// m < 2^31 always (checked on the host): 32-bit unsigned division
__device__ __forceinline__ long row_offset(const RowsD& r, unsigned m) {
    if (r.batch == 1) return (long)m * r.rs;
    const unsigned b = m / (unsigned)r.rpb;
    return (long)b * r.bs + (long)(m - b * (unsigned)r.rpb) * r.rs;
}
__device__ __forceinline__ long row_offset(const RowsOutD& r, unsigned m) {
    if (r.batch == 1) return (long)m * r.rs;
    const unsigned b = m / (unsigned)r.rpb;
    return (long)b * r.bs + (long)(m - b * (unsigned)r.rpb) * r.rs;
}

__device__ __forceinline__ float apply_epi(float v, int epi, float bias, const float* aux, long idx,
                                           const float* dst) {
    switch (epi) {
        case LIDBOX_EPI_BIAS: return v + bias;
        case LIDBOX_EPI_BIAS_RELU: return fmaxf(v + bias, 0.f);
        case LIDBOX_EPI_RELU_MASK: return aux[idx] > 0.f ? v : 0.f;
        case LIDBOX_EPI_ACCUM: return v + *dst;
        case LIDBOX_EPI_ACCUM_RELU_MASK: return *dst + (aux[idx] > 0.f ? v : 0.f);
        case LIDBOX_EPI_ACCUM_RELU: return fmaxf(*dst + v, 0.f);
        case LIDBOX_EPI_RELU: return fmaxf(v, 0.f);
        default: return v;
    }
}

template <bool V>
struct FarTag {
    static constexpr bool value = V;
};
template <int V>
struct IntTag {
    static constexpr int value = V;
};
#ifndef LBX_EPI_INNER
#define LBX_EPI_INNER 1            // 0: A/B aid (every block through the predicated epilogue)
#endif

// Epilogue of one workgroup tile held as MI x NJ accumulator blocks per wave (waves 2 x 2).
// Rows of this launch are [m_beg, M).  gridDim.y > 1 = split along K: raw partial sums go to
// P[split][row - m_beg][n] and rows_reduce_kernel applies the epilogue.
// EXT (bf16-storage kernels only; the fp32 instantiations compile exactly as before): Cd.base may be NULL -- the finished
// values then exist only as the bf16 shadow -- and mask16, when given, is the ReLU mask source read as bfloat16 at C's
// element offsets (the shadow of the activation) instead of the fp32 aux.
template <int MI, int NJ, bool EXT = false>
__device__ __forceinline__ void store_rows_tile(const f32x16 (&acc)[MI][NJ], long m0, int n0, int wm, int wn, int lane,
                                                long m_beg, long M, int N, int epi, const float* __restrict__ aux,
                                                const RowsOutD& Cd, float* __restrict__ P, int split,
                                                unsigned long long mask_bits = 0ull, bool have_mask_bits = false,
                                                unsigned short* __restrict__ shadow = nullptr,
                                                const unsigned short* __restrict__ mask16 = nullptr,
                                                int partial_override = -1) {
    // shadow (bf16-storage GEMMs, gemm_bf16.hip): a bfloat16 copy of every finished value at the same element offset as C
    // (round-to-nearest-even) -- the operand the next GEMM reads; never written for split-K partial sums.
    // Epilogue kind as four uniform flags (no per-element switch); bias and column state hoisted.
    const int h = lane >> 5, l = lane & 31;
    // partial_override: 0 / 1 from kernels whose K splits are not grid.y (gemm_dma.h: several GEMMs in one launch)
    const bool partial = partial_override < 0 ? gridDim.y > 1 : partial_override != 0;
    const bool has_bias = !partial && (epi == LIDBOX_EPI_BIAS || epi == LIDBOX_EPI_BIAS_RELU);
    const bool do_relu = !partial && (epi == LIDBOX_EPI_BIAS_RELU || epi == LIDBOX_EPI_ACCUM_RELU || epi == LIDBOX_EPI_RELU);
    const bool has_mask = !partial && (epi == LIDBOX_EPI_RELU_MASK || epi == LIDBOX_EPI_ACCUM_RELU_MASK);
    const bool accum = !partial && (epi == LIDBOX_EPI_ACCUM || epi == LIDBOX_EPI_ACCUM_RELU_MASK || epi == LIDBOX_EPI_ACCUM_RELU);
    int col[NJ];
    bool colok[NJ];
    float bias[NJ];
#pragma unroll
    for (int bj = 0; bj < NJ; ++bj) {
        col[bj] = n0 + wn * (32 * NJ) + bj * 32 + l;
        colok[bj] = col[bj] < N;
        bias[bj] = (has_bias && colok[bj]) ? aux[col[bj]] : 0.f;
    }
    float* const out_base = partial ? P + ((long)split * (M - m_beg) - m_beg) * N : Cd.base;   // P[split][row - m_beg][n]
    const long out_rs = partial ? (long)N : Cd.rs;
    const bool batched = !partial && Cd.batch != 1;
    // Loads and stores of a 32x32 block are issued in BATCHES of eight rows: the eight mask values and the eight old
    // values of a column first, then the arithmetic, then the eight stores.  Written row by row (load mask -> select
    // -> load old -> add -> store) the compiler cannot move row r+1's loads above row r's store (the pointers may
    // alias), so every row paid two or three dependent round trips to memory while the workgroup held its slot
    // without issuing MFMAs (96 round trips per wave of a 128x64 tile; now 8).  Rows / columns outside the matrix
    // read a safe address (offset 0) and only their store is predicated: no control flow between the loads.  Eight
    // rows at a time keeps the temporaries inside the registers the K loop no longer needs (occupancy unchanged).
    if (!has_mask && !accum) {
        // store-only epilogues (forward: bias / bias + ReLU, partial sums): nothing to batch, rows go out as they come.
        // INNER (wave-uniform, per 32-row block): all rows inside the matrix, all columns inside N, utterances of >= 28 rows
        // (at most one boundary inside the block, a select): no per-lane branches around the stores, 32-bit offsets.
        auto fast_block = [&](auto inner_tag, auto nowrap_tag, auto bi_tag) {
            constexpr bool INNER = decltype(inner_tag)::value;
            constexpr bool NOWRAP = decltype(nowrap_tag)::value;      // INNER only: no utterance boundary inside the block
            constexpr int bi = decltype(bi_tag)::value;
            const long rbase = m0 + wm * (32 * MI) + bi * 32 + 4 * h;
            unsigned b0 = 0, t0 = (unsigned)rbase;
            if (batched) { b0 = (unsigned)rbase / (unsigned)Cd.rpb; t0 = (unsigned)rbase - b0 * (unsigned)Cd.rpb; }
            const long off0 = batched ? (long)b0 * Cd.bs + (long)t0 * Cd.rs : rbase * out_rs;
            const unsigned t_wrap = batched ? (unsigned)Cd.rpb : 0xffffffffu;
            const long wrap = batched ? Cd.bs - (long)Cd.rpb * Cd.rs : 0;
            if constexpr (INNER) {
                // one 64-bit base per lane and column block, 32-bit row offsets (the inner test bounds them): per row a
                // compare + select + add where an utterance boundary may fall inside the block, a scalar otherwise
                float* pj[NJ];
                unsigned short* sj[NJ];
#pragma unroll
                for (int bj = 0; bj < NJ; ++bj) {
                    pj[bj] = out_base + off0 + col[bj];
                    sj[bj] = shadow + off0 + col[bj];
                }
                const unsigned rs32 = (unsigned)out_rs, wrap32 = (unsigned)wrap;
                const bool st32 = !EXT || out_base, st16 = shadow && !partial;
                const unsigned wfrom = t_wrap - t0;                      // rows from this lane's first row to the boundary
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    const unsigned drs = (unsigned)dr * rs32;                                         // uniform: a scalar product
                    const unsigned sel = NOWRAP ? 0u : ((unsigned)dr >= wfrom ? wrap32 : 0u);           // per lane: behind the boundary
#pragma unroll
                    for (int bj = 0; bj < NJ; ++bj) {
                        float x = acc[bi][bj][r] + bias[bj];
                        if (do_relu) x = fmaxf(x, 0.f);
                        if (st32) (pj[bj] + drs)[sel] = x;
                        if (st16) (sj[bj] + drs)[sel] = __builtin_bit_cast(unsigned short, (__bf16)x);
                    }
                }
                return;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                const long row = rbase + dr;
                if (row >= M) continue;
                long off = off0 + dr * out_rs;
                if (batched && t0 + dr >= (unsigned)Cd.rpb) off = row_offset(Cd, (unsigned)row);   // crossed an utterance
#pragma unroll
                for (int bj = 0; bj < NJ; ++bj) {
                    if (!colok[bj]) continue;
                    float x = acc[bi][bj][r] + bias[bj];
                    if (do_relu) x = fmaxf(x, 0.f);
                    if (!EXT || out_base) out_base[off + col[bj]] = x;
                    if (shadow && !partial) shadow[off + col[bj]] = __builtin_bit_cast(unsigned short, (__bf16)x);
                }
            }
        };
        auto fast_one = [&](auto bi_tag) {
            constexpr int bi = decltype(bi_tag)::value;
            const long rb0 = m0 + wm * (32 * MI) + bi * 32;                 // the block's first row (wave-uniform)
            const long wrap_u = batched ? Cd.bs - (long)Cd.rpb * Cd.rs : 0;
            const bool inner = LBX_EPI_INNER && rb0 + 32 <= M && n0 + wn * (32 * NJ) + 32 * NJ <= N && (!batched || (unsigned)Cd.rpb >= 28) &&
                               out_rs > 0 && out_rs < (1L << 25) && wrap_u >= 0 && wrap_u < (1L << 30);
            if (!inner) { fast_block(FarTag<false>{}, FarTag<false>{}, bi_tag); return; }
            const bool nowrap = !batched || (unsigned)rb0 % (unsigned)Cd.rpb + 32u <= (unsigned)Cd.rpb;
            if (nowrap) fast_block(FarTag<true>{}, FarTag<true>{}, bi_tag);
            else fast_block(FarTag<true>{}, FarTag<false>{}, bi_tag);
        };
        fast_one(IntTag<0>{});
        if constexpr (MI > 1) fast_one(IntTag<1>{});
        return;
    }
    const bool far = batched && (unsigned)Cd.rpb < 28;          // wave-uniform
    // INNER (wave-uniform, per 32-row block): every row of the block is inside the matrix and every column inside N -- the
    // common case.  Neither the loads nor the stores are predicated then (an utterance boundary inside the block stays a
    // select); measured: the predicated form costs the mask epilogues 6 % of a K = 512 launch.
    auto block = [&](auto far_tag, auto inner_tag, auto bi_tag) {
        constexpr bool FAR = decltype(far_tag)::value;
        constexpr bool INNER = decltype(inner_tag)::value;
        constexpr int bi = decltype(bi_tag)::value;
        const long rbase = m0 + wm * (32 * MI) + bi * 32 + 4 * h;
        // (b, t) of the block's first row; the other 15 rows are <= 27 below it
        unsigned b0 = 0, t0 = (unsigned)rbase;
        if (batched) { b0 = (unsigned)rbase / (unsigned)Cd.rpb; t0 = (unsigned)rbase - b0 * (unsigned)Cd.rpb; }
        const long off0 = batched ? (long)b0 * Cd.bs + (long)t0 * Cd.rs : rbase * out_rs;
        // An utterance boundary inside the block: rows behind it restart in the next batch element.  With >= 28 rows
        // per utterance there is at most one boundary and the offset is a select (no control flow between the loads);
        // shorter utterances (wave-uniform test) take the general formula.
        const unsigned t_wrap = batched ? (unsigned)Cd.rpb : 0xffffffffu;
        const long wrap = batched ? Cd.bs - (long)Cd.rpb * Cd.rs : 0;
        // element `dr` rows below the lane's first row, column c, of an array laid out like C: INNER blocks use one 64-bit base
        // per array, a scalar row product and a 32-bit select behind an utterance boundary (the inner test bounds them)
        const unsigned rs32 = (unsigned)out_rs, wrap32 = (unsigned)wrap, wfrom = t_wrap - t0;
        auto row_off = [&](int dr) -> long {
            long o;
            if (FAR) o = rbase + dr < M ? row_offset(Cd, (unsigned)(rbase + dr)) : 0;
            else o = off0 + dr * out_rs + (t0 + dr >= t_wrap ? wrap : 0);
            return rbase + dr < M ? o : 0;
        };
#pragma unroll
        for (int bj = 0; bj < NJ; ++bj) {
            const int c = (INNER || colok[bj]) ? col[bj] : 0;
            auto at = [&](auto* base, int dr) {
                if constexpr (INNER) return (base + off0 + c + (unsigned)dr * rs32) + ((unsigned)dr >= wfrom ? wrap32 : 0u);
                else return base + row_off(dr) + c;
            };
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 8) {
                float v[8], mv[8], ov[8];
                if (EXT && has_mask && mask16) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)                  // sign and zero-ness are all that is looked at: bits << 16 is the value
                        mv[i] = __builtin_bit_cast(float, (unsigned)*at(mask16, ((r0 + i) & 3) + 8 * ((r0 + i) >> 2)) << 16);
                } else if (has_mask && !have_mask_bits) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) mv[i] = *at(aux, ((r0 + i) & 3) + 8 * ((r0 + i) >> 2));
                }
                if (accum) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) ov[i] = *at(out_base, ((r0 + i) & 3) + 8 * ((r0 + i) >> 2));
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = r0 + i;
                    float x = acc[bi][bj][r] + bias[bj];
                    if (has_mask) {
                        // have_mask_bits: the caller read the mask during its K loop; bit (bi*NJ + bj)*16 + r
                        const bool keep = have_mask_bits ? ((mask_bits >> ((bi * NJ + bj) * 16 + r)) & 1ull) != 0 : mv[i] > 0.f;
                        x = keep ? x : 0.f;
                    }
                    if (accum) x += ov[i];
                    if (do_relu) x = fmaxf(x, 0.f);
                    v[i] = x;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int dr = ((r0 + i) & 3) + 8 * ((r0 + i) >> 2);
                    if (INNER || (rbase + dr < M && colok[bj])) {
                        if (!EXT || out_base) *at(out_base, dr) = v[i];
                        if (shadow) *at(shadow, dr) = __builtin_bit_cast(unsigned short, (__bf16)v[i]);
                    }
                }
            }
        }
    };
    auto run = [&](auto far_tag) {
        auto one = [&](auto bi_tag) {
            constexpr int bi = decltype(bi_tag)::value;
            const long rb0 = m0 + wm * (32 * MI) + bi * 32;                 // the block's first row (wave-uniform)
            const long wrap_u = batched ? Cd.bs - (long)Cd.rpb * Cd.rs : 0;
            const bool inner = LBX_EPI_INNER && !decltype(far_tag)::value && rb0 + 32 <= M && n0 + wn * (32 * NJ) + 32 * NJ <= N &&
                               out_rs > 0 && out_rs < (1L << 25) && wrap_u >= 0 && wrap_u < (1L << 30);
            if (inner) block(far_tag, FarTag<true>{}, bi_tag);
            else block(far_tag, FarTag<false>{}, bi_tag);
        };
        one(IntTag<0>{});
        if constexpr (MI > 1) one(IntTag<1>{});
        static_assert(MI <= 2, "store_rows_tile: MI <= 2");
    };
    if (far) run(FarTag<true>{});
    else run(FarTag<false>{});
}

// wgrad tiles: the wave's MI x NJ accumulator blocks -> Pd[row][col] (one slice's partial sums, leading dimension N), rows
// < K1, columns < N.  A 32 x 32 block wholly inside takes unpredicated stores (wave-uniform test).
template <int MI, int NJ>
__device__ __forceinline__ void store_partial_blocks(float* __restrict__ Pd, const f32x16 (&acc)[MI][NJ], int i0, int n0, int wm, int wn, int lane,
                                                     int K1, int N) {
    const int h = lane >> 5, l = lane & 31;
#pragma unroll
    for (int bj = 0; bj < NJ; ++bj) {
        const int cb = n0 + wn * (32 * NJ) + bj * 32, col = cb + l;
#pragma unroll
        for (int bi = 0; bi < MI; ++bi) {
            const int rb = i0 + wm * (32 * MI) + bi * 32;
            if (LBX_EPI_INNER && rb + 32 <= K1 && cb + 32 <= N) {
                float* p = Pd + (long)(rb + 4 * h) * N + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) p[(long)((r & 3) + 8 * (r >> 2)) * N] = acc[bi][bj][r];
            } else if (col < N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rb + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < K1) Pd[(long)row * N + col] = acc[bi][bj][r];
                }
            }
        }
    }
}

// C rows [m_beg, m_beg + Msub) = epi( sum_s P[s][Msub][N] ), fixed order
// (bf16-storage launches: Cd.base may be NULL, mask16 = the ReLU mask source as bfloat16 -- see store_rows_tile)
__global__ void rows_reduce_kernel(const float* __restrict__ P, int splits, long m_beg, long Msub, int N, RowsOutD Cd,
                                   int epi, const float* __restrict__ aux, unsigned short* __restrict__ shadow = nullptr,
                                   const unsigned short* __restrict__ mask16 = nullptr) {
    const long total = Msub * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += P[(long)k * total + i];
        const long row = i / N;
        const int col = (int)(i - row * N);
        const long off = row_offset(Cd, (unsigned)(m_beg + row));
        const float bias = (epi == LIDBOX_EPI_BIAS || epi == LIDBOX_EPI_BIAS_RELU) ? aux[col] : 0.f;
        float* dst = Cd.base + off + col;
        float x;
        if (mask16 && epi == LIDBOX_EPI_RELU_MASK) x = __builtin_bit_cast(float, (unsigned)mask16[off + col] << 16) > 0.f ? s : 0.f;
        else if (mask16 && epi == LIDBOX_EPI_ACCUM_RELU_MASK)
            x = *dst + (__builtin_bit_cast(float, (unsigned)mask16[off + col] << 16) > 0.f ? s : 0.f);
        else x = apply_epi(s, epi, bias, aux, off + col, dst);
        if (Cd.base) *dst = x;
        if (shadow) shadow[off + col] = __builtin_bit_cast(unsigned short, (__bf16)x);
    }
}

// C[i] (+)= sum_s P[s][i]; bias_grad[n] (+)= sum_s Pc[s][n]   (fixed order)
__global__ void splitk_reduce_kernel(const float* __restrict__ P, const float* __restrict__ Pc, int splits,
                                     long n, int N, float* __restrict__ Cm, long ldc, int accumulate,
                                     float* __restrict__ bias_grad) {
    const long total = n + (bias_grad ? N : 0);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        if (i < n) {
            for (int k = 0; k < splits; ++k) s += P[(long)k * n + i];
            const long row = i / N, col = i - row * N;
            float* d = Cm + row * ldc + col;
            *d = accumulate ? *d + s : s;
        } else {
            const long c = i - n;
            for (int k = 0; k < splits; ++k) s += Pc[(long)k * N + c];
            bias_grad[c] = accumulate ? bias_grad[c] + s : s;
        }
    }
}

// A pending fixed-order reduce of wgrad slices (what splitk_reduce4_kernel does), as a value that can travel: run by its own
// launch, or by the LEADING `nblocks` workgroups of a later GEMM launch on the same stream ("carried" reduce, gemm_dma.h /
// gemm16_dma.h: the bandwidth-bound sums then run beside that launch's MFMA-bound tiles instead of between two launches).
// Same element -> thread map, same slice order 0 .. splits-1 for every element whoever runs it: bit-identical results.
struct ReduceJob {
    const float* P = nullptr;       // [splits][n]
    const float* Pc = nullptr;      // [splits][N] (bias gradient slices) or NULL
    float* Cm = nullptr;
    float* bias_grad = nullptr;
    long n = 0, ldc = 0;
    int splits = 0, N = 0, accumulate = 0;
    unsigned nblocks = 0;           // workgroups of 256 threads that share the job; 0 = no job
};

// one matrix of slices X[splits][n] -> C quads, two quads per pass (eight 16-byte loads in flight per thread)
__device__ __forceinline__ void reduce_slices(const float* X, long n, int splits, long nquads, unsigned blk, unsigned nblocks, int N, float* Cm,
                                              long ldc, int accumulate) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    // buffer loads: one wave-uniform descriptor of the slices, the slice offset k * n * 4 as the scalar offset, a 32-bit byte
    // offset per lane -- no per-load 64-bit addresses (the GEMM kernels that carry a job have 80 registers)
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, 0xffffffffu, 0x00020000);
    const unsigned slice = (unsigned)(n * 4);
    const long stride_q = (long)nblocks * 256;
    auto put = [&](long q, f4 s) {
        const long i = 4 * q, row = i / N, col = i - row * N;
        f4* d4 = reinterpret_cast<f4*>(Cm + row * ldc + col);
        *d4 = accumulate ? *d4 + s : s;
    };
#define LBX_RJ_LD(v, o) __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (v), (o), 0))
    long q0 = (long)blk * 256 + threadIdx.x;
    for (; q0 + stride_q < nquads; q0 += 2 * stride_q) {             // two quads per pass
        const long q1 = q0 + stride_q;
        const unsigned v0 = (unsigned)(16 * q0), v1 = (unsigned)(16 * q1);
        f4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        unsigned so = 0;
        for (; k + 4 <= splits; k += 4, so += 4 * slice) {
            const f4 a0 = LBX_RJ_LD(v0, so), b0 = LBX_RJ_LD(v0, so + slice), c0 = LBX_RJ_LD(v0, so + 2 * slice), d0 = LBX_RJ_LD(v0, so + 3 * slice);
            const f4 a1 = LBX_RJ_LD(v1, so), b1 = LBX_RJ_LD(v1, so + slice), c1 = LBX_RJ_LD(v1, so + 2 * slice), d1 = LBX_RJ_LD(v1, so + 3 * slice);
            s0 = (((s0 + a0) + b0) + c0) + d0;
            s1 = (((s1 + a1) + b1) + c1) + d1;
        }
        for (; k < splits; ++k, so += slice) {
            s0 += LBX_RJ_LD(v0, so);
            s1 += LBX_RJ_LD(v1, so);
        }
        put(q0, s0);
        put(q1, s1);
    }
    if (q0 < nquads) {                                                // a thread's last, single quad: eight slices in flight
        const unsigned v0 = (unsigned)(16 * q0);
        f4 s0 = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        unsigned so = 0;
        for (; k + 8 <= splits; k += 8, so += 8 * slice) {
            const f4 a0 = LBX_RJ_LD(v0, so), b0 = LBX_RJ_LD(v0, so + slice), c0 = LBX_RJ_LD(v0, so + 2 * slice), d0 = LBX_RJ_LD(v0, so + 3 * slice);
            const f4 e0 = LBX_RJ_LD(v0, so + 4 * slice), f0 = LBX_RJ_LD(v0, so + 5 * slice), g0 = LBX_RJ_LD(v0, so + 6 * slice),
                     h0 = LBX_RJ_LD(v0, so + 7 * slice);
            s0 = (((((((s0 + a0) + b0) + c0) + d0) + e0) + f0) + g0) + h0;
        }
        for (; k + 4 <= splits; k += 4, so += 4 * slice) {
            const f4 a0 = LBX_RJ_LD(v0, so), b0 = LBX_RJ_LD(v0, so + slice), c0 = LBX_RJ_LD(v0, so + 2 * slice), d0 = LBX_RJ_LD(v0, so + 3 * slice);
            s0 = (((s0 + a0) + b0) + c0) + d0;
        }
        for (; k < splits; ++k, so += slice) s0 += LBX_RJ_LD(v0, so);
        put(q0, s0);
    }
#undef LBX_RJ_LD
}

// C[i] (+)= sum_s P[s][i]; bias_grad[n] (+)= sum_s Pc[s][n], four consecutive elements per thread as 16-byte accesses (every
// element adds its slices in the order 0 .. splits-1: bit-identical to the scalar kernel).  `blk` of `j.nblocks` workgroups.
// A carried job runs on few workgroups beside a GEMM's tiles, and a pass is a memory round trip under load (microseconds),
// so the dependent passes per thread are what counts (segment1's 1.5 M-element job on 96 workgroups, one quad per pass: 16
// passes made those workgroups the launch's long pole) -- hence two quads per pass.
__device__ __forceinline__ void reduce_job_run(const ReduceJob& j, unsigned blk) {
    reduce_slices(j.P, j.n, j.splits, j.n >> 2, blk, j.nblocks, j.N, j.Cm, j.ldc, j.accumulate);
    if (j.bias_grad) reduce_slices(j.Pc, (long)j.N, j.splits, (long)(j.N >> 2), blk, j.nblocks, j.N, j.bias_grad, (long)j.N, j.accumulate);
}

// up to two pending jobs sharing the leading workgroups of one launch (by-value kernel arguments live in SGPRs: a third job no longer leaves the LDS-DMA statements their scalar operands): job i owns j[i].nblocks consecutive blocks
constexpr int MAX_CARRY = 2;
struct ReduceJobs {
    ReduceJob j[MAX_CARRY];
    unsigned total = 0;             // sum of j[i].nblocks
};
__device__ __forceinline__ void reduce_jobs_run(const ReduceJobs& js, unsigned blk) {
#pragma unroll
    for (int i = 0; i < MAX_CARRY; ++i) {
        if (blk < js.j[i].nblocks) {
            reduce_job_run(js.j[i], blk);
            return;
        }
        blk -= js.j[i].nblocks;
    }
}

// A job with splits == -1 is not a sum but the optimizer's per-step scalar work (lidbox_adam_prepare_job, nnops.hip: advance the
// device-side step counter, publish the bias-corrected learning rate): one thread of the stand-alone launch does it, so the
// launch that finishes the step's last wgrad also prepares Adam (round 3 had a one-thread launch for that).  Cm = the 16-byte
// state {int64 step, float lr_t, float lr_now}; n / ldc carry the bits of lr, beta_1 / beta_2.  GEMM launches never carry it
// (float64 pow in their leading blocks would set their register count).
struct AdamStateD {
    long long step;
    float lr_t;
    float lr_now;
};
__device__ __forceinline__ void adam_prepare_run(const ReduceJob& j) {
    AdamStateD* st = reinterpret_cast<AdamStateD*>(j.Cm);
    const float lr = __uint_as_float((unsigned)(j.n & 0xffffffffL)), b1 = __uint_as_float((unsigned)((unsigned long)j.n >> 32));
    const float b2 = __uint_as_float((unsigned)(j.ldc & 0xffffffffL));
    const long long t = ++st->step;
    const double c = sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t));
    const float base = __float_as_uint(st->lr_now) != 0u ? fabsf(st->lr_now) : lr;
    st->lr_t = (float)((double)base * c);
}

__global__ __launch_bounds__(256) void splitk_reduce4_kernel(ReduceJobs js) {
    unsigned blk = blockIdx.x;
#pragma unroll
    for (int i = 0; i < MAX_CARRY; ++i) {
        if (blk < js.j[i].nblocks) {
            if (js.j[i].splits == -1) {
                if (blk == 0 && threadIdx.x == 0) adam_prepare_run(js.j[i]);
            } else {
                reduce_job_run(js.j[i], blk);
            }
            return;
        }
        blk -= js.j[i].nblocks;
    }
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// whether the 16-byte form applies: every quad aligned and inside one row
inline bool reduce_job_vec_ok(const float* P, const float* Pc, int splits, long n, int N, const float* Cm, long ldc, const float* bias_grad) {
    // (the 16-byte kernel addresses the slices with 32-bit byte offsets)
    return N % 4 == 0 && ldc % 4 == 0 && n % 4 == 0 && aligned16(P) && aligned16(Cm) && (!bias_grad || (aligned16(Pc) && aligned16(bias_grad))) &&
           (double)splits * (double)n * 4.0 < 4.0e9;
}

// the job of a wgrad's slices; nblocks as the stand-alone launch uses it
inline ReduceJob make_reduce_job(const float* P, const float* Pc, int splits, long n, int N, float* Cm, long ldc, int accumulate, float* bias_grad) {
    ReduceJob j;
    j.P = P; j.Pc = bias_grad ? Pc : nullptr; j.Cm = Cm; j.bias_grad = bias_grad;
    j.n = n; j.ldc = ldc; j.splits = splits; j.N = N; j.accumulate = accumulate;
    long g = lbx_cdiv((n + N) / 4, 256);
    if (g > 2048) g = 2048;
    j.nblocks = (unsigned)(g < 1 ? 1 : g);
    return j;
}

// `njobs` pending jobs as the leading workgroups of a GEMM launch: `cap` blocks (a multiple of 8: block % 8 stays the XCD of
// the tiles behind them) dealt in proportion to the load batches of each job (quads x ceil(slices / 4)), at least 8 each.
inline ReduceJobs pack_carry(const ReduceJob* jobs, int njobs, long cap) {
    ReduceJobs js;
    double bytes[MAX_CARRY], all = 0.0;
    int m = 0;
    for (int i = 0; i < njobs && m < MAX_CARRY; ++i) {
        if (jobs[i].nblocks == 0) continue;
        js.j[m] = jobs[i];
        const int batches = (jobs[i].splits + 3) / 4;                     // load batches: what a job's workgroups spend their time on
        bytes[m] = (double)jobs[i].n * (batches > 0 ? batches : 1);       // (a zero fill -- no slices -- counts its stores)
        all += bytes[m];
        ++m;
    }
    for (int i = 0; i < m; ++i) {
        long quads = lbx_cdiv((js.j[i].n + (js.j[i].bias_grad ? js.j[i].N : 0)) / 4, 256);
        long g = (long)((double)cap * bytes[i] / all + 0.5);
        if (g > quads) g = quads;
        if (g < 8) g = 8;
        js.j[i].nblocks = (unsigned)((g + 7) & ~7L);
        js.total += js.j[i].nblocks;
    }
    return js;
}

// C (+)= sum of the wgrad slices, bias gradient likewise: the 16-byte kernel when every quad is aligned and inside one row
inline void launch_splitk_reduce(const float* P, const float* Pc, int splits, long n, int N, float* Cm, long ldc, int accumulate,
                                 float* bias_grad, hipStream_t st) {
    if (reduce_job_vec_ok(P, Pc, splits, n, N, Cm, ldc, bias_grad)) {
        ReduceJobs js;
        js.j[0] = make_reduce_job(P, Pc, splits, n, N, Cm, ldc, accumulate, bias_grad);
        js.total = js.j[0].nblocks;
        hipLaunchKernelGGL(splitk_reduce4_kernel, dim3(js.total), dim3(256), 0, st, js);
        return;
    }
    long g = lbx_cdiv(n + N, 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, P, Pc, splits, n, N, Cm, ldc, accumulate, bias_grad);
}

inline bool rows_aligned(const lidbox_rows_t& r) {
    return aligned16(r.base) && r.row_stride % 4 == 0 && (r.batch == 1 || r.batch_stride % 4 == 0);
}

inline RowsD to_dev(const lidbox_rows_t& r) { return RowsD{r.base, r.batch_stride, r.row_stride, r.batch, r.rows_per_batch}; }

int check_rows(const char* fn, const void* base, long bs, long rs, int batch, int rpb) {
    if (!base || batch < 0 || rpb < 0 || rs < 0 || bs < 0 || (long)batch * rpb > 0x7fffffffL) {
        lidbox_set_error("%s: invalid rows descriptor", fn);
        return LIDBOX_E_INVALID;
    }
    return LIDBOX_OK;
}

int validate_rows_call(const char* fn, const lidbox_rows_t& A, const float* Bm, long ldb, const lidbox_rows_out_t& C,
                       int K, int N, int epilogue, const float* aux, long ldb_min) {
    if (check_rows(fn, A.base, A.batch_stride, A.row_stride, A.batch, A.rows_per_batch)) return LIDBOX_E_INVALID;
    if (check_rows(fn, C.base, C.batch_stride, C.row_stride, C.batch, C.rows_per_batch)) return LIDBOX_E_INVALID;
    const bool needs_aux = epilogue == LIDBOX_EPI_BIAS || epilogue == LIDBOX_EPI_BIAS_RELU ||
                           epilogue == LIDBOX_EPI_RELU_MASK || epilogue == LIDBOX_EPI_ACCUM_RELU_MASK;
    const char* msg = nullptr;
    if (!Bm || K < 1 || N < 0 || ldb < ldb_min) msg = "B != NULL, K >= 1, N >= 0, ldb large enough";
    else if ((long)A.batch * A.rows_per_batch != (long)C.batch * C.rows_per_batch) msg = "A and C row counts differ";
    else if (epilogue < LIDBOX_EPI_NONE || epilogue > LIDBOX_EPI_RELU) msg = "epilogue";
    else if (needs_aux && !aux) msg = "aux required by this epilogue";
    if (msg) {
        lidbox_set_error("%s: invalid argument: %s", fn, msg);
        return LIDBOX_E_INVALID;
    }
    return LIDBOX_OK;
}

}  // namespace
