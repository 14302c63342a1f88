// gemm16_pp.h -- 256-row, eight-wave tile of the bf16-storage rows GEMM (lidbox_gemm_bf16s_nt: reference xvector.py:38-43,53-64
// under a bfloat16 compute policy, see gemm_bf16.hip), included by gemm_bf16.hip.
//
// Why (round 5): the four-wave LDS-DMA tiles of gemm16_dma.h top out at 64 x 128 ... 128 x 128 -- 24 ... 32 KB of operands
// through L2 -> LDS and 48 ... 64 KB of LDS reads per 1 ... 2 MFLOP of a K step, i.e. the loop is bound by the LDS / L2
// paths at roughly half the matrix rate (frame2's forward: 0.27 of the bf16 peak), and every wave meets its K step's barrier
// in the same state, so the matrix pipe idles while all of them wait for their first operand reads.  Here ONE workgroup of
// eight waves owns a 256 x 256 (or 256 x 128) tile of a CU -- half the operand bytes per flop -- and its two waves per SIMD
// run PING-PONG: the waves are two groups of four (one per SIMD each), offset by one phase; while a group issues the MFMAs
// of a sub-step from registers, the other group fetches ITS operand registers of that sub-step from LDS and issues its
// share of the next K step's LDS-DMA pieces.  A phase ends in one raw s_barrier; the DMAs stay in flight across barriers
// and are waited for (counted vmcnt, own pieces) only in the phase before their stage is read.
//
//   phase      0         1         2         3         4    ...
//   group 0    LOAD(0)   COMP(0)   LOAD(1)   COMP(1)   LOAD(2)
//   group 1    --        LOAD(0)   COMP(0)   LOAD(1)   COMP(1)
//
// A sub-step is a K step (64 k) cut into SUB parts (SUB = 1: 32 MFMAs per COMP at 256 x 256, 96 operand registers; SUB = 2:
// 16 MFMAs, 48).  Stage ring of two (2 x 64 KB at 256 x 256): the stage of step t + 1 is free once group 1 has finished
// LOAD of step t - 1's last sub-step, i.e. from the phase in which group 0 starts LOAD(t) -- that is where group 0 issues its
// pieces of step t + 1, group 1 one phase later; both wait for them in the last phase of step t.
//
// Layout: gemm16_dma.h's (a stage = A rows then B rows, 128 B = 64 k per row, 16-byte chunks XOR-swizzled by row & 7 on the
// source side of the DMA, conflict-free ds_read_b128 operand fetches).  Epilogue: gemm_shared.h's store_rows_tile per
// 64-row half of a wave's tile.  K tails, rows past M / N, K splits over grid.y and carried reduce jobs as in gemm16_dma.h.
#pragma once

#include "gemm16_dma.h"

namespace {

// this wave's NPC pieces (8 rows x 128 B each = one wave DMA instruction = 1 KB of LDS) of one operand: pieces first ..
// first + NPC - 1 of the operand's tile rows; piece p lands at p * 1024 of the operand's stage
template <int NPC>
struct PpPieces {
    const float* sb;                        // wave-uniform byte base (+ k of the next step to issue)
    unsigned vo[NPC];                       // this lane's byte offsets: row offset + swizzled chunk
    unsigned kqs;                           // 3 bits per piece: the chunk this lane fetches
    int p0;

    __device__ __forceinline__ void init(const RowsH& X, long row0, long nrows, int kbeg, int lane, int first) {
        static_assert(NPC <= 10, "kqs holds ten pieces");
        sb = sk_uniform(reinterpret_cast<const float*>(X.base + kbeg));
        kqs = 0;
        p0 = first;
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int trow = 8 * (first + i) + (lane >> 3);            // row inside the tile = LDS row
            const int chunk = (lane & 7) ^ d16_swz(trow);
            kqs |= (unsigned)chunk << (3 * i);
            long r = row0 + trow;
            if (r >= nrows) r = row0;
            vo[i] = (unsigned)((row_offset(X, (unsigned)r) + chunk * 8) * 2);
        }
    }
    // every piece of the current step into the operand stage at LDS byte address `stage_lds`, then on to the next step;
    // kvalid < 64: the K range's last, partial step (chunks at or past kvalid come from the zero source)
    __device__ __forceinline__ void issue_step(unsigned stage_lds, int kvalid) {
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const unsigned dst = stage_lds + (unsigned)(p0 + i) * 1024u;
            if (kvalid < D16_BK) {
                const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(sb) + vo[i]);
                sk_dma_f((int)((kqs >> (3 * i)) & 7u) * 8 < kvalid ? p : g_sk_zero, dst);
            } else {
                sk_dma_s(sb, vo[i], dst);
            }
        }
        sb += D16_ROW_BYTES / 4;
    }
};

// operand registers of block b (32 rows from `row`) for k slice ks (16 k): lane -> row lane & 31, k 16 ks + 8 (lane >> 5) ..+7
__device__ __forceinline__ bf16x8 pp_read(const char* op, int row, int lane, int b, int ks) {
    const int pos = (2 * ks + (lane >> 5)) ^ d16_swz(lane & 31);       // row and b * 32 are multiples of 32: the key is the lane's
    return *reinterpret_cast<const bf16x8*>(op + (row + b * 32 + (lane & 31)) * D16_ROW_BYTES + pos * 16);
}

__device__ __forceinline__ void pp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void pp_loop_barrier();
__device__ __forceinline__ void pp_wait_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

#ifndef LBX_PP_ABLATE
#define LBX_PP_ABLATE 0
#endif
__device__ __forceinline__ void pp_loop_barrier() {
    if (LBX_PP_ABLATE & 8) __builtin_amdgcn_sched_barrier(0);
    else pp_barrier();
}

// Epilogue of one wave's MI x NJ accumulator blocks through a wave-private LDS strip: a 32-row x (32 NJ)-column strip is
// written in the accumulator layout (lane = column), read back row-wise -- a lane then holds 8 consecutive columns of one row
// -- and leaves as 16-byte stores (two for the fp32 value, one for the bf16 shadow), the mask / old values arrive as 16-byte
// loads.  Why: with one workgroup per CU nothing overlaps the epilogue, and gemm_shared.h's store_rows_tile issues one 2- or
// 4-byte access per lane and element (128 + 128 store instructions per wave of a 128 x 64 tile: store-issue bound, measured
// ~30 us of a launch); here it is 16 + 32.  Same arithmetic per element, in the same order, as store_rows_tile:
// x = acc + bias; mask; + old; ReLU.  Vector accesses need 16-byte aligned fp32 / 8-byte aligned bf16 bases and row / batch
// strides that are multiples of 4 elements (true of every buffer of the model; anything else takes the element-wise path
// below, wave-uniformly).
constexpr int PP_EPI_LDW = 68;                                   // floats per strip row: 64 + 4 (conflict-free both ways)
constexpr int PP_EPI_BYTES = 32 * PP_EPI_LDW * 4;                // 8 704 bytes per wave

template <int MI, int NJ>
__device__ __forceinline__ void pp_store_tile(const f32x16 (&acc)[MI][NJ], float* __restrict__ wlds, long mrow0, int ncol0, int lane, long m_beg,
                                              long M, int N, int epi, const float* __restrict__ aux, const RowsOutD& Cd, float* __restrict__ P,
                                              int split, unsigned short* __restrict__ shadow, const unsigned short* __restrict__ mask16,
                                              const unsigned (&mbits_v)[MI], bool mbits, int partial_override = -1) {
    // mbits: mbits_v holds the ReLU decisions of this lane's chunks, bit 8 c + j = element j of chunk c = 4 strip + 2 half + i,
    // collected during the K loop (pp_mask_prefetch below) -- the strips that take the vector path then load no mask
    static_assert(NJ == 2, "pp_store_tile: 64-column strips");
    const int h = lane >> 5, l = lane & 31;
    // partial_override: 0 / 1 from kernels whose splits are not grid.y (the wgrad tile: raw sums to P[split][row][n])
    const bool partial = partial_override < 0 ? gridDim.y > 1 : partial_override != 0;
    const bool has_bias = !partial && (epi == LIDBOX_EPI_BIAS || epi == LIDBOX_EPI_BIAS_RELU);
    const bool do_relu = !partial && (epi == LIDBOX_EPI_BIAS_RELU || epi == LIDBOX_EPI_ACCUM_RELU || epi == LIDBOX_EPI_RELU);
    const bool has_mask = !partial && (epi == LIDBOX_EPI_RELU_MASK || epi == LIDBOX_EPI_ACCUM_RELU_MASK);
    const bool accum = !partial && (epi == LIDBOX_EPI_ACCUM || epi == LIDBOX_EPI_ACCUM_RELU_MASK || epi == LIDBOX_EPI_ACCUM_RELU);
    float bias[NJ];
#pragma unroll
    for (int bj = 0; bj < NJ; ++bj) {
        const int c = ncol0 + bj * 32 + l;
        bias[bj] = (has_bias && c < N) ? aux[c] : 0.f;
    }
    float* const out_base = partial ? P + ((long)split * (M - m_beg) - m_beg) * N : Cd.base;    // P[split][row - m_beg][n]
    unsigned short* const sh = partial ? nullptr : shadow;
    const bool batched = !partial && Cd.batch != 1;
    const long rs = partial ? (long)N : Cd.rs;
    // wave-uniform: may the arrays laid out like C be accessed in vectors?  Row / batch strides that are multiples of 4 elements
    // put every 8-column chunk on a 16-byte boundary of the fp32 arrays and an 8-byte boundary of the bf16 ones (frame5's 1 500
    // channels: 6 000- / 3 000-byte rows); multiples of 8 elements make the bf16 chunks 16-byte accesses too.
    const bool str4 = rs % 4 == 0 && (!batched || Cd.bs % 4 == 0);
    const bool v32 = str4 && (((uintptr_t)out_base) & 15) == 0 && (!(has_mask && !mask16) || (((uintptr_t)aux) & 15) == 0);
    const bool v16 = str4 && (((uintptr_t)sh) & 7) == 0 && (((uintptr_t)mask16) & 7) == 0;
    const bool vec = v32 && v16;
    const bool w16 = rs % 8 == 0 && (!batched || Cd.bs % 8 == 0) && (((uintptr_t)sh) & 15) == 0 && (((uintptr_t)mask16) & 15) == 0;   // 16-byte bf16 accesses
    auto ld16 = [&](const unsigned short* p) -> u32x4_t {           // 8 bf16 at an 8-byte (w16: 16-byte) aligned address
        if (w16) return *reinterpret_cast<const u32x4_t*>(p);
        const uint2 a = *reinterpret_cast<const uint2*>(p), b = *reinterpret_cast<const uint2*>(p + 4);
        return u32x4_t{a.x, a.y, b.x, b.y};
    };
    auto st16 = [&](unsigned short* p, const float (&x)[8]) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)x[j];
        if (w16) {
            *reinterpret_cast<bf16x8*>(p) = o;
        } else {
            const u32x4_t u = __builtin_bit_cast(u32x4_t, o);
            *reinterpret_cast<uint2*>(p) = make_uint2(u[0], u[1]);
            *reinterpret_cast<uint2*>(p + 4) = make_uint2(u[2], u[3]);
        }
    };
#pragma unroll
    for (int bi = 0; bi < MI; ++bi) {
#pragma unroll
        for (int bj = 0; bj < NJ; ++bj)
#pragma unroll
            for (int r = 0; r < 16; ++r) wlds[((r & 3) + 8 * (r >> 2) + 4 * h) * PP_EPI_LDW + bj * 32 + l] = acc[bi][bj][r] + bias[bj];
        wave_lds_sync();
        // INNER (wave-uniform): the strip lies wholly inside the matrix and vector accesses apply -- no per-lane predicates, and the
        // four chunks of a lane go in batches (offsets, then every mask / old-value load, then the arithmetic, then the stores):
        // chunk by chunk, a mask epilogue is sixteen dependent round trips to memory per wave with nothing else on the CU to hide them
        if (vec && mrow0 + bi * 32 + 32 <= M && ncol0 + 64 <= N) {
#pragma unroll
          for (int half = 0; half < 2; ++half) {               // two chunks per batch: the temporaries of four do not fit beside 128 accumulators
            long off[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const long R = mrow0 + bi * 32 + (lane >> 3) + 8 * (2 * half + i);
                off[i] = (partial ? R * (long)N : row_offset(Cd, (unsigned)R)) + ncol0 + (lane & 7) * 8;
            }
            u32x4_t mk[2];
            f32x4 ma[2], mb[2], oa[2], ob[2];
            if (has_mask && !mbits) {
                if (mask16) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) mk[i] = ld16(mask16 + off[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        ma[i] = *reinterpret_cast<const f32x4*>(aux + off[i]);
                        mb[i] = *reinterpret_cast<const f32x4*>(aux + off[i] + 4);
                    }
                }
            }
            if (accum) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    oa[i] = *reinterpret_cast<const f32x4*>(out_base + off[i]);
                    ob[i] = *reinterpret_cast<const f32x4*>(out_base + off[i] + 4);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* src = wlds + ((lane >> 3) + 8 * (2 * half + i)) * PP_EPI_LDW + (lane & 7) * 8;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
                float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (has_mask && mbits) {
                    const int c = bi * 4 + 2 * half + i;
                    const unsigned byte = mbits_v[c >> 2] >> (8 * (c & 3));
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = ((byte >> j) & 1u) ? x[j] : 0.f;
                } else if (has_mask) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float mvj = mask16 ? __builtin_bit_cast(float, (j & 1) ? (mk[i][j >> 1] & 0xffff0000u) : (mk[i][j >> 1] << 16))
                                                 : (j < 4 ? ma[i][j & 3] : mb[i][j & 3]);
                        x[j] = mvj > 0.f ? x[j] : 0.f;
                    }
                }
                if (accum) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] += j < 4 ? oa[i][j & 3] : ob[i][j & 3];
                }
                if (do_relu) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = fmaxf(x[j], 0.f);
                }
                if (out_base) {
                    *reinterpret_cast<f32x4*>(out_base + off[i]) = f32x4{x[0], x[1], x[2], x[3]};
                    *reinterpret_cast<f32x4*>(out_base + off[i] + 4) = f32x4{x[4], x[5], x[6], x[7]};
                }
                if (sh) st16(sh + off[i], x);
            }
          }
            wave_lds_sync();
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i, row = c >> 3, cc = c & 7;
            const long R = mrow0 + bi * 32 + row;
            const int col = ncol0 + cc * 8;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(wlds + row * PP_EPI_LDW + cc * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(wlds + row * PP_EPI_LDW + cc * 8 + 4);
            float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            if (R >= M || col >= N) continue;
            const long off = (partial ? R * (long)N : row_offset(Cd, (unsigned)R)) + col;
            if (vec && col + 8 <= N) {
                if (has_mask) {
                    if (mask16) {
                        const u32x4_t mk = ld16(mask16 + off);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {          // sign and zero-ness are all that is looked at: bits << 16 is the value
                            const float mvj = __builtin_bit_cast(float, (j & 1) ? (mk[j >> 1] & 0xffff0000u) : (mk[j >> 1] << 16));
                            x[j] = mvj > 0.f ? x[j] : 0.f;
                        }
                    } else {
                        const f32x4 m0v = *reinterpret_cast<const f32x4*>(aux + off), m1v = *reinterpret_cast<const f32x4*>(aux + off + 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            x[j] = m0v[j] > 0.f ? x[j] : 0.f;
                            x[4 + j] = m1v[j] > 0.f ? x[4 + j] : 0.f;
                        }
                    }
                }
                if (accum) {
                    const f32x4 o0 = *reinterpret_cast<const f32x4*>(out_base + off), o1 = *reinterpret_cast<const f32x4*>(out_base + off + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        x[j] += o0[j];
                        x[4 + j] += o1[j];
                    }
                }
                if (do_relu) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = fmaxf(x[j], 0.f);
                }
                if (out_base) {
                    *reinterpret_cast<f32x4*>(out_base + off) = f32x4{x[0], x[1], x[2], x[3]};
                    *reinterpret_cast<f32x4*>(out_base + off + 4) = f32x4{x[4], x[5], x[6], x[7]};
                }
                if (sh) st16(sh + off, x);
            } else {
                for (int j = 0; j < 8 && col + j < N; ++j) {
                    float v = x[j];
                    if (has_mask) {
                        const float mvj = mask16 ? __builtin_bit_cast(float, (unsigned)mask16[off + j] << 16) : aux[off + j];
                        v = mvj > 0.f ? v : 0.f;
                    }
                    if (accum) v += out_base[off + j];
                    if (do_relu) v = fmaxf(v, 0.f);
                    if (out_base) out_base[off + j] = v;
                    if (sh) sh[off + j] = __builtin_bit_cast(unsigned short, (__bf16)v);
                }
            }
        }
        wave_lds_sync();
    }
}

// the same epilogue for the four-wave LDS-DMA tiles (gemm16_dma.h declares it ahead of its kernel)
template <int MI, int NJ>
__device__ __forceinline__ void d16_store_tile_wide(const f32x16 (&acc)[MI][NJ], float* __restrict__ wlds, long mrow0, int ncol0, int lane,
                                                    long m_beg, long M, int N, int epi, const float* __restrict__ aux, const RowsOutD& Cd,
                                                    float* __restrict__ P, int split, unsigned short* __restrict__ shadow,
                                                    const unsigned short* __restrict__ mask16) {
    const unsigned none[MI] = {};
    pp_store_tile<MI, NJ>(acc, wlds, mrow0, ncol0, lane, m_beg, M, N, epi, aux, Cd, P, split, shadow, mask16, none, false);
}

// The ReLU mask of a dgrad epilogue, fetched DURING the K loop.  With one workgroup per CU every tile of a round reaches its
// epilogue at the same time: 256 x (128 KB of mask + 128 KB of output) hit HBM in one burst (~6 us of a ~27 us K = 512 round for the
// mask alone) while the memory system idles under the MFMA phases before it.  A lane's epilogue chunks (8 consecutive columns of a
// row: pp_store_tile's vector path) are known up front, and only the SIGN of the mask matters: up to four 16-byte loads per batch
// go out as inline-asm global loads right BEFORE a batch of LDS-DMA pieces whose covering counted `vmcnt` retires them as well
// (VMEM operations retire in order), and are folded into one bit per element -- 4 x 32 bits per lane for a 128 x 64 wave tile.
// Inline asm because hipcc waits vmcnt(0) for an ordinary load's first use, which would drain the DMA ring every step.
template <int NCH>
struct PpMaskPrefetch {
    static constexpr int BATCH = 4;
    unsigned bits[NCH / 4];
    u32x4_t pend[BATCH];
    int issued, folded;                     // chunks whose load has been issued / folded into bits (wave-uniform)
    bool on;

    __device__ __forceinline__ void init(bool enable) {
        on = enable;
        issued = folded = 0;
#pragma unroll
        for (int i = 0; i < NCH / 4; ++i) bits[i] = 0u;
    }
    // the next batch of loads: chunk c = 4 strip + k covers row mrow0 + 32 strip + (lane >> 3) + 8 k, columns ncol0 + 8 (lane & 7) ..
    __device__ __forceinline__ void issue(const unsigned short* mask16, const RowsOutD& Cd, long mrow0, int ncol0, int lane) {
        if (!on || issued >= NCH) return;
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int c = issued + u;
            if (c < NCH) {
                const long R = mrow0 + (c >> 2) * 32 + (lane >> 3) + 8 * (c & 3);
                const unsigned short* ptr = mask16 + row_offset(Cd, (unsigned)R) + ncol0 + (lane & 7) * 8;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pend[u]) : "v"(ptr) : "memory");
            }
        }
        issued = min(NCH, issued + BATCH);
    }
    // after the counted wait that covers the last issue(): fold the landed chunks into bits
    __device__ __forceinline__ void fold() {
        if (!on || folded >= issued) return;
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int c = folded + u;
            if (c < issued) {
                asm volatile("" : "+v"(pend[u]));                    // the value is defined only behind the wait (asm statements keep their order)
                unsigned byte = 0u;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned w = pend[u][j >> 1];
                    const float mv = __builtin_bit_cast(float, (j & 1) ? (w & 0xffff0000u) : (w << 16));
                    byte |= (mv > 0.f ? 1u : 0u) << j;
                }
                // the chunk index is wave-uniform but not a compile-time constant, and an indexed register array would go to scratch:
                // the words form one shift register -- a byte enters at the top, everything moves down 8 bits; after NCH chunks
                // chunk c sits at bits 8 c .. 8 c + 7
#pragma unroll
                for (int wd = 0; wd + 1 < NCH / 4; ++wd) bits[wd] = __builtin_amdgcn_alignbyte(bits[wd + 1], bits[wd], 1);
                bits[NCH / 4 - 1] = (bits[NCH / 4 - 1] >> 8) | (byte << 24);
            }
        }
        folded = issued;
    }
    __device__ __forceinline__ bool complete() const { return on && folded >= NCH; }
};

// C[M,N] = epi(A[M,K] . B[N,K]^T), bf16 operands, fp32 accumulate; 512 threads, tile 256 x BN (BN = 256: waves 2 x 4, 128 x 64
// each; BN = 128: waves 4 x 2, 64 x 64 each); grid.x = [carried reduce blocks] + tiles (XCD-chunk remapped), grid.y = K splits.
// LDS: an A ring of THREE stages (3 x 32 KB) and a B ring of two (2 x BN x 128 B): 160 KB at BN = 256, the whole CU.
// LBX_PP_ABLATE (measurement builds only, results are wrong): 1 = no LDS-DMA issue after the prologue, 2 = operand fetches of the
// first sub-step only, 4 = no MFMAs, 8 = no barriers inside the loop, 16 = no epilogue, 32 = no DMA at all (with 1), 64 = no mask prefetch
#ifndef LBX_PP_ABLATE
#define LBX_PP_ABLATE 0
#endif

template <int BN>
constexpr int pp_lds_bytes() {
    return 3 * 256 * D16_ROW_BYTES + 2 * BN * D16_ROW_BYTES;
}

// one tile (index `tile` of `ntiles`, K range of split blockIdx.y) of one problem: the body of both kernels below
template <int BN, int SUB>
__device__ __forceinline__ void pp_tile(const RowsH& A, const RowsH& Bw, const RowsOutD& Cd, unsigned short* __restrict__ C16,
                                        float* __restrict__ P, long m_beg, long M, int K, int N, int epi,
                                        const float* __restrict__ aux, int tiles_n, unsigned ntiles, int k_per_split,
                                        const unsigned short* __restrict__ mask16, unsigned tile) {
    constexpr int BM = 256;
    constexpr int WN = BN / 64, WM = 8 / WN;                  // waves along N / M
    constexpr int MI = BM / WM / 32, NJ = 2;                  // 32 x 32 blocks per wave
    constexpr int A_ST = BM * D16_ROW_BYTES, B_ST = BN * D16_ROW_BYTES, B_RING = 3 * A_ST;
    constexpr int NA = BM / 64, NB = BN / 64;                 // DMA pieces per wave and step: 4 of A, 4 | 2 of B
    static_assert(SUB == 1 || SUB == 2, "sub-steps");
    constexpr int KH = 4 / SUB;                               // k slices (16 k) per sub-step
    extern __shared__ __attribute__((aligned(16))) char smem16p[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wv >> 2;
    const int wm = wv / WN, wn = wv % WN;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem16p);
    const unsigned chunk = xcd_chunk_id(tile, ntiles);
    const int tn = chunk % tiles_n;
    const long m0 = m_beg + (long)(chunk / tiles_n) * BM;
    const int n0 = tn * BN;
    const int split = blockIdx.y;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    const int n = (kend - kbeg + D16_BK - 1) / D16_BK;
    const int ktail = kend - kbeg - (n - 1) * D16_BK;          // valid k of the last step: 8 .. 64
    const int rowA = wm * (BM / WM), rowB = wn * 64;           // this wave's first row inside an A / B stage

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // the dgrad epilogue's ReLU mask rides in the K loop when the wave's whole tile takes the epilogue's vector path
    const bool mask_epi = gridDim.y == 1 && mask16 && (epi == LIDBOX_EPI_RELU_MASK || epi == LIDBOX_EPI_ACCUM_RELU_MASK);
    PpMaskPrefetch<MI * 4> mp;
    // Built, parity-green, and SLOWER (profiles/r05_bf16_pp_mask_prefetch_ab.txt: dgrads +4 ... +6 us at bs 512, the configs[4] step 1.091 vs
    // 1.070 ms): 16 row-offset divisions, 16 loads and ~400 fold instructions per lane and 28 more registers inside the K loop cost
    // more than the 6 us of mask traffic they take out of the epilogue burst.  Compiled out (-DLBX_PP_MASK_PREFETCH=1 brings it back).
#ifndef LBX_PP_MASK_PREFETCH
#define LBX_PP_MASK_PREFETCH 0
#endif
    mp.init(LBX_PP_MASK_PREFETCH && !(LBX_PP_ABLATE & 64) && mask_epi && n >= 4 && m0 + rowA + 32 * MI <= M && n0 + rowB + 64 <= N && Cd.rs % 8 == 0 &&
            (Cd.batch == 1 || Cd.bs % 8 == 0) && (((uintptr_t)mask16) & 15) == 0);
    PpPieces<NA> pa;
    PpPieces<NB> pb;
    pa.init(A, m0, M, kbeg, lane, wv * NA);
    pb.init(Bw, n0, N, kbeg, lane, wv * NB);
    // A of step `step` into A stage step % 3, B of step `step` into B stage step & 1 (the lists advance with every issue: A runs
    // two steps ahead of the step being computed, B one)
    auto issue_a = [&](int step, int sa) {
        if ((LBX_PP_ABLATE & 32) || ((LBX_PP_ABLATE & 1) && step > 1)) return;
        pa.issue_step(lds0 + (unsigned)(sa * A_ST), step == n - 1 ? ktail : D16_BK);
    };
    auto issue_b = [&](int step, int sb) {
        if ((LBX_PP_ABLATE & 32) || ((LBX_PP_ABLATE & 1) && step > 0)) return;
        pb.issue_step(lds0 + (unsigned)(B_RING + sb * B_ST), step == n - 1 ? ktail : D16_BK);
    };

    auto body = [&](auto grp_tag) {
        constexpr int G = decltype(grp_tag)::value;
        bf16x8 a[MI][KH], b[NJ][KH];
        bool first_load = true;
        auto load = [&](int sa, int sb, int j) {
            if (LBX_PP_ABLATE & 2) {
                if (!first_load) {
#pragma unroll
                    for (int kk = 0; kk < KH; ++kk) {
#pragma unroll
                        for (int bj = 0; bj < NJ; ++bj) asm volatile("" : "+v"(b[bj][kk]));
#pragma unroll
                        for (int bi = 0; bi < MI; ++bi) asm volatile("" : "+v"(a[bi][kk]));
                    }
                    return;
                }
                first_load = false;
            }
            const char* sta = smem16p + sa * A_ST;
            const char* stb = smem16p + B_RING + sb * B_ST;
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) {
#pragma unroll
                for (int bj = 0; bj < NJ; ++bj) b[bj][kk] = pp_read(stb, rowB, lane, bj, j * KH + kk);
#pragma unroll
                for (int bi = 0; bi < MI; ++bi) a[bi][kk] = pp_read(sta, rowA, lane, bi, j * KH + kk);
            }
        };
        auto comp = [&]() {
            if (LBX_PP_ABLATE & 4) {
#pragma unroll
                for (int kk = 0; kk < KH; ++kk) {
#pragma unroll
                    for (int bj = 0; bj < NJ; ++bj) asm volatile("" ::"v"(b[bj][kk]));
#pragma unroll
                    for (int bi = 0; bi < MI; ++bi) asm volatile("" ::"v"(a[bi][kk]));
                }
                return;
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < KH; ++kk)
#pragma unroll
                for (int bi = 0; bi < MI; ++bi)
#pragma unroll
                    for (int bj = 0; bj < NJ; ++bj)
                        acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[bi][kk], b[bj][kk], acc[bi][bj], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        // prologue: A(0), B(0), then A(1), which stays in flight across the first hand-off
        issue_a(0, 0);
        issue_b(0, 0);
        mp.issue(mask16, Cd, m0 + rowA, n0 + rowB, lane);      // older than A(1): the wait below retires it too
        if (n > 1) {
            issue_a(1, 1);
            sk_wait_vm<NA>();
        } else {
            sk_wait_vm<0>();
        }
        mp.fold();
        pp_barrier();
        if (G == 1) pp_barrier();                              // phase 0 belongs to group 0 alone
        int sa = 0, sb = 0;                                    // A / B stages of the step being computed
        for (int t = 0; t < n; ++t) {
            const bool more_b = t + 1 < n, more_a = t + 2 < n;
            int sa2 = sa + 2;
            if (sa2 >= 3) sa2 -= 3;
#pragma unroll
            for (int j = 0; j < SUB; ++j) {
                load(sa, sb, j);                               // operand fetches first: they complete under the DMA issues
                // B(t + 1) in the step's first LOAD phase (its stage was read last in the phase before), A(t + 2) in the last one
                // (no deadline until two steps on): every LOAD phase of either group carries its share of the DMA traffic
                if (j == 0 && more_b) issue_b(t + 1, sb ^ 1);
                if (j == SUB - 1 && more_b) mp.issue(mask16, Cd, m0 + rowA, n0 + rowB, lane);   // ahead of A(t + 2): retired by this step's wait
                if (j == SUB - 1 && more_a) issue_a(t + 2, sa2);
                pp_wait_lds();
                if (G == 1 && j == SUB - 1 && more_b) {        // step t + 1's operands: everything but the A pieces just issued
                    if (more_a) sk_wait_vm<NA>();
                    else sk_wait_vm<0>();
                    mp.fold();
                }
                pp_loop_barrier();
                comp();
                if (G == 0) {
                    if (j == SUB - 1 && more_b) {
                        if (more_a) sk_wait_vm<NA>();
                        else sk_wait_vm<0>();
                        mp.fold();
                    }
                    pp_loop_barrier();
                } else if (!(j == SUB - 1 && !more_b)) {
                    pp_loop_barrier();
                }
            }
            sa = sa + 1 == 3 ? 0 : sa + 1;
            sb ^= 1;
        }
    };
    if (grp == 0) body(IntTag<0>{});
    else body(IntTag<1>{});

    // epilogue through this wave's LDS strip: no wave reads a stage any more once group 0 has passed the last barrier (group
    // 1's last sub-step runs from registers), and no DMA is in flight
    if (LBX_PP_ABLATE & 16) {                                  // no epilogue: one store per wave keeps the accumulators alive
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
        if (sacc == 12345.f) P[tid] = sacc;
        return;
    }
    pp_store_tile<MI, NJ>(acc, reinterpret_cast<float*>(smem16p + wv * PP_EPI_BYTES), m0 + rowA, n0 + rowB, lane, m_beg, M, N, epi, aux, Cd, P,
                          split, C16, mask16, mp.bits, mp.complete());
}

template <int BN, int SUB>
__global__ __launch_bounds__(512, 2) void gemm16s_rows_pp_kernel(RowsH A, RowsH Bw, RowsOutD Cd, unsigned short* __restrict__ C16,
                                                                  float* __restrict__ P, long m_beg, long M, int K, int N, int epi,
                                                                  const float* __restrict__ aux, int tiles_n, unsigned ntiles,
                                                                  int k_per_split, const unsigned short* __restrict__ mask16, ReduceJobs rj) {
    if (blockIdx.x < rj.total) {
        if (blockIdx.y == 0 && threadIdx.x < 256) reduce_jobs_run(rj, blockIdx.x);
        return;
    }
    pp_tile<BN, SUB>(A, Bw, Cd, C16, P, m_beg, M, K, N, epi, aux, tiles_n, ntiles, k_per_split, mask16, blockIdx.x - rj.total);
}

// Two independent problems in ONE grid (round 6: the two row residues of frame2's output-stationary dgrad, 492 tiles 1 024 deep and
// 396 tiles 512 deep -- 1.92 + 1.55 rounds of 256 CUs as two launches, 3.47 as one): the first problem's tiles take the leading
// workgroups (the caller passes the deeper contraction first: long tiles first packs best), no K splits, no slice workspace.
struct PpProblem {
    RowsH A, Bw;
    RowsOutD Cd;
    unsigned short* C16;
    float* P;                            // NULL (no K splits); a kernel argument rather than a literal: hipcc 7.2's SimplifyCFG crashes on the latter
    const unsigned short* mask16;
    const float* aux;
    long M;
    int K, N, epi, tiles_n;
    unsigned ntiles;
};
template <int BN, int SUB>
__global__ __launch_bounds__(512, 2) void gemm16s_rows_pp2_kernel(PpProblem p0, PpProblem p1, ReduceJobs rj) {
    if (blockIdx.x < rj.total) {
        if (threadIdx.x < 256) reduce_jobs_run(rj, blockIdx.x);
        return;
    }
    const unsigned t = blockIdx.x - rj.total;
    if (t < p0.ntiles)
        pp_tile<BN, SUB>(p0.A, p0.Bw, p0.Cd, p0.C16, p0.P, 0L, p0.M, p0.K, p0.N, p0.epi, p0.aux, p0.tiles_n, p0.ntiles, p0.K + D16_BK, p0.mask16, t);
    else
        pp_tile<BN, SUB>(p1.A, p1.Bw, p1.Cd, p1.C16, p1.P, 0L, p1.M, p1.K, p1.N, p1.epi, p1.aux, p1.tiles_n, p1.ntiles, p1.K + D16_BK, p1.mask16,
                         t - p0.ntiles);
}

}  // namespace
