// gemm_dma.h -- the classic decomposition (one tile per workgroup, grid = tiles x K splits, many small workgroups per CU at
// different phases: gemm.hip) on the operand path of gemm_sk.h: LDS-DMA ring in the saddr form, K-inner operands XOR-swizzled
// on the source side and read with ds_read_b128, K-outer operands read with ds_read_b32.  Tiles 64 x 64, 128 x 64, 64 x 128,
// 128 x 128 (four waves 2 x 2, MI x NJ blocks of 32 x 32 per wave).  Same contracts / epilogues as gemm_rows_kernel
// (reference xvector.py:38-43,53-64, cnn.py:32-41); 16-byte aligned operands only.
//   rows_dma_body / gemm_rows_dma_kernel   forward (nn) and dgrad (nt) with the fused epilogues; K splits through P, or the last
//                                          partial round of tiles streamed along K inside the launch (DmaStream)
//   tn_dma_body / gemm_tn_dma_kernel       wgrad over M slices, P[split][K1][N] + bias-gradient partials into gemm_shared.h's reduce
//   gemm_nt_tn_pair_kernel                 a layer's dgrad + wgrad (64 x 64 tiles) as one launch (lidbox_gemm_nt_tn)
// The bodies are device functions so that several GEMMs can share a launch; measurements behind every choice: DESIGN.md 4.2c.
#pragma once

#include "gemm_sk.h"

namespace {

#ifndef LBX_DMA_STAGES
#define LBX_DMA_STAGES 3
#endif
#ifndef LBX_DMA_LATE_READ
#define LBX_DMA_LATE_READ 0                 // rows kernel only: see the variant in its K loop
#endif
constexpr int DMA_STAGES = LBX_DMA_STAGES;

// K-inner operand, ROWS = 64 or 128 rows x 16 k per step; a wave issues ROWS / 64 pieces of 16 rows (four lanes per row)
template <int ROWS>
struct DmaInner {
    static constexpr int PW = ROWS / 64;          // pieces per wave per step
    const float* sb;
    unsigned vo[PW];
    int rd;
    __device__ __forceinline__ void init(const float* base, const long (&roff)[PW], int k0, int lane, int wsub) {
        sb = sk_uniform(base + k0);
        const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
        for (int i = 0; i < PW; ++i) vo[i] = (unsigned)((roff[i] + chunk * 4) * 4);
        rd = (wsub * (ROWS / 2) + (lane & 31)) * 16;
    }
    __device__ __forceinline__ void issue(int i, unsigned dst) const { sk_dma_s(sb, vo[i], dst); }
    __device__ __forceinline__ void issue_tail(int i, unsigned dst, int kvalid, int lane) const {
        const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
        const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(sb) + vo[i]);
        sk_dma_f(chunk * 4 < kvalid ? p : g_sk_zero, dst);
    }
    __device__ __forceinline__ void advance() { sb += SK_BK; }
    template <int NB>
    __device__ __forceinline__ void read(const float* st, int lane, int s2, float (&v)[NB][4]) const {
        const int slot = (2 * s2 + (lane >> 5)) ^ ((lane >> 2) & 3);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const f32x4_t x = *reinterpret_cast<const f32x4_t*>(st + rd + b * 32 * 16 + slot * 4);
            v[b][0] = x[0]; v[b][1] = x[1]; v[b][2] = x[2]; v[b][3] = x[3];
        }
    }
};

// K-outer operand of a plain matrix X[k][ld], 16 k x COLS columns per step; LDS image [k][COLS]; a piece = 1 KB =
// 1024 / (4 COLS) k rows
template <int COLS>
struct DmaOuter {
    static constexpr int PW = COLS / 64;
    static constexpr int LPR = COLS / 4;          // lanes per k row
    static constexpr int RPP = 64 / LPR;          // k rows per piece: 4 (64 columns) or 2 (128)
    const float* sb;
    unsigned vo[PW];
    long step;
    int rd;
    __device__ __forceinline__ void init(const float* base, long ld, int col0, int ncols, int k0, int lane, int wv, int wsub) {
        sb = sk_uniform(base + (long)k0 * ld + col0);
        int c = (lane % LPR) * 4;
        if (col0 + c >= ncols) c = 0;
#pragma unroll
        for (int i = 0; i < PW; ++i) vo[i] = (unsigned)(((long)(RPP * (wv * PW + i) + lane / LPR) * ld + c) * 4);
        step = (long)SK_BK * ld;
        rd = (4 * (lane >> 5)) * COLS + wsub * (COLS / 2) + (lane & 31);
    }
    __device__ __forceinline__ void issue(int i, unsigned dst) const { sk_dma_s(sb, vo[i], dst); }
    __device__ __forceinline__ void issue_tail(int i, unsigned dst, int kvalid, int lane, int wv) const {
        const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(sb) + vo[i]);
        sk_dma_f(RPP * (wv * PW + i) + lane / LPR < kvalid ? p : g_sk_zero, dst);
    }
    __device__ __forceinline__ void advance() { sb += step; }
    template <int NB>
    __device__ __forceinline__ void read(const float* st, int lane, int s2, float (&v)[NB][4]) const {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[b][j] = st[rd + (8 * s2 + j) * COLS + b * 32];
    }
};

template <int MI, int NJ, int J0, int J1>
__device__ __forceinline__ void dma_mma(const float (&a)[MI][4], const float (&b)[NJ][4], f32x16 (&acc)[MI][NJ]) {
#pragma unroll
    for (int j = J0; j < J1; ++j)
#pragma unroll
        for (int bi = 0; bi < MI; ++bi)
#pragma unroll
            for (int bj = 0; bj < NJ; ++bj)
                acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[bi][j], b[bj][j], acc[bi][bj], 0, 0, 0);
}

// Streamed remainder (stream-K for the last partial round only).  One tile per workgroup costs ceil(tiles / 256) tile
// times: the CU that got one tile more than the others finishes alone (measured staircase, profiles/r03_dma_staircase.txt:
// 64 x 64 x 1536 tiles, 4 per CU 107 us, 4.125 per CU 124 us, 5 per CU 124 us).  The R = tiles % 256 tiles of the last
// partial round are therefore cut along K into g pieces each; the R * g pieces take the FIRST blockIdx values (they start
// with the first wave of whole tiles, and their hand-off is over long before the launch ends), each piece leaves its
// partial sums in a slab of the workspace (accumulator order, write-through: gemm_sk.h) and takes a ticket on the tile's
// arrival counter; the last arriver sums the g slabs in k order -- a fixed order whoever it is -- and runs the epilogue.
struct DmaStream {
    unsigned npad = 0;          // blockIdx.x < npad: piece workgroups (pieces rounded up to a multiple of 8: keeps block % 8 = XCD
                                // for the whole tiles behind them); the padding blocks exit
    unsigned pieces = 0;        // R * g
    int g = 1;
    unsigned first_tile = 0;    // chunk index of the first streamed tile (= number of whole tiles)
    unsigned epoch = 0;
    unsigned* counters = nullptr;
    float* slabs = nullptr;     // [pieces][BM * BN]
};

template <int MI, int NJ>
__device__ __forceinline__ void dma_slab_store(float* slab, const f32x16 (&acc)[MI][NJ], int wv, int lane) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sk_uniform(slab)), 0, MI * NJ * 4096 * 4, 0x00020000);
#pragma unroll
    for (int bi = 0; bi < MI; ++bi)
#pragma unroll
        for (int bj = 0; bj < NJ; ++bj)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4_t x = {acc[bi][bj][4 * r4], acc[bi][bj][4 * r4 + 1], acc[bi][bj][4 * r4 + 2], acc[bi][bj][4 * r4 + 3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, x), r,
                                                       ((((wv * (MI * NJ) + bi * NJ + bj) * 4 + r4) * 64 + lane) * 4) * 4, 0, /*sc1*/ 16);
            }
}
template <int MI, int NJ>
__device__ __forceinline__ void dma_slab_add(const float* slab, f32x16 (&acc)[MI][NJ], int wv, int lane) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sk_uniform(slab)), 0, MI * NJ * 4096 * 4, 0x00020000);
#pragma unroll
    for (int bi = 0; bi < MI; ++bi)
#pragma unroll
        for (int bj = 0; bj < NJ; ++bj) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4_t x = __builtin_bit_cast(
                    f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, ((((wv * (MI * NJ) + bi * NJ + bj) * 4 + r4) * 64 + lane) * 4) * 4, 0, /*sc1*/ 16));
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[bi][bj][4 * r4 + j] += x[j];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
}

// C[M,N] = epi(A[M,K] . B), grid.x = [streamed pieces] + whole tiles (XCD-chunk remapped), grid.y = K splits (partials to P,
// rows_reduce_kernel finishes; never together with streamed pieces)
template <int BM, int BN, bool B_KINNER>
__device__ __forceinline__ void rows_dma_body(const RowsD& A, const float* __restrict__ Bm, long ldb, const RowsOutD& Cd, float* __restrict__ P,
                                              long m_beg, long M, int K, int N, int epi, const float* __restrict__ aux, int tiles_n,
                                              unsigned ntiles, int k_per_split, const DmaStream& sp, unsigned bx, unsigned by, int partial,
                                              float* smem) {
    constexpr int MI = BM / 64, NJ = BN / 64;
    constexpr int A_ST = BM * SK_BK, B_ST = SK_BK * BN, ST = A_ST + B_ST;
    constexpr int PA = BM / 64, PB = BN / 64;                // DMA pieces per wave per step

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
    const int split = (int)by;
    unsigned chunk;
    int kbeg, kend, part = -1;
    unsigned rem_t = 0;
    if (bx < sp.npad) {                               // wave-uniform: a piece of a streamed tile
        if (bx >= sp.pieces) return;
        rem_t = bx / (unsigned)sp.g;
        part = (int)(bx - rem_t * (unsigned)sp.g);
        chunk = sp.first_tile + rem_t;
        const int nk = (K + SK_BK - 1) / SK_BK;
        kbeg = (part * nk / sp.g) * SK_BK;
        kend = min(K, ((part + 1) * nk / sp.g) * SK_BK);
    } else {
        chunk = xcd_chunk_id(bx - sp.npad, ntiles);
        kbeg = split * k_per_split;
        kend = min(K, kbeg + k_per_split);
    }
    const int tn = chunk % tiles_n;
    const long m0 = m_beg + (long)(chunk / tiles_n) * BM;
    const int n0 = tn * BN;
    const int n = (kend - kbeg + SK_BK - 1) / SK_BK;
    const int ktail = kend - kbeg - (n - 1) * SK_BK;          // valid k of the last step (4 .. 16)

    DmaInner<BM> oa;
    {
        long roff[PA];
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            long r = m0 + 16 * (wv * PA + i) + (lane >> 2);
            if (r >= M) r = m0;
            roff[i] = row_offset(A, (unsigned)r);
        }
        oa.init(A.base, roff, kbeg, lane, wm);
    }
    DmaInner<BN> obi;
    DmaOuter<BN> obo;
    if (B_KINNER) {
        long roff[PB];
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            long r = n0 + 16 * (wv * PB + i) + (lane >> 2);
            if (r >= N) r = n0;
            roff[i] = r * ldb;
        }
        obi.init(Bm, roff, kbeg, lane, wn);
    } else {
        obo.init(Bm, ldb, n0, N, kbeg, lane, wv, wn);
    }
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // piece pc of this wave for step `step` into stage `stage`: pieces 0 .. PA-1 of A, then PB of B
    auto issue = [&](int pc, int step, int stage) {
        const bool tail = step == n - 1 && ktail < SK_BK;
        if (pc < PA) {
            const unsigned d = lds0 + (unsigned)((stage * ST + (wv * PA + pc) * 256) * 4);
            if (tail) oa.issue_tail(pc, d, ktail, lane);
            else oa.issue(pc, d);
        } else {
            const int q = pc - PA;
            const unsigned d = lds0 + (unsigned)((stage * ST + A_ST + (wv * PB + q) * 256) * 4);
            if (B_KINNER) { if (tail) obi.issue_tail(q, d, ktail, lane); else obi.issue(q, d); }
            else { if (tail) obo.issue_tail(q, d, ktail, lane, wv); else obo.issue(q, d); }
        }
    };
    auto next = [&]() {
        oa.advance();
        if (B_KINNER) obi.advance();
        else obo.advance();
    };
    auto ra = [&](const float* st, int s2, float (&v)[MI][4]) { oa.template read<MI>(st, lane, s2, v); };
    auto rb = [&](const float* st, int s2, float (&v)[NJ][4]) {
        if (B_KINNER) obi.template read<NJ>(st, lane, s2, v);
        else obo.template read<NJ>(st, lane, s2, v);
    };
    constexpr int NP = PA + PB;                               // 2 .. 4 pieces per wave per step
#pragma unroll
    for (int s = 0; s < DMA_STAGES - 1; ++s)
        if (s < n) {
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) issue(pc, s, s);
            next();
        }
    if (n >= DMA_STAGES - 1) sk_wait_vm<(DMA_STAGES - 2) * NP>();
    else sk_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    float a0[MI][4], b0[NJ][4], a1[MI][4], b1[NJ][4];
#if !LBX_DMA_LATE_READ
    if (n > 0) {
        ra(smem, 0, a0);
        rb(smem + A_ST, 0, b0);
    }
#endif
    int cur = 0;
#if LBX_DMA_LATE_READ
    // A/B variant, measured and not adopted (profiles/r03_dma_loop_variants_ab.txt): no operand registers carried across
    // the barrier -- the wait then only covers step t (the pieces of step t + 1 stay in flight: two steps of prefetch
    // distance on three stages), at the price of an exposed LDS read behind every barrier
    for (int t = 0; t < n; ++t) {
        if (t + 1 < n) sk_wait_vm<NP>();
        else sk_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        const bool more = t + DMA_STAGES - 1 < n;
        int tgt = cur + DMA_STAGES - 1;
        if (tgt >= DMA_STAGES) tgt -= DMA_STAGES;
        const float* st = smem + cur * ST;
        ra(st, 0, a0);
        rb(st + A_ST, 0, b0);
        ra(st, 1, a1);
        rb(st + A_ST, 1, b1);
        if (more) issue(0, t + DMA_STAGES - 1, tgt);
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<MI, NJ, 0, 2>(a0, b0, acc);
        if (more) issue(1, t + DMA_STAGES - 1, tgt);
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<MI, NJ, 2, 4>(a0, b0, acc);
        if (more && NP > 2) issue(2, t + DMA_STAGES - 1, tgt);
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<MI, NJ, 0, 2>(a1, b1, acc);
        if (more) {
            if (NP > 3) issue(3, t + DMA_STAGES - 1, tgt);
            next();
        }
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<MI, NJ, 2, 4>(a1, b1, acc);
        ++cur;
        if (cur == DMA_STAGES) cur = 0;
    }
#else
    for (int t = 0; t < n; ++t) {
        if (t + 1 < n) {
            if (DMA_STAGES >= 4 && t + DMA_STAGES - 2 < n) sk_wait_vm<(DMA_STAGES >= 4 ? (DMA_STAGES - 3) * NP : 0)>();
            else sk_wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();
        int nxt = cur + 1;
        if (nxt == DMA_STAGES) nxt = 0;
        const bool more = t + DMA_STAGES - 1 < n;
        int tgt = cur + DMA_STAGES - 1;
        if (tgt >= DMA_STAGES) tgt -= DMA_STAGES;
        const float* st = smem + cur * ST;
        dma_mma<MI, NJ, 0, 1>(a0, b0, acc);
        __builtin_amdgcn_sched_barrier(0);
        ra(st, 1, a1);
        rb(st + A_ST, 1, b1);
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<MI, NJ, 1, 2>(a0, b0, acc);
        if (more) issue(0, t + DMA_STAGES - 1, tgt);
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<MI, NJ, 2, 3>(a0, b0, acc);
        if (more) issue(1, t + DMA_STAGES - 1, tgt);
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<MI, NJ, 3, 4>(a0, b0, acc);
        if (more && NP > 2) issue(2, t + DMA_STAGES - 1, tgt);
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<MI, NJ, 0, 1>(a1, b1, acc);
        if (more) {
            if (NP > 3) issue(3, t + DMA_STAGES - 1, tgt);
            next();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < n) {
            const float* sn = smem + nxt * ST;
            ra(sn, 0, a0);
            rb(sn + A_ST, 0, b0);
        }
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<MI, NJ, 1, 4>(a1, b1, acc);
        cur = nxt;
    }
#endif
    if (part >= 0) {
        constexpr int SLAB = BM * BN;
        dma_slab_store<MI, NJ>(sp.slabs + (size_t)bx * SLAB, acc, wv, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                      // every wave's slab rows are out, and nobody reads the ring any more
        unsigned* flag = reinterpret_cast<unsigned*>(smem);
        if (tid == 0) {
            // arrival counter = (launch epoch << 8) | arrivals; a word of another epoch counts as zero (gemm_sk.h)
            unsigned old = __hip_atomic_load(sp.counters + rem_t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned cnt;
            for (;;) {
                cnt = (old >> 8) == sp.epoch ? (old & 255u) : 0u;
                if (__hip_atomic_compare_exchange_strong(sp.counters + rem_t, &old, (sp.epoch << 8) | (cnt + 1u), __ATOMIC_RELAXED,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                    break;
            }
            *flag = cnt;
        }
        __syncthreads();
        if (*flag != (unsigned)(sp.g - 1)) return;            // not the last arriver
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int q = 0; q < sp.g; ++q)                        // k order: the same sum whoever arrives last
            dma_slab_add<MI, NJ>(sp.slabs + ((size_t)rem_t * sp.g + q) * SLAB, acc, wv, lane);
        if (tid == 0) __hip_atomic_store(sp.counters + rem_t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    store_rows_tile<MI, NJ>(acc, m0, n0, wm, wn, lane, m_beg, M, N, epi, aux, Cd, P, split, 0ull, false, nullptr, nullptr, partial);
}

template <int BM, int BN>
constexpr int dma_stage_floats() { return (BM + BN) * SK_BK; }

template <int BM, int BN, bool B_KINNER>
__global__ __launch_bounds__(256, (BM * BN >= 16384 ? 3 : (BM * BN >= 8192 ? 4 : 6))) void gemm_rows_dma_kernel(
    RowsD A, const float* __restrict__ Bm, long ldb, RowsOutD Cd, float* __restrict__ P, long m_beg, long M, int K, int N, int epi,
    const float* __restrict__ aux, int tiles_n, unsigned ntiles, int k_per_split, DmaStream sp, ReduceJobs rj) {
    __shared__ __attribute__((aligned(16))) float smem[DMA_STAGES * dma_stage_floats<BM, BN>()];
    // carried reduces (gemm_shared.h: ReduceJobs): the first rj.total workgroups (a multiple of 8: block % 8 stays the XCD of
    // the tiles behind them) sum pending wgrads' slices -- bandwidth-bound work beside this launch's MFMA-bound tiles
    if (blockIdx.x < rj.total) {
        if (blockIdx.y == 0) reduce_jobs_run(rj, blockIdx.x);
        return;
    }
    rows_dma_body<BM, BN, B_KINNER>(A, Bm, ldb, Cd, P, m_beg, M, K, N, epi, aux, tiles_n, ntiles, k_per_split, sp, blockIdx.x - rj.total,
                                    blockIdx.y, gridDim.y > 1 ? 1 : 0, smem);
}

// K-outer operand whose k index is an implicit activation row (tn: the contraction runs over the rows), 16 rows x COLS
// columns per step; LDS image [k][COLS]; a piece = 1 KB = RPP rows.  32-bit element offsets advanced with adds only
// (SkOuterRows of gemm_sk.h for any tile width).
template <int COLS>
struct DmaOuterRows {
    static constexpr int PW = COLS / 64;
    static constexpr int LPR = COLS / 4;          // lanes per row
    static constexpr int RPP = 64 / LPR;          // rows per piece: 4 (64 columns) or 2 (128)
    const float* sb;
    int off[PW];
    unsigned tt[PW];
    int a_step, a_wrap;
    unsigned rpb;
    int mleft[PW];
    int rd;

    __device__ __forceinline__ void init(const RowsD& X, int col0, int ncols, long mbeg, long mend, int lane, int wv, int wsub) {
        sb = sk_uniform(X.base + col0);
        int c = (lane % LPR) * 4;
        if (col0 + c >= ncols) c = 0;
        a_step = (int)(SK_BK * X.rs);
        a_wrap = X.batch == 1 ? 0 : (int)(X.bs - (long)X.rpb * X.rs);
        rpb = X.batch == 1 ? 0xffffffffu : (unsigned)X.rpb;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const long m = mbeg + RPP * (wv * PW + i) + lane / LPR;
            const unsigned bq = X.batch == 1 ? 0u : (unsigned)m / rpb;
            tt[i] = (unsigned)m - (X.batch == 1 ? 0u : bq * rpb);
            off[i] = (int)((long)bq * X.bs + (long)tt[i] * X.rs) + c;
            mleft[i] = (int)(mend - m);
        }
        rd = (4 * (lane >> 5)) * COLS + wsub * (COLS / 2) + (lane & 31);
    }
    __device__ __forceinline__ void issue(int i, unsigned dst, bool tail) {
        if (!tail) {
            sk_dma_s(sb, (unsigned)off[i] * 4u, dst);
        } else {
            const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(sb) + (size_t)((unsigned)off[i] * 4u));
            sk_dma_f(mleft[i] > 0 ? p : g_sk_zero, dst);
        }
        mleft[i] -= SK_BK;
        tt[i] += SK_BK;
        off[i] += a_step;
        while (tt[i] >= rpb) { tt[i] -= rpb; off[i] += a_wrap; }
    }
    template <int NB>
    __device__ __forceinline__ void read(const float* st, int lane, int s2, float (&v)[NB][4]) const {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[b][j] = st[rd + (8 * s2 + j) * COLS + b * 32];
    }
};

// wgrad: P[split][K1][N] = A[Mslice, K1]^T . B[Mslice, N] (+ Pc[split][N] = column sums of B[Mslice] from the stages in
// LDS), the decomposition of gemm_tn_kernel (gemm.hip) on the LDS-DMA operand path with the tile shape as a template
// parameter; grid.x = nblk = tiles x splits.  Block -> (slice, tile) through the XCD-chunk remap: the tiles of one slice run on
// ONE XCD (block % 8), so a slice's activation / gradient panels are fetched by one L2 instead of eight (round 3 measured
// 2 x the input bytes at the fabric: every panel went to four or eight L2s)
template <int BM, int BN>
__device__ __forceinline__ void tn_dma_body(const RowsD& A, const RowsD& Bd, float* __restrict__ P, float* __restrict__ Pc, long M, int K1, int N,
                                            int tiles_n, int ntiles, long rows_per_split, unsigned bx, unsigned nblk, float* smem) {
    constexpr int MI = BM / 64, NJ = BN / 64;
    constexpr int A_ST = SK_BK * BM, B_ST = SK_BK * BN, ST = A_ST + B_ST;
    constexpr int PA = BM / 64, PB = BN / 64, NP = PA + PB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
    const unsigned vb = xcd_chunk_id(bx, nblk);
    const int tile = (int)(vb % (unsigned)ntiles);
    const int split = (int)(vb / (unsigned)ntiles);
    const int tn = tile % tiles_n, tk = tile / tiles_n;
    const int i0 = tk * BM, n0 = tn * BN;
    const long mbeg = (long)split * rows_per_split;
    long mend = mbeg + rows_per_split;
    if (mend > M) mend = M;
    const int n = (int)((mend - mbeg + SK_BK - 1) / SK_BK);
    const int tail_step = ((mend - mbeg) % SK_BK != 0) ? n - 1 : -1;

    DmaOuterRows<BM> oa;
    DmaOuterRows<BN> ob;
    oa.init(A, i0, K1, mbeg, mend, lane, wv, wm);
    ob.init(Bd, n0, N, mbeg, mend, lane, wv, wn);
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float csum = 0.f;
    const bool do_csum = (Pc != nullptr) && tk == 0 && tid < BN;

    auto issue = [&](int pc, int step, int stage) {
        if (pc < PA) oa.issue(pc, lds0 + (unsigned)((stage * ST + (wv * PA + pc) * 256) * 4), step == tail_step);
        else ob.issue(pc - PA, lds0 + (unsigned)((stage * ST + A_ST + (wv * PB + (pc - PA)) * 256) * 4), step == tail_step);
    };
    auto ra = [&](const float* st, int s2, float (&v)[MI][4]) { oa.template read<MI>(st, lane, s2, v); };
    auto rb = [&](const float* st, int s2, float (&v)[NJ][4]) {
        ob.template read<NJ>(st, lane, s2, v);
        if (do_csum && s2 == 0) {
#pragma unroll
            for (int kk = 0; kk < SK_BK; ++kk) csum += st[kk * BN + tid];
        }
    };
#pragma unroll
    for (int s = 0; s < DMA_STAGES - 1; ++s)
        if (s < n) {
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) issue(pc, s, s);
        }
    if (n >= DMA_STAGES - 1) sk_wait_vm<(DMA_STAGES - 2) * NP>();
    else sk_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    float a0[MI][4], b0[NJ][4], a1[MI][4], b1[NJ][4];
    if (n > 0) {
        ra(smem, 0, a0);
        rb(smem + A_ST, 0, b0);
    }
    int cur = 0;
    for (int t = 0; t < n; ++t) {
        if (t + 1 < n) sk_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        int nxt = cur + 1;
        if (nxt == DMA_STAGES) nxt = 0;
        const bool more = t + DMA_STAGES - 1 < n;
        int tgt = cur + DMA_STAGES - 1;
        if (tgt >= DMA_STAGES) tgt -= DMA_STAGES;
        const float* st = smem + cur * ST;
        dma_mma<MI, NJ, 0, 1>(a0, b0, acc);
        __builtin_amdgcn_sched_barrier(0);
        ra(st, 1, a1);
        rb(st + A_ST, 1, b1);
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<MI, NJ, 1, 2>(a0, b0, acc);
        if (more) issue(0, t + DMA_STAGES - 1, tgt);
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<MI, NJ, 2, 3>(a0, b0, acc);
        if (more) issue(1, t + DMA_STAGES - 1, tgt);
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<MI, NJ, 3, 4>(a0, b0, acc);
        if (more && NP > 2) issue(2, t + DMA_STAGES - 1, tgt);
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<MI, NJ, 0, 1>(a1, b1, acc);
        if (more && NP > 3) issue(3, t + DMA_STAGES - 1, tgt);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < n) {
            const float* sn = smem + nxt * ST;
            ra(sn, 0, a0);
            rb(sn + A_ST, 0, b0);
        }
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<MI, NJ, 1, 4>(a1, b1, acc);
        cur = nxt;
    }
    float* Pd = P + (long)split * K1 * N;
    store_partial_blocks<MI, NJ>(Pd, acc, i0, n0, wm, wn, lane, K1, N);
    if (do_csum && n0 + tid < N) Pc[(long)split * N + n0 + tid] = csum;
}

template <int BM, int BN>
__global__ __launch_bounds__(256, (BM * BN >= 16384 ? 3 : (BM * BN >= 8192 ? 4 : 6))) void gemm_tn_dma_kernel(
    RowsD A, RowsD Bd, float* __restrict__ P, float* __restrict__ Pc, long M, int K1, int N, int tiles_n, int ntiles, long rows_per_split) {
    __shared__ __attribute__((aligned(16))) float smem[DMA_STAGES * dma_stage_floats<BM, BN>()];
    tn_dma_body<BM, BN>(A, Bd, P, Pc, M, K1, N, tiles_n, ntiles, rows_per_split, blockIdx.x, gridDim.x, smem);
}

// Two independent GEMMs that read the same output gradient -- a layer's dgrad (nt) and its wgrad (tn) -- in ONE launch of
// 64 x 64 tiles: block b < rows_blocks runs the dgrad's block (b % rows_nx, b / rows_nx), the others the wgrad's.  For the
// dense head (M = batch rows: each of the two alone puts a handful of workgroups on the chip for 10-19 us of launch and
// round-trip latency) the pair costs what the longer one costs.  Same bodies, same block -> tile maps, same summation
// orders as the separate launches: bit-identical results.
struct PairRows {
    RowsD A;
    const float* Bm;
    long ldb;
    RowsOutD Cd;
    float* P;
    long M;
    int K, N, epi;
    const float* aux;
    int tiles_n;
    unsigned ntiles;
    int k_per_split;
    unsigned nx;                // blocks per K split (pieces + whole tiles)
    int partial;                // K splits > 1: raw partial sums to P
    DmaStream sp;
};
struct PairTn {
    RowsD A, Bd;
    float* P;
    float* Pc;
    long M;
    int K1, N, tiles_n, ntiles;
    long rows_per_split;
};
__global__ __launch_bounds__(256, 6) void gemm_nt_tn_pair_kernel(PairRows r, PairTn t, unsigned rows_blocks, ReduceJobs rj) {
    __shared__ __attribute__((aligned(16))) float smem[DMA_STAGES * dma_stage_floats<64, 64>()];
    if (blockIdx.x < rj.total) {                        // carried reduces of earlier layers (gemm_rows_dma_kernel)
        reduce_jobs_run(rj, blockIdx.x);
        return;
    }
    const unsigned bx = blockIdx.x - rj.total, nb = gridDim.x - rj.total;
    if (bx < rows_blocks) {
        rows_dma_body<64, 64, true>(r.A, r.Bm, r.ldb, r.Cd, r.P, 0, r.M, r.K, r.N, r.epi, r.aux, r.tiles_n, r.ntiles, r.k_per_split, r.sp,
                                    bx % r.nx, bx / r.nx, r.partial, smem);
    } else {
        tn_dma_body<64, 64>(t.A, t.Bd, t.P, t.Pc, t.M, t.K1, t.N, t.tiles_n, t.ntiles, t.rows_per_split, bx - rows_blocks, nb - rows_blocks, smem);
    }
}

}  // namespace
