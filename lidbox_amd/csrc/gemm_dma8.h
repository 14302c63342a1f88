// gemm_dma8.h -- the LDS-DMA tile kernel of gemm_dma.h worked by EIGHT waves (round 4): a 128-row tile, waves 4 (rows) x 2
// (columns), every wave one 32 x 32 accumulator block per 64 columns (BN = 64: 16 accumulator registers, BN = 128: 32).
// Why: the step's K = 512 / K = 1500 dgrads run on 64 x 64 tiles because only those reach six workgroups per CU (24 waves)
// -- launches of 8-32 K steps with a ReLU-mask / accumulate epilogue need that many waves to cover each other's epilogues
// and barriers -- at the price of the highest L2 -> LDS traffic per flop (16 B/clk/CU at the full MFMA rate).  A 128 x 64
// tile moves 12 B/clk/CU, and with one accumulator block per wave it fits 64 registers: four workgroups of eight waves = 32
// waves per CU.  Same operand path (saddr LDS-DMA ring of three stages, K-inner operands XOR-swizzled on the source side and
// read with ds_read_b128), same epilogues (store_rows_tile), same streamed remainder and carried reduces as
// gemm_rows_dma_kernel; nt (dgrads) and nn (forward: B K-outer, read with ds_read_b32).  Reference: the backward of Conv1D / Dense,
// lidbox/models/xvector.py:38-43,53-64.
#pragma once

#include "gemm_dma.h"

namespace {

// K-inner operand worked by 8 waves: ROWS = 128 or 64 rows x 16 k per step = ROWS / 16 pieces of 16 rows; wave w issues
// piece w (ROWS = 64: waves 0 .. 3 only)
template <int ROWS>
struct DmaInner8 {
    static constexpr int NPIECE = ROWS / 16;
    const float* sb;
    unsigned vo;
    int rd;
    bool mine;
    __device__ __forceinline__ void init(const float* base, long roff, int k0, int lane, int wv, int wsub, int rows_per_wave) {
        sb = sk_uniform(base + k0);
        const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
        vo = (unsigned)((roff + chunk * 4) * 4);
        rd = (wsub * rows_per_wave + (lane & 31)) * 16;
        mine = wv < NPIECE;
    }
    __device__ __forceinline__ void issue(unsigned dst) const { sk_dma_s(sb, vo, dst); }
    __device__ __forceinline__ void issue_tail(unsigned dst, int kvalid, int lane) const {
        const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
        const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(sb) + vo);
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

// K-outer operand of a plain matrix X[k][ld] worked by 8 waves (B of nn): 16 k x COLS columns per step, LDS image [k][COLS],
// a piece = 1 KB = RPP k rows; wave w issues piece w (COLS = 64: waves 0 .. 3 only)
template <int COLS>
struct DmaOuter8 {
    static constexpr int NPIECE = COLS / 16;
    static constexpr int LPR = COLS / 4;          // lanes per k row
    static constexpr int RPP = 64 / LPR;          // k rows per piece: 4 (64 columns) or 2 (128)
    const float* sb;
    unsigned vo;
    long step;
    int rd;
    int krow;                                     // this lane's k row inside a step
    bool mine;
    __device__ __forceinline__ void init(const float* base, long ld, int col0, int ncols, int k0, int lane, int wv, int wsub) {
        sb = sk_uniform(base + (long)k0 * ld + col0);
        int c = (lane % LPR) * 4;
        if (col0 + c >= ncols) c = 0;
        const int p = wv % NPIECE;
        krow = RPP * p + lane / LPR;
        vo = (unsigned)(((long)krow * ld + c) * 4);
        step = (long)SK_BK * ld;
        rd = (4 * (lane >> 5)) * COLS + wsub * (COLS / 2) + (lane & 31);
        mine = wv < NPIECE;
    }
    __device__ __forceinline__ void issue(unsigned dst) const { sk_dma_s(sb, vo, dst); }
    __device__ __forceinline__ void issue_tail(unsigned dst, int kvalid, int lane) const {
        const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(sb) + vo);
        sk_dma_f(krow < kvalid ? p : g_sk_zero, dst);
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

// slabs of the streamed remainder, accumulator order of EIGHT waves: [(wave * NJ + bj) * 4 + r4][lane][4]
template <int NJ>
__device__ __forceinline__ void dma8_slab_store(float* slab, const f32x16 (&acc)[1][NJ], int wv, int lane) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sk_uniform(slab)), 0, 8 * NJ * 4096, 0x00020000);
#pragma unroll
    for (int bj = 0; bj < NJ; ++bj)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4_t x = {acc[0][bj][4 * r4], acc[0][bj][4 * r4 + 1], acc[0][bj][4 * r4 + 2], acc[0][bj][4 * r4 + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, x), r, ((((wv * NJ + bj) * 4 + r4) * 64 + lane) * 4) * 4, 0, /*sc1*/ 16);
        }
}
template <int NJ>
__device__ __forceinline__ void dma8_slab_add(const float* slab, f32x16 (&acc)[1][NJ], int wv, int lane) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sk_uniform(slab)), 0, 8 * NJ * 4096, 0x00020000);
#pragma unroll
    for (int bj = 0; bj < NJ; ++bj) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4_t x =
                __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, ((((wv * NJ + bj) * 4 + r4) * 64 + lane) * 4) * 4, 0, /*sc1*/ 16));
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0][bj][4 * r4 + j] += x[j];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// C[M,N] = epi(A[M,K] . B), B_KINNER: B[N][K] (nt) else B[K][N] (nn); BM = 128, BN = 64 NJ;
// grid.x = [carried reduce blocks] + [streamed pieces] + whole tiles
template <int BN, bool B_KINNER>
__global__ __launch_bounds__(512, BN == 64 ? 8 : 4) void gemm_rows_dma8_kernel(RowsD A, const float* __restrict__ Bm, long ldb, RowsOutD Cd,
                                                                              float* __restrict__ P, long m_beg, long M, int K, int N, int epi,
                                                                              const float* __restrict__ aux, int tiles_n, unsigned ntiles,
                                                                              int k_per_split, DmaStream sp, ReduceJobs rj) {
    constexpr int BM = 128, NJ = BN / 64;
    constexpr int A_ST = BM * SK_BK, B_ST = SK_BK * BN, ST = A_ST + B_ST;
    __shared__ __attribute__((aligned(16))) float smem[DMA_STAGES * ST];
    if (blockIdx.x < rj.total) {                        // carried reduces (gemm_shared.h): the first 256 threads run them
        if (blockIdx.y == 0 && threadIdx.x < 256) reduce_jobs_run(rj, blockIdx.x);
        return;
    }
    const unsigned bx = blockIdx.x - rj.total;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
    const int split = (int)blockIdx.y;
    unsigned chunk;
    int kbeg, kend, part = -1;
    unsigned rem_t = 0;
    if (bx < sp.npad) {
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
    const int ktail = kend - kbeg - (n - 1) * SK_BK;

    DmaInner8<BM> oa;
    {
        long r = m0 + 16 * wv + (lane >> 2);
        if (r >= M) r = m0;
        oa.init(A.base, row_offset(A, (unsigned)r), kbeg, lane, wv, wm, 32);
    }
    DmaInner8<BN> obi;
    DmaOuter8<BN> obo;
    if (B_KINNER) {
        long r = n0 + 16 * (wv % (BN / 16)) + (lane >> 2);
        if (r >= N) r = n0;
        obi.init(Bm, r * ldb, kbeg, lane, wv, wn, 32 * NJ);
    } else {
        obo.init(Bm, ldb, n0, N, kbeg, lane, wv, wn);
    }
    const bool has_b = B_KINNER ? obi.mine : obo.mine;       // wave-uniform: this wave also moves a piece of B every step
    f32x16 acc[1][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

    auto issue_a = [&](int step, int stage) {
        const unsigned d = lds0 + (unsigned)((stage * ST + wv * 256) * 4);
        if (step == n - 1 && ktail < SK_BK) oa.issue_tail(d, ktail, lane);
        else oa.issue(d);
    };
    auto issue_b = [&](int step, int stage) {
        const unsigned d = lds0 + (unsigned)((stage * ST + A_ST + wv * 256) * 4);
        const bool tail = step == n - 1 && ktail < SK_BK;
        if (B_KINNER) { if (tail) obi.issue_tail(d, ktail, lane); else obi.issue(d); }
        else { if (tail) obo.issue_tail(d, ktail, lane); else obo.issue(d); }
    };
    auto adv_b = [&]() { if (B_KINNER) obi.advance(); else obo.advance(); };
    auto rb = [&](const float* stg, int s2, float (&v)[NJ][4]) {
        if (B_KINNER) obi.template read<NJ>(stg, lane, s2, v);
        else obo.template read<NJ>(stg, lane, s2, v);
    };
#pragma unroll
    for (int s = 0; s < DMA_STAGES - 1; ++s)
        if (s < n) {
            issue_a(s, s);
            if (has_b) issue_b(s, s);
            oa.advance();
            adv_b();
        }
    // stage 0 must have landed; stage 1's pieces (one or two per wave) may stay in flight
    if (n >= DMA_STAGES - 1) {
        if (has_b) sk_wait_vm<(DMA_STAGES - 2) * 2>();
        else sk_wait_vm<(DMA_STAGES - 2) * 1>();
    } else {
        sk_wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    float a0[1][4], b0[NJ][4], a1[1][4], b1[NJ][4];
    if (n > 0) {
        oa.template read<1>(smem, lane, 0, a0);
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
        dma_mma<1, NJ, 0, 1>(a0, b0, acc);
        __builtin_amdgcn_sched_barrier(0);
        oa.template read<1>(st, lane, 1, a1);
        rb(st + A_ST, 1, b1);
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<1, NJ, 1, 2>(a0, b0, acc);
        if (more) issue_a(t + DMA_STAGES - 1, tgt);
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<1, NJ, 2, 3>(a0, b0, acc);
        if (more && has_b) issue_b(t + DMA_STAGES - 1, tgt);
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<1, NJ, 3, 4>(a0, b0, acc);
        if (more) {
            oa.advance();
            adv_b();
        }
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<1, NJ, 0, 1>(a1, b1, acc);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < n) {
            const float* sn = smem + nxt * ST;
            oa.template read<1>(sn, lane, 0, a0);
            rb(sn + A_ST, 0, b0);
        }
        __builtin_amdgcn_sched_barrier(0);
        dma_mma<1, NJ, 1, 4>(a1, b1, acc);
        cur = nxt;
    }
    if (part >= 0) {
        constexpr int SLAB = BM * BN;
        dma8_slab_store<NJ>(sp.slabs + (size_t)bx * SLAB, acc, wv, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(smem);
        if (tid == 0) {
            unsigned old = __hip_atomic_load(sp.counters + rem_t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned cnt;
            for (;;) {
                cnt = (old >> 8) == sp.epoch ? (old & 255u) : 0u;
                if (__hip_atomic_compare_exchange_strong(sp.counters + rem_t, &old, (sp.epoch << 8) | (cnt + 1u), __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT))
                    break;
            }
            *flag = cnt;
        }
        __syncthreads();
        if (*flag != (unsigned)(sp.g - 1)) return;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
        for (int q = 0; q < sp.g; ++q) dma8_slab_add<NJ>(sp.slabs + ((size_t)rem_t * sp.g + q) * SLAB, acc, wv, lane);
        if (tid == 0) __hip_atomic_store(sp.counters + rem_t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    store_rows_tile<1, NJ>(acc, m0, n0, wm, wn, lane, m_beg, M, N, epi, aux, Cd, P, split, 0ull, false, nullptr, nullptr, gridDim.y > 1 ? 1 : 0);
}

}  // namespace
