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

// this wave's NPC pieces (8 rows x 128 B each) of one operand, pieces first .. first + NPC - 1 of the tile's rows
template <int NPC>
struct PpPieces {
    const float* sb;                        // wave-uniform byte base (+ k of the next step to issue)
    unsigned vo[NPC > 0 ? NPC : 1];         // this lane's byte offsets: row offset + swizzled chunk
    int kq;

    __device__ __forceinline__ void init(const RowsH& X, long row0, long nrows, int kbeg, int lane, int first) {
        sb = sk_uniform(reinterpret_cast<const float*>(X.base + kbeg));
        const int chunk = (lane & 7) ^ (lane >> 3);
        kq = chunk * 8;
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            long r = row0 + 8 * (first + i) + (lane >> 3);
            if (r >= nrows) r = row0;
            vo[i] = (unsigned)((row_offset(X, (unsigned)r) + chunk * 8) * 2);
        }
    }
    __device__ __forceinline__ void issue(int i, unsigned dst) const { sk_dma_s(sb, vo[i], dst); }
    __device__ __forceinline__ void issue_tail(int i, unsigned dst, int kvalid) const {
        const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(sb) + vo[i]);
        sk_dma_f(kq < kvalid ? p : g_sk_zero, dst);
    }
    __device__ __forceinline__ void advance() { sb += D16_ROW_BYTES / 4; }
};

// operand registers of block b (32 rows from `row`) for k slice ks (16 k): lane -> row lane & 31, k 16 ks + 8 (lane >> 5) ..+7
__device__ __forceinline__ bf16x8 pp_read(const char* op, int row, int lane, int b, int ks) {
    const int pos = (2 * ks + (lane >> 5)) ^ (lane & 7);
    return *reinterpret_cast<const bf16x8*>(op + (row + b * 32 + (lane & 31)) * D16_ROW_BYTES + pos * 16);
}

__device__ __forceinline__ void pp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void pp_wait_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// C[M,N] = epi(A[M,K] . B[N,K]^T), bf16 operands, fp32 accumulate; 512 threads, tile 256 x BN (BN = 256: waves 2 x 4, 128 x 64
// each; BN = 128: waves 4 x 2, 64 x 64 each); grid.x = [carried reduce blocks] + tiles (XCD-chunk remapped), grid.y = K splits.
// PA0 / PB0: A / B pieces a group-0 wave issues per step (group 1 takes the rest: its pieces have one phase less to land).
template <int BN, int SUB, int PA0, int PB0>
__global__ __launch_bounds__(512, 2) void gemm16s_rows_pp_kernel(RowsH A, RowsH Bw, RowsOutD Cd, unsigned short* __restrict__ C16,
                                                                  float* __restrict__ P, long m_beg, long M, int K, int N, int epi,
                                                                  const float* __restrict__ aux, int tiles_n, unsigned ntiles,
                                                                  int k_per_split, const unsigned short* __restrict__ mask16, ReduceJobs rj) {
    if (blockIdx.x < rj.total) {
        if (blockIdx.y == 0 && threadIdx.x < 256) reduce_jobs_run(rj, blockIdx.x);
        return;
    }
    constexpr int BM = 256;
    constexpr int WN = BN / 64, WM = 8 / WN;                  // waves along N / M
    constexpr int MI = BM / WM / 32, NJ = 2;                  // 32 x 32 blocks per wave
    constexpr int A_ST = BM * D16_ROW_BYTES, ST = (BM + BN) * D16_ROW_BYTES;
    constexpr int TA = BM / 8, TB = BN / 8;                   // pieces per stage
    constexpr int PA1 = TA / 4 - PA0, PB1 = TB / 4 - PB0;
    static_assert(PA1 >= 0 && PB1 >= 0 && (SUB == 1 || SUB == 2), "piece split");
    constexpr int KH = 4 / SUB;                               // k slices (16 k) per sub-step
    extern __shared__ __attribute__((aligned(16))) char smem16p[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wv >> 2, q = wv & 3;
    const int wm = wv / WN, wn = wv % WN;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem16p);
    const unsigned chunk = xcd_chunk_id(blockIdx.x - rj.total, ntiles);
    const int tn = chunk % tiles_n;
    const long m0 = m_beg + (long)(chunk / tiles_n) * BM;
    const int n0 = tn * BN;
    const int split = blockIdx.y;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    const int n = (kend - kbeg + D16_BK - 1) / D16_BK;
    const int ktail = kend - kbeg - (n - 1) * D16_BK;          // valid k of the last step: 8 .. 64
    const int rowA = wm * (BM / WM), rowB = wn * 64;           // this wave's first row inside the A / B part of a stage

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto body = [&](auto grp_tag) {
        constexpr int G = decltype(grp_tag)::value;
        constexpr int NA = G == 0 ? PA0 : PA1, NB = G == 0 ? PB0 : PB1;
        PpPieces<NA> pa;
        PpPieces<NB> pb;
        const int fa = G == 0 ? q * PA0 : 4 * PA0 + q * PA1, fb = G == 0 ? q * PB0 : 4 * PB0 + q * PB1;
        pa.init(A, m0, M, kbeg, lane, fa);
        pb.init(Bw, n0, N, kbeg, lane, fb);
        // this wave's pieces of step `step` into stage `stage`
        auto issue = [&](int step, int stage) {
            const bool tail = step == n - 1 && ktail < D16_BK;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const unsigned d = lds0 + (unsigned)(stage * ST + (fa + i) * 1024);
                if (tail) pa.issue_tail(i, d, ktail);
                else pa.issue(i, d);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const unsigned d = lds0 + (unsigned)(stage * ST + A_ST + (fb + i) * 1024);
                if (tail) pb.issue_tail(i, d, ktail);
                else pb.issue(i, d);
            }
            pa.advance();
            pb.advance();
        };
        bf16x8 a[MI][KH], b[NJ][KH];
        auto load = [&](int stage, int j) {
            const char* st = smem16p + stage * ST;
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) {
#pragma unroll
                for (int bj = 0; bj < NJ; ++bj) b[bj][kk] = pp_read(st + A_ST, rowB, lane, bj, j * KH + kk);
#pragma unroll
                for (int bi = 0; bi < MI; ++bi) a[bi][kk] = pp_read(st, rowA, lane, bi, j * KH + kk);
            }
        };
        auto comp = [&]() {
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
        // step 0 into stage 0: every wave its pieces, then the first hand-off
        issue(0, 0);
        sk_wait_vm<0>();
        pp_barrier();
        if (G == 1) pp_barrier();                              // phase 0 belongs to group 0 alone
        int cur = 0;
        for (int t = 0; t < n; ++t) {
            const bool more = t + 1 < n;
#pragma unroll
            for (int j = 0; j < SUB; ++j) {
                if (j == 0 && more) issue(t + 1, cur ^ 1);
                load(cur, j);
                pp_wait_lds();
                if (G == 1 && j == SUB - 1 && more) sk_wait_vm<0>();
                pp_barrier();
                comp();
                if (G == 0) {
                    if (j == SUB - 1 && more) sk_wait_vm<0>();
                    pp_barrier();
                } else if (!(j == SUB - 1 && !more)) {
                    pp_barrier();
                }
            }
            cur ^= 1;
        }
    };
    if (grp == 0) body(IntTag<0>{});
    else body(IntTag<1>{});

    // epilogue: the wave's tile as 64-row halves (store_rows_tile addresses rows as m0 + wm * 64 + block * 32)
#pragma unroll
    for (int hh = 0; hh < MI / 2; ++hh)
        store_rows_tile<2, NJ, true>(reinterpret_cast<const f32x16(&)[2][NJ]>(acc[2 * hh]), m0 + rowA + 64 * hh, n0, 0, wn, lane, m_beg, M, N, epi,
                                     aux, Cd, P, split, 0ull, false, C16, mask16);
}

}  // namespace
