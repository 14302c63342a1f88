// gemm16_dma.h -- bf16-storage rows GEMM (lidbox_gemm_bf16s_nt: reference xvector.py:38-43,53-64 under a bfloat16 compute
// policy, see gemm_bf16.hip) on the LDS-DMA operand path, included by gemm_bf16.hip.
//
// Why: the launches of one bf16 train step are SHORT (8 448 x 512 x 512 at bs 256 = 4.4 GFLOP, 23 us on the 128 x 128
// register-staged kernel = 0.08 of the bf16 MFMA peak): one or two K steps deep per CU, they spend their time in the first
// round trip to memory, the staging chain global -> VGPR -> ds_write -> barrier, and the C epilogue of the one workgroup a
// CU holds.  Here both operand tiles of a step go from global memory to LDS with global_load_lds_dwordx4 (no staging
// registers, no ds_write) into a ring of S stages, the tile shape is a template parameter (64 x 64 tiles put 4+ independent
// workgroups on a CU: their round trips and epilogues overlap), and one s_barrier per step publishes a stage.
//
// Layout.  Both operands have the contraction index contiguous in HBM (A = bf16 shadow rows, B = weight shadow [N][K]).
// A step is 64 k = one 128-byte line per row = 8 chunks of 16 bytes.  The LDS image of a DMA is lane-linear (lane i lands
// at base + 16 i), so a wave instruction moves 8 rows x 128 B and the swizzle is applied on the SOURCE side: lane i =
// (row i >> 3, position i & 7) fetches chunk (i & 7) ^ d16_swz(row) of its row.  The MFMA operand of v_mfma_f32_32x32x16_bf16
// (lane -> row lane & 31, 8 consecutive k at 16 ks + 8 (lane >> 5)) is then one ds_read_b128 at position chunk ^ d16_swz(row)
// -- conflict-free for the lane groups the LDS serves a b128 access in (d16_swz below).
// K tails (K % 64 != 0; K % 8 == 0 always) take their missing chunks from a 16-byte zero source (flat-form DMA on that
// step only); rows past M read the tile's first row (never stored).
#pragma once

#include "lds_dma.h"

namespace {

constexpr int D16_BK = 64;                  // bf16 per row per step
constexpr int D16_ROW_BYTES = D16_BK * 2;   // 128

// Swizzle key of a tile row: a row's 16-byte chunk c sits at position c ^ d16_swz(row) of its 128-byte LDS row.
// The operand fetch is one ds_read_b128 per lane with lane -> row lane & 31, and the LDS serves a b128 wave access in four
// groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS table) -- each
// covering the 64 banks once when its 16 lanes hit 16 different 16-byte slots mod 256 bytes: slot = (row & 1) * 8 + position.
// The 8 even (odd) rows of a group differ exactly in bits 1, 3, 4 of the row index, so THOSE bits are the key.  Round 3's
// key, row & 7 (bits 0-2), assumed groups of 8 consecutive lanes: rows 12 and 20 (4 and 28, ...) of one group then share
// key and parity -> every operand fetch was a 2-way bank conflict (LBX_D16_SWZ=0 rebuilds that layout: A/B aid).
#ifndef LBX_D16_SWZ
#define LBX_D16_SWZ 1
#endif
__device__ __forceinline__ int d16_swz(int row) {
#if LBX_D16_SWZ
    return ((row >> 1) & 1) | ((row >> 2) & 2) | ((row >> 2) & 4);
#else
    return row & 7;
#endif
}

template <int ROWS>
struct Dma16Operand {
    static constexpr int PW = ROWS / 32;    // pieces (8 rows x 128 B) per wave per step
    const float* sb;                        // wave-uniform byte base (+ k of the next step to issue)
    unsigned vo[PW];                        // this lane's byte offsets: row offset + swizzled chunk
    unsigned kqs;                           // 3 bits per piece: the chunk this lane fetches (first k of it = 8 x that)
    int rd;                                 // LDS read base (bytes inside the operand's stage): row of the first block

    __device__ __forceinline__ void init(const RowsH& X, long row0, long nrows, int kbeg, int lane, int wv, int wsub) {
        sb = sk_uniform(reinterpret_cast<const float*>(X.base + kbeg));
        kqs = 0;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int trow = 8 * (wv * PW + i) + (lane >> 3);          // row inside the tile = LDS row
            const int chunk = (lane & 7) ^ d16_swz(trow);
            kqs |= (unsigned)chunk << (3 * i);
            long r = row0 + trow;
            if (r >= nrows) r = row0;
            vo[i] = (unsigned)((row_offset(X, (unsigned)r) + chunk * 8) * 2);
        }
        rd = (wsub * (ROWS / 2) + (lane & 31)) * D16_ROW_BYTES;
    }
    __device__ __forceinline__ void issue(int i, unsigned dst) const { sk_dma_s(sb, vo[i], dst); }
    // last step of a K range that is not a multiple of 64: chunks at or past kvalid come from the zero source
    __device__ __forceinline__ void issue_tail(int i, unsigned dst, int kvalid) const {
        const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(sb) + vo[i]);
        sk_dma_f((int)((kqs >> (3 * i)) & 7u) * 8 < kvalid ? p : g_sk_zero, dst);
    }
    __device__ __forceinline__ void advance() { sb += D16_ROW_BYTES / 4; }
    // operand registers of block b (32 rows) for k slice ks (16 k): lane -> row lane & 31, k 16 ks + 8 (lane >> 5) ..+7
    __device__ __forceinline__ bf16x8 read(const char* st, int lane, int b, int ks) const {
        const int pos = (2 * ks + (lane >> 5)) ^ d16_swz(lane & 31);
        return *reinterpret_cast<const bf16x8*>(st + rd + b * 32 * D16_ROW_BYTES + pos * 16);
    }
};

constexpr int D16_EPI_BYTES = 32 * 68 * 4;                  // one wave's 32-row x 64-column fp32 strip (gemm16_pp.h: PP_EPI_*)
template <int MI, int NJ>
__device__ __forceinline__ void d16_store_tile_wide(const f32x16 (&acc)[MI][NJ], float* __restrict__ wlds, long mrow0, int ncol0, int lane,
                                                    long m_beg, long M, int N, int epi, const float* __restrict__ aux, const RowsOutD& Cd,
                                                    float* __restrict__ P, int split, unsigned short* __restrict__ shadow,
                                                    const unsigned short* __restrict__ mask16);

template <int N>
__device__ __forceinline__ void d16_wait_le() {
    sk_wait_vm<N>();
}

// C[M,N] = epi(A[M,K] . B[N,K]^T), bf16 operands, fp32 accumulate; grid.x = tiles (XCD-chunk remapped), grid.y = K splits
template <int BM, int BN, int STAGES, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm16s_rows_dma_kernel(RowsH A, RowsH Bw, RowsOutD Cd, unsigned short* __restrict__ C16,
                                                                    float* __restrict__ P, long m_beg, long M, int K, int N, int epi,
                                                                    const float* __restrict__ aux, int tiles_n, unsigned ntiles,
                                                                    int k_per_split, const unsigned short* __restrict__ mask16, ReduceJobs rj) {
    // carried reduces (gemm_shared.h: ReduceJobs): the leading rj.total workgroups sum pending wgrads' slices
    if (blockIdx.x < rj.total) {
        if (blockIdx.y == 0) reduce_jobs_run(rj, blockIdx.x);
        return;
    }
    constexpr int MI = BM / 64, NJ = BN / 64;
    constexpr int A_ST = BM * D16_ROW_BYTES, ST = (BM + BN) * D16_ROW_BYTES;      // bytes
    constexpr int PA = BM / 32, PB = BN / 32, NP = PA + PB;                       // DMA pieces per wave per step: 4 .. 8
    constexpr int NKS = D16_BK / 16;
    extern __shared__ __attribute__((aligned(16))) char smem16d[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem16d);
    const unsigned chunk = xcd_chunk_id(blockIdx.x - rj.total, ntiles);
    const int tn = chunk % tiles_n;
    const long m0 = m_beg + (long)(chunk / tiles_n) * BM;
    const int n0 = tn * BN;
    const int split = blockIdx.y;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    const int n = (kend - kbeg + D16_BK - 1) / D16_BK;
    const int ktail = kend - kbeg - (n - 1) * D16_BK;          // valid k of the last step: 8 .. 64

    Dma16Operand<BM> oa;
    Dma16Operand<BN> ob;
    oa.init(A, m0, M, kbeg, lane, wv, wm);
    ob.init(Bw, n0, N, kbeg, lane, wv, wn);

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // piece pc of this wave for step `step` into stage `stage`: pieces 0 .. PA-1 of A, then PB of B
    auto issue = [&](int pc, int step, int stage) {
        const bool tail = step == n - 1 && ktail < D16_BK;
        if (pc < PA) {
            const unsigned d = lds0 + (unsigned)(stage * ST + (wv * PA + pc) * 1024);
            if (tail) oa.issue_tail(pc, d, ktail);
            else oa.issue(pc, d);
        } else {
            const int q = pc - PA;
            const unsigned d = lds0 + (unsigned)(stage * ST + A_ST + (wv * PB + q) * 1024);
            if (tail) ob.issue_tail(q, d, ktail);
            else ob.issue(q, d);
        }
    };
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < n) {
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) issue(pc, s, s);
            oa.advance();
            ob.advance();
        }
    int cur = 0;
    for (int t = 0; t < n; ++t) {
        // step t's pieces have landed (this wave's; the barrier covers the others'): the steps issued after it may stay in flight
        const int later = min(STAGES - 2, n - 1 - t);
        if (later >= 2) d16_wait_le<2 * NP>();
        else if (later == 1) d16_wait_le<NP>();
        else d16_wait_le<0>();
        __builtin_amdgcn_s_barrier();
        const bool more = t + STAGES - 1 < n;
        int tgt = cur + STAGES - 1;
        if (tgt >= STAGES) tgt -= STAGES;                       // the stage step t - 1 was read from: free since the barrier
        const char* st = smem16d + cur * ST;
        bf16x8 a[2][MI], b[2][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) a[0][i] = oa.read(st, lane, i, 0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[0][j] = ob.read(st + A_ST, lane, j, 0);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int c = ks & 1, nx = c ^ 1;
            if (ks + 1 < NKS) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[nx][i] = oa.read(st, lane, i, ks + 1);
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[nx][j] = ob.read(st + A_ST, lane, j, ks + 1);
            }
            // the DMAs of step t + STAGES - 1 go out between the MFMA groups (an issue holds the wave ~56 cycles)
            if (more) {
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
                    if (pc * NKS / NP == ks) issue(pc, t + STAGES - 1, tgt);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            oa.advance();
            ob.advance();
        }
        ++cur;
        if (cur == STAGES) cur = 0;
    }
#ifndef LBX_D16_WIDE_EPI
#define LBX_D16_WIDE_EPI 1
#endif
    // Epilogues that READ per element (ReLU mask, accumulate: the dgrads) take the 16-byte epilogue of the eight-wave tile
    // (gemm16_pp.h: pp_store_tile) through a wave-private strip of the ring -- frame5 / frame4 dgrad 30.6 -> 28.8 / 15.4 -> 12.9 us at
    // bs 256, 55.5 -> 49.2 / 26.7 -> 22.3 us at bs 512; store-only epilogues keep the row loop, which the other resident workgroups
    // hide and which costs no LDS round trip (forward launches were 1-5 us slower on the strip: profiles/r05_bf16_wide_epilogue_ab.txt)
    const bool reads = gridDim.y == 1 && (epi == LIDBOX_EPI_RELU_MASK || epi == LIDBOX_EPI_ACCUM_RELU_MASK || epi == LIDBOX_EPI_ACCUM ||
                                          epi == LIDBOX_EPI_ACCUM_RELU);
    if constexpr (LBX_D16_WIDE_EPI && NJ == 2 && STAGES * ST >= 4 * D16_EPI_BYTES) {
        if (reads) {
            // nobody reads the ring any more behind this barrier (the last step's DMAs were waited for before its own barrier)
            __builtin_amdgcn_s_barrier();
            d16_store_tile_wide<MI, NJ>(acc, reinterpret_cast<float*>(smem16d + wv * D16_EPI_BYTES), m0 + wm * (32 * MI), n0 + wn * (32 * NJ), lane,
                                        m_beg, M, N, epi, aux, Cd, P, split, C16, mask16);
            return;
        }
    }
    (void)reads;
    store_rows_tile<MI, NJ, true>(acc, m0, n0, wm, wn, lane, m_beg, M, N, epi, aux, Cd, P, split, 0ull, false, C16, mask16);
}

}  // namespace
