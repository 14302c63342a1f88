// gemm.hip -- fp32 MFMA GEMM family behind Conv1D(padding="causal") / Dense forward, dgrad,
// wgrad and linear_to_mel (gfx950).
//
// Replaces (reference file:line):
//   lidbox/models/xvector.py:38-43,53-64   frame_layer (Conv1D causal, strided) / segment_layer (Dense)
//   lidbox/models/cnn.py:32-41             Conv1D / Dense stack of the CNN classifier
//   lidbox/features/audio.py:261           tf.tensordot(spectrograms, mel_weights, 1)
// and the backward passes Keras derives for them (keras_utils.py:191-203, Model.fit).
//
// Design
//   * Activations live in HBM as [B, (k-1) zero rows + T, C]; a causal window of the NEXT layer
//     is then a contiguous run of k*C floats, so Conv1D is a GEMM whose A rows are addressed
//     as base + b*batch_stride + t*row_stride (lidbox_rows_t) -- no im2col buffer, no bounds
//     logic in the inner loop.  Strided layers simply use row_stride = s*C.
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  Block tile BM x BN in
//     {128,64}^2, 4 waves as 2x2, each wave (BM/2)x(BN/2) = up to 2x2 MFMA blocks.
//   * LDS tiles are K-outer ([k][row]) for both operands so every MFMA operand fetch is one
//     conflict-free ds_read_b32 of 32 consecutive floats per half-wave.  Operands whose
//     contraction index is contiguous in HBM (A of NN/NT, B of NT) are transposed on the way
//     in (row stride = rows + 2 keeps those scattered ds_write_b32 conflict-free); the others
//     are copied with ds_write_b128.
//   * global -> register prefetch of tile t+1 is issued before the MFMAs of tile t and
//     written to the other LDS buffer afterwards: one barrier per K step; 33 KB LDS per
//     workgroup keeps 4 workgroups per CU resident to cover each other's barriers.
//   * Work decomposition is chosen per launch by a small cost model (choose_rows / tn_plan):
//     the chip has 256 CUs and an fp32 MFMA tile is long, so tile COUNT quantisation is the
//     first-order effect (264 tiles of 128x128 run at 52 %).  Small-M problems (Dense layers,
//     M = 256) are split along K into partial sums that a reduce kernel sums in a fixed
//     order and finishes with the fused epilogue (deterministic; no float atomics).
//   * block id -> tile uses the XCD-chunk remap so the n-tiles that share an A row panel run
//     on one XCD's L2; the weight matrix (<= 3 MB) is L2-resident everywhere.
//   * wgrad contracts over M = B*T (up to 50 688): always split; the bias gradient (column
//     sums of dY) is accumulated from the dY tiles already in LDS by the k1-tile-0 workgroups.
//   * Round 3: which KERNEL runs a decomposition.  16-byte aligned operands with 32-bit extents -- every layer of the models
//     -- go through the LDS-DMA operand path: gemm_dma.h (the decompositions above, one tile per workgroup, with the last
//     partial round of tiles streamed along K inside the launch; a layer's small dgrad + wgrad as one launch:
//     lidbox_gemm_nt_tn) or, for long-K forward GEMMs and large wgrads, the persistent stream-K kernels of gemm_sk.h.  The
//     register-staged kernels of THIS file keep everything else (unaligned operands, LIDBOX_GEMM_DMA=0) and remain the
//     reference implementation the A/B tools compare against.  Dispatch: launch_rows / lidbox_gemm_tn below.
// Roofline: MFMA fp32, 157.3 TFLOP/s.
#include <stdlib.h>

#include <atomic>

#include "gemm_shared.h"
#include "gemm_sk.h"
#include <string.h>
#include "gemm_dma.h"
#include "gemm_dma8.h"
#include "gemm_tuned.h"

namespace {

#ifndef LBX_GEMM_BK
#define LBX_GEMM_BK 16
#endif
#ifndef LBX_GEMM_PRIO
#define LBX_GEMM_PRIO 3                    // s_setprio level of the MFMA phase (0 = leave the default)
#endif
constexpr int BK = LBX_GEMM_BK;           // K depth of one LDS tile (tuning aid: -DLBX_GEMM_BK=32)
#ifndef LBX_GEMM_WAVES_HINT
#define LBX_GEMM_WAVES_HINT 1              // occupancy hint per tile shape: the register allocator then keeps the batched
                                           // epilogue (gemm_shared.h) inside the K loop's budget -- 64x64: 56 registers
                                           // (8 waves/SIMD), 128x64: 73 (6), 128x128: 116 (4); without it 72 / 104 / 180
#endif
#ifndef LBX_GEMM_MASK_PREFETCH
#define LBX_GEMM_MASK_PREFETCH 0           // 1: fetch the ReLU mask as bits inside the K loop (A/B: 2375 vs 2355 us -- the mask
                                           // costs bandwidth / issue slots, not epilogue latency)
#endif
#ifndef LBX_GEMM_WAVES_128
#define LBX_GEMM_WAVES_128 3               // waves/SIMD asked for 128x128 tiles (4 would need <= 128 registers)
#endif
#ifndef LBX_GEMM_TN_HINT
#define LBX_GEMM_TN_HINT 1                 // 1: hint on the 128x128 wgrad kernel only (-2.5 % on the three launches that use it;
                                           // 2: on every wgrad tile -- 64x64 and 128x64 measured 8-27 % SLOWER with it)
#endif
#if LBX_GEMM_WAVES_HINT
#define LBX_ROWS_BOUNDS(BM, BN) __launch_bounds__(256, ((BM) * (BN) >= 16384 ? LBX_GEMM_WAVES_128 : ((BM) * (BN) >= 8192 ? 6 : 8)))
#else
#define LBX_ROWS_BOUNDS(BM, BN) __launch_bounds__(256)
#endif
#if LBX_GEMM_TN_HINT == 2
#define LBX_TN_BOUNDS(BM, BN) LBX_ROWS_BOUNDS(BM, BN)
#elif LBX_GEMM_TN_HINT == 1
#define LBX_TN_BOUNDS(BM, BN) __launch_bounds__(256, ((BM) * (BN) >= 16384 ? LBX_GEMM_WAVES_128 : 1))
#else
#define LBX_TN_BOUNDS(BM, BN) __launch_bounds__(256)
#endif

// ---- K-inner operand (contraction index contiguous in HBM): ROWS x BK tile, transposed into
//      LDS [BK][ROWS + 2].  Each thread keeps one source pointer per pass and bumps it by BK.
//      Rows outside the matrix are CLAMPED to row 0 (finite data whose products only reach
//      output rows/columns that are never stored), so interior K-steps carry no predicates at
//      all; only the tail step (CHECK) tests k and zero-fills.
template <int ROWS, bool ALIGNED, int NT = 256>
struct KInnerLoader {
    static constexpr int F4R = BK / 4;                // float4 per tile row: 4 (BK 16) or 8 (BK 32)
    static constexpr int RPP = NT / F4R;              // tile rows per pass: 64 or 32 (256 threads), 128 (512)
    static constexpr int PASSES = ROWS >= RPP ? ROWS / RPP : 1;
    static constexpr bool PARTIAL = ROWS < RPP;       // more threads than float4s in the tile: the upper ones idle
    static constexpr int LD = ROWS + 8 / F4R;         // = 2 or 1 (mod 32): conflict-free transposed stores
    float4 v[PASSES];
    const float* ptr[PASSES];
    int c4, r0, k;

    __device__ __forceinline__ void init(const RowsD& rows, long row_base, long nrows, int tid, int kbeg) {
        c4 = tid % F4R;
        r0 = tid / F4R;
        k = kbeg + c4 * 4;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const long r = row_base + r0 + p * RPP;
            ptr[p] = rows.base + ((r < nrows && !(PARTIAL && r0 >= ROWS)) ? row_offset(rows, (unsigned)r) : 0) + k;
        }
    }
    // loads the tile starting at the current k, then advances by BK
    template <bool CHECK>
    __device__ __forceinline__ void load(int kend) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            float4 x;
            if (!CHECK) {
                if (ALIGNED) x = *reinterpret_cast<const float4*>(ptr[p]);
                else x = make_float4(ptr[p][0], ptr[p][1], ptr[p][2], ptr[p][3]);
            } else {
                x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ALIGNED) {
                    if (k < kend) x = *reinterpret_cast<const float4*>(ptr[p]);
                } else {
                    if (k + 0 < kend) x.x = ptr[p][0];
                    if (k + 1 < kend) x.y = ptr[p][1];
                    if (k + 2 < kend) x.z = ptr[p][2];
                    if (k + 3 < kend) x.w = ptr[p][3];
                }
            }
            ptr[p] += BK;
            v[p] = x;
        }
        k += BK;
    }
    __device__ __forceinline__ void store(float* tile) const {
        if (PARTIAL && r0 >= ROWS) return;               // idle threads loaded row 0 (in bounds) and store nothing
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            float* d = tile + (c4 * 4) * LD + r0 + p * RPP;
            d[0 * LD] = v[p].x;
            d[1 * LD] = v[p].y;
            d[2 * LD] = v[p].z;
            d[3 * LD] = v[p].w;
        }
    }
};

// ---- K-outer operand: BK x COLS tile copied into LDS [BK][COLS]; rows are the contraction index.
//      Columns outside the matrix are clamped to column 0 on the aligned path (never stored);
//      the unaligned path keeps per-element column tests (it must not read past a row's end).
template <int COLS, bool ALIGNED, int NT = 256>
struct KOuterLoader {
    static constexpr int F4 = COLS / 4;                // float4 per tile row: 32 or 16
    static constexpr int RPP = NT / F4;                // tile rows per pass: 8 or 16 (256 threads)
    static constexpr int PASSES = BK >= RPP ? BK / RPP : 1;   // 2 or 1
    static constexpr bool PARTIAL = BK < RPP;          // more threads than float4s in the tile: the upper ones idle
    static constexpr int LD = COLS;
    float4 v[PASSES];
    const float* ptr[PASSES];                          // plain-matrix mode: bumped by BK*ld per step
    long step;
    int c4, kk0, c, cload, kcur, ncols_;

    __device__ __forceinline__ void init(int tid, int col0, int ncols) {
        c4 = tid % F4;
        kk0 = tid / F4;
        if (PARTIAL && kk0 >= BK) kk0 -= BK;           // surplus threads mirror an active one: same loads, same LDS stores
        c = col0 + c4 * 4;
        ncols_ = ncols;
        cload = (ALIGNED && c >= ncols) ? 0 : c;       // clamped column used for addressing
    }
    // plain matrix: row k at base + k*ld
    __device__ __forceinline__ void init_plain(const float* base, long ld, int kbeg, int col0, int ncols, int tid) {
        init(tid, col0, ncols);
        kcur = kbeg + kk0;
        step = (long)BK * ld;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) ptr[p] = base + (long)(kcur + RPP * p) * ld + cload;
    }
    template <bool CHECK>
    __device__ __forceinline__ void load_plain(int kend) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            v[p] = fetch<CHECK>(ptr[p], (kcur + RPP * p) < kend);
            ptr[p] += step;
        }
        kcur += BK;
    }
    // implicit rows: caller supplies the per-pass row offsets (tracked incrementally)
    template <bool CHECK>
    __device__ __forceinline__ void load_rows(const float* base, const long (&off)[PASSES],
                                              const bool (&ok)[PASSES]) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) v[p] = fetch<CHECK>(base + off[p] + cload, ok[p]);
    }
    template <bool CHECK>
    __device__ __forceinline__ float4 fetch(const float* src, bool row_ok) const {
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ALIGNED) {
            if (!CHECK || row_ok) x = *reinterpret_cast<const float4*>(src);
        } else if (!CHECK || row_ok) {
            if (c + 0 < ncols_) x.x = src[0];
            if (c + 1 < ncols_) x.y = src[1];
            if (c + 2 < ncols_) x.z = src[2];
            if (c + 3 < ncols_) x.w = src[3];
        }
        return x;
    }
    __device__ __forceinline__ void store(float* tile) const {
#pragma unroll
        for (int p = 0; p < PASSES; ++p)
            *reinterpret_cast<float4*>(tile + (kk0 + RPP * p) * LD + c4 * 4) = v[p];
    }
};

// (BM/2)x(BN/2) per wave: MI x NJ blocks of v_mfma_f32_32x32x2_f32 over one BK-deep LDS tile pair.
// Operand registers are double-buffered by hand: the ds_reads of k-pair kk+2 are issued before
// the MFMAs of k-pair kk, so LDS latency hides under 4 x 64 cycles of matrix work instead of
// serialising read -> wait -> MFMA.
template <int MI, int NJ, int LDA, int LDB>
__device__ __forceinline__ void mma_tile(const float* As, const float* Bs, int wm, int wn, int lane,
                                         f32x16 (&acc)[MI][NJ]) {
    const int h = lane >> 5, l = lane & 31;
    const float* ap = As + h * LDA + wm * (32 * MI) + l;
    const float* bp = Bs + h * LDB + wn * (32 * NJ) + l;
    float a[2][MI], b[2][NJ];
    // Waves in their MFMA phase outrank co-resident waves that are in the load/store/barrier phase
    // (measured in tools/micro/gemm_loop.hip: +5 % at 3 workgroups/CU, +15 % at 1).
    if (LBX_GEMM_PRIO) __builtin_amdgcn_s_setprio(LBX_GEMM_PRIO);
#pragma unroll
    for (int i = 0; i < MI; ++i) a[0][i] = ap[32 * i];
#pragma unroll
    for (int j = 0; j < NJ; ++j) b[0][j] = bp[32 * j];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
        const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
        if (kk + 2 < BK) {
#pragma unroll
            for (int i = 0; i < MI; ++i) a[nxt][i] = ap[(kk + 2) * LDA + 32 * i];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[nxt][j] = bp[(kk + 2) * LDB + 32 * j];
        }
        __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ahead of this k-pair's MFMAs
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
    }
    if (LBX_GEMM_PRIO) __builtin_amdgcn_s_setprio(0);
}

#ifdef LBX_GEMM_TIMING
// Debug builds only (tools/gemm_phases.py): per-wave s_memtime samples of the rows kernels.
//   [0] kernel entry  [1] first tile staged (after the first barrier)  [2] K loop done  [3] epilogue done
//   [4] cycles inside mma_tile, [5] issuing the next tile's global loads, [6] LDS stores (incl. the wait for those loads),
//   [7] waiting at the barrier -- summed over the K steps; [8] SIMD/CU/XCD id word (HW_ID)
__device__ long long* g_gemm_stamps = nullptr;
#define LBX_T() ((long long)__builtin_amdgcn_s_memtime())
#endif

// ------------------------------------------------------------------------------------------------
// C[M,N] = epi(A[M,K] . B)    B_KINNER = false: B[K][N] (NN)   true: B[N][K] (NT)
// grid.x = tiles (XCD-chunk remapped), grid.y = K splits.  splits > 1: raw partial sums go to
// P[split][M][N] and rows_reduce_kernel finishes.
// ------------------------------------------------------------------------------------------------
// WAVES = 4: waves 2 x 2.  WAVES = 8 (gemm_rows8_kernel): waves 4 x 2 over the same tile -- half the accumulators per
// wave, so twice the waves fit a SIMD at the L2 -> LDS traffic of the larger tile.
template <int BM, int BN, bool B_KINNER, bool ALIGNED, int WAVES>
__device__ __forceinline__ void gemm_rows_body(const RowsD& A, const float* __restrict__ Bm, long ldb,
                                               const RowsOutD& Cd, float* __restrict__ P, long m_beg, long M, int K,
                                               int N, int epi, const float* __restrict__ aux,
                                               int tiles_n, unsigned ntiles, int k_per_split) {
    constexpr int WN = 2, WM = WAVES / WN, NT = 64 * WAVES;
    constexpr int MI = BM / (32 * WM), NJ = BN / (32 * WN);
    using LA = KInnerLoader<BM, ALIGNED, NT>;
    using LBI = KInnerLoader<BN, ALIGNED, NT>;
    using LBO = KOuterLoader<BN, ALIGNED, NT>;
    constexpr int LDA = LA::LD;
    constexpr int LDB = B_KINNER ? LBI::LD : LBO::LD;
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef LBX_GEMM_TIMING
    long long ts0 = LBX_T(), t_mma = 0, t_ld = 0, t_st = 0, t_bar = 0;
#endif
    const int wm = wave / WN, wn = wave % WN;
    const unsigned chunk = xcd_chunk_id(blockIdx.x, ntiles);
    const int tn = chunk % tiles_n;
    const long tm = chunk / tiles_n;
    const long m0 = m_beg + tm * BM;            // this launch covers rows [m_beg, M)
    const int n0 = tn * BN;
    const int split = blockIdx.y;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);

    LA la;
    la.init(A, m0, M, tid, kbeg);
    LBI lbi;
    LBO lbo;
    RowsD Brows{Bm, 0, ldb, 1, 0};
    if (B_KINNER) lbi.init(Brows, n0, N, tid, kbeg);
    else lbo.init_plain(Bm, ldb, kbeg, n0, N, tid);

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        la.template load<true>(kend);
        if (B_KINNER) lbi.template load<true>(kend);
        else lbo.template load_plain<true>(kend);
        la.store(As[0]);
        if (B_KINNER) lbi.store(Bs[0]);
        else lbo.store(Bs[0]);
    }
    __syncthreads();
#ifdef LBX_GEMM_TIMING
    const long long ts1 = LBX_T();
#endif

    // Backward epilogues multiply by the ReLU mask (aux > 0, aux = the forward activation, C's layout).  Reading
    // it in the epilogue costs one dependent HBM round trip per output row while the workgroup holds its slot
    // without issuing MFMAs (PMC: 31 % of wave cycles parked in the NT launches vs 18 % in NN).  Instead every
    // lane fetches its 16*MI*NJ mask values two per K-step, ahead of that step's tile loads (so they are back
    // when the tile is), and keeps them as bits: the epilogue is store-only.
    constexpr int NPRE = 16 * MI * NJ;
    const bool pre_mask = LBX_GEMM_MASK_PREFETCH && gridDim.y == 1 &&
                          (epi == LIDBOX_EPI_RELU_MASK || epi == LIDBOX_EPI_ACCUM_RELU_MASK);
    unsigned long long mbits = 0ull;
    auto mask_ptr = [&](int idx) -> const float* {
        const int r = idx & 15, blk = idx >> 4, bj = blk % NJ, bi = blk / NJ;
        const long row = m0 + wm * (32 * MI) + bi * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
        const int col = n0 + wn * (32 * NJ) + bj * 32 + (lane & 31);
        return aux + ((row < M && col < N) ? row_offset(Cd, (unsigned)row) + col : 0);
    };

    // The prefetch of step kt targets tile kt+1; only the LAST tile can be partial, so interior
    // prefetches carry no predicates (a wave-uniform scalar branch picks the variant).  One MMA
    // site keeps the accumulators in place.
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool mload = pre_mask && 2 * kt < NPRE;        // wave-uniform
        float mv0 = 0.f, mv1 = 0.f;
        if (mload) {
            mv0 = *mask_ptr(2 * kt);
            mv1 = *mask_ptr(2 * kt + 1);
        }
#ifdef LBX_GEMM_TIMING
        __builtin_amdgcn_sched_barrier(0);
        const long long q0 = LBX_T();
#endif
        if (kt + 2 < nk) {
            la.template load<false>(kend);
            if (B_KINNER) lbi.template load<false>(kend);
            else lbo.template load_plain<false>(kend);
        } else if (kt + 1 < nk) {
            la.template load<true>(kend);
            if (B_KINNER) lbi.template load<true>(kend);
            else lbo.template load_plain<true>(kend);
        }
#ifdef LBX_GEMM_TIMING
        __builtin_amdgcn_sched_barrier(0);
        const long long q1 = LBX_T();
#endif
        mma_tile<MI, NJ, LDA, LDB>(As[cur], Bs[cur], wm, wn, lane, acc);
#ifdef LBX_GEMM_TIMING
        __builtin_amdgcn_sched_barrier(0);
        const long long q2 = LBX_T();
#endif
        if (kt + 1 < nk) {
            la.store(As[cur ^ 1]);
            if (B_KINNER) lbi.store(Bs[cur ^ 1]);
            else lbo.store(Bs[cur ^ 1]);
        }
        if (mload)
            mbits |= ((unsigned long long)(mv0 > 0.f) << (2 * kt)) | ((unsigned long long)(mv1 > 0.f) << (2 * kt + 1));
#ifdef LBX_GEMM_TIMING
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const long long q3 = LBX_T();
#endif
        __syncthreads();
#ifdef LBX_GEMM_TIMING
        __builtin_amdgcn_sched_barrier(0);
        const long long q4 = LBX_T();
        t_ld += q1 - q0; t_mma += q2 - q1; t_st += q3 - q2; t_bar += q4 - q3;
#endif
    }
#ifdef LBX_GEMM_TIMING
    const long long ts2 = LBX_T();
#endif
    if (pre_mask)                                            // short K: the values no K-step fetched
        for (int idx = 2 * nk; idx < NPRE; ++idx) mbits |= (unsigned long long)(*mask_ptr(idx) > 0.f) << idx;

    store_rows_tile<MI, NJ>(acc, m0, n0, wm, wn, lane, m_beg, M, N, epi, aux, Cd, P, split, mbits, pre_mask);
#ifdef LBX_GEMM_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && g_gemm_stamps) {
        long long* o = g_gemm_stamps + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * (NT / 64) + wave) * 10;
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = LBX_T(); o[4] = t_mma; o[5] = t_ld; o[6] = t_st; o[7] = t_bar; o[8] = hwid; o[9] = nk;
    }
#endif
}

template <int BM, int BN, bool B_KINNER, bool ALIGNED>
__global__ LBX_ROWS_BOUNDS(BM, BN) void gemm_rows_kernel(RowsD A, const float* __restrict__ Bm, long ldb,
                                                        RowsOutD Cd, float* __restrict__ P, long m_beg, long M, int K,
                                                        int N, int epi, const float* __restrict__ aux,
                                                        int tiles_n, unsigned ntiles, int k_per_split) {
    gemm_rows_body<BM, BN, B_KINNER, ALIGNED, 4>(A, Bm, ldb, Cd, P, m_beg, M, K, N, epi, aux, tiles_n, ntiles, k_per_split);
}

template <int BM, int BN, bool B_KINNER, bool ALIGNED>
__global__ __launch_bounds__(512, (BM * BN >= 16384 ? 6 : 8)) void gemm_rows8_kernel(RowsD A, const float* __restrict__ Bm, long ldb,
                                                            RowsOutD Cd, float* __restrict__ P, long m_beg, long M, int K,
                                                            int N, int epi, const float* __restrict__ aux,
                                                            int tiles_n, unsigned ntiles, int k_per_split) {
    gemm_rows_body<BM, BN, B_KINNER, ALIGNED, 8>(A, Bm, ldb, Cd, P, m_beg, M, K, N, epi, aux, tiles_n, ntiles, k_per_split);
}

// ------------------------------------------------------------------------------------------------
// wgrad: P[split][K1][N] = A[Mslice, K1]^T . B[Mslice, N];  Pc[split][N] = column sums of B[Mslice]
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, bool ALIGNED>
__global__ LBX_TN_BOUNDS(BM, BN) void gemm_tn_kernel(RowsD A, RowsD Bd, float* __restrict__ P,
                                                      float* __restrict__ Pc, long M, int K1, int N,
                                                      int tiles_n, int ntiles, long rows_per_split) {
    constexpr int MI = BM / 64, NJ = BN / 64;
    using LAo = KOuterLoader<BM, ALIGNED>;
    using LBo = KOuterLoader<BN, ALIGNED>;
    __shared__ __attribute__((aligned(16))) float As[2][BK * LAo::LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LBo::LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // consecutive block ids = the tiles of one M slice: they read the same A/B rows (L2 reuse)
    const int tile = blockIdx.x % ntiles;
    const int split = blockIdx.x / ntiles;
    const int tn = tile % tiles_n, tk = tile / tiles_n;
    const int i0 = tk * BM, n0 = tn * BN;
    const long mbeg = (long)split * rows_per_split;
    long mend = mbeg + rows_per_split;
    if (mend > M) mend = M;

    LAo la;
    LBo lb;
    la.init(tid, i0, K1);
    lb.init(tid, n0, N);
    // per-pass offset of the contraction row this thread fetches; advanced by BK rows per step with
    // adds only: off += BK*rs, and "+= bs - rpb*rs" whenever t crosses into the next utterance
    long am[LAo::PASSES], bm_[LBo::PASSES], aoff[LAo::PASSES], boff[LBo::PASSES];
    unsigned at[LAo::PASSES], bt[LBo::PASSES];
    const long a_step = (long)BK * A.rs, b_step = (long)BK * Bd.rs;
    const long a_wrap = A.batch == 1 ? 0 : A.bs - (long)A.rpb * A.rs;
    const long b_wrap = Bd.batch == 1 ? 0 : Bd.bs - (long)Bd.rpb * Bd.rs;
    const unsigned a_rpb = A.batch == 1 ? 0xffffffffu : (unsigned)A.rpb;
    const unsigned b_rpb = Bd.batch == 1 ? 0xffffffffu : (unsigned)Bd.rpb;
#pragma unroll
    for (int p = 0; p < LAo::PASSES; ++p) {
        am[p] = mbeg + la.kk0 + LAo::RPP * p;
        const unsigned bq = A.batch == 1 ? 0u : (unsigned)am[p] / a_rpb;
        at[p] = (unsigned)am[p] - (A.batch == 1 ? 0u : bq * a_rpb);
        aoff[p] = (long)bq * A.bs + (long)at[p] * A.rs;
    }
#pragma unroll
    for (int p = 0; p < LBo::PASSES; ++p) {
        bm_[p] = mbeg + lb.kk0 + LBo::RPP * p;
        const unsigned bq = Bd.batch == 1 ? 0u : (unsigned)bm_[p] / b_rpb;
        bt[p] = (unsigned)bm_[p] - (Bd.batch == 1 ? 0u : bq * b_rpb);
        boff[p] = (long)bq * Bd.bs + (long)bt[p] * Bd.rs;
    }
    // CHECK = the tile may run past mend (only the last tile of the slice): row tests + zero fill
#define LBX_TN_FETCH(CHECK)                                                             \
    {                                                                                   \
        bool oka[LAo::PASSES], okb[LBo::PASSES];                                        \
        long offa[LAo::PASSES], offb[LBo::PASSES];                                      \
        _Pragma("unroll") for (int p = 0; p < LAo::PASSES; ++p) {                       \
            oka[p] = am[p] < mend;                                                      \
            offa[p] = (CHECK && !oka[p]) ? 0 : aoff[p];                                 \
            am[p] += BK;                                                                \
            at[p] += BK;                                                                \
            aoff[p] += a_step;                                                          \
            while (at[p] >= a_rpb) { at[p] -= a_rpb; aoff[p] += a_wrap; }               \
        }                                                                               \
        _Pragma("unroll") for (int p = 0; p < LBo::PASSES; ++p) {                       \
            okb[p] = bm_[p] < mend;                                                     \
            offb[p] = (CHECK && !okb[p]) ? 0 : boff[p];                                 \
            bm_[p] += BK;                                                               \
            bt[p] += BK;                                                                \
            boff[p] += b_step;                                                          \
            while (bt[p] >= b_rpb) { bt[p] -= b_rpb; boff[p] += b_wrap; }               \
        }                                                                               \
        la.template load_rows<CHECK>(A.base, offa, oka);                                \
        lb.template load_rows<CHECK>(Bd.base, offb, okb);                               \
    }

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float csum = 0.f;
    const bool do_csum = (Pc != nullptr) && tk == 0 && tid < BN;

    const int nk = (int)((mend - mbeg + BK - 1) / BK);
#ifdef LBX_GEMM_TIMING
    long long ts0 = LBX_T(), t_mma = 0, t_ld = 0, t_st = 0, t_bar = 0;
#endif
    if (nk > 0) {
        LBX_TN_FETCH(true)
        la.store(As[0]);
        lb.store(Bs[0]);
    }
    __syncthreads();
#ifdef LBX_GEMM_TIMING
    const long long ts1 = LBX_T();
#endif
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
#ifdef LBX_GEMM_TIMING
        __builtin_amdgcn_sched_barrier(0);
        const long long q0 = LBX_T();
#endif
        if (kt + 2 < nk) LBX_TN_FETCH(false)
        else if (kt + 1 < nk) LBX_TN_FETCH(true)
#ifdef LBX_GEMM_TIMING
        __builtin_amdgcn_sched_barrier(0);
        const long long q1 = LBX_T();
#endif
        mma_tile<MI, NJ, LAo::LD, LBo::LD>(As[cur], Bs[cur], wm, wn, lane, acc);
        if (do_csum) {
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) csum += Bs[cur][kk * LBo::LD + tid];
        }
#ifdef LBX_GEMM_TIMING
        __builtin_amdgcn_sched_barrier(0);
        const long long q2 = LBX_T();
#endif
        if (kt + 1 < nk) {
            la.store(As[cur ^ 1]);
            lb.store(Bs[cur ^ 1]);
        }
#ifdef LBX_GEMM_TIMING
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const long long q3 = LBX_T();
#endif
        __syncthreads();
#ifdef LBX_GEMM_TIMING
        __builtin_amdgcn_sched_barrier(0);
        const long long q4 = LBX_T();
        t_ld += q1 - q0; t_mma += q2 - q1; t_st += q3 - q2; t_bar += q4 - q3;
#endif
    }
#ifdef LBX_GEMM_TIMING
    const long long ts2 = LBX_T();
#endif
#undef LBX_TN_FETCH
    float* Pd = P + (long)split * K1 * N;
    const int h = lane >> 5, l = lane & 31;
#pragma unroll
    for (int bj = 0; bj < NJ; ++bj) {
        const int col = n0 + wn * (32 * NJ) + bj * 32 + l;
        if (col >= N) continue;
#pragma unroll
        for (int bi = 0; bi < MI; ++bi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * (32 * MI) + bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < K1) Pd[(long)row * N + col] = acc[bi][bj][r];
            }
    }
    if (do_csum && n0 + tid < N) Pc[(long)split * N + n0 + tid] = csum;
#ifdef LBX_GEMM_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && g_gemm_stamps) {
        long long* o = g_gemm_stamps + ((long)blockIdx.x * 4 + wave) * 10;
        o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = LBX_T(); o[4] = t_mma; o[5] = t_ld; o[6] = t_st; o[7] = t_bar; o[8] = 0; o[9] = nk;
    }
#endif
}

// column sums (standalone): stage 1 partial[rs][N] over row slices, stage 2 fixed-order reduce
__global__ __launch_bounds__(256) void colsum_stage1(RowsD A, long M, int N, long rows_per_slice,
                                                     float* __restrict__ partial) {
    __shared__ float red[256];
    const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + col;
    const long mbeg = (long)blockIdx.y * rows_per_slice;
    long mend = mbeg + rows_per_slice;
    if (mend > M) mend = M;
    float s = 0.f;
    if (c < N)
        for (long m = mbeg + g; m < mend; m += 4) s += A.base[row_offset(A, (unsigned)m) + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && c < N)
        partial[(long)blockIdx.y * N + c] = red[col] + red[col + 64] + red[col + 128] + red[col + 192];
}

__global__ void colsum_stage2(const float* __restrict__ partial, int slices, int N, float* __restrict__ out,
                              int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    float s = 0.f;
    for (int k = 0; k < slices; ++k) s += partial[(long)k * N + c];
    out[c] = accumulate ? out[c] + s : s;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// persistent stream-K kernels (gemm_sk.h): eligibility, plans, launches
// ------------------------------------------------------------------------------------------------
// LIDBOX_GEMM_SK=0 keeps every launch on the kernels of this file (A/B aid).  LIDBOX_GEMM_SK_GRID / _MIN_FLOP are test
// aids: a small persistent grid and no size floor let small problems exercise every branch of the stream-K schedule.
inline bool sk_enabled() {
    const char* e = getenv("LIDBOX_GEMM_SK");
    return !(e && atoi(e) == 0);
}
inline unsigned sk_grid() {
    if (const char* e = getenv("LIDBOX_GEMM_SK_GRID")) {
        const int g = atoi(e);
        if (g >= 1 && g <= NUM_CU * SK_WGCU) return (unsigned)g;
    }
    return NUM_CU * SK_WGCU;
}
inline double sk_min_flop() {
    if (const char* e = getenv("LIDBOX_GEMM_SK_MIN_FLOP")) return atof(e);
    return 3.0e9;                              // below this the slab traffic of 768 workgroups outweighs the balance
}
constexpr int SK_MIN_SPAN = 8;                 // K steps per workgroup
constexpr int SK_PART_STEPS = 16;              // K steps of one part of a remainder tile

// launch epoch of the arrival counters (gemm_sk.h): 1 .. 2^24 - 1
inline unsigned sk_next_epoch() {
    static std::atomic<unsigned> e{0};
    unsigned v;
    do v = e.fetch_add(1u) + 1u; while ((v & 0xffffffu) == 0u);
    return v & 0xffffffu;
}

struct SkRows {
    bool ok = false;
    SkPlan pl{};
    unsigned P = 0;
    size_t ws_need = 0;
};

// Where the stream-K rows kernel is the default (measured per x-vector / CNN layer at bs 256, profiles/r03_gemm_sk_ab.txt):
// forward GEMMs with a long contraction.  With K = 512 a tile's epilogue is a fifth of its life and the whole-tile rounds
// of a persistent grid run their epilogues in phase, and the dgrad epilogues (ReLU mask / accumulate: dependent loads of
// the old values at the end of a tile) sit on the critical path of the LAST contributor of a streamed tile -- the classic
// kernels, eight small tiles per CU at different phases, hide both better.  LIDBOX_GEMM_SK_ALL=1 lifts the policy (tests,
// A/B runs).
inline bool sk_all() {
    const char* e = getenv("LIDBOX_GEMM_SK_ALL");
    return e && atoi(e) != 0;
}
inline bool sk_rows_policy(int kind, int K) { return sk_all() || (kind == 0 && K >= 1024); }
// a measured classic decomposition of the shape (gemm_tuned.h, swept with the stream-K kernels as "model") wins over stream-K
inline bool sk_tuned_out(int kind, long M, int N, int K) { return !sk_all() && tuned_gemm(kind, M, N, K) != nullptr; }

// pipelined-epilogue variant (gemm_skp_rows_kernel): 0 off, 1 where the policy says, 2 every eligible launch (A/B aid).
// MEASURED AND NOT ADOPTED (profiles/r03_gemm_skp_ab.txt, bs 256): correct through every schedule branch
// (LIDBOX_GEMM_SKP=2 pytest tests/test_gemm_sk_gpu.py) but slower than the classic kernels on every x-vector layer --
// frame2 dgrad0 289 vs 241 us, frame5 fwd 145 vs 122, frame2 fwd 346 vs 321 (306 on the plain stream-K kernel): the second
// accumulator set costs the third resident workgroup, and eight stores + the staging traffic per K step between the
// MFMA groups cost the wave more issue time than the epilogue they hide.  The policy below therefore selects nothing.
inline int skp_mode() {
    if (const char* e = getenv("LIDBOX_GEMM_SKP")) return atoi(e);
    return 1;
}

// where the pipelined variant is the default (measured, profiles/r03_gemm_skp_ab.txt); LIDBOX_GEMM_SKP=2: everywhere
inline bool skp_policy(int kind, long M, int N, int K) {
    (void)kind; (void)M; (void)N; (void)K;
    return skp_mode() == 2;
}

SkRows sk_rows_plan(int kind, long M, int N, int K, bool pipelined = false) {
    SkRows c;
    if (!sk_enabled() || M < SK_BM || K % 4 != 0 || 2.0 * (double)M * N * K < sk_min_flop()) return c;
    if (!pipelined && !sk_rows_policy(kind, K)) return c;
    unsigned P = sk_grid();
    if (pipelined && P == (unsigned)(NUM_CU * SK_WGCU)) P = NUM_CU * SKP_WGCU;
    if (P % 8 != 0) return c;
    SkPlan& pl = c.pl;
    pl.tiles_n = (int)lbx_cdiv(N, SK_BN);
    const long T = lbx_cdiv(M, SK_BM) * pl.tiles_n;
    if (T > 0x3fffffffL) return c;
    pl.ntiles = (int)T;
    pl.nk = (int)lbx_cdiv(K, SK_BK);
    pl.ktail = K - (pl.nk - 1) * SK_BK;
    // Whole rounds of tiles go to the workgroups one tile each.  What remains:
    //   * after >= 1 whole round: each remaining tile is cut into g parts of >= SK_PART_STEPS K steps for g workgroups of one
    //     XCD (workgroups of a CU share its matrix pipes, so a few longer workgroups cost only the efficiency of the tail,
    //     while every slab costs 64 KB each way);
    //   * fewer tiles than workgroups: every workgroup takes the same number of K steps (stream-K proper).
    pl.dp_rounds = (int)(T / P);
    pl.sk_first = pl.dp_rounds * (int)P;
    pl.sk_tiles = (int)T - pl.sk_first;
    pl.parts = 0;
    if (pl.dp_rounds > 0 && pl.sk_tiles > 0) {
        long g = pl.nk / SK_PART_STEPS;
        if (g > (long)(P / 8) / lbx_cdiv(pl.sk_tiles, 8)) g = (long)(P / 8) / lbx_cdiv(pl.sk_tiles, 8);
        if (g < 1) g = 1;
        if (g > 200) g = 200;
        // K steps of the busiest CU (its resident workgroups share the matrix pipes): parts are dealt one per workgroup, so a
        // big remainder with g = 1 is a half-empty extra round (792 tiles on 512 workgroups: +29 %); equal spans over the
        // remainder plus one whole round balance exactly and cost two slabs per workgroup (charged 5 %)
        const long wg_cu = P / NUM_CU > 0 ? P / NUM_CU : 1;
        const double cu_parts = (double)pl.dp_rounds * pl.nk * wg_cu + (double)lbx_cdiv(pl.sk_tiles * g, NUM_CU) * ((double)pl.nk / g);
        const double cu_spans = 1.05 * (double)T * pl.nk / NUM_CU;
        if (cu_parts <= cu_spans) {
            pl.parts = (int)g;
        } else {
            --pl.dp_rounds;
            pl.sk_first = pl.dp_rounds * (int)P;
            pl.sk_tiles = (int)T - pl.sk_first;
        }
    }
    if (pl.parts == 0 && pl.sk_tiles > 0 && (long)pl.sk_tiles * pl.nk < (long)P * SK_MIN_SPAN) return c;
    if (pl.parts == 0 && pl.sk_tiles > 0 && (long)pl.nk * P / ((long)pl.sk_tiles * pl.nk) + 2 > 250) return c;      // arrivals of a tile fit 8 bits
    if ((size_t)pl.sk_tiles * sizeof(unsigned) > SK_COUNTER_BYTES) return c;
    c.P = P;
    c.ws_need = SK_COUNTER_BYTES + (size_t)P * 2 * SK_SLAB * sizeof(float) + (pipelined ? SKP_DUMMY_BYTES : 0);
    c.ok = true;
    return c;
}

// every lane offset of the saddr loads must fit 32 bits
inline bool sk_extent_ok(const lidbox_rows_t& r, long cols) {
    const double last = (double)(r.batch > 0 ? r.batch - 1 : 0) * r.batch_stride + (double)(r.rows_per_batch > 0 ? r.rows_per_batch - 1 : 0) * r.row_stride + cols + 64;
    return last * 4.0 < 4.0e9;
}

// the B operand's 32-bit lane offsets: B[N][K] (nt) spans N rows of ldb floats, B[K][N] (nn) K rows (+ the 16-row lane term)
inline bool sk_b_extent_ok(bool b_kinner, int K, int N, long ldb) {
    return ((double)(b_kinner ? N : K) + 16.0) * (double)ldb * 4.0 < 4.0e9;
}

void sk_set_lds_attr() {
    static const bool done = [] {
        (void)hipFuncSetAttribute((const void*)gemm_sk_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SK_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_sk_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SK_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_sk_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SK_LDS_BYTES);
#define LBX_SKP_ATTR(B_, M_, A_) (void)hipFuncSetAttribute((const void*)gemm_skp_rows_kernel<B_, M_, A_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SKP_LDS_BYTES)
        LBX_SKP_ATTR(false, false, false); LBX_SKP_ATTR(false, true, false); LBX_SKP_ATTR(false, false, true); LBX_SKP_ATTR(false, true, true);
        LBX_SKP_ATTR(true, false, false); LBX_SKP_ATTR(true, true, false); LBX_SKP_ATTR(true, false, true); LBX_SKP_ATTR(true, true, true);
#undef LBX_SKP_ATTR
        return true;
    }();
    (void)done;
}

struct SkTn {
    bool ok = false;
    int splits = 0, tiles_n = 0, ntiles = 0;
    long rows_per_split = 0;
    size_t ws_need = 0;
};

SkTn sk_tn_plan(long M, int K1, int N) {
    SkTn c;
    // (frame4's wgrad, 4.4 GFLOP at bs 256, measured 55 vs 52 us on the classic kernel: the floor is twice the rows kernels')
    if (!sk_enabled() || K1 % 4 != 0 || N % 4 != 0 || 2.0 * (double)M * N * K1 < 2.0 * sk_min_flop()) return c;
    c.tiles_n = (int)lbx_cdiv(N, SK_BN);
    c.ntiles = (int)lbx_cdiv(K1, SK_BM) * c.tiles_n;
    long g = (long)sk_grid() / c.ntiles;
    if (g < 1) g = 1;
    long rps = lbx_cdiv(lbx_cdiv(M, g), SK_BK) * SK_BK;
    if (rps < 8 * SK_BK) return c;
    // one fp32 accumulation chain per slice: keep it short enough that the 8-shard == global gradient identity of
    // tests/test_fullsize_gpu.py (rel 1e-5 at 2048 utterances on one GPU) holds; bs <= 512 per GPU never gets here
    if (rps > 8192) rps = 8192;
    c.rows_per_split = rps;
    c.splits = (int)lbx_cdiv(M, rps);
    c.ws_need = ((size_t)c.splits * K1 * N + (size_t)c.splits * N) * sizeof(float);
    c.ok = true;
    return c;
}

struct RowsChoice {
    int bm, bn, splits, k_per_split;
    bool no_tail_split = false;
    int waves = 4;                 // 8: gemm_rows8_kernel (128-row tiles only)
};

// what the calling thread's most recent lidbox_gemm_nn / _nt / _tn call launched (lidbox_gemm_last_launches):
// {kernels of the instantiation plan_query names, GEMM kernels of another instantiation (a tail-split remainder planned
// on its own), reduce kernels}
thread_local int g_last_launches[3] = {0, 0, 0};
thread_local int g_last_family = 0;       // 0 register-staged kernels of this file, 1 LDS-DMA (gemm_dma.h), 2 stream-K (gemm_sk.h)
thread_local int g_first_tile[2] = {0, 0};
thread_local int g_last_carried = 0;      // whether the most recent lidbox_gemm_nt_carry ran its job inside the GEMM launch

// Cost model in "K-steps of a 128x128 tile at the full fp32 MFMA rate" (~1 us each per CU).
// A CU's resident workgroups share its four matrix pipes, so a CU's time is the SUM of its
// tiles' MFMA time; co-resident workgroups hide each other's barrier / staging stalls, a lone
// workgroup cannot (measured: 1 WG/CU keeps the pipe ~50 % busy, 3+ WG/CU ~90 %).
const int CAND[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
// steady state of the LDS-DMA kernels (gemm_dma.h; profiles/r03_dma_staircase.txt): 137 / 134 / 133 / 131 TFLOP/s
const double TILE_EFF[4] = {1.0, 0.98, 0.97, 0.955};
const int RESIDENT[4] = {BK == 16 ? 3 : 2, BK == 16 ? 4 : 3, BK == 16 ? 4 : 3, BK == 16 ? 6 : 4};   // workgroups per CU (VGPR / LDS limited)
const double FIXED_STEPS[4] = {6.0, 5.0, 5.0, 3.0};       // prologue + epilogue of one tile, in K-steps (fewer residents hide less of it)

inline double conc_eff(double w) {
    if (w <= 1.0) return 0.55;
    if (w <= 2.0) return 0.55 + 0.25 * (w - 1.0);
    if (w <= 3.0) return 0.80 + 0.10 * (w - 2.0);
    return 0.92;
}

inline double launch_cost(int c, long wgs, double ksteps) {
    const int bm = CAND[c][0], bn = CAND[c][1];
    const double tile_t = (bm * bn / 16384.0) * (ksteps + FIXED_STEPS[c]) / TILE_EFF[c];
    // one tile per workgroup costs ceil(tiles / CUs) tile times; past one whole round the remainder is streamed along K
    // (DmaStream) and costs about its share
    double per_cu = (double)lbx_cdiv(wgs, NUM_CU);
    if (wgs > NUM_CU && wgs % NUM_CU != 0) {
        const double fluid = (double)wgs / NUM_CU + 0.12;
        if (fluid < per_cu) per_cu = fluid;
    }
    const double conc = per_cu < RESIDENT[c] ? (double)wgs / NUM_CU : (double)RESIDENT[c];
    return per_cu * tile_t / conc_eff(conc < 1.0 ? 1.0 : conc);
}

// kind: 0 nn, 1 nt.  Measured decompositions (gemm_tuned.h) come first, the cost model covers every other shape.
RowsChoice choose_rows(int kind, long M, int N, int K, size_t ws_bytes) {
    if (const TunedGemm* t = tuned_gemm(kind, M, N, K)) {
        const int kps = (int)(lbx_cdiv(lbx_cdiv(K, t->splits), BK) * BK);
        const int splits = (int)lbx_cdiv(K, kps);
        if (!(BK > 16 && t->bm == 128 && t->bn == 128) &&
            (splits == 1 || (size_t)splits * M * N * sizeof(float) <= ws_bytes))
            return RowsChoice{t->bm, t->bn, splits, kps, t->no_tail_split != 0, (t->waves == 8 && t->bm == 128 && BK == 16) ? 8 : 4};
    }
    RowsChoice best{128, 128, 1, K};
    double best_cost = 1e30;
    for (int c = (BK > 16 ? 1 : 0); c < 4; ++c) {
        const int bm = CAND[c][0], bn = CAND[c][1];
        const long tiles = lbx_cdiv(M, bm) * lbx_cdiv(N, bn);
        for (int s = 1; s <= 64; s *= 2) {
            int kps = (int)(lbx_cdiv(lbx_cdiv(K, s), BK) * BK);
            const int splits = (int)lbx_cdiv(K, kps);
            if (s > 1 && (splits < 2 || kps < 4 * BK)) break;
            if (splits > 1 && (size_t)splits * M * N * sizeof(float) > ws_bytes) break;
            double cost = launch_cost(c, tiles * splits, (double)kps / BK);
            if (splits > 1) cost += 5.0 + (double)M * N * 4.0 * (splits + 1) / 3.0e6;   // reduce pass
            if (cost < best_cost) { best_cost = cost; best = RowsChoice{bm, bn, splits, kps}; }
            if (splits < s) break;
        }
    }
    return best;
}

// the decomposition a launch uses: the planner's, or the tuning overrides LIDBOX_GEMM_PLAN / LIDBOX_GEMM_TILE
RowsChoice choose_rows_env(int kind, long M, int N, int K, size_t wsb) {
    RowsChoice ch = choose_rows(kind, M, N, K, wsb);
    if (M >= 4096) {                  // A/B aids: conv-size dgrads (LIDBOX_GEMM_NT8) / forward GEMMs (LIDBOX_GEMM_NN8) on the eight-wave 128 x bn tiles
        if (const char* f = getenv(kind == 1 ? "LIDBOX_GEMM_NT8" : "LIDBOX_GEMM_NN8")) {
            const int bn = atoi(f);
            if (bn == 64 || bn == 128) return RowsChoice{128, bn, 1, K, false, 8};
        }
    }
    if (const char* f = getenv("LIDBOX_GEMM_PLAN")) {             // tuning aid (tools/gemm_sweep.py): "bm,bn,splits"
        int bm = 0, bn = 0, sp = 0, wv = 4;
        if (sscanf(f, "%d,%d,%d,%d", &bm, &bn, &sp, &wv) >= 3 && (bm == 64 || bm == 128) && (bn == 64 || bn == 128) && sp >= 1 &&
            !(BK > 16 && bm == 128 && bn == 128)) {
            const int kps = (int)(lbx_cdiv(lbx_cdiv(K, sp), BK) * BK);
            const int splits = (int)lbx_cdiv(K, kps);
            if (splits == 1 || (size_t)splits * M * N * sizeof(float) <= wsb)
                ch = RowsChoice{bm, bn, splits, kps, false, (wv == 8 && bm == 128 && BK == 16) ? 8 : 4};
        }
    } else if (const char* f = getenv("LIDBOX_GEMM_TILE")) {      // tuning aid: "128x128" etc.
        int bm = 0, bn = 0;
        if (sscanf(f, "%dx%d", &bm, &bn) == 2 && (bm == 64 || bm == 128) && (bn == 64 || bn == 128)) {
            ch.bm = bm; ch.bn = bn; ch.splits = 1; ch.k_per_split = K; ch.no_tail_split = false;
            if (BK > 16 && bm == 128 && bn == 128) ch.bn = 64;
        }
    }
    return ch;
}

template <int BM, int BN, bool B_KINNER>
void launch_rows8_t(bool al, dim3 grid, hipStream_t st, RowsD Ad, const float* Bm, long ldb, RowsOutD Co, float* P,
                    long m_beg, long m_end, int K, int N, int epi, const float* aux, int tiles_n, unsigned ntiles, int kps) {
    if (al)
        hipLaunchKernelGGL((gemm_rows8_kernel<BM, BN, B_KINNER, true>), grid, dim3(512), 0, st, Ad, Bm, ldb, Co, P, m_beg,
                           m_end, K, N, epi, aux, tiles_n, ntiles, kps);
    else
        hipLaunchKernelGGL((gemm_rows8_kernel<BM, BN, B_KINNER, false>), grid, dim3(512), 0, st, Ad, Bm, ldb, Co, P, m_beg,
                           m_end, K, N, epi, aux, tiles_n, ntiles, kps);
}

template <int BM, int BN, bool B_KINNER>
void launch_rows_t(bool al, dim3 grid, hipStream_t st, RowsD Ad, const float* Bm, long ldb, RowsOutD Co, float* P,
                   long m_beg, long m_end, int K, int N, int epi, const float* aux, int tiles_n, unsigned ntiles, int kps) {
    if (al)
        hipLaunchKernelGGL((gemm_rows_kernel<BM, BN, B_KINNER, true>), grid, dim3(256), 0, st, Ad, Bm, ldb, Co, P, m_beg,
                           m_end, K, N, epi, aux, tiles_n, ntiles, kps);
    else
        hipLaunchKernelGGL((gemm_rows_kernel<BM, BN, B_KINNER, false>), grid, dim3(256), 0, st, Ad, Bm, ldb, Co, P, m_beg,
                           m_end, K, N, epi, aux, tiles_n, ntiles, kps);
}

// The classic decompositions run on the LDS-DMA operand path (gemm_dma.h) wherever the operands allow (16-byte aligned, 32-bit
// extents); LIDBOX_GEMM_DMA=0 keeps the register-staged kernels of this file (A/B aid, and what unaligned problems use).
inline int dma_mode() {
    if (const char* e = getenv("LIDBOX_GEMM_DMA")) return atoi(e);
    return 1;
}

template <int BM, int BN, bool B_KINNER>
void launch_rows_dma_t(dim3 grid, hipStream_t st, RowsD Ad, const float* Bm, long ldb, RowsOutD Co, float* P, long m_beg, long m_end,
                       int K, int N, int epi, const float* aux, int tiles_n, unsigned ntiles, int kps, const DmaStream& sp, const ReduceJobs& rj) {
    hipLaunchKernelGGL((gemm_rows_dma_kernel<BM, BN, B_KINNER>), grid, dim3(256), 0, st, Ad, Bm, ldb, Co, P, m_beg, m_end, K, N, epi,
                       aux, tiles_n, ntiles, kps, sp, rj);
}

// workgroups the carried reduces of a GEMM launch share (pack_carry).  Measured per launch at bs 256 (profiles/
// r04_carry_blocks.txt): 64 .. 160 workgroups hide the sums beside the tiles (-6 .. -10 us per layer against reduce launches),
// 256 and more delay the tiles by most of what the reduce takes -- their loads starve the tiles' operand DMA.
// LIDBOX_GEMM_CARRY_BLOCKS overrides (A/B aid).
inline long carry_cap() {
    if (const char* e = getenv("LIDBOX_GEMM_CARRY_BLOCKS")) { const long v = atol(e); if (v >= 8) return v; }
    return 96;
}
struct Carry {
    const ReduceJob* jobs = nullptr;
    int njobs = 0;
    bool carried = false;           // set by the launch that took the jobs
};

// Streamed remainder of an unsplit decomposition (gemm_dma.h: DmaStream): g pieces per tile of the last partial round,
// chosen for the fewest rounds of pieces (in tile times), pieces of at least DMA_STREAM_MIN_STEPS K steps.
// LIDBOX_GEMM_STREAM_TAIL=0 switches it off (A/B aid), =g forces g pieces.
constexpr int DMA_STREAM_MIN_STEPS = 8;
constexpr int DMA_STREAM_MAX_G = 8;
struct DmaStreamPlan {
    int g = 0;                  // 0: no streamed remainder
    long whole = 0, rem = 0;    // whole tiles, streamed tiles
    size_t ws_need = 0;
};
inline DmaStreamPlan dma_stream_plan(int bm, int bn, long M, int N, int K) {
    DmaStreamPlan pl;
    int force = -1;
    if (const char* e = getenv("LIDBOX_GEMM_STREAM_TAIL")) force = atoi(e);
    if (force == 0) return pl;
    const long tiles = lbx_cdiv(M, bm) * lbx_cdiv(N, bn);
    const long rem = tiles % NUM_CU;
    if (tiles < NUM_CU || rem == 0) return pl;
    const int nk = (int)lbx_cdiv(K, SK_BK);
    int best = 1;
    double best_cost = 1.0;                                   // unstreamed: the remainder costs one tile time
    for (int g = 2; g <= DMA_STREAM_MAX_G && nk / g >= DMA_STREAM_MIN_STEPS; ++g) {
        // rounds of pieces, each 1 / g of a tile time, plus the hand-off (slab write, the last arriver reads g slabs)
        const double cost = (double)lbx_cdiv(rem * g, (long)NUM_CU) / g + 0.03 + 0.01 * g;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = g; }
    }
    if (force > 1 && nk / force >= 2 && force <= 64) best = force;
    else if (best_cost > 0.9) return pl;
    if (best < 2) return pl;
    pl.g = best; pl.rem = rem; pl.whole = tiles - rem;
    pl.ws_need = SK_COUNTER_BYTES + (size_t)rem * best * bm * bn * sizeof(float);
    return pl;
}

// one launch (+ its split-K reduce) over the row range [m_beg, m_end) with a given decomposition
template <bool B_KINNER>
int launch_rows_range(const RowsChoice& ch, bool al, RowsD Ad, const float* Bm, long ldb, RowsOutD Co, long m_beg,
                      long m_end, int K, int N, int epi, const float* aux, float* P, hipStream_t st, bool dma_ok = false,
                      const DmaStreamPlan* stream = nullptr, void* ws = nullptr, Carry* carry = nullptr) {
    const long Msub = m_end - m_beg;
    if (Msub <= 0) return LIDBOX_OK;
    const int tiles_n = (int)lbx_cdiv(N, ch.bn);
    const long ntiles = lbx_cdiv(Msub, ch.bm) * tiles_n;
    dim3 grid((unsigned)ntiles, (unsigned)ch.splits);
    if (g_last_launches[0] == 0) { g_first_tile[0] = ch.bm; g_first_tile[1] = ch.bn; }
    ++g_last_launches[(ch.bm == g_first_tile[0] && ch.bn == g_first_tile[1]) ? 0 : 1];
    if (ch.splits > 1) ++g_last_launches[2];
#define LBX_ROWS(BM_, BN_) launch_rows_t<BM_, BN_, B_KINNER>(al, grid, st, Ad, Bm, ldb, Co, P, m_beg, m_end, K, N, epi, aux, tiles_n, (unsigned)ntiles, ch.k_per_split)
#define LBX_ROWS8(BM_, BN_) launch_rows8_t<BM_, BN_, B_KINNER>(al, grid, st, Ad, Bm, ldb, Co, P, m_beg, m_end, K, N, epi, aux, tiles_n, (unsigned)ntiles, ch.k_per_split)
#define LBX_ROWS_DMA(BM_, BN_) launch_rows_dma_t<BM_, BN_, B_KINNER>(grid, st, Ad, Bm, ldb, Co, P, m_beg, m_end, K, N, epi, aux, tiles_n, ntiles_k, ch.k_per_split, sp, rj)
    if (dma_ok && al && dma_mode() != 0) {
        g_last_family = 1;
        DmaStream sp;
        // eight-wave tiles (gemm_dma8.h): nt launches planned as 128 x 64 / 128 x 128 with waves = 8
        const bool dma8 = ch.waves == 8 && ch.bm == 128;
        ReduceJobs rj;                              // total = 0: nothing carried
        if (carry && !carry->carried && carry->njobs > 0) {
            rj = pack_carry(carry->jobs, carry->njobs, carry_cap());
            carry->carried = rj.total > 0;
        }
        unsigned ntiles_k = (unsigned)ntiles;
        if (stream && stream->g > 1 && ch.splits == 1) {
            sp.pieces = (unsigned)(stream->rem * stream->g);
            sp.npad = (sp.pieces + 7u) & ~7u;
            sp.g = stream->g;
            sp.first_tile = (unsigned)stream->whole;
            sp.epoch = sk_next_epoch();
            sp.counters = (unsigned*)ws;
            sp.slabs = (float*)((char*)ws + SK_COUNTER_BYTES);
            ntiles_k = (unsigned)stream->whole;
            grid.x = sp.npad + ntiles_k;
        }
        grid.x += rj.total;
        if (dma8) {
            g_last_family = 3;
            if (ch.bn == 128)
                hipLaunchKernelGGL((gemm_rows_dma8_kernel<128, B_KINNER>), grid, dim3(512), 0, st, Ad, Bm, ldb, Co, P, m_beg, m_end, K, N, epi, aux, tiles_n,
                                   ntiles_k, ch.k_per_split, sp, rj);
            else
                hipLaunchKernelGGL((gemm_rows_dma8_kernel<64, B_KINNER>), grid, dim3(512), 0, st, Ad, Bm, ldb, Co, P, m_beg, m_end, K, N, epi, aux, tiles_n,
                                   ntiles_k, ch.k_per_split, sp, rj);
        } else
        if (ch.bm == 128 && ch.bn == 128) LBX_ROWS_DMA(128, 128);
        else if (ch.bm == 128) LBX_ROWS_DMA(128, 64);
        else if (ch.bn == 128) LBX_ROWS_DMA(64, 128);
        else LBX_ROWS_DMA(64, 64);
    } else
#if LBX_GEMM_BK == 16
    if (ch.waves == 8 && ch.bm == 128 && ch.bn == 128) LBX_ROWS8(128, 128);
    else if (ch.waves == 8 && ch.bm == 128 && ch.bn == 64) LBX_ROWS8(128, 64);
    else if (ch.bm == 128 && ch.bn == 128) LBX_ROWS(128, 128);
    else
#endif
    if (ch.bm == 128) LBX_ROWS(128, 64);
    else if (ch.bn == 128) LBX_ROWS(64, 128);
    else LBX_ROWS(64, 64);
#undef LBX_ROWS
#undef LBX_ROWS8
    LBX_LAUNCH_OK();
    if (ch.splits > 1) {
        long g = lbx_cdiv(Msub * N, 256);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(rows_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)P, ch.splits, m_beg,
                           Msub, N, Co, epi, aux);
        LBX_LAUNCH_OK();
    }
    return LIDBOX_OK;
}

inline int cand_index(int bm, int bn) { return bm == 128 ? (bn == 128 ? 0 : 1) : (bn == 128 ? 2 : 3); }

template <bool B_KINNER>
int launch_rows(lidbox_rows_t A, const float* Bm, long ldb, lidbox_rows_out_t Cd, int K, int N, int epi,
                const float* aux, void* ws, size_t ws_bytes, hipStream_t st, Carry* carry = nullptr) {
    const long M = (long)A.batch * A.rows_per_batch;
    g_last_launches[0] = g_last_launches[1] = g_last_launches[2] = 0;
    g_last_family = 0;
    if (M == 0 || N == 0) return LIDBOX_OK;
    const size_t wsb = ws ? ws_bytes : 0;
    const RowsChoice ch = choose_rows_env(B_KINNER ? 1 : 0, M, N, K, wsb);
    // float4 paths: A rows and K multiple of 4; B: NN needs N%4 (columns), NT needs K%4 (rows)
    const bool al = rows_aligned(A) && K % 4 == 0 && aligned16(Bm) && ldb % 4 == 0 &&
                    (B_KINNER ? true : N % 4 == 0);
    RowsD Ad = to_dev(A);
    RowsOutD Co{Cd.base, Cd.batch_stride, Cd.row_stride, Cd.batch, Cd.rows_per_batch};
    float* P = (float*)ws;

    // persistent stream-K kernel (gemm_sk.h): aligned problems that fill the chip, workspace permitting
    if (al && !getenv("LIDBOX_GEMM_PLAN") && !getenv("LIDBOX_GEMM_TILE") && aligned16(ws) && !sk_tuned_out(B_KINNER ? 1 : 0, M, N, K) &&
        sk_b_extent_ok(B_KINNER, K, N, ldb)) {
        // the pipelined variant: its in-loop epilogue addresses C rows with at most one utterance wrap per 32-row block
        // (its drain stages the mask / old values of C through LDS-DMA: 16-byte aligned C rows, whole 16-byte column chunks)
        const bool has_mask_ = epi == LIDBOX_EPI_RELU_MASK || epi == LIDBOX_EPI_ACCUM_RELU_MASK;
        const bool skp_ok = skp_mode() != 0 && (Cd.batch == 1 || Cd.rows_per_batch >= 32) && N % 4 == 0 && aligned16(Cd.base) &&
                            Cd.row_stride % 4 == 0 && (Cd.batch == 1 || Cd.batch_stride % 4 == 0) && (!has_mask_ || aligned16(aux)) &&
                            skp_policy(B_KINNER ? 1 : 0, M, N, K);
        if (skp_ok) {
            const SkRows sk = sk_rows_plan(B_KINNER ? 1 : 0, M, N, K, true);
            if (sk.ok && wsb >= sk.ws_need && sk_extent_ok(A, K)) {
                sk_set_lds_attr();
                g_last_launches[0] = 1;
                g_last_family = 2;
                g_first_tile[0] = SK_BM; g_first_tile[1] = SK_BN;
                unsigned* counters = (unsigned*)ws;
                float* slabs = (float*)((char*)ws + SK_COUNTER_BYTES);
                float* dummy = slabs + (size_t)sk.P * 2 * SK_SLAB;
                const bool has_mask = epi == LIDBOX_EPI_RELU_MASK || epi == LIDBOX_EPI_ACCUM_RELU_MASK;
                const bool accum = epi == LIDBOX_EPI_ACCUM || epi == LIDBOX_EPI_ACCUM_RELU_MASK || epi == LIDBOX_EPI_ACCUM_RELU;
                const unsigned ep = sk_next_epoch();
#define LBX_SKP(MASK_, ACC_)                                                                                                \
    hipLaunchKernelGGL((gemm_skp_rows_kernel<B_KINNER, MASK_, ACC_>), dim3(sk.P), dim3(256), (MASK_ || ACC_) ? SKP_LDS_BYTES : SK_LDS_BYTES, st, Ad, Bm, ldb, Co, M, K, \
                       N, epi, aux, sk.pl, ep, counters, slabs, dummy)
                if (has_mask && accum) LBX_SKP(true, true);
                else if (has_mask) LBX_SKP(true, false);
                else if (accum) LBX_SKP(false, true);
                else LBX_SKP(false, false);
#undef LBX_SKP
                LBX_LAUNCH_OK();
                return LIDBOX_OK;
            }
        }
        const SkRows sk = sk_rows_plan(B_KINNER ? 1 : 0, M, N, K);
        if (sk.ok && wsb >= sk.ws_need && sk_extent_ok(A, K)) {
            sk_set_lds_attr();
            g_last_launches[0] = 1;
            g_last_family = 2;
            g_first_tile[0] = SK_BM; g_first_tile[1] = SK_BN;
            unsigned* counters = (unsigned*)ws;
            float* slabs = (float*)((char*)ws + SK_COUNTER_BYTES);
            hipLaunchKernelGGL((gemm_sk_rows_kernel<B_KINNER>), dim3(sk.P), dim3(256), SK_LDS_BYTES, st, Ad, Bm, ldb, Co, M, K, N, epi,
                               aux, sk.pl, sk_next_epoch(), counters, slabs);
            LBX_LAUNCH_OK();
            return LIDBOX_OK;
        }
    }

    const bool dma_ok = al && sk_extent_ok(A, K) && sk_b_extent_ok(B_KINNER, K, N, ldb);
    // Tail quantisation: with W workgroups on 256 CUs the last partial round runs at the pace of a
    // full one (measured: the last 3 % of frame2's rows cost 21 % of its time).  When the main
    // decomposition is unsplit and leaves such a tail, launch it over the largest row prefix whose
    // workgroup count is a whole number of rounds, and hand the few remaining rows to a second
    // launch planned on its own (small tiles, split along K) -- if the cost model agrees.
    if (dma_ok && dma_mode() != 0 && ch.splits == 1 && aligned16(ws)) {
        const DmaStreamPlan spl = dma_stream_plan(ch.bm, ch.bn, M, N, K);
        if (spl.g > 1 && wsb >= spl.ws_need)
            return launch_rows_range<B_KINNER>(ch, al, Ad, Bm, ldb, Co, 0, M, K, N, epi, aux, P, st, dma_ok, &spl, ws, carry);
    }
    const bool no_tail_split = ch.no_tail_split || getenv("LIDBOX_GEMM_NO_TAIL_SPLIT") != nullptr;
    if (!no_tail_split && ch.splits == 1) {
        const int tiles_n = (int)lbx_cdiv(N, ch.bn);
        const long tiles_m = lbx_cdiv(M, ch.bm);
        const long wgs = tiles_m * tiles_n;
        const long rounds = wgs / NUM_CU;                       // whole rounds
        if (rounds >= 1 && wgs % NUM_CU != 0) {
            const long main_tiles_m = rounds * NUM_CU / tiles_n;  // row tiles of the prefix (rounded down)
            const long m_main = main_tiles_m * ch.bm;
            const long m_rem = M - m_main;
            if (main_tiles_m >= 1 && m_rem > 0) {
                const RowsChoice rem = choose_rows(B_KINNER ? 1 : 0, m_rem, N, K, wsb);
                const int c = cand_index(ch.bm, ch.bn), cr = cand_index(rem.bm, rem.bn);
                const double ksteps = (double)lbx_cdiv(K, BK);
                const double whole = launch_cost(c, wgs, ksteps);
                double split = launch_cost(c, main_tiles_m * tiles_n, ksteps) + 4.0 +
                               launch_cost(cr, lbx_cdiv(m_rem, rem.bm) * lbx_cdiv(N, rem.bn) * rem.splits,
                                           (double)rem.k_per_split / BK);
                if (rem.splits > 1) split += 5.0 + (double)m_rem * N * 4.0 * (rem.splits + 1) / 3.0e6;
                if (split < 0.97 * whole) {
                    int rc = launch_rows_range<B_KINNER>(ch, al, Ad, Bm, ldb, Co, 0, m_main, K, N, epi, aux, P, st, dma_ok, nullptr, nullptr, carry);
                    if (rc) return rc;
                    return launch_rows_range<B_KINNER>(rem, al, Ad, Bm, ldb, Co, m_main, M, K, N, epi, aux, P, st, dma_ok);
                }
            }
        }
    }
    return launch_rows_range<B_KINNER>(ch, al, Ad, Bm, ldb, Co, 0, M, K, N, epi, aux, P, st, dma_ok, nullptr, nullptr, carry);
}

struct TnPlan {
    int bm, bn, splits;
    long rows_per_split;
};

TnPlan tn_plan(long M, int K1, int N) {
    if (const char* f = getenv("LIDBOX_GEMM_TN_PLAN")) {          // tuning aid (tools/gemm_sweep.py): "bm,bn,splits"
        int bm = 0, bn = 0;
        long sp = 0;
        if (sscanf(f, "%d,%d,%ld", &bm, &bn, &sp) == 3 && (bm == 64 || bm == 128) && (bn == 64 || bn == 128) && sp >= 1 &&
            !(BK > 16 && bm == 128 && bn == 128)) {
            const long rps = lbx_cdiv(lbx_cdiv(M, sp), BK) * BK;
            return TnPlan{bm, bn, (int)lbx_cdiv(M, rps), rps};
        }
    }
    if (const TunedGemm* t = tuned_gemm(2, M, N, K1)) {
        if (!(BK > 16 && t->bm == 128 && t->bn == 128)) {
            const long rps = lbx_cdiv(lbx_cdiv(M, (long)t->splits), BK) * BK;
            return TnPlan{t->bm, t->bn, (int)lbx_cdiv(M, rps), rps};
        }
    }
    TnPlan best{128, 128, 1, M};
    double best_cost = 1e30;
    for (int c = (BK > 16 ? 1 : 0); c < 4; ++c) {
        const int bm = CAND[c][0], bn = CAND[c][1];
        const long tiles = lbx_cdiv(K1, bm) * lbx_cdiv(N, bn);
        for (long target = NUM_CU; target <= 8 * NUM_CU; target += NUM_CU / 2) {
            long s = target / tiles;
            if (s < 1) s = 1;
            const long max_s = lbx_cdiv(M, 4 * BK);
            if (s > max_s) s = max_s;
            if (s < 1) s = 1;
            long rps = lbx_cdiv(lbx_cdiv(M, s), BK) * BK;
            s = lbx_cdiv(M, rps);
            double cost = launch_cost(c, tiles * s, (double)rps / BK);
            cost += 5.0 + (double)K1 * N * 4.0 * (s + 1) / 3.0e6;
            if (cost < best_cost) { best_cost = cost; best = TnPlan{bm, bn, (int)s, rps}; }
        }
    }
    return best;
}

template <int BM, int BN>
void launch_tn_t(bool al, unsigned grid, hipStream_t st, RowsD A, RowsD Bd, float* P, float* Pc, long M, int K1, int N,
                 int tiles_n, int ntiles, long rps) {
    if (al)
        hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, true>), dim3(grid), dim3(256), 0, st, A, Bd, P, Pc, M, K1, N, tiles_n,
                           ntiles, rps);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, false>), dim3(grid), dim3(256), 0, st, A, Bd, P, Pc, M, K1, N, tiles_n,
                           ntiles, rps);
}

}  // namespace

extern "C" int lidbox_gemm_plan_query(int kind, long M, int N, int K, size_t workspace_bytes, int* out4) {
    LBX_ARG(out4 && M > 0 && N > 0 && K > 0 && kind >= 0 && kind <= 2, "kind in {0 nn, 1 nt, 2 tn}, positive sizes");
    if (kind == 2) {           // tn: (M rows contracted, K = K1, N)
        const TnPlan pl = tn_plan(M, K, N);
        out4[0] = pl.bm; out4[1] = pl.bn; out4[2] = pl.splits; out4[3] = (int)pl.rows_per_split;
    } else {
        const RowsChoice ch = choose_rows_env(kind, M, N, K, workspace_bytes);
        out4[0] = ch.bm; out4[1] = ch.bn; out4[2] = ch.splits; out4[3] = ch.k_per_split;
    }
    return LIDBOX_OK;
}

extern "C" int lidbox_gemm_plan_is_stream_k(int kind, long M, int N, int K, size_t workspace_bytes) {
    if (M <= 0 || N <= 0 || K <= 0 || kind < 0 || kind > 2) return 0;
    if (kind == 2) {
        const SkTn t = sk_tn_plan(M, K, N);
        return t.ok && workspace_bytes >= t.ws_need && !sk_tuned_out(2, M, N, K);
    }
    {
        const SkRows r = sk_rows_plan(kind, M, N, K);
        return r.ok && workspace_bytes >= r.ws_need && (kind == 1 || N % 4 == 0) && !sk_tuned_out(kind, M, N, K);
    }
    return LIDBOX_OK;
}

extern "C" int lidbox_gemm_plan_waves(int kind, long M, int N, int K, size_t workspace_bytes) {
    if (M <= 0 || N <= 0 || K <= 0 || kind < 0 || kind > 1) return 4;
    return choose_rows_env(kind, M, N, K, workspace_bytes).waves;
}

extern "C" int lidbox_gemm_last_family(void) { return g_last_family; }

extern "C" int lidbox_gemm_plan_stream_tail(int kind, long M, int N, int K, size_t workspace_bytes) {
    if (M <= 0 || N <= 0 || K <= 0 || kind < 0 || kind > 1 || dma_mode() == 0) return 0;
    if (!getenv("LIDBOX_GEMM_PLAN") && !getenv("LIDBOX_GEMM_TILE") && lidbox_gemm_plan_is_stream_k(kind, M, N, K, workspace_bytes)) return 0;
    const RowsChoice ch = choose_rows_env(kind, M, N, K, workspace_bytes);
    if (ch.splits != 1) return 0;
    const DmaStreamPlan spl = dma_stream_plan(ch.bm, ch.bn, M, N, K);
    return (spl.g > 1 && workspace_bytes >= spl.ws_need) ? spl.g : 0;
}

extern "C" int lidbox_gemm_last_launches(int* out3) {
    LBX_ARG(out3, "out3 != NULL");
    out3[0] = g_last_launches[0]; out3[1] = g_last_launches[1]; out3[2] = g_last_launches[2];
    return LIDBOX_OK;
}

extern "C" size_t lidbox_gemm_rows_workspace(long M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    // room for the deepest split the cost model may pick for a small-M problem (capped at 64 MiB)
    size_t need = 0;
    for (int kind = 0; kind < 2; ++kind) {
        const RowsChoice ch = choose_rows(kind, M, N, K, (size_t)64 << 20);
        if (ch.splits > 1 && (size_t)ch.splits * M * N * sizeof(float) > need) need = (size_t)ch.splits * M * N * sizeof(float);
    }
    for (int bm = 64; bm <= 128; bm += 64)               // streamed remainder of any tile shape the planner may pick
        for (int bn = 64; bn <= 128; bn += 64) {
            const DmaStreamPlan spl = dma_stream_plan(bm, bn, M, N, K);
            if (spl.g > 1 && spl.ws_need > need) need = spl.ws_need;
        }
    for (int kind = 0; kind < 2; ++kind) {
        const SkRows sk = sk_rows_plan(kind, M, N, K);
        if (sk.ok && sk.ws_need > need) need = sk.ws_need;
        if (skp_mode() != 0 && skp_policy(kind, M, N, K)) {
            const SkRows skp = sk_rows_plan(kind, M, N, K, true);
            if (skp.ok && skp.ws_need > need) need = skp.ws_need;
        }
    }
    return need;
}

extern "C" int lidbox_gemm_nn(lidbox_rows_t A, const float* Bm, long ldb, lidbox_rows_out_t C, int K, int N,
                              int epilogue, const float* aux, void* workspace, size_t workspace_bytes,
                              lidbox_stream_t stream) {
    if (validate_rows_call(__func__, A, Bm, ldb, C, K, N, epilogue, aux, N)) return LIDBOX_E_INVALID;
    return launch_rows<false>(A, Bm, ldb, C, K, N, epilogue, aux, workspace, workspace_bytes, (hipStream_t)stream);
}

#ifdef LBX_GEMM_TIMING
extern "C" int lidbox_gemm_debug_set_stamps(void* device_ptr) {          // timing builds only (not in include/lidbox_hip.h)
    LBX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), &device_ptr, sizeof(void*)));
    return LIDBOX_OK;
}
#endif

extern "C" int lidbox_gemm_nt(lidbox_rows_t A, const float* Bm, long ldb, lidbox_rows_out_t C, int K, int N,
                              int epilogue, const float* aux, void* workspace, size_t workspace_bytes,
                              lidbox_stream_t stream) {
    if (validate_rows_call(__func__, A, Bm, ldb, C, K, N, epilogue, aux, K)) return LIDBOX_E_INVALID;
    return launch_rows<true>(A, Bm, ldb, C, K, N, epilogue, aux, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" size_t lidbox_gemm_tn_workspace(int M, int K1, int N) {
    if (M <= 0 || K1 <= 0 || N <= 0) return 0;
    const TnPlan pl = tn_plan(M, K1, N);
    size_t need = ((size_t)pl.splits * K1 * N + (size_t)pl.splits * N) * sizeof(float);
    const SkTn sk = sk_tn_plan(M, K1, N);
    if (sk.ok && sk.ws_need > need) need = sk.ws_need;
    return need;
}

namespace {

// the wgrad GEMM of lidbox_gemm_tn without its reduce: the slices' partial sums land in the workspace and *job describes the
// fixed-order reduce that finishes C / bias_grad (nblocks == 0: unaligned case, the scalar reduce was launched here)
int tn_partial_launch(const char* fn, lidbox_rows_t A, lidbox_rows_t Bd, float* Cm, long ldc, int K1, int N, int accumulate,
                      float* bias_grad, void* workspace, size_t workspace_bytes, hipStream_t st, ReduceJob* job) {
    *job = ReduceJob{};
    if (check_rows(fn, A.base, A.batch_stride, A.row_stride, A.batch, A.rows_per_batch)) return LIDBOX_E_INVALID;
    if (check_rows(fn, Bd.base, Bd.batch_stride, Bd.row_stride, Bd.batch, Bd.rows_per_batch)) return LIDBOX_E_INVALID;
    LBX_ARG(Cm && K1 >= 1 && N >= 1 && ldc >= N, "C != NULL, K1, N >= 1, ldc >= N");
    const long M = (long)A.batch * A.rows_per_batch;
    LBX_ARG(M == (long)Bd.batch * Bd.rows_per_batch, "A and B row counts differ");
    LBX_ARG(M >= 1, "M >= 1");
    auto finish = [&](const float* P, const float* Pc, int splits) {
        const long n = (long)K1 * N;
        if (reduce_job_vec_ok(P, Pc, splits, n, N, Cm, ldc, bias_grad)) *job = make_reduce_job(P, Pc, splits, n, N, Cm, ldc, accumulate, bias_grad);
        else launch_splitk_reduce(P, Pc, splits, n, N, Cm, ldc, accumulate, bias_grad, st);
    };
    {
        // persistent-body kernel (gemm_sk.h) over a regular split of the contraction rows
        const SkTn sk = sk_tn_plan(M, K1, N);
        if (sk.ok && !getenv("LIDBOX_GEMM_TN_PLAN") && !sk_tuned_out(2, M, N, K1) && workspace && aligned16(workspace) && workspace_bytes >= sk.ws_need &&
            rows_aligned(A) && rows_aligned(Bd) && sk_extent_ok(A, K1) && sk_extent_ok(Bd, N)) {
            sk_set_lds_attr();
            g_last_launches[0] = 1; g_last_launches[1] = 0; g_last_launches[2] = 1;
            g_last_family = 2;
            float* P = (float*)workspace;
            float* Pc = bias_grad ? P + (size_t)sk.splits * K1 * N : nullptr;
            hipLaunchKernelGGL(gemm_sk_tn_kernel, dim3((unsigned)(sk.ntiles * sk.splits)), dim3(256), SK_LDS_BYTES, st, to_dev(A),
                               to_dev(Bd), P, Pc, M, K1, N, sk.tiles_n, sk.ntiles, sk.rows_per_split);
            LBX_LAUNCH_OK();
            finish(P, Pc, sk.splits);
            LBX_LAUNCH_OK();
            return LIDBOX_OK;
        }
    }
    const TnPlan pl = tn_plan(M, K1, N);
    const size_t need = ((size_t)pl.splits * K1 * N + (size_t)pl.splits * N) * sizeof(float);
    LBX_ARG(workspace && workspace_bytes >= need, "workspace too small (lidbox_gemm_tn_workspace)");
    g_last_launches[0] = 1; g_last_launches[1] = 0; g_last_launches[2] = 1;
    g_last_family = 0;
    const int tiles_n = (int)lbx_cdiv(N, pl.bn);
    const int ntiles = (int)(lbx_cdiv(K1, pl.bm) * tiles_n);
    const bool al = rows_aligned(A) && rows_aligned(Bd) && K1 % 4 == 0 && N % 4 == 0;
    float* P = (float*)workspace;
    float* Pc = bias_grad ? P + (size_t)pl.splits * K1 * N : nullptr;
    const unsigned grid = (unsigned)(ntiles * pl.splits);
    // the same decomposition on the LDS-DMA operand path (gemm_dma.h): 16-byte aligned operands with 32-bit extents
    if (al && dma_mode() != 0 && sk_extent_ok(A, K1) && sk_extent_ok(Bd, N) && pl.rows_per_split % SK_BK == 0) {
        g_last_family = 1;
#define LBX_TN_DMA(BM_, BN_) hipLaunchKernelGGL((gemm_tn_dma_kernel<BM_, BN_>), dim3(grid), dim3(256), 0, st, to_dev(A), to_dev(Bd), P, Pc, M, K1, N, tiles_n, ntiles, pl.rows_per_split)
        if (pl.bm == 128 && pl.bn == 128) LBX_TN_DMA(128, 128);
        else if (pl.bm == 128) LBX_TN_DMA(128, 64);
        else if (pl.bn == 128) LBX_TN_DMA(64, 128);
        else LBX_TN_DMA(64, 64);
#undef LBX_TN_DMA
        LBX_LAUNCH_OK();
        finish(P, Pc, pl.splits);
        LBX_LAUNCH_OK();
        return LIDBOX_OK;
    }
#define LBX_TN(BM_, BN_) launch_tn_t<BM_, BN_>(al, grid, st, to_dev(A), to_dev(Bd), P, Pc, M, K1, N, tiles_n, ntiles, pl.rows_per_split)
#if LBX_GEMM_BK == 16
    if (pl.bm == 128 && pl.bn == 128) LBX_TN(128, 128);
    else
#endif
    if (pl.bm == 128) LBX_TN(128, 64);
    else if (pl.bn == 128) LBX_TN(64, 128);
    else LBX_TN(64, 64);
#undef LBX_TN
    LBX_LAUNCH_OK();
    finish(P, Pc, pl.splits);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

static_assert(sizeof(lidbox_reduce_job_t) == sizeof(ReduceJob), "lidbox_reduce_job_t mirrors ReduceJob");
inline ReduceJob job_in(const lidbox_reduce_job_t* j) {
    ReduceJob r;
    if (j) memcpy(&r, j, sizeof r);
    return r;
}
// the non-empty jobs of a caller's list (at most MAX_CARRY are kept; -1: too many)
inline int jobs_in(const lidbox_reduce_job_t* jobs, int njobs, ReduceJob (&out)[MAX_CARRY]) {
    int m = 0;
    for (int i = 0; jobs && i < njobs; ++i) {
        if (jobs[i].nblocks == 0) continue;
        if (m == MAX_CARRY) return -1;
        out[m++] = job_in(jobs + i);
    }
    return m;
}

// the jobs as one launch of their own
int run_reduce_jobs(const ReduceJob* jobs, int njobs, hipStream_t st) {
    ReduceJobs js;
    for (int i = 0, m = 0; i < njobs && m < MAX_CARRY; ++i) {
        if (jobs[i].nblocks == 0) continue;
        js.j[m++] = jobs[i];
        js.total += jobs[i].nblocks;
    }
    if (js.total == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(splitk_reduce4_kernel, dim3(js.total), dim3(256), 0, st, js);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

}  // namespace

extern "C" int lidbox_gemm_tn(lidbox_rows_t A, lidbox_rows_t Bd, float* Cm, long ldc, int K1, int N,
                              int accumulate, float* bias_grad, void* workspace, size_t workspace_bytes,
                              lidbox_stream_t stream) {
    ReduceJob job;
    int rc = tn_partial_launch(__func__, A, Bd, Cm, ldc, K1, N, accumulate, bias_grad, workspace, workspace_bytes, (hipStream_t)stream, &job);
    if (rc) return rc;
    return run_reduce_jobs(&job, 1, (hipStream_t)stream);
}

// The two halves of lidbox_gemm_tn as separate calls, so that the reduce can be CARRIED by a later GEMM launch on the same
// stream (lidbox_gemm_nt_carry): the workspace holds the slices until the job has run.
extern "C" int lidbox_gemm_tn_partial(lidbox_rows_t A, lidbox_rows_t Bd, float* Cm, long ldc, int K1, int N, int accumulate,
                                      float* bias_grad, void* workspace, size_t workspace_bytes, lidbox_reduce_job_t* job,
                                      lidbox_stream_t stream) {
    LBX_ARG(job, "job != NULL");
    ReduceJob j;
    int rc = tn_partial_launch(__func__, A, Bd, Cm, ldc, K1, N, accumulate, bias_grad, workspace, workspace_bytes, (hipStream_t)stream, &j);
    memcpy(job, &j, sizeof j);
    return rc;
}

// A zero fill in the shape of a job (splits = 0: the "sum" of no slices): `batch` runs of row_floats zeros, batch_stride floats
// apart.  The engine carries it like a reduce: the rows a strided layer's accumulating dgrad groups do not all cover are
// cleared in the leading workgroups of the group-0 launch instead of by a fill launch of their own.
extern "C" int lidbox_zero_job(float* base, long batch_stride, long row_floats, int batch, lidbox_reduce_job_t* job) {
    LBX_ARG(job && base && batch >= 1 && row_floats >= 4 && batch_stride >= row_floats, "base, job != NULL; batch >= 1; row_floats >= 4");
    LBX_ARG(aligned16(base) && row_floats % 4 == 0 && batch_stride % 4 == 0 && row_floats <= 0x7fffffffL,
            "16-byte aligned base, row_floats and batch_stride multiples of 4");
    ReduceJob j;
    j.Cm = base; j.n = (long)batch * row_floats; j.N = (int)row_floats; j.ldc = batch_stride; j.splits = 0; j.accumulate = 0;
    long g = lbx_cdiv(j.n / 4, 256);
    if (g > 2048) g = 2048;
    j.nblocks = (unsigned)g;
    memcpy(job, &j, sizeof j);
    return LIDBOX_OK;
}

extern "C" int lidbox_reduce_jobs_run(const lidbox_reduce_job_t* jobs, int njobs, lidbox_stream_t stream) {
    LBX_ARG(njobs >= 0 && (jobs || njobs == 0), "jobs != NULL");
    ReduceJob js[MAX_CARRY];
    const int m = jobs_in(jobs, njobs, js);
    LBX_ARG(m >= 0, "at most 2 non-empty jobs per call");
    return run_reduce_jobs(js, m, (hipStream_t)stream);
}

// lidbox_gemm_nt whose launch also runs the pending `jobs` (NULL / 0 / empty ones: plain lidbox_gemm_nt): in its leading
// workgroups when the launch is an LDS-DMA tile launch (gemm_dma.h), else as a launch of their own behind it.  Results of
// both are bit-identical to the separate calls.
extern "C" int lidbox_gemm_nt_carry(lidbox_rows_t A, const float* Bm, long ldb, lidbox_rows_out_t C, int K, int N, int epilogue,
                                    const float* aux, void* workspace, size_t workspace_bytes, const lidbox_reduce_job_t* jobs,
                                    int njobs, lidbox_stream_t stream) {
    if (validate_rows_call(__func__, A, Bm, ldb, C, K, N, epilogue, aux, K)) return LIDBOX_E_INVALID;
    ReduceJob js[MAX_CARRY];
    const int m = jobs_in(jobs, njobs, js);
    LBX_ARG(m >= 0, "at most 2 non-empty jobs per call");
    for (int i = 0; i < m; ++i) LBX_ARG(js[i].splits >= 0, "an optimizer-prepare job runs through lidbox_reduce_jobs_run only");
    for (int i = 0; i < m; ++i) LBX_ARG((const void*)js[i].P != workspace, "a job's slices live in this call's workspace");
    Carry carry;
    carry.jobs = js;
    carry.njobs = (m > 0 && getenv("LIDBOX_GEMM_NO_CARRY") == nullptr) ? m : 0;
    int rc = launch_rows<true>(A, Bm, ldb, C, K, N, epilogue, aux, workspace, workspace_bytes, (hipStream_t)stream, &carry);
    if (rc) return rc;
    g_last_carried = carry.carried ? m : 0;
    if (!carry.carried) return run_reduce_jobs(js, m, (hipStream_t)stream);
    return LIDBOX_OK;
}

extern "C" int lidbox_gemm_last_carried(void) { return g_last_carried; }

inline long pair_max_blocks() {
    if (const char* e = getenv("LIDBOX_GEMM_PAIR_MAX_BLOCKS")) return atol(e);     // A/B aid
    return 12L * NUM_CU;           // two rounds of resident workgroups: frame4 at bs 256 (97 -> 88 us); beyond that the pair gains nothing
}

// whether dgrad (M x N, K = Co) + wgrad (K1 x Co over M rows) of 16-byte aligned operands go out as one launch
bool pair_plan(long M, int Co, int N, int K1, size_t ws_nt_bytes, size_t ws_tn_bytes, RowsChoice* ch_out, TnPlan* pl_out) {
    if (dma_mode() == 0 || getenv("LIDBOX_GEMM_NO_PAIR") != nullptr || M < 1 || N < 1 || K1 < 1 || Co < 1) return false;
    if (Co % 4 != 0 || K1 % 4 != 0) return false;
    const RowsChoice ch = choose_rows_env(1, M, N, Co, ws_nt_bytes);
    const TnPlan pl = tn_plan(M, K1, Co);
    const SkTn sk = sk_tn_plan(M, K1, Co);
    const bool sk_rows = !getenv("LIDBOX_GEMM_PLAN") && !getenv("LIDBOX_GEMM_TILE") && lidbox_gemm_plan_is_stream_k(1, M, N, Co, ws_nt_bytes);
    const bool sk_tn = sk.ok && !getenv("LIDBOX_GEMM_TN_PLAN") && !sk_tuned_out(2, M, Co, K1) && ws_tn_bytes >= sk.ws_need;
    const size_t tn_need = ((size_t)pl.splits * K1 * Co + (size_t)pl.splits * Co) * sizeof(float);
    const long rows_blocks = lbx_cdiv(M, 64L) * lbx_cdiv((long)N, 64L) * ch.splits;
    const long tn_blocks = lbx_cdiv((long)K1, 64L) * lbx_cdiv((long)Co, 64L) * pl.splits;
    if (!(ch.bm == 64 && ch.bn == 64 && ch.waves == 4 && pl.bm == 64 && pl.bn == 64 && !sk_rows && !sk_tn && ws_tn_bytes >= tn_need &&
          (ch.splits == 1 || (size_t)ch.splits * M * N * sizeof(float) <= ws_nt_bytes) && rows_blocks + tn_blocks <= pair_max_blocks()))
        return false;
    if (ch_out) *ch_out = ch;
    if (pl_out) *pl_out = pl;
    return true;
}

// A layer's dgrad and wgrad, both reading the output gradient dY [M, Co]:
//     dX rows = epi(dY . W^T)      (lidbox_gemm_nt:  A = dY, B = W [N][Co], C = dX, K = Co)
//     dW [K1, Co] = X^T . dY, db   (lidbox_gemm_tn:  A = X [M, K1], B = dY)
// When both are small 64 x 64-tile launches of the LDS-DMA family (the dense head: M = batch rows) they go out as ONE launch
// (gemm_nt_tn_pair_kernel, gemm_dma.h) followed by their reduces; otherwise this is exactly the two calls.  Results are
// bit-identical to the two calls either way (same bodies, tiles and summation orders).
// + jobs_in: pending reduces of EARLIER layers, carried by this call's launch (the pair kernel or the dgrad launch);
// job_out (may be NULL): where this call's own wgrad reduce goes when it could not be placed inside this call's launches (the
// one-launch pair: the slices are complete only when that launch ends) -- the caller hands it to a later carry call or to
// lidbox_reduce_jobs_run; with job_out == NULL the reduce is launched here.  *job_out comes back empty otherwise.
extern "C" int lidbox_gemm_nt_tn_carry(lidbox_rows_t dY, const float* W, long ldb, lidbox_rows_out_t dX, int Co, int N, int epilogue,
                                       const float* aux, void* ws_nt, size_t ws_nt_bytes, lidbox_rows_t X, float* dW, long ldc, int K1,
                                       int accumulate, float* bias_grad, void* ws_tn, size_t ws_tn_bytes,
                                       const lidbox_reduce_job_t* jobs, int njobs, lidbox_reduce_job_t* job_out, lidbox_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    if (job_out) memset(job_out, 0, sizeof *job_out);
    ReduceJob pend[MAX_CARRY];
    const int npend = jobs_in(jobs, njobs, pend);
    LBX_ARG(npend >= 0 && npend < MAX_CARRY, "at most 1 non-empty pending job per call");
    for (int i = 0; i < npend; ++i)
        LBX_ARG((const void*)pend[i].P != ws_nt && (const void*)pend[i].P != ws_tn, "a pending job's slices live in this call's workspaces");
    for (int i = 0; i < npend; ++i) LBX_ARG(pend[i].splits >= 0, "an optimizer-prepare job runs through lidbox_reduce_jobs_run only");
    const long M = (long)dY.batch * dY.rows_per_batch;
    bool pair = dma_mode() != 0 && getenv("LIDBOX_GEMM_NO_PAIR") == nullptr && M >= 1 && N >= 1 && K1 >= 1 && Co >= 1 && W && dW && ws_tn &&
                ws_nt != ws_tn;
    if (pair) {
        if (validate_rows_call(__func__, dY, W, ldb, dX, Co, N, epilogue, aux, Co)) return LIDBOX_E_INVALID;
        if (check_rows(__func__, X.base, X.batch_stride, X.row_stride, X.batch, X.rows_per_batch)) return LIDBOX_E_INVALID;
        LBX_ARG((long)X.batch * X.rows_per_batch == M && ldc >= Co, "X and dY row counts differ, or ldc < Co");
    }
    RowsChoice ch{};
    TnPlan pl{};
    if (pair) {
        const size_t wsb = ws_nt ? ws_nt_bytes : 0;
        const bool al_rows = rows_aligned(dY) && aligned16(W) && ldb % 4 == 0 && sk_extent_ok(dY, Co) && sk_b_extent_ok(true, Co, N, ldb) &&
                             aligned16(ws_nt);
        const bool al_tn = rows_aligned(X) && sk_extent_ok(X, K1) && aligned16(ws_tn);
        pair = al_rows && al_tn && pair_plan(M, Co, N, K1, wsb, ws_tn_bytes, &ch, &pl);
    }
    if (!pair) {
        // wgrad GEMM first, then the dgrad launch carries the wgrad's reduce in its leading workgroups
        lidbox_reduce_job_t all[MAX_CARRY];
        for (int i = 0; i < npend; ++i) memcpy(&all[i], &pend[i], sizeof(ReduceJob));
        int rc = lidbox_gemm_tn_partial(X, dY, dW, ldc, K1, Co, accumulate, bias_grad, ws_tn, ws_tn_bytes, &all[npend], stream);
        if (rc) return rc;
        if (ws_nt == ws_tn) {               // one shared workspace: the slices must be consumed before the dgrad may use it
            rc = lidbox_reduce_jobs_run(all, npend + 1, stream);
            if (rc) return rc;
            return lidbox_gemm_nt(dY, W, ldb, dX, Co, N, epilogue, aux, ws_nt, ws_nt_bytes, stream);
        }
        return lidbox_gemm_nt_carry(dY, W, ldb, dX, Co, N, epilogue, aux, ws_nt, ws_nt_bytes, all, npend + 1, stream);
    }
    PairRows r;
    r.A = to_dev(dY); r.Bm = W; r.ldb = ldb;
    r.Cd = RowsOutD{dX.base, dX.batch_stride, dX.row_stride, dX.batch, dX.rows_per_batch};
    r.P = (float*)ws_nt; r.M = M; r.K = Co; r.N = N; r.epi = epilogue; r.aux = aux;
    r.tiles_n = (int)lbx_cdiv((long)N, 64L);
    r.ntiles = (unsigned)(lbx_cdiv(M, 64L) * r.tiles_n);
    r.k_per_split = ch.k_per_split; r.nx = r.ntiles; r.partial = ch.splits > 1 ? 1 : 0;
    if (ch.splits == 1 && ws_nt) {                              // the streamed remainder of the separate launch, same pieces
        const DmaStreamPlan spl = dma_stream_plan(64, 64, M, N, Co);
        if (spl.g > 1 && ws_nt_bytes >= spl.ws_need) {
            r.sp.pieces = (unsigned)(spl.rem * spl.g);
            r.sp.npad = (r.sp.pieces + 7u) & ~7u;
            r.sp.g = spl.g;
            r.sp.first_tile = (unsigned)spl.whole;
            r.sp.epoch = sk_next_epoch();
            r.sp.counters = (unsigned*)ws_nt;
            r.sp.slabs = (float*)((char*)ws_nt + SK_COUNTER_BYTES);
            r.ntiles = (unsigned)spl.whole;
            r.nx = r.sp.npad + r.ntiles;
        }
    }
    PairTn t;
    t.A = to_dev(X); t.Bd = to_dev(dY);
    t.P = (float*)ws_tn; t.Pc = bias_grad ? t.P + (size_t)pl.splits * K1 * Co : nullptr;
    t.M = M; t.K1 = K1; t.N = Co; t.tiles_n = (int)lbx_cdiv((long)Co, 64L);
    t.ntiles = (int)(lbx_cdiv((long)K1, 64L) * t.tiles_n);
    t.rows_per_split = pl.rows_per_split;
    const unsigned rows_blocks = r.nx * (unsigned)ch.splits;
    ReduceJobs rj;                                              // pending reduces of earlier layers ride in the leading blocks
    if (npend > 0 && getenv("LIDBOX_GEMM_NO_CARRY") == nullptr) rj = pack_carry(pend, npend, carry_cap());
    const unsigned grid = rj.total + rows_blocks + (unsigned)(t.ntiles * pl.splits);
    g_last_launches[0] = 1; g_last_launches[1] = 0; g_last_launches[2] = (job_out ? 0 : 1) + (ch.splits > 1 ? 1 : 0);
    g_last_family = 1;
    g_last_carried = rj.total > 0 ? npend : 0;
    hipLaunchKernelGGL(gemm_nt_tn_pair_kernel, dim3(grid), dim3(256), 0, st, r, t, rows_blocks, rj);
    LBX_LAUNCH_OK();
    if (npend > 0 && rj.total == 0) {
        int rc = run_reduce_jobs(pend, npend, st);
        if (rc) return rc;
    }
    if (ch.splits > 1) {
        long g = lbx_cdiv(M * N, 256);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(rows_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)r.P, ch.splits, 0L, M, N, r.Cd, epilogue,
                           aux);
        LBX_LAUNCH_OK();
    }
    if (job_out && reduce_job_vec_ok(t.P, t.Pc, pl.splits, (long)K1 * Co, Co, dW, ldc, bias_grad)) {
        const ReduceJob j = make_reduce_job(t.P, t.Pc, pl.splits, (long)K1 * Co, Co, dW, ldc, accumulate, bias_grad);
        memcpy(job_out, &j, sizeof j);
        return LIDBOX_OK;
    }
    launch_splitk_reduce((const float*)t.P, (const float*)t.Pc, pl.splits, (long)K1 * Co, Co, dW, ldc, accumulate, bias_grad, st);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_gemm_nt_tn(lidbox_rows_t dY, const float* W, long ldb, lidbox_rows_out_t dX, int Co, int N, int epilogue,
                                 const float* aux, void* ws_nt, size_t ws_nt_bytes, lidbox_rows_t X, float* dW, long ldc, int K1,
                                 int accumulate, float* bias_grad, void* ws_tn, size_t ws_tn_bytes, lidbox_stream_t stream) {
    return lidbox_gemm_nt_tn_carry(dY, W, ldb, dX, Co, N, epilogue, aux, ws_nt, ws_nt_bytes, X, dW, ldc, K1, accumulate, bias_grad, ws_tn,
                                   ws_tn_bytes, nullptr, 0, nullptr, stream);
}

extern "C" int lidbox_gemm_plan_is_pair(long M, int Co, int N, int K1, size_t ws_nt_bytes, size_t ws_tn_bytes) {
    return pair_plan(M, Co, N, K1, ws_nt_bytes, ws_tn_bytes, nullptr, nullptr) ? 1 : 0;
}

extern "C" size_t lidbox_colsum_workspace(long M, int N) {
    long slices = lbx_cdiv(M, 256);
    if (slices > 128) slices = 128;
    if (slices < 1) slices = 1;
    return (size_t)slices * N * sizeof(float);
}

extern "C" int lidbox_colsum(lidbox_rows_t A, int N, float* out, int accumulate, void* workspace,
                             size_t workspace_bytes, lidbox_stream_t stream) {
    if (check_rows(__func__, A.base, A.batch_stride, A.row_stride, A.batch, A.rows_per_batch)) return LIDBOX_E_INVALID;
    LBX_ARG(out && N >= 1, "out != NULL, N >= 1");
    const long M = (long)A.batch * A.rows_per_batch;
    long slices = lbx_cdiv(M, 256);
    if (slices > 128) slices = 128;
    if (slices < 1) slices = 1;
    LBX_ARG(workspace && workspace_bytes >= (size_t)slices * N * sizeof(float), "workspace too small (lidbox_colsum_workspace)");
    const long rps = lbx_cdiv(M, slices);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_stage1, dim3((unsigned)lbx_cdiv(N, 64), (unsigned)slices), dim3(256), 0, st,
                       to_dev(A), M, N, rps, (float*)workspace);
    LBX_LAUNCH_OK();
    hipLaunchKernelGGL(colsum_stage2, dim3((unsigned)lbx_cdiv(N, 256)), dim3(256), 0, st, (const float*)workspace,
                       (int)slices, N, out, accumulate);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}
