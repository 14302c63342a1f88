// gemm_sk.h -- persistent stream-K fp32 MFMA GEMM kernels (round 3), included by gemm.hip.
//
// Same contracts as gemm_rows_kernel / gemm_tn_kernel (gemm.hip): Conv1D(padding="causal") / Dense forward, dgrad and
// wgrad of lidbox/models/xvector.py:38-43,53-64 and cnn.py:32-41 as implicit-row GEMMs.  Used for 16-byte aligned
// problems large enough to fill the chip; everything else stays on the kernels of gemm.hip.
//
// What is different (measurements: profiles/r03_skgemm_*.txt, prototype tools/micro/skgemm.hip)
//   * Operand tiles reach LDS by LDS-DMA (global_load_lds_dwordx4, saddr form: wave-uniform 64-bit base in SGPRs + one
//     32-bit byte offset per lane) into a ring of SK_STAGES stages of BK = 16: no VGPR staging, no ds_write, and the loads
//     of step t+2 are in flight while step t multiplies.  One s_barrier per K step.
//   * K-inner operands ([row][k] in HBM: A of nn / nt, B of nt) land as [row][16 floats]; the four 16-byte chunks of a
//     row are XOR-swizzled by (row >> 2) & 3 on the SOURCE side (the LDS image of a DMA is lane-linear) and read with
//     conflict-free ds_read_b128: lanes 0-31 take chunk 2s, lanes 32-63 chunk 2s+1, so MFMA j of sub-step s contracts
//     k = {8s + j, 8s + 4 + j} -- a permutation of the contraction index applied to both operands alike.
//     K-outer operands ([k][col]: B of nn, both operands of tn) land as [k][128] and are read with ds_read_b32.
//   * The four DMA pieces of a wave go out one per MFMA group and the operand reads of the next sub-step behind the
//     first group of the current one: an LDS-DMA issue holds its wave ~56 cycles, four in a row left the matrix pipe
//     idle (ablation: profiles/r03_skgemm_ablation_*.txt).
//   * nn / nt: persistent grid of 256 x SK_WGCU workgroups.  Tiles beyond the whole rounds are streamed (stream-K):
//     every workgroup gets the same number of K steps of them, partial tiles go to slabs in accumulator order, and the
//     LAST contributor of a tile (write-through slab stores -> arrival ticket -> write-through loads; nobody ever waits,
//     no cache-wide fence) sums the slabs in k order
//     -- deterministic -- and runs the fused epilogue.  The streamed spans start at different k offsets, so epilogues and
//     prologues of co-resident workgroups fall into each other's K loops instead of hitting HBM all at once.
//   * Co-resident workgroups of a CU rotate one s_setprio step (slot = blockIdx.x / 256): between equal priorities the
//     matrix pipe goes to the OLDEST wave, which let the first resident run ahead and left the last one alone at the end
//     (profiles/r03_skgemm_timeline_per_cu.txt).
//   * tn (wgrad): the same body over a regular split of the contraction rows; slabs in P[split][K1][N] order feed the
//     fixed-order reduce of gemm_shared.h.
#pragma once

#include "gemm_shared.h"
#include "lds_dma.h"

namespace {

constexpr int SK_BM = 128, SK_BN = 128, SK_BK = 16;
constexpr int SK_A_STAGE = SK_BM * SK_BK, SK_B_STAGE = SK_BK * SK_BN, SK_STAGE = SK_A_STAGE + SK_B_STAGE;   // floats
#ifndef LBX_SK_STAGES
#define LBX_SK_STAGES 3
#endif
#ifndef LBX_SK_WGCU
#define LBX_SK_WGCU 3
#endif
#ifndef LBX_SK_PRIO_ROTATE
#define LBX_SK_PRIO_ROTATE 1               // 0: leave the arbitration to age (A/B aid)
#endif
constexpr int SK_STAGES = LBX_SK_STAGES;          // LDS ring depth: 48 KB per workgroup, three workgroups per CU
constexpr int SK_WGCU = LBX_SK_WGCU;              // resident workgroups per CU the persistent grid is sized for
constexpr int SK_SLAB = SK_BM * SK_BN;            // floats of one partial tile
constexpr size_t SK_COUNTER_BYTES = 16384;        // head of the workspace: one arrival counter per streamed tile (<= 4096)
constexpr size_t SK_LDS_BYTES = (size_t)SK_STAGES * SK_STAGE * sizeof(float);

struct SkPlan {
    int tiles_n, ntiles;
    int nk;                 // K steps per tile (the last one may be partial)
    int ktail;              // valid k of the last step: 4 .. 16, a multiple of 4
    int dp_rounds;          // whole tiles per workgroup
    int sk_tiles, sk_first; // tiles [sk_first, sk_first + sk_tiles) are streamed
    int parts;              // 0: every workgroup takes an equal span of the streamed K steps (fewer tiles than workgroups);
                            // g >= 1: each streamed tile is cut into g equal parts, one workgroup each (the remainder of
                            //         whole rounds: few slabs, the other workgroups go straight to their whole tiles)
};

// ---- K-inner operand: 128 rows x 16 k per step.  Piece i of wave w = rows 16 (2w + i) .. + 15, four lanes per row.
struct SkInner {
    const float* sb;        // wave-uniform: base + k of the next step to issue
    unsigned vo[2];         // this lane's byte offsets of its two pieces (row offset + swizzled chunk)
    int rd;                 // LDS read base of this lane (floats inside the operand's stage)

    // roff[i]: element offset of this lane's row of piece i relative to base (rows outside the matrix already clamped)
    __device__ __forceinline__ void init(const float* base, const long (&roff)[2], int k0, int lane, int wsub) {
        sb = sk_uniform(base + k0);
        const int chunk = (lane & 3) ^ ((lane >> 4) & 3);             // row within the piece = lane >> 2
#pragma unroll
        for (int i = 0; i < 2; ++i) vo[i] = (unsigned)((roff[i] + chunk * 4) * 4);
        rd = (wsub * 64 + (lane & 31)) * 16;
    }
    __device__ __forceinline__ void issue(int i, unsigned dst) const { sk_dma_s(sb, vo[i], dst); }
    // the step holds only kvalid < 16 contraction values: later chunks come from the zero chunk
    __device__ __forceinline__ void issue_tail(int i, unsigned dst, int kvalid, int lane) const {
        const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
        const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(sb) + vo[i]);
        sk_dma_f(chunk * 4 < kvalid ? p : g_sk_zero, dst);
    }
    __device__ __forceinline__ void advance() { sb += SK_BK; }
    __device__ __forceinline__ void read(const float* st, int lane, int s2, float (&v)[2][4]) const {
        const int slot = (2 * s2 + (lane >> 5)) ^ ((lane >> 2) & 3);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4_t x = *reinterpret_cast<const f32x4_t*>(st + rd + b * 32 * 16 + slot * 4);
            v[b][0] = x[0]; v[b][1] = x[1]; v[b][2] = x[2]; v[b][3] = x[3];
        }
    }
};

// ---- K-outer operand of a plain matrix X[k][ld]: 16 k x 128 columns per step.  Piece i of wave w = k rows 2 (2w + i), + 1.
struct SkOuter {
    const float* sb;
    unsigned vo[2];
    long step;
    int rd;

    __device__ __forceinline__ void init(const float* base, long ld, int col0, int ncols, int k0, int lane, int wv, int wsub) {
        sb = sk_uniform(base + (long)k0 * ld + col0);
        int c = (lane & 31) * 4;
        if (col0 + c >= ncols) c = 0;                                // columns outside the matrix: never stored
#pragma unroll
        for (int i = 0; i < 2; ++i) vo[i] = (unsigned)(((long)(2 * (wv * 2 + i) + (lane >> 5)) * ld + c) * 4);
        step = (long)SK_BK * ld;
        rd = (4 * (lane >> 5)) * 128 + wsub * 64 + (lane & 31);
    }
    __device__ __forceinline__ void issue(int i, unsigned dst) const { sk_dma_s(sb, vo[i], dst); }
    __device__ __forceinline__ void issue_tail(int i, unsigned dst, int kvalid, int lane, int wv) const {
        const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(sb) + vo[i]);
        sk_dma_f(2 * (wv * 2 + i) + (lane >> 5) < kvalid ? p : g_sk_zero, dst);
    }
    __device__ __forceinline__ void advance() { sb += step; }
    __device__ __forceinline__ void read(const float* st, int lane, int s2, float (&v)[2][4]) const {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[b][j] = st[rd + (8 * s2 + j) * 128 + b * 32];
    }
};

// ---- K-outer operand whose k index is an implicit row (tn: the contraction runs over the rows of the activations).
//      Offsets are 32-bit element offsets (the host checks the extent), advanced with adds only.
struct SkOuterRows {
    const float* sb;        // base + first column of the tile
    int off[2];             // element offset of this lane's row of piece i for the next step to issue (+ its column)
    unsigned tt[2];         // its position inside the utterance
    int a_step, a_wrap;
    unsigned rpb;
    int mleft[2];           // rows of the slice at or after this lane's row (<= 0: the row is past the slice)
    int rd;

    __device__ __forceinline__ void init(const RowsD& X, int col0, int ncols, long mbeg, long mend, int lane, int wv, int wsub) {
        sb = sk_uniform(X.base + col0);
        int c = (lane & 31) * 4;
        if (col0 + c >= ncols) c = 0;
        a_step = (int)(SK_BK * X.rs);
        a_wrap = X.batch == 1 ? 0 : (int)(X.bs - (long)X.rpb * X.rs);
        rpb = X.batch == 1 ? 0xffffffffu : (unsigned)X.rpb;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long m = mbeg + 2 * (wv * 2 + i) + (lane >> 5);
            const unsigned bq = X.batch == 1 ? 0u : (unsigned)m / rpb;
            tt[i] = (unsigned)m - (X.batch == 1 ? 0u : bq * rpb);
            off[i] = (int)((long)bq * X.bs + (long)tt[i] * X.rs) + c;
            mleft[i] = (int)(mend - m);
        }
        rd = (4 * (lane >> 5)) * 128 + wsub * 64 + (lane & 31);
    }
    // tail (wave-uniform): the step may hold rows past the slice, which contribute zeros
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
    __device__ __forceinline__ void read(const float* st, int lane, int s2, float (&v)[2][4]) const {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[b][j] = st[rd + (8 * s2 + j) * 128 + b * 32];
    }
};

template <int J0, int J1>
__device__ __forceinline__ void sk_mma(const float (&a)[2][4], const float (&b)[2][4], f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int j = J0; j < J1; ++j)
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 2; ++bj)
                acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[bi][j], b[bj][j], acc[bi][bj], 0, 0, 0);
}

// The K loop over n steps of one work item.  issue(piece, step, lds_dst) puts piece 0..3 (A0, A1, B0, B1) of step `step`
// (0-based inside the item) on its way, next() is called once all four pieces of a step are out; reads come through
// ra / rb (stage base, sub-step) -> registers.
// drain(i), i = 0 .. 7, is called behind each of the eight MFMA groups of a step and step_end() once per step: the pipelined
// kernel issues one piece of the PREVIOUS tile's epilogue there (gemm_skp_rows_kernel); the others pass no-ops.
// epi_ops(): vector-memory instructions the drains of the coming step will issue, EXACTLY (wave-uniform): they sit between
// the ring's DMAs in the vmcnt queue (gfx9 returns vector memory operations in issue order), so the wait that publishes a
// stage allows for the ones issued after that stage's DMAs -- a count that is too high would read the stage early.
struct SkNoDrain {
    __device__ __forceinline__ void operator()(int) const {}
};
struct SkNoEpiOps {
    __device__ __forceinline__ int operator()() const { return 0; }
};
__device__ __forceinline__ void sk_wait_vm_dyn(int nops) {
    switch (nops) {
        case 0: sk_wait_vm<0>(); break;
        case 4: sk_wait_vm<4>(); break;
        case 8: sk_wait_vm<8>(); break;
        case 12: sk_wait_vm<12>(); break;
        case 16: sk_wait_vm<16>(); break;
        case 20: sk_wait_vm<20>(); break;
        case 24: sk_wait_vm<24>(); break;
        case 28: sk_wait_vm<28>(); break;
        default: sk_wait_vm<0>(); break;
    }
}
template <int STAGES, int WGCU, class IssueF, class NextF, class ReadA, class ReadB, class DrainF, class StepEndF, class EpiOpsF>
__device__ __forceinline__ void sk_kloop(float* smem, unsigned lds0, int wv, int n, int prio_slot, IssueF issue, NextF next,
                                         ReadA ra, ReadB rb, f32x16 (&acc)[2][2], DrainF drain, StepEndF step_end, EpiOpsF epi_ops) {
    auto dst = [&](int stage, int piece) -> unsigned {
        return lds0 + (unsigned)((stage * SK_STAGE + (piece >> 1) * SK_A_STAGE + (wv * 2 + (piece & 1)) * 256) * 4);
    };
    // every wave is past the previous item's LDS reads once it arrives here (its MFMAs consumed them)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < n) {
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) issue(pc, s, dst(s, pc));
            next();
        }
    if (n >= STAGES - 1) sk_wait_vm<(STAGES - 2) * 4>();
    else sk_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    float a0[2][4], b0[2][4], a1[2][4], b1[2][4];
    ra(smem, 0, a0);
    rb(smem + SK_A_STAGE, 0, b0);
    int cur = 0;
    int e_prev = 0;                     // epilogue operations issued during the previous step
    for (int t = 0; t < n; ++t) {
        // publish step t+1: its pieces were issued STAGES-2 steps ago; what was issued since may stay in flight:
        // the DMAs of the steps after it (STAGES >= 4) and the previous step's epilogue operations
        if (t + 1 < n) {
            if (STAGES >= 4 && t + STAGES - 2 < n) sk_wait_vm_dyn((STAGES - 3) * 4 + (STAGES >= 4 ? e_prev : 0));
            else sk_wait_vm_dyn(STAGES >= 4 ? e_prev : 0);
        }
        e_prev = epi_ops();
        __builtin_amdgcn_s_barrier();
        int nxt = cur + 1;
        if (nxt == STAGES) nxt = 0;
        const bool more = t + STAGES - 1 < n;
        int tgt = cur + STAGES - 1;
        if (tgt >= STAGES) tgt -= STAGES;
        const float* st = smem + cur * SK_STAGE;
        if (WGCU > 0 && LBX_SK_PRIO_ROTATE) {
            if ((unsigned)(t + prio_slot) % (unsigned)(WGCU > 0 ? WGCU : 1) == 0) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
        }
        sk_mma<0, 1>(a0, b0, acc);
        drain(0);
        __builtin_amdgcn_sched_barrier(0);
        ra(st, 1, a1);
        rb(st + SK_A_STAGE, 1, b1);
        __builtin_amdgcn_sched_barrier(0);
        sk_mma<1, 2>(a0, b0, acc);
        if (more) issue(0, t + STAGES - 1, dst(tgt, 0));
        drain(1);
        __builtin_amdgcn_sched_barrier(0);
        sk_mma<2, 3>(a0, b0, acc);
        if (more) issue(1, t + STAGES - 1, dst(tgt, 1));
        drain(2);
        __builtin_amdgcn_sched_barrier(0);
        sk_mma<3, 4>(a0, b0, acc);
        if (more) issue(2, t + STAGES - 1, dst(tgt, 2));
        drain(3);
        __builtin_amdgcn_sched_barrier(0);
        sk_mma<0, 1>(a1, b1, acc);
        if (more) {
            issue(3, t + STAGES - 1, dst(tgt, 3));
            next();
        }
        drain(4);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < n) {
            const float* sn = smem + nxt * SK_STAGE;
            ra(sn, 0, a0);
            rb(sn + SK_A_STAGE, 0, b0);
        }
        __builtin_amdgcn_sched_barrier(0);
        sk_mma<1, 2>(a1, b1, acc);
        drain(5);
        __builtin_amdgcn_sched_barrier(0);
        sk_mma<2, 3>(a1, b1, acc);
        drain(6);
        __builtin_amdgcn_sched_barrier(0);
        sk_mma<3, 4>(a1, b1, acc);
        drain(7);
        step_end();
        cur = nxt;
    }
    if (WGCU > 0 && LBX_SK_PRIO_ROTATE) __builtin_amdgcn_s_setprio(0);
}

// partial tile <-> slab, accumulator order: [(wave * 4 + block) * 4 + r4][lane][4] -- 1 KB per wave access.
// Slabs travel WRITE-THROUGH (sc1 stores, sc1 loads): a hand-off needs no agent-scope release / acquire fence then -- a
// release (buffer_wbl2) writes back EVERY dirty line of the XCD's L2, the output tiles of 95 other workgroups included,
// and cost more than the K loop of a short tile (frame4, bs 256: 103 us with fences vs 51 us on the classic kernels).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sk_slab_rsrc(const float* slab) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sk_uniform(slab)), 0, SK_SLAB * 4, 0x00020000);
}
__device__ __forceinline__ void sk_slab_store(float* slab, const f32x16 (&acc)[2][2], int wv, int lane) {
    const __amdgpu_buffer_rsrc_t r = sk_slab_rsrc(slab);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4_t x = {acc[bi][bj][4 * r4], acc[bi][bj][4 * r4 + 1], acc[bi][bj][4 * r4 + 2], acc[bi][bj][4 * r4 + 3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, x), r,
                                                       ((((wv * 4 + bi * 2 + bj) * 4 + r4) * 64 + lane) * 4) * 4, 0, /*sc1*/ 16);
            }
}
__device__ __forceinline__ void sk_slab_add(const float* slab, f32x16 (&acc)[2][2], int wv, int lane) {
    const __amdgpu_buffer_rsrc_t r = sk_slab_rsrc(slab);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4_t x = __builtin_bit_cast(
                    f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, ((((wv * 4 + bi * 2 + bj) * 4 + r4) * 64 + lane) * 4) * 4, 0, /*sc1*/ 16));
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[bi][bj][4 * r4 + j] += x[j];
            }
            // four 16-byte loads in flight at a time: left alone the scheduler hoists all sixteen (64 registers) above the adds
            __builtin_amdgcn_sched_barrier(0);
        }
}

// ------------------------------------------------------------------------------------------------
// C[M,N] = epi(A[M,K] . B)    B_KINNER = false: B[K][N] (nn)   true: B[N][K] (nt)
// grid = 256 x SK_WGCU persistent workgroups.  ws: [SK_COUNTER_BYTES of arrival counters][2 slabs per workgroup];
// epoch: 1 .. 2^24 - 1, different for launches that may find each other's counters.
// ------------------------------------------------------------------------------------------------
template <bool B_KINNER>
__global__ __launch_bounds__(256, SK_WGCU) void gemm_sk_rows_kernel(RowsD A, const float* __restrict__ Bm, long ldb, RowsOutD Cd,
                                                                    long M, int K, int N, int epi, const float* __restrict__ aux,
                                                                    SkPlan pl, unsigned epoch, unsigned* __restrict__ counters,
                                                                    float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float sk_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const unsigned P = gridDim.x;
    const unsigned pid = xcd_chunk_id(blockIdx.x, P);
    const int prio_slot = (int)(blockIdx.x >> 8);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)sk_smem);
    const int nk = pl.nk;
    const long total = (long)pl.sk_tiles * nk;

    // one work item: K steps [kb, ke) of a tile; returns with the partial sums in acc
    auto run_item = [&](int tile, int kb, int ke, f32x16 (&acc)[2][2], long& m0, int& n0) {
        const int tn = tile % pl.tiles_n, tm = tile / pl.tiles_n;
        m0 = (long)tm * SK_BM;
        n0 = tn * SK_BN;
        SkInner oa;
        {
            long roff[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                long r = m0 + 16 * (wv * 2 + i) + (lane >> 2);
                if (r >= M) r = m0;
                roff[i] = row_offset(A, (unsigned)r);
            }
            oa.init(A.base, roff, kb * SK_BK, lane, wm);
        }
        SkInner obi;
        SkOuter obo;
        if (B_KINNER) {
            long roff[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                long r = n0 + 16 * (wv * 2 + i) + (lane >> 2);
                if (r >= N) r = n0;
                roff[i] = r * ldb;
            }
            obi.init(Bm, roff, kb * SK_BK, lane, wn);
        } else {
            obo.init(Bm, ldb, n0, N, kb * SK_BK, lane, wv, wn);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int n = ke - kb;
        const int tail_step = (ke == nk && pl.ktail < SK_BK) ? n - 1 : -1;       // step of this item that is partial in k
        auto issue = [&](int pc, int step, unsigned d) {
            if (step == tail_step) {
                if (pc < 2) oa.issue_tail(pc, d, pl.ktail, lane);
                else if (B_KINNER) obi.issue_tail(pc - 2, d, pl.ktail, lane);
                else obo.issue_tail(pc - 2, d, pl.ktail, lane, wv);
            } else {
                if (pc < 2) oa.issue(pc, d);
                else if (B_KINNER) obi.issue(pc - 2, d);
                else obo.issue(pc - 2, d);
            }
        };
        auto next = [&]() {
            oa.advance();
            if (B_KINNER) obi.advance();
            else obo.advance();
        };
        auto ra = [&](const float* st, int s2, float (&v)[2][4]) { oa.read(st, lane, s2, v); };
        auto rb = [&](const float* st, int s2, float (&v)[2][4]) {
            if (B_KINNER) obi.read(st, lane, s2, v);
            else obo.read(st, lane, s2, v);
        };
        sk_kloop<SK_STAGES, SK_WGCU>(sk_smem, lds0, wv, n, prio_slot, issue, next, ra, rb, acc, SkNoDrain{}, [] {}, SkNoEpiOps{});
    };

    f32x16 acc[2][2];
    long m0;
    int n0;

    // Work items of this workgroup: its share of the streamed tiles first (their fix-ups then overlap the whole tiles of
    // the other workgroups), then its whole tiles.  ONE call site each for the K loop and the epilogue.
    long it = 0, end = 0;
    // parts mode: block b = (xcd, idx): tile (idx / g) * 8 + xcd, part idx % g -- the parts of a tile sit on one XCD (their
    // slabs and the A panel in one L2), the streamed tiles are dealt round-robin over the XCDs and CUs
    const int g = pl.parts;
    int pt_tile = -1, pt_part = 0;
    if (pl.sk_tiles > 0) {
        if (g == 0) {
            it = (long)pid * total / P;
            end = (long)(pid + 1) * total / P;
        } else {
            const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
            const int t = (idx / g) * 8 + xcd;
            if (t < pl.sk_tiles) { pt_tile = t; pt_part = idx % g; }
        }
    }
    int which = 0, d = 0;
    for (;;) {
        int tile, kb, ke, t = 0;
        if (pt_tile >= 0) {
            t = pt_tile;
            pt_tile = -1;
            kb = (int)((long)pt_part * nk / g);
            ke = (int)((long)(pt_part + 1) * nk / g);
            tile = pl.sk_first + t;
        } else if (it < end) {
            t = (int)(it / nk);
            kb = (int)(it - (long)t * nk);
            ke = kb + (int)(end - it);
            if (ke > nk) ke = nk;
            it += ke - kb;
            tile = pl.sk_first + t;
        } else if (d < pl.dp_rounds) {
            tile = d * (int)P + (int)pid;
            kb = 0;
            ke = nk;
            ++d;
        } else {
            break;
        }
        run_item(tile, kb, ke, acc, m0, n0);
        // Opaque per-item copies of what the epilogue code addresses with: its per-lane / per-row address terms otherwise
        // count as invariants of the item loop, get hoisted above the K loop and spilled (60 VGPRs + 110 SGPRs of scratch;
        // a scratch-using launch cost ~70 us each).
        int lane_e = lane, wm_e = wm, wn_e = wn, wv_e = wv;
        RowsOutD Cd_e = Cd;
        asm volatile("" : "+v"(lane_e), "+s"(wm_e), "+s"(wn_e), "+s"(wv_e), "+s"(Cd_e.rs), "+s"(Cd_e.bs));
        bool finish = kb == 0 && ke == nk;
        bool reducer = false;
        if (!finish) {
            // ---- partial tile: slab, ticket, and the last arriver finishes the tile
            const long my_slab = g == 0 ? (long)pid * 2 + which : (long)blockIdx.x * 2;
            sk_slab_store(slabs + my_slab * SK_SLAB, acc, wv_e, lane_e);
            ++which;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned* flag = reinterpret_cast<unsigned*>(sk_smem);
            if (tid == 0) {
                // arrival counter = (launch epoch << 8) | arrivals.  A word that carries another epoch -- zero after a finished
                // tile, anything at all in a workspace that was never used -- counts as zero arrivals, so the workspace
                // needs no initialisation; the last arriver leaves the word zero (a graph replay, whose epoch is frozen,
                // starts from there).
                unsigned old = __hip_atomic_load(counters + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned cnt;
                for (;;) {
                    cnt = (old >> 8) == epoch ? (old & 255u) : 0u;
                    if (__hip_atomic_compare_exchange_strong(counters + t, &old, (epoch << 8) | (cnt + 1u), __ATOMIC_RELAXED,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        break;
                }
                *flag = cnt;
            }
            __syncthreads();
            const unsigned ticket = *flag;
            // contributors of tile t, in k order: parts mode: the g blocks of the tile; span mode: workgroups pa .. pb
            long pa = 0, pb = g - 1;
            if (g == 0) {
                const long ib = (long)t * nk, ie = ib + nk;
                pa = ib * P / total;
                while (pa > 0 && pa * total / P > ib) --pa;
                while ((pa + 1) * total / P <= ib) ++pa;
                pb = (ie - 1) * P / total;
                while (pb + 1 < (long)P && (pb + 1) * total / P < ie) ++pb;
                while (pb * total / P >= ie) --pb;
            }
            __syncthreads();                                    // everybody has read the flag word
            if (ticket == (unsigned)(pb - pa)) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                for (long p = pa; p <= pb; ++p) {               // k order: a fixed summation order whoever arrives last
                    long sl;
                    if (g == 0) {
                        const long b = p * total / P, e = (p + 1) * total / P;
                        const long t0 = b / nk;
                        const bool first_partial = !(b == t0 * nk && e >= (t0 + 1) * nk);
                        sl = p * 2 + ((t0 == t) ? 0 : (first_partial ? 1 : 0));
                    } else {
                        const long idx0 = (long)(t >> 3) * g;       // first idx of the tile's blocks on XCD t & 7
                        sl = (((idx0 + p) << 3) | (t & 7)) * 2;
                    }
                    sk_slab_add(slabs + sl * SK_SLAB, acc, wv_e, lane_e);
                }
                finish = true;
                reducer = true;
            }
        }
        if (finish) store_rows_tile<2, 2>(acc, m0, n0, wm_e, wn_e, lane_e, 0, M, N, epi, aux, Cd_e, nullptr, 0);
        if (reducer && tid == 0) __hip_atomic_store(counters + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------------
// Pipelined variant of the rows kernel: the epilogue of a finished tile is NOT run behind its K loop.  The tile's
// accumulators are copied to a second register set and its epilogue is issued behind the MFMA groups of the NEXT item's K
// loop, one (32 x 32 block, 8-row) group per K step:
//     step s     : the mask / old values of group s start on their way into a per-wave LDS staging area BY LDS-DMA
//     step s + 2 : they are read from LDS, group s is finished and its eight rows are stored
// so the only vector-memory operations inside the K loop are LDS-DMAs and stores: nothing the compiler has to wait for (a
// VGPR load carried around the loop makes hipcc drain the whole queue, ring DMAs included, at every use), and every drain
// step issues an EXACT number of them -- lanes outside the matrix read a safe address and store to a scratch word -- which
// the counted vmcnt wait of the ring accounts for (sk_kloop: epi_ops).  A tile with K = 512 otherwise spends a fifth of its
// life in a mask epilogue (dependent load -> select -> store chains with idle matrix pipes, all resident workgroups of a
// persistent grid at once).  Two workgroups per CU (256 registers per lane, 48 KB ring + 32 KB staging each = all 160 KB);
// no priority rotation: between equal priorities the older workgroup runs ahead, which staggers the two residents' tile
// boundaries.  Streamed (partial) tiles keep the synchronous slab / ticket path; the tile a last arriver sums takes the
// pipelined road like a whole one.
// ------------------------------------------------------------------------------------------------
constexpr int SKP_WGCU = 2;
constexpr int SKP_STAGE_FLOATS = 512;          // one group of one kind for one wave: 16 rows x 32 columns
constexpr size_t SKP_LDS_BYTES = SK_LDS_BYTES + (size_t)4 * 4 * SKP_STAGE_FLOATS * sizeof(float);   // + [wave][buffer][kind]
constexpr size_t SKP_DUMMY_BYTES = (size_t)NUM_CU * SK_WGCU * 256 * sizeof(float);   // where lanes outside the matrix "store"

struct SkPend {
    int s;                  // drain step of the pending tile: 0 .. 9; 10: nothing pending
    long m0;
    int n0;
};

struct SkEpiCtx {
    RowsOutD Cd;
    long M;
    int N;
    const float* aux;
    bool has_bias, do_relu;
    long wrap;
    unsigned t_wrap;
    float bias0, bias1;     // bias of this lane's two columns of the pending tile
    float* dummy;           // this lane's scratch word: the store target of lanes outside the matrix (every drain store is issued)
    float* stage;           // this wave's LDS staging area [buffer 2][kind 2][16 rows][32 columns]
    unsigned stage_lds;     // its LDS byte address
};

// values of group g (block bi = g >> 2, bj = (g >> 1) & 1, rows r0 = 8 (g & 1) .. + 7) out of the accumulators: the only place
// that indexes them with the group, as a switch over compile-time indices (a runtime index would put them in scratch)
__device__ __forceinline__ void skp_extract(const f32x16 (&acc)[2][2], int g, float (&val)[8]) {
#define SKP_CASE(G)                                                                        \
    case G:                                                                                \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) val[i] = acc[(G) >> 2][((G) >> 1) & 1][8 * ((G) & 1) + i]; \
        break;
    switch (g) {
        SKP_CASE(0) SKP_CASE(1) SKP_CASE(2) SKP_CASE(3) SKP_CASE(4) SKP_CASE(5) SKP_CASE(6) SKP_CASE(7)
        default: break;
    }
#undef SKP_CASE
}

// element offset of row (rb + dr) of C / the mask (negative: outside the matrix).  rb is the first row of a 32-row block and
// WAVE-UNIFORM (the division runs on the scalar unit); dr in 0 .. 31 is the lane's part; an utterance holds >= 32 rows
// (checked on the host), so a block crosses at most one utterance boundary.
__device__ __forceinline__ long skp_row_off(long rb, int dr, const SkEpiCtx& c) {
    long o;
    const unsigned rbu = (unsigned)__builtin_amdgcn_readfirstlane((int)rb);
    if (c.Cd.batch != 1) {
        const unsigned b0 = rbu / (unsigned)c.Cd.rpb;
        const unsigned t0 = rbu - b0 * (unsigned)c.Cd.rpb;
        o = ((long)b0 * c.Cd.bs + (long)t0 * c.Cd.rs) + (long)dr * c.Cd.rs + (t0 + (unsigned)dr >= c.t_wrap ? c.wrap : 0);
    } else {
        o = (long)(rbu + (unsigned)dr) * c.Cd.rs;
    }
    return (long)rbu + dr < c.M ? o : -1;
}

// the mask / old values of group g of the pending tile start towards staging buffer g & 1: per kind two LDS-DMAs, one per
// half wave h of the CONSUMER layout (lane -> row slot lane >> 3 = row i of the group, 16-byte chunk lane & 7 of its 32 columns)
template <bool HAS_MASK, bool ACCUM>
__device__ __forceinline__ void skp_issue_loads(const SkPend& pd, int g, const SkEpiCtx& c, int wm, int wn, int lane) {
    const int bi = g >> 2, bj = (g >> 1) & 1;
    const int i = lane >> 3, chunk = lane & 7;
    const int dr = (i & 3) + 8 * (i >> 2) + 16 * (g & 1);
    const int col = pd.n0 + wn * 64 + bj * 32 + chunk * 4;
    const long rb = pd.m0 + wm * 64 + bi * 32;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const long o = skp_row_off(rb, dr + 4 * h, c);
        const long a = (o >= 0 && col < c.N) ? o + col : 0;                  // a safe address for what lies outside the matrix
        const unsigned dst = c.stage_lds + (unsigned)((((g & 1) * 2) * SKP_STAGE_FLOATS + h * 256) * 4);
        if (HAS_MASK) sk_dma_f(c.aux + a, dst);
        if (ACCUM) sk_dma_f(c.Cd.base + a, dst + SKP_STAGE_FLOATS * 4);
    }
}

// group g of the pending tile: staged values -> registers (before the buffer is handed to group g + 2's DMAs)
template <bool HAS_MASK, bool ACCUM>
__device__ __forceinline__ void skp_read_stage(int g, const SkEpiCtx& c, int lane, float (&mv)[8], float (&ov)[8]) {
    const float* st = c.stage + ((g & 1) * 2) * SKP_STAGE_FLOATS + (lane >> 5) * 256 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (HAS_MASK) mv[i] = st[i * 32];
        if (ACCUM) ov[i] = st[SKP_STAGE_FLOATS + i * 32];
    }
}

// rows i0 .. i0 + 3 of group g: finish and store (every lane stores: outside the matrix to its scratch word)
template <bool HAS_MASK, bool ACCUM>
__device__ __forceinline__ void skp_store4(const SkPend& pd, int g, int i0, const SkEpiCtx& c, int wm, int wn, int lane,
                                           const float (&val)[8], const float (&mv)[8], const float (&ov)[8]) {
    const int bi = g >> 2, bj = (g >> 1) & 1;
    const int col = pd.n0 + wn * 64 + bj * 32 + (lane & 31);
    const float bias = bj ? c.bias1 : c.bias0;
    const long rb = pd.m0 + wm * 64 + bi * 32;
#pragma unroll
    for (int i = i0; i < i0 + 4; ++i) {
        const int dr = (i & 3) + 8 * (i >> 2) + 16 * (g & 1) + 4 * (lane >> 5);
        const long o = skp_row_off(rb, dr, c);
        float x = val[i] + bias;
        if (HAS_MASK) x = mv[i] > 0.f ? x : 0.f;
        if (ACCUM) x += ov[i];
        if (c.do_relu) x = fmaxf(x, 0.f);
        float* dst = (o >= 0 && col < c.N) ? c.Cd.base + o + col : c.dummy;
        *dst = x;
    }
}

template <bool B_KINNER, bool HAS_MASK, bool ACCUM>
__global__ __launch_bounds__(256, SKP_WGCU) void gemm_skp_rows_kernel(RowsD A, const float* __restrict__ Bm, long ldb, RowsOutD Cd,
                                                                      long M, int K, int N, int epi, const float* __restrict__ aux,
                                                                      SkPlan pl, unsigned epoch, unsigned* __restrict__ counters,
                                                                      float* __restrict__ slabs, float* __restrict__ dummy) {
    extern __shared__ __attribute__((aligned(16))) float sk_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const unsigned P = gridDim.x;
    const unsigned pid = xcd_chunk_id(blockIdx.x, P);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)sk_smem);
    const int nk = pl.nk;
    const long total = (long)pl.sk_tiles * nk;

    SkEpiCtx ec;
    ec.Cd = Cd; ec.M = M; ec.N = N; ec.aux = aux;
    ec.has_bias = epi == LIDBOX_EPI_BIAS || epi == LIDBOX_EPI_BIAS_RELU;
    ec.do_relu = epi == LIDBOX_EPI_BIAS_RELU || epi == LIDBOX_EPI_ACCUM_RELU || epi == LIDBOX_EPI_RELU;
    ec.wrap = Cd.batch != 1 ? Cd.bs - (long)Cd.rpb * Cd.rs : 0;
    ec.t_wrap = Cd.batch != 1 ? (unsigned)Cd.rpb : 0xffffffffu;
    ec.bias0 = ec.bias1 = 0.f;
    ec.dummy = dummy + (long)blockIdx.x * 256 + tid;
    ec.stage = sk_smem + SK_STAGES * SK_STAGE + wv * 4 * SKP_STAGE_FLOATS;
    ec.stage_lds = lds0 + (unsigned)((SK_STAGES * SK_STAGE + wv * 4 * SKP_STAGE_FLOATS) * 4);

    SkPend pd;
    pd.s = 10;
    pd.m0 = 0; pd.n0 = 0;
    float val[8], mv[8], ov[8];      // the group being stored in the current drain step
#pragma unroll
    for (int i = 0; i < 8; ++i) { val[i] = 0.f; mv[i] = 0.f; ov[i] = 0.f; }
    // one drain step = three pieces, placed behind the ring's DMAs of the K step (MFMA groups 5, 6, 7)
    auto drain_piece = [&](int which, const f32x16 (&prev)[2][2]) {
        const int ds = __builtin_amdgcn_readfirstlane(pd.s);          // wave-uniform, and provably so: scalar branches
        if (ds >= 10) return;
        if (which == -1) {
            if (ds >= 2) {                                    // group s - 2: staged two steps ago, landed (counted wait / flush wait)
                skp_read_stage<HAS_MASK, ACCUM>(ds - 2, ec, lane, mv, ov);
                skp_extract(prev, ds - 2, val);
            }
        } else if (which == 0) {
            // the staging reads (issued one MFMA group ago) are back before their buffer is refilled
            if ((HAS_MASK || ACCUM) && ds >= 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (ds < 8) skp_issue_loads<HAS_MASK, ACCUM>(pd, ds, ec, wm, wn, lane);
        } else if (ds >= 2) {
            skp_store4<HAS_MASK, ACCUM>(pd, ds - 2, which == 1 ? 0 : 4, ec, wm, wn, lane, val, mv, ov);
        }
    };
    auto drain_ops = [&]() -> int {
        const int ds = __builtin_amdgcn_readfirstlane(pd.s);
        if (ds >= 10) return 0;
        return (ds < 8 ? 2 * ((HAS_MASK ? 1 : 0) + (ACCUM ? 1 : 0)) : 0) + (ds >= 2 ? 8 : 0);
    };
    // outside a K loop: the rest of the pending epilogue, step by step (each step waits for everything in flight)
    auto flush = [&](const f32x16 (&prev)[2][2]) {
        while (pd.s < 10) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            drain_piece(-1, prev);
            drain_piece(0, prev);
            drain_piece(1, prev);
            drain_piece(2, prev);
            ++pd.s;
        }
    };

    // work items as in gemm_sk_rows_kernel
    long it = 0, end = 0;
    const int g = pl.parts;
    int pt_tile = -1, pt_part = 0;
    if (pl.sk_tiles > 0) {
        if (g == 0) {
            it = (long)pid * total / P;
            end = (long)(pid + 1) * total / P;
        } else {
            const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
            const int t = (idx / g) * 8 + xcd;
            if (t < pl.sk_tiles) { pt_tile = t; pt_part = idx % g; }
        }
    }
    int which = 0, d = 0;

    // One K loop and one accumulator set `cur`; a finished whole tile is COPIED to `prev` (64 register moves per tile) and
    // drains from there while the next item accumulates into `cur`.
    f32x16 cur[2][2], prev[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) prev[i][j][r] = 0.f;
    for (;;) {
        int tile, kb, ke, t = 0;
        if (pt_tile >= 0) {
            t = pt_tile;
            pt_tile = -1;
            kb = (int)((long)pt_part * nk / g);
            ke = (int)((long)(pt_part + 1) * nk / g);
            tile = pl.sk_first + t;
        } else if (it < end) {
            t = (int)(it / nk);
            kb = (int)(it - (long)t * nk);
            ke = kb + (int)(end - it);
            if (ke > nk) ke = nk;
            it += ke - kb;
            tile = pl.sk_first + t;
        } else if (d < pl.dp_rounds) {
            tile = d * (int)P + (int)pid;
            kb = 0;
            ke = nk;
            ++d;
        } else {
            break;
        }
        const int tn = tile % pl.tiles_n, tm = tile / pl.tiles_n;
        const long m0 = (long)tm * SK_BM;
        const int n0 = tn * SK_BN;
        SkInner oa;
        {
            long roff[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                long r = m0 + 16 * (wv * 2 + i) + (lane >> 2);
                if (r >= M) r = m0;
                roff[i] = row_offset(A, (unsigned)r);
            }
            oa.init(A.base, roff, kb * SK_BK, lane, wm);
        }
        SkInner obi;
        SkOuter obo;
        if (B_KINNER) {
            long roff[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                long r = n0 + 16 * (wv * 2 + i) + (lane >> 2);
                if (r >= N) r = n0;
                roff[i] = r * ldb;
            }
            obi.init(Bm, roff, kb * SK_BK, lane, wn);
        } else {
            obo.init(Bm, ldb, n0, N, kb * SK_BK, lane, wv, wn);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) cur[i][j][r] = 0.f;
        const int n = ke - kb;
        const int tail_step = (ke == nk && pl.ktail < SK_BK) ? n - 1 : -1;
        auto issue = [&](int pc, int step, unsigned dd) {
            if (step == tail_step) {
                if (pc < 2) oa.issue_tail(pc, dd, pl.ktail, lane);
                else if (B_KINNER) obi.issue_tail(pc - 2, dd, pl.ktail, lane);
                else obo.issue_tail(pc - 2, dd, pl.ktail, lane, wv);
            } else {
                if (pc < 2) oa.issue(pc, dd);
                else if (B_KINNER) obi.issue(pc - 2, dd);
                else obo.issue(pc - 2, dd);
            }
        };
        auto next = [&]() {
            oa.advance();
            if (B_KINNER) obi.advance();
            else obo.advance();
        };
        auto ra = [&](const float* st, int s2, float (&v)[2][4]) { oa.read(st, lane, s2, v); };
        auto rb = [&](const float* st, int s2, float (&v)[2][4]) {
            if (B_KINNER) obi.read(st, lane, s2, v);
            else obo.read(st, lane, s2, v);
        };
        auto drain = [&](int i) {
            if (i >= 4) drain_piece(i - 5, prev);
        };
        auto step_end = [&]() {
            if (pd.s < 10) ++pd.s;
        };
        sk_kloop<SK_STAGES, 0>(sk_smem, lds0, wv, n, 0, issue, next, ra, rb, cur, drain, step_end, drain_ops);
        // what the K loop was too short to drain (prev is about to be overwritten / the drain registers reused)
        flush(prev);
        if (kb == 0 && ke == nk) {
            // whole tile: its epilogue rides the next item's K loop
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) prev[i][j] = cur[i][j];
            pd.s = 0;
            pd.m0 = m0;
            pd.n0 = n0;
            if (ec.has_bias) {
                const int c0 = n0 + wn * 64 + (lane & 31);
                ec.bias0 = c0 < N ? aux[c0] : 0.f;
                ec.bias1 = c0 + 32 < N ? aux[c0 + 32] : 0.f;
            }
            continue;
        }
        // ---- partial tile: slab, ticket, and the last arriver finishes the tile (synchronously, on `cur`)
        // nothing is pending here (flushed above): say so to the register allocator -- `prev` and the drain state are dead
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) prev[i][j][r] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { val[i] = 0.f; mv[i] = 0.f; ov[i] = 0.f; }
        int lane_e = lane, wv_e = wv;
        asm volatile("" : "+v"(lane_e), "+s"(wv_e));
        const long my_slab = g == 0 ? (long)pid * 2 + which : (long)blockIdx.x * 2;
        sk_slab_store(slabs + my_slab * SK_SLAB, cur, wv_e, lane_e);
        ++which;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(sk_smem);
        if (tid == 0) {
            unsigned old = __hip_atomic_load(counters + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned cnt;
            for (;;) {
                cnt = (old >> 8) == epoch ? (old & 255u) : 0u;
                if (__hip_atomic_compare_exchange_strong(counters + t, &old, (epoch << 8) | (cnt + 1u), __ATOMIC_RELAXED,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                    break;
            }
            *flag = cnt;
        }
        __syncthreads();
        const unsigned ticket = *flag;
        long pa = 0, pb = g - 1;
        if (g == 0) {
            const long ib = (long)t * nk, ie = ib + nk;
            pa = ib * P / total;
            while (pa > 0 && pa * total / P > ib) --pa;
            while ((pa + 1) * total / P <= ib) ++pa;
            pb = (ie - 1) * P / total;
            while (pb + 1 < (long)P && (pb + 1) * total / P < ie) ++pb;
            while (pb * total / P >= ie) --pb;
        }
        __syncthreads();
        if (ticket == (unsigned)(pb - pa)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) cur[i][j][r] = 0.f;
            for (long p = pa; p <= pb; ++p) {
                long sl;
                if (g == 0) {
                    const long b = p * total / P, e = (p + 1) * total / P;
                    const long t0 = b / nk;
                    const bool first_partial = !(b == t0 * nk && e >= (t0 + 1) * nk);
                    sl = p * 2 + ((t0 == t) ? 0 : (first_partial ? 1 : 0));
                } else {
                    const long idx0 = (long)(t >> 3) * g;
                    sl = (((idx0 + p) << 3) | (t & 7)) * 2;
                }
                sk_slab_add(slabs + sl * SK_SLAB, cur, wv_e, lane_e);
            }
            if (tid == 0) __hip_atomic_store(counters + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the summed tile takes the same road as a whole one: its epilogue rides the next item's K loop
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) prev[i][j] = cur[i][j];
            pd.s = 0;
            pd.m0 = m0;
            pd.n0 = n0;
            if (ec.has_bias) {
                const int c0 = n0 + wn * 64 + (lane & 31);
                ec.bias0 = c0 < N ? aux[c0] : 0.f;
                ec.bias1 = c0 + 32 < N ? aux[c0 + 32] : 0.f;
            }
        }
    }
    // the last finished tile
    flush(prev);
}

// ------------------------------------------------------------------------------------------------
// wgrad: P[split][K1][N] = A[Mslice, K1]^T . B[Mslice, N];  Pc[split][N] = column sums of B[Mslice]
// grid = ntiles x splits (the tiles of one slice share an XCD), rows_per_split a multiple of 16.
// (Slabs in accumulator order with a matching reduce kernel were tried: 16-byte slab stores, but the reduce then scatters
// four rows per thread -- 135 vs 124 us on frame1's wgrad, 59 vs 55 on frame4's; not kept.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, SK_WGCU) void gemm_sk_tn_kernel(RowsD A, RowsD Bd, float* __restrict__ P, float* __restrict__ Pc,
                                                                  long M, int K1, int N, int tiles_n, int ntiles, long rows_per_split) {
    extern __shared__ __attribute__((aligned(16))) float sk_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)sk_smem);
    // block -> (slice, tile) through the XCD-chunk remap: all tiles of a slice on ONE XCD (block % 8), so its panels are fetched
    // by one L2 (round 3: 2 x the input bytes at the fabric with the tiles of a slice dealt over all eight)
    const unsigned vb = xcd_chunk_id(blockIdx.x, gridDim.x);
    const int tile = (int)(vb % (unsigned)ntiles);
    const int split = (int)(vb / (unsigned)ntiles);
    const int tn = tile % tiles_n, tk = tile / tiles_n;
    const int i0 = tk * SK_BM, n0 = tn * SK_BN;
    const long mbeg = (long)split * rows_per_split;
    long mend = mbeg + rows_per_split;
    if (mend > M) mend = M;
    const int n = (int)((mend - mbeg + SK_BK - 1) / SK_BK);
    const int prio_slot = (int)(blockIdx.x >> 8);

    SkOuterRows oa, ob;
    oa.init(A, i0, K1, mbeg, mend, lane, wv, wm);
    ob.init(Bd, n0, N, mbeg, mend, lane, wv, wn);
    const int tail_step = ((mend - mbeg) % SK_BK != 0) ? n - 1 : -1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float csum = 0.f;
    const bool do_csum = (Pc != nullptr) && tk == 0 && tid < SK_BN;
    auto issue = [&](int pc, int step, unsigned d) {
        if (pc < 2) oa.issue(pc, d, step == tail_step);
        else ob.issue(pc - 2, d, step == tail_step);
    };
    auto next = [&]() {};
    auto ra = [&](const float* st, int s2, float (&v)[2][4]) { oa.read(st, lane, s2, v); };
    // the bias gradient (column sums of dY) comes from the B stage in LDS: the sub-step 0 read of a stage adds its 16 rows
    auto rb = [&](const float* st, int s2, float (&v)[2][4]) {
        ob.read(st, lane, s2, v);
        if (do_csum && s2 == 0) {
#pragma unroll
            for (int kk = 0; kk < SK_BK; ++kk) csum += st[kk * 128 + tid];
        }
    };
    sk_kloop<SK_STAGES, SK_WGCU>(sk_smem, lds0, wv, n, prio_slot, issue, next, ra, rb, acc, SkNoDrain{}, [] {}, SkNoEpiOps{});

    float* Pd = P + (long)split * K1 * N;
    store_partial_blocks<2, 2>(Pd, acc, i0, n0, wm, wn, lane, K1, N);
    if (do_csum && n0 + tid < N) Pc[(long)split * N + n0 + tid] = csum;
}

}  // namespace
