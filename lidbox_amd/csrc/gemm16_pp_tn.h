// gemm16_pp_tn.h -- 256 x 256, eight-wave ping-pong tile of the bf16-storage wgrad (lidbox_gemm_bf16s_tn: the weight gradient of
// the reference's Conv1D / Dense layers, xvector.py:38-43,53-64 under tf.GradientTape, see gemm_bf16.hip), included by gemm_bf16.hip.
//
// P[slice][K1][N] = A[rows of the slice, K1]^T . B[rows of the slice, N]: the contraction index is the ROW of both operands, so a
// K step is 64 rows of A (256 columns = 512 B each) and 64 rows of B.  Same skeleton as gemm16_pp.h -- two groups of four waves
// one phase apart, an A ring of three stages and a B ring of two (5 x 32 KB = the CU's LDS), LDS-DMA pieces that stay in flight
// across the phase barriers and are waited for with counted vmcnt -- with these differences:
//   * a DMA piece (one wave instruction, 1 KB) is TWO whole rows of 512 B: lane l -> row l >> 5, 16-byte chunk l & 31.  The rows
//     keep their memory order in LDS (stride 512 B); what is swizzled is the chunk: position q of row r holds source chunk
//     q ^ 4 (r & 3), applied on the source side (the DMA's destination is lane-linear);
//   * the MFMA operands (lane -> column, 8 consecutive rows) come out of LDS through ds_read_b64_tr_b16 as in gemm16s_tn_kernel:
//     a 32-lane service group reads 4 rows x 64 B, which the chunk swizzle spreads over the four 64-byte quarters of the 256-byte
//     bank row (unswizzled, the four rows of a 512-byte stride would hit the same quarter);
//   * the row offsets of a step's pieces are not a table: the lane keeps (row, utterance, row inside the utterance) of its first
//     piece for the A and the B stream and steps them by 64 rows per issue (implicit-row descriptors: utterances of rpb rows,
//     batch stride bs, row stride rs);
//   * the bias gradient (column sums of B) rides on the matrix pipe: on the tiles of the first K1 panel every wave feeds ONE of
//     its B blocks against an all-ones A operand (one MFMA in 9), every row of that product is the column sum.
// The slices are the launch's second dimension folded into grid.x (slice-major through the XCD-chunk remap: the tiles of a slice
// share an L2).  Raw fp32 sums go to P through pp_store_tile (16-byte stores); the fixed-order slice sum is the caller's carried
// reduce job, as for gemm16s_tn_kernel.
#pragma once

#include "gemm16_pp.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_pp;

#ifndef LBX_PPT_ABLATE
#define LBX_PPT_ABLATE 0                                // measurement builds only (wrong results): 1 no DMA after the prologue, 2 operand fetches
#endif                                                  // of the first step only, 4 no MFMAs, 16 no epilogue, 32 every step fetches the same rows, 64 DMA addresses but no DMA, 128 no column sums

constexpr int PPT_BM = 32;                            // contraction rows per K step (= per LDS stage)
constexpr int PPT_BT = 256;                           // tile edge (K1 and N)
constexpr int PPT_ROW_BYTES = PPT_BT * 2;             // 512
constexpr int PPT_ST = PPT_BM * PPT_ROW_BYTES;        // 16 KB per operand stage
constexpr int PPT_D = 5;                              // stages per operand ring: a step's pieces are issued four steps ahead
constexpr int PPT_LDS_BYTES = 2 * PPT_D * PPT_ST;     // 160 KB
constexpr int PPT_NP = 2;                             // DMA pieces per wave, step and operand

// The lane's pieces of the NEXT step to issue: per piece the row inside its utterance and the two operands' byte offsets, stepped by
// 32 rows per issue with adds only (one utterance counter for both operands: the host checks that they have the same utterance
// length, and that every offset fits 32 bits).
struct PptStream {
    const float* sa;                        // wave-uniform byte bases: X.base + first column of the tile
    const float* sb;
    unsigned astep, bstep, awrap, bwrap;    // bytes per 32 rows; bytes from the end of an utterance to the start of the next
    int rpb, m;                             // rows per utterance; row of piece 0
    int tt[PPT_NP];
    unsigned va[PPT_NP], vb[PPT_NP];

    __device__ __forceinline__ void init(const RowsH& A, const RowsH& B, long mbeg, int i0, int K1, int n0, int N, int lane, int wv) {
        sa = sk_uniform(reinterpret_cast<const float*>(A.base + i0));
        sb = sk_uniform(reinterpret_cast<const float*>(B.base + n0));
        const unsigned ars2 = (unsigned)(A.rs * 2), brs2 = (unsigned)(B.rs * 2);
        const bool flat = A.batch == 1;
        rpb = flat ? 0x7fffffff : A.rpb;
        astep = PPT_BM * ars2;
        bstep = PPT_BM * brs2;
        awrap = flat ? 0u : (unsigned)(A.bs * 2) - (unsigned)rpb * ars2;
        bwrap = flat ? 0u : (unsigned)(B.bs * 2) - (unsigned)rpb * brs2;
        m = (int)mbeg + wv * (2 * PPT_NP) + (lane >> 5);
#pragma unroll
        for (int i = 0; i < PPT_NP; ++i) {
            const int mi = m + 2 * i;
            const int bi = flat ? 0 : mi / rpb;
            tt[i] = mi - (flat ? 0 : bi * rpb);
            const int chunk = (lane & 31) ^ (4 * ((2 * i + (lane >> 5)) & 3));
            const int c = chunk * 8;                                   // column inside the tile; past the matrix: any valid one
            va[i] = (unsigned)bi * (unsigned)(A.bs * 2) + (unsigned)tt[i] * ars2 + (unsigned)((i0 + c < K1 ? c : 0) * 2);
            vb[i] = (unsigned)bi * (unsigned)(B.bs * 2) + (unsigned)tt[i] * brs2 + (unsigned)((n0 + c < N ? c : 0) * 2);
        }
    }
    // the wave's pieces (rows 4 wv + 2 i + {0, 1} of the step) of both operands into their stages at LDS byte addresses sta / stb
    __device__ __forceinline__ void issue_step(unsigned sta, unsigned stb, int wv, int mend, bool tail) {
#pragma unroll
        for (int i = 0; i < PPT_NP; ++i) {
            const unsigned off = (unsigned)(wv * PPT_NP + i) * 1024u;
            if (LBX_PPT_ABLATE & 64) {                                 // addresses only
                asm volatile("" ::"v"(va[i]), "v"(vb[i]), "s"(sta + off));
            } else if (tail) {
                const bool in = m + 2 * i < mend;
                sk_dma_f(in ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(sa) + va[i]) : g_sk_zero, sta + off);
                sk_dma_f(in ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(sb) + vb[i]) : g_sk_zero, stb + off);
            } else {
                sk_dma_s(sa, va[i], sta + off);
                sk_dma_s(sb, vb[i], stb + off);
            }
        }
        if (LBX_PPT_ABLATE & 32) return;                               // the same rows every step: L2 hits
        m += PPT_BM;
#pragma unroll
        for (int i = 0; i < PPT_NP; ++i) {
            tt[i] += PPT_BM;
            va[i] += astep;
            vb[i] += bstep;
            while (tt[i] >= rpb) { tt[i] -= rpb; va[i] += awrap; vb[i] += bwrap; }
        }
    }
};

__device__ __forceinline__ bf16x8 ppt_read(const char* p) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_pp*)(p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_pp*)(p + 4 * PPT_ROW_BYTES));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// own pieces of everything but the `ahead` most recently issued steps have landed
__device__ __forceinline__ void ppt_wait_steps(int ahead) {
    if (ahead >= 3) sk_wait_vm<3 * 2 * PPT_NP>();
    else if (ahead == 2) sk_wait_vm<2 * 2 * PPT_NP>();
    else if (ahead == 1) sk_wait_vm<1 * 2 * PPT_NP>();
    else sk_wait_vm<0>();
}

__global__ __launch_bounds__(512, 2) void gemm16s_tn_pp_kernel(RowsH A, RowsH Bd, float* __restrict__ P, float* __restrict__ Pc, long M,
                                                                int K1, int N, int tiles_n, int ntiles, long rows_per_split) {
    constexpr int MI = 4, NJ = 2, KH = PPT_BM / 16;           // wave tile 128 x 64; two 16-row slices per step
    constexpr int D = PPT_D;
    static_assert(D == 5, "ppt_wait_steps counts up to D - 2 = 3 steps in flight");
    constexpr int B_RING = D * PPT_ST;
    extern __shared__ __attribute__((aligned(16))) char smem16q[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wv >> 2;
    const int wm = wv >> 2, wn = wv & 3;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem16q);
    const unsigned vb = xcd_chunk_id(blockIdx.x, gridDim.x);
    const int tile = (int)(vb % (unsigned)ntiles);
    const int split = (int)(vb / (unsigned)ntiles);
    const int tn = tile % tiles_n, tk = tile / tiles_n;
    const int i0 = tk * PPT_BT, n0 = tn * PPT_BT;
    const long mbeg = (long)split * rows_per_split;
    long mend_l = mbeg + rows_per_split;
    if (mend_l > M) mend_l = M;
    const int mend = (int)mend_l;
    const int n = (int)((mend_l - mbeg + PPT_BM - 1) / PPT_BM);
    const bool ragged = (mend_l - mbeg) % PPT_BM != 0;        // the last step has rows past the slice: zero source
    const int rowA = wm * 128, rowB = wn * 64;                // this wave's first column inside an A / B stage

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x16 csacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) csacc[r] = 0.f;
    const bool do_csum = (Pc != nullptr) && tk == 0;          // workgroup-uniform
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    PptStream ps;
    ps.init(A, Bd, mbeg, i0, K1, n0, N, lane, wv);
    auto issue = [&](int step, int stage) {
        if ((LBX_PPT_ABLATE & 1) && step >= D - 1) return;
        ps.issue_step(lds0 + (unsigned)(stage * PPT_ST), lds0 + (unsigned)(B_RING + stage * PPT_ST), wv, mend, ragged && step == n - 1);
    };

    // operand fetch addresses: lane 16 g + j reads the 4-column piece (j & 3) of row 8 (g >> 1) + (j >> 2) in the block of columns
    // 16 (g & 1) .. +15 -- chunk 2 (g & 1) + ((j & 3) >> 1) of the block's four, XOR-swizzled by the row's low two bits
    unsigned aoff[MI], boff[NJ];
    {
        const int g = lane >> 4, j = lane & 15;
        const unsigned base = (unsigned)((8 * (g >> 1) + (j >> 2)) * PPT_ROW_BYTES + 8 * (j & 1));
        const unsigned cl = (unsigned)(2 * (g & 1) + ((j & 3) >> 1)), sw = (unsigned)(4 * ((j >> 2) & 3));
#pragma unroll
        for (int bi = 0; bi < MI; ++bi) aoff[bi] = base + ((((unsigned)(rowA >> 3) + 4u * bi) ^ sw) + cl) * 16u;
#pragma unroll
        for (int bj = 0; bj < NJ; ++bj) boff[bj] = B_RING + base + ((((unsigned)(rowB >> 3) + 4u * bj) ^ sw) + cl) * 16u;
    }

    auto body = [&](auto grp_tag) {
        constexpr int G = decltype(grp_tag)::value;
        bf16x8 a[MI][KH], b[NJ][KH];
        bool first_load = true;
        auto load = [&](int stage) {
            if (LBX_PPT_ABLATE & 2) {
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
            const char* st = smem16q + stage * PPT_ST;
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) {
#pragma unroll
                for (int bj = 0; bj < NJ; ++bj) b[bj][kk] = ppt_read(st + kk * 16 * PPT_ROW_BYTES + boff[bj]);
#pragma unroll
                for (int bi = 0; bi < MI; ++bi) a[bi][kk] = ppt_read(st + kk * 16 * PPT_ROW_BYTES + aoff[bi]);
            }
        };
        auto comp = [&]() {
            if (LBX_PPT_ABLATE & 4) {
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
            for (int kk = 0; kk < KH; ++kk) {
#pragma unroll
                for (int bi = 0; bi < MI; ++bi)
#pragma unroll
                    for (int bj = 0; bj < NJ; ++bj)
                        acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[bi][kk], b[bj][kk], acc[bi][bj], 0, 0, 0);
                if (do_csum && !(LBX_PPT_ABLATE & 128)) csacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, wm ? b[1][kk] : b[0][kk], csacc, 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        };
        // prologue: steps 0 .. D - 2 in flight, step 0 landed
        const int npro = n < D - 1 ? n : D - 1;
        for (int s = 0; s < npro; ++s) issue(s, s);
        ppt_wait_steps(npro - 1);
        pp_barrier();
        if (G == 1) pp_barrier();                              // phase 0 belongs to group 0 alone
        int stage = 0;                                         // stage of the step being computed; step t + D - 1 goes to stage - 1
        for (int t = 0; t < n; ++t) {
            load(stage);                                       // operand fetches first: they complete under the DMA issues
            const int prev = stage == 0 ? D - 1 : stage - 1;   // read last for step t - 1, by both groups before this phase
            if (t + D - 1 < n) issue(t + D - 1, prev);
            pp_wait_lds();
            // own pieces of step t + 1 before the barrier that opens the phase in which group 0 fetches them
            const int ahead = (n - 1 < t + D - 1 ? n - 1 : t + D - 1) - (t + 1);
            if (G == 1 && t + 1 < n) ppt_wait_steps(ahead);
            pp_barrier();
            comp();
            if (G == 0) {
                if (t + 1 < n) ppt_wait_steps(ahead);
                pp_barrier();
            } else if (t + 1 < n) {
                pp_barrier();
            }
            stage = stage + 1 == D ? 0 : stage + 1;
        }
    };
    if (grp == 0) body(IntTag<0>{});
    else body(IntTag<1>{});

    if (LBX_PPT_ABLATE & 16) {
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
    // raw sums to P[split][i][n] through this wave's LDS strip (no wave reads a stage any more, no DMA is in flight)
    const RowsOutD none{nullptr, 0, (long)N, 1, K1};
    const unsigned nobits[MI] = {};
    pp_store_tile<MI, NJ>(acc, reinterpret_cast<float*>(smem16q + wv * PP_EPI_BYTES), (long)i0 + rowA, n0 + rowB, lane, 0, (long)K1, N,
                          LIDBOX_EPI_NONE, nullptr, none, P, split, nullptr, nullptr, nobits, false, 1);
    if (do_csum && lane < 32) {
        const int c = n0 + rowB + wm * 32 + lane;             // row 0 of the ones-product: lanes 0 .. 31, element 0
        if (c < N) Pc[(long)split * N + c] = csacc[0];
    }
}

}  // namespace
