// gemm_bf16.hip -- bf16-MFMA GEMM family (gfx950): the "bf16 compute / fp32 master" variant of the
// Conv1D / Dense forward, dgrad and wgrad GEMMs in gemm.hip (BASELINE config 5).
//
// Replaces (reference file:line): the same call sites as gemm.hip --
//   lidbox/models/xvector.py:38-43,53-64 (frame_layer / segment_layer), lidbox/models/cnn.py:32-41
// when the model is run under a bfloat16 compute policy.  The reference has no such switch of its
// own (Keras' mixed_bfloat16 policy would be set outside lidbox); the numerical contract here is:
//   * with the fp32-source entry points (lidbox_gemm_bf16_nn/_nt/_tn) every tensor in HBM stays fp32 (weights = the fp32
//     master copy, activations, gradients); the storage entry points further down read and write bf16 shadows instead;
//   * both GEMM operands are rounded to bfloat16, round-to-nearest-even, as they are staged into
//     LDS (v_cvt_pk_bf16_f32);
//   * products are accumulated in fp32 by v_mfma_f32_32x32x16_bf16; bias / ReLU / mask / accumulate
//     epilogues, bias gradients, pooling, losses and Adam are unchanged fp32 code.
// So a result equals the fp32 GEMM of the bf16-rounded operands up to fp32 summation order, which
// is what tests/test_gemm_bf16_gpu.py checks.
//
// Design
//   * Same implicit-row addressing as gemm.hip (lidbox_rows_t), same C ABI shape, same split
//     decompositions with fixed-order reduce kernels (gemm_shared.h).
//   * Workgroup tile 128 x 128, K depth 32 per LDS tile = two MFMAs (K = 16 each) per 32x32 block;
//     4 waves as 2 x 2, each 64 x 64 = 2 x 2 blocks.  At 32 cycles per MFMA a K-step is 256 matrix
//     cycles per wave against 32 KB of fp32 operand data, i.e. the loop is bound by the global-load
//     path, not by the 2.5 PFLOP/s pipe: keeping fp32 in HBM trades peak rate for leaving every
//     other kernel and buffer untouched.  Two tiles are kept in flight per workgroup (register ring
//     of depth 2, ~190 VGPRs, 2 workgroups per CU) because one 256-cycle K-step cannot cover a loaded
//     global round trip (measured in tools/micro/bf16_loop.hip).
//   * LDS tiles are [row][k] bf16 with an 80-byte row stride for BOTH operands: the MFMA operand
//     fetch (lane -> row lane&31, 8 consecutive k at 8*(lane>>5)) is one ds_read_b128, conflict-free
//     over the four 16-lane groups the hardware serves it in; staging stores are ds_write_b64 of
//     4 consecutive k, and each 16-lane store group covers 8 k-quads x 2 rows that differ by 4
//     (80*4 = 64 mod 128 bytes) -> conflict-free too.
//       - operands whose contraction index is contiguous in HBM (A of NN/NT, B of NT): one
//         float4 -> 4 bf16, no transpose;
//       - operands whose contraction index is the row (B of NN, both operands of TN): a thread
//         loads a 4(k) x 4(col) block as four float4 and writes its transpose.
//   * wgrad's bias gradient (column sums of dY) is accumulated in fp32 from the staged float4
//     registers BEFORE rounding, so it is bit-for-bit the quantity the fp32 path computes up to
//     summation order.
//   * bf16-STORAGE variants further down (gemm16s_rows_kernel, gemm16s_tn_kernel): operands that already are bf16 in HBM
//     (shadows written by the producing epilogues) go to LDS unchanged; wgrad's row-contracted operands come out of LDS
//     through ds_read_b64_tr_b16.
// Requirements (checked, LIDBOX_E_INVALID otherwise): 16-byte aligned bases, K (K1) and N multiples
// of 4, row/batch strides multiples of 4 -- true of every layer of the x-vector / CNN models.
// Roofline: MFMA bf16 dense, 2.5 PFLOP/s (MI355X_MICROARCH.md); practical bound = L2->LDS traffic.
#include <string.h>

#include <atomic>

#include "gemm_shared.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef LBX16_TAIL_MIN_ROUNDS
#define LBX16_TAIL_MIN_ROUNDS 1
#endif
#ifndef LBX16_TAIL_MIN_K
#define LBX16_TAIL_MIN_K 1024              // tail split of the rows launches: measured a loss at K = 512 (bs 256), a gain at K >= 1500
#endif
constexpr int BK = 32;        // contraction depth of one LDS tile
constexpr int BT = 128;       // tile rows (M side) = tile columns (N side)
constexpr int LDT = 40;       // LDS row stride in bf16: 32 + 8 pad = 80 bytes
constexpr int TILE_ELEMS = BT * LDT;

// Staging registers are native 4-float vectors (not HIP's float4 struct): the 4x4 transpose below is then
// plain register picks -- with struct temporaries the compiler built it through scratch memory.
__device__ __forceinline__ bf16x4 to_bf16(f32x4 x) { return __builtin_convertvector(x, bf16x4); }
__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// Staging payload of one operand tile per thread: four float4 (16 VGPRs).  It lives OUTSIDE the loaders so
// that the kernels can keep two tiles in flight (register ring of depth 2, see the K loops).
struct Tile4 {
    f32x4 v0, v1, v2, v3;
};

// ---- operand with the contraction index contiguous in HBM: 128 rows x 32 k.
//      thread -> k-quad (tid & 7) of rows r0 + 32*{0,1,2,3}; a wave's load covers 8 rows x 128 B.
//      r0 swaps bits 0 and 2 of (tid >> 3) so that the two rows of a 16-lane store group differ by 4.
struct KInner16 {
    const float* ptr[4];
    int kq, r0, k;

    __device__ __forceinline__ void init(const RowsD& rows, long row_base, long nrows, int tid, int kbeg) {
        kq = tid & 7;
        const int rs = tid >> 3;
        r0 = (rs & 0x1a) | ((rs & 1) << 2) | ((rs >> 2) & 1);
        k = kbeg + 4 * kq;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const long r = row_base + r0 + 32 * p;
            // rows outside the matrix are clamped to row 0: finite data that only reaches outputs never stored
            ptr[p] = rows.base + (r < nrows ? row_offset(rows, (unsigned)r) : 0) + k;
        }
    }
    template <bool CHECK>
    __device__ __forceinline__ void load(Tile4& t, int kend) {
        const bool in = !CHECK || k < kend;
        t.v0 = in ? ldg4(ptr[0]) : zero4();
        t.v1 = in ? ldg4(ptr[1]) : zero4();
        t.v2 = in ? ldg4(ptr[2]) : zero4();
        t.v3 = in ? ldg4(ptr[3]) : zero4();
#pragma unroll
        for (int p = 0; p < 4; ++p) ptr[p] += BK;
        k += BK;
    }
    __device__ __forceinline__ void store(__bf16* tile, const Tile4& t) const {
        __bf16* d = tile + r0 * LDT + 4 * kq;
        *reinterpret_cast<bf16x4*>(d + 0 * 32 * LDT) = to_bf16(t.v0);
        *reinterpret_cast<bf16x4*>(d + 1 * 32 * LDT) = to_bf16(t.v1);
        *reinterpret_cast<bf16x4*>(d + 2 * 32 * LDT) = to_bf16(t.v2);
        *reinterpret_cast<bf16x4*>(d + 3 * 32 * LDT) = to_bf16(t.v3);
    }
};

// ---- operand whose ROWS are the contraction index: 32 k x 128 columns.
//      thread -> k rows 4*kq .. 4*kq+3 (kq = lane & 7), columns 4*nq .. 4*nq+3 (nq = tid >> 3);
//      each of the four loads of a wave covers 8 k-rows x 128 B.  store() writes the 4x4 transpose.
struct KOuter16 {
    int kq, nq, cload;

    __device__ __forceinline__ void init(int tid, int col0, int ncols) {
        kq = tid & 7;
        nq = tid >> 3;
        const int c = col0 + 4 * nq;
        cload = c < ncols ? c : 0;              // clamped column (results never stored)
    }
    __device__ __forceinline__ void store(__bf16* tile, const Tile4& t) const {
        __bf16* d = tile + (4 * nq) * LDT + 4 * kq;
        *reinterpret_cast<bf16x4*>(d + 0 * LDT) = to_bf16(f32x4{t.v0.x, t.v1.x, t.v2.x, t.v3.x});
        *reinterpret_cast<bf16x4*>(d + 1 * LDT) = to_bf16(f32x4{t.v0.y, t.v1.y, t.v2.y, t.v3.y});
        *reinterpret_cast<bf16x4*>(d + 2 * LDT) = to_bf16(f32x4{t.v0.z, t.v1.z, t.v2.z, t.v3.z});
        *reinterpret_cast<bf16x4*>(d + 3 * LDT) = to_bf16(f32x4{t.v0.w, t.v1.w, t.v2.w, t.v3.w});
    }
    // fp32 column sums of a staged block (wgrad's bias gradient)
    static __device__ __forceinline__ f32x4 colsum(const Tile4& t) { return (t.v0 + t.v1) + (t.v2 + t.v3); }
};

// plain matrix B[K][N] (NN): row k at base + k*ld
struct KOuterPlain : KOuter16 {
    const float* ptr;
    long ld, step;
    int k;
    __device__ __forceinline__ void init_plain(const float* base, long ld_, int kbeg, int col0, int ncols, int tid) {
        init(tid, col0, ncols);
        ld = ld_;
        k = kbeg + 4 * kq;
        ptr = base + (long)k * ld + cload;
        step = (long)BK * ld;
    }
    template <bool CHECK>
    __device__ __forceinline__ void load(Tile4& t, int kend) {
        t.v0 = (!CHECK || k + 0 < kend) ? ldg4(ptr) : zero4();
        t.v1 = (!CHECK || k + 1 < kend) ? ldg4(ptr + ld) : zero4();
        t.v2 = (!CHECK || k + 2 < kend) ? ldg4(ptr + 2 * ld) : zero4();
        t.v3 = (!CHECK || k + 3 < kend) ? ldg4(ptr + 3 * ld) : zero4();
        ptr += step;
        k += BK;
    }
};

// implicit rows (TN): contraction row m at row_offset(rows, m); the first of the thread's four rows is
// tracked incrementally (adds only), the other three follow at +rs unless they cross an utterance.
struct KOuterRows : KOuter16 {
    RowsD rows;
    long m, off, step, wrap;
    unsigned t, rpb;
    __device__ __forceinline__ void init_rows(const RowsD& r, long mbeg, int col0, int ncols, int tid) {
        init(tid, col0, ncols);
        rows = r;
        m = mbeg + 4 * kq;
        rpb = r.batch == 1 ? 0xffffffffu : (unsigned)r.rpb;
        const unsigned b = r.batch == 1 ? 0u : (unsigned)m / rpb;
        t = (unsigned)m - (r.batch == 1 ? 0u : b * rpb);
        off = (long)b * r.bs + (long)t * r.rs;
        step = (long)BK * r.rs;
        wrap = r.batch == 1 ? 0 : r.bs - (long)r.rpb * r.rs;
    }
    template <bool CHECK>
    __device__ __forceinline__ f32x4 fetch(int j, bool fast, long mend) const {
        if (CHECK && m + j >= mend) return zero4();
        const long o = fast ? off + j * rows.rs : row_offset(rows, (unsigned)(m + j));
        return ldg4(rows.base + o + cload);
    }
    template <bool CHECK>
    __device__ __forceinline__ void load(Tile4& x, long mend) {
        const bool fast = t + 3 < rpb;
        x.v0 = fetch<CHECK>(0, fast, mend);
        x.v1 = fetch<CHECK>(1, fast, mend);
        x.v2 = fetch<CHECK>(2, fast, mend);
        x.v3 = fetch<CHECK>(3, fast, mend);
        m += BK;
        t += BK;
        off += step;
        while (t >= rpb) { t -= rpb; off += wrap; }
    }
};

// 64 x 64 per wave: 2 x 2 blocks, two K=16 MFMAs each, over one LDS tile pair
__device__ __forceinline__ void mma_tile16(const __bf16* As, const __bf16* Bs, int wm, int wn, int lane,
                                           f32x16 (&acc)[2][2]) {
    const int h = lane >> 5, l = lane & 31;
    const __bf16* ap = As + (wm * 64 + l) * LDT + 8 * h;
    const __bf16* bp = Bs + (wn * 64 + l) * LDT + 8 * h;
    bf16x8 a[2][2], b[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[ks][i] = *reinterpret_cast<const bf16x8*>(ap + i * 32 * LDT + ks * 16);
            b[ks][i] = *reinterpret_cast<const bf16x8*>(bp + i * 32 * LDT + ks * 16);
        }
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// ------------------------------------------------------------------------------------------------
// C[M,N] = epi(A[M,K] . B)    B_KINNER = false: B[K][N] (NN)   true: B[N][K] (NT)
// grid.x = tiles (XCD-chunk remapped), grid.y = K splits (partials to P, rows_reduce_kernel finishes)
// ------------------------------------------------------------------------------------------------
template <bool B_KINNER>
__global__ __launch_bounds__(256, 2) void gemm16_rows_kernel(RowsD A, const float* __restrict__ Bm, long ldb, RowsOutD Cd,
                                                          float* __restrict__ P, long m_beg, long M, int K, int N, int epi,
                                                          const float* __restrict__ aux, int tiles_n, unsigned ntiles,
                                                          int k_per_split) {
    // this launch covers rows [m_beg, M)
    __shared__ __attribute__((aligned(16))) __bf16 As[2][TILE_ELEMS];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][TILE_ELEMS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned chunk = xcd_chunk_id(blockIdx.x, ntiles);
    const int tn = chunk % tiles_n;
    const long m0 = m_beg + (long)(chunk / tiles_n) * BT;
    const int n0 = tn * BT;
    const int split = blockIdx.y;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);

    KInner16 la;
    la.init(A, m0, M, tid, kbeg);
    KInner16 lbi;
    KOuterPlain lbo;
    const RowsD Brows{Bm, 0, ldb, 1, 0};
    if (B_KINNER) lbi.init(Brows, n0, N, tid, kbeg);
    else lbo.init_plain(Bm, ldb, kbeg, n0, N, tid);

    f32x16 acc[2][2];
    zero_acc(acc);

    // Register ring of depth 2: while tile kt is multiplied out of LDS, tile kt+1 sits in registers (loaded one
    // step ago, so its wait is short) and the loads of tile kt+2 are issued.  A 32-deep bf16 K-step is only
    // 256 matrix cycles, far less than a loaded global round trip, so one tile in flight leaves the loop
    // latency-bound (tools/micro/bf16_loop.hip: +58 % with one workgroup per CU, +13..22 % with 2-3).
    Tile4 ra[2], rb[2];
    const int nk = (kend - kbeg + BK - 1) / BK;
    auto fetch = [&](Tile4& a, Tile4& b, int t) {                // tile t; only the last tile can be partial
        if (t + 1 < nk) {
            la.load<false>(a, kend);
            if (B_KINNER) lbi.load<false>(b, kend);
            else lbo.load<false>(b, kend);
        } else {
            la.load<true>(a, kend);
            if (B_KINNER) lbi.load<true>(b, kend);
            else lbo.load<true>(b, kend);
        }
    };
    auto stage = [&](int buf, const Tile4& a, const Tile4& b) {
        la.store(As[buf], a);
        if (B_KINNER) lbi.store(Bs[buf], b);
        else lbo.store(Bs[buf], b);
    };
    if (nk > 0) {
        fetch(ra[0], rb[0], 0);
        if (nk > 1) fetch(ra[1], rb[1], 1);
        stage(0, ra[0], rb[0]);
    }
    __syncthreads();
    // step kt (parity PAR): registers [PAR^1] hold tile kt+1; tile kt+2 is loaded into registers [PAR]
#define LBX16_STEP(PAR)                                                   \
    {                                                                     \
        if (kt + 2 < nk) fetch(ra[PAR], rb[PAR], kt + 2);                 \
        mma_tile16(As[PAR], Bs[PAR], wm, wn, lane, acc);                  \
        if (kt + 1 < nk) stage((PAR) ^ 1, ra[(PAR) ^ 1], rb[(PAR) ^ 1]);  \
        __syncthreads();                                                  \
        ++kt;                                                             \
    }
    for (int kt = 0; kt < nk;) {
        LBX16_STEP(0)
        if (kt < nk) LBX16_STEP(1)
    }
#undef LBX16_STEP
    store_rows_tile<2, 2>(acc, m0, n0, wm, wn, lane, m_beg, M, N, epi, aux, Cd, P, split);
}

// ------------------------------------------------------------------------------------------------
// wgrad: P[split][K1][N] = A[Mslice, K1]^T . B[Mslice, N];  Pc[split][N] = fp32 column sums of B[Mslice]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm16_tn_kernel(RowsD A, RowsD Bd, float* __restrict__ P,
                                                        float* __restrict__ Pc, long M, int K1, int N, int tiles_n,
                                                        int ntiles, long rows_per_split) {
    __shared__ __attribute__((aligned(16))) __bf16 As[2][TILE_ELEMS];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][TILE_ELEMS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // consecutive block ids = the tiles of one M slice: they read the same A/B rows (L2 reuse)
    const int tile = blockIdx.x % ntiles;
    const int split = blockIdx.x / ntiles;
    const int tn = tile % tiles_n, tk = tile / tiles_n;
    const int i0 = tk * BT, n0 = tn * BT;
    const long mbeg = (long)split * rows_per_split;
    long mend = mbeg + rows_per_split;
    if (mend > M) mend = M;

    KOuterRows la, lb;
    la.init_rows(A, mbeg, i0, K1, tid);
    lb.init_rows(Bd, mbeg, n0, N, tid);

    f32x16 acc[2][2];
    zero_acc(acc);
    f32x4 csum = zero4();
    const bool do_csum = (Pc != nullptr) && tk == 0;

    Tile4 ra[2], rb[2];                                          // register ring of depth 2 (see gemm16_rows_kernel)
    const int nk = (int)((mend - mbeg + BK - 1) / BK);
    auto fetch = [&](Tile4& a, Tile4& b, int t) {
        if (t + 1 < nk) { la.load<false>(a, mend); lb.load<false>(b, mend); }
        else { la.load<true>(a, mend); lb.load<true>(b, mend); }
    };
    auto stage = [&](int buf, const Tile4& a, const Tile4& b) {
        if (do_csum) csum += KOuter16::colsum(b);                // fp32, before rounding; every tile exactly once
        la.store(As[buf], a);
        lb.store(Bs[buf], b);
    };
    if (nk > 0) {
        fetch(ra[0], rb[0], 0);
        if (nk > 1) fetch(ra[1], rb[1], 1);
        stage(0, ra[0], rb[0]);
    }
    __syncthreads();
#define LBX16_STEP(PAR)                                                   \
    {                                                                     \
        if (kt + 2 < nk) fetch(ra[PAR], rb[PAR], kt + 2);                 \
        mma_tile16(As[PAR], Bs[PAR], wm, wn, lane, acc);                  \
        if (kt + 1 < nk) stage((PAR) ^ 1, ra[(PAR) ^ 1], rb[(PAR) ^ 1]);  \
        __syncthreads();                                                  \
        ++kt;                                                             \
    }
    for (int kt = 0; kt < nk;) {
        LBX16_STEP(0)
        if (kt < nk) LBX16_STEP(1)
    }
#undef LBX16_STEP
    float* Pd = P + (long)split * K1 * N;
    store_partial_blocks<2, 2>(Pd, acc, i0, n0, wm, wn, lane, K1, N);
    if (do_csum) {
        // the 8 k-quad lanes of one column quad are lanes (lane & ~7) + 0..7: fixed-order butterfly
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            f32x4 other;
            other.x = __shfl_xor(csum.x, o, 64);
            other.y = __shfl_xor(csum.y, o, 64);
            other.z = __shfl_xor(csum.z, o, 64);
            other.w = __shfl_xor(csum.w, o, 64);
            csum += other;
        }
        const int c = n0 + 4 * lb.nq;
        if (lb.kq == 0 && c < N) *reinterpret_cast<f32x4*>(Pc + (long)split * N + c) = csum;
    }
}

// ------------------------------------------------------------------------------------------------
// bf16-STORAGE variant (lidbox_gemm_bf16s_nt): both operands already are bfloat16 in HBM with the contraction index
// contiguous -- A = the bf16 shadow of an activation / gradient buffer (same [B, pad + T, C] element layout as the fp32
// buffer, written by the producing GEMM's epilogue), B = a bf16 weight shadow [N][K].  Half the bytes of the fp32-source
// kernels on the L2 -> LDS path (their bound), no convert / transpose work in staging: a thread moves two 16-byte pieces
// (8 k each) per operand tile from global memory to LDS unchanged.  Same tile (128 x 128 x 32), LDS layout, MFMA loop,
// split decompositions and epilogues as gemm16_rows_kernel; the epilogue can write the bf16 shadow of C as well.
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef LBX16S_BK
#define LBX16S_BK 64                        // K depth of one LDS tile of the storage kernel: 64 bf16 = one 128-byte line per row
#endif
constexpr int BKS = LBX16S_BK;
constexpr int LDS_S = BKS + 8;              // LDS row stride in bf16 (80 or 144 bytes: conflict-free ds_read_b128 / ds_write_b128)
constexpr int TILE_S = BT * LDS_S;
constexpr int OCT = BKS / 8;                // 16-byte pieces per tile row (4 or 8 lanes cover one row)
constexpr int RPP = 256 / OCT;              // tile rows per pass of the 256 threads
constexpr int NPASS = BT / RPP;             // passes per operand tile (2 or 4)

struct RowsH {
    const __bf16* base;
    long bs, rs;
    int batch, rpb;
};

__device__ __forceinline__ long row_offset(const RowsH& r, unsigned m) {
    if (r.batch == 1) return (long)m * r.rs;
    const unsigned b = m / (unsigned)r.rpb;
    return (long)b * r.bs + (long)(m - b * (unsigned)r.rpb) * r.rs;
}

struct TileS {
    u32x4 v[NPASS];
};

// thread -> 16-byte piece (tid % OCT) of rows r0 + RPP * {0 .. NPASS-1}; with BKS = 64 the 8 lanes of a row fetch one whole
// 128-byte line (a wave-load touches 8 lines; with 32-deep tiles it touched 16 half lines, which the L1/L2 path serves at
// half the rate -- tools/micro/l2_rate.hip: 28 vs 52 B/clk/CU).  r0 permutes the low row bits so that the rows of an 8-lane
// ds_write_b128 group land on different banks.
struct KInner16S {
    const __bf16* ptr[NPASS];
    int ko, r0, k;

    __device__ __forceinline__ void init(const RowsH& rows, long row_base, long nrows, int tid, int kbeg) {
        ko = tid % OCT;
        const int rs = tid / OCT;
        r0 = OCT == 4 ? ((rs & ~7) | ((rs & 1) << 2) | ((rs >> 1) & 3)) : rs;
        k = kbeg + 8 * ko;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const long r = row_base + r0 + RPP * p;
            ptr[p] = rows.base + (r < nrows ? row_offset(rows, (unsigned)r) : 0) + k;     // outside rows: row 0 (never stored)
        }
    }
    template <bool CHECK>
    __device__ __forceinline__ void load(TileS& t, int kend) {
        const bool in = !CHECK || k < kend;                     // K is a multiple of 8: an octet is inside or outside as a whole
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            t.v[p] = in ? *reinterpret_cast<const u32x4*>(ptr[p]) : z;
            ptr[p] += BKS;
        }
        k += BKS;
    }
    __device__ __forceinline__ void store(__bf16* tile, const TileS& t) const {
        __bf16* d = tile + r0 * LDS_S + 8 * ko;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) *reinterpret_cast<u32x4*>(d + p * RPP * LDS_S) = t.v[p];
    }
};

// 64 x 64 per wave over one LDS tile pair of the storage kernel: BKS / 16 MFMAs per 32 x 32 block
__device__ __forceinline__ void mma_tile16s(const __bf16* As, const __bf16* Bs, int wm, int wn, int lane, f32x16 (&acc)[2][2]) {
    const int h = lane >> 5, l = lane & 31;
    const __bf16* ap = As + (wm * 64 + l) * LDS_S + 8 * h;
    const __bf16* bp = Bs + (wn * 64 + l) * LDS_S + 8 * h;
    constexpr int NKS = BKS / 16;
    bf16x8 a[2][2], b[2][2];                                     // operand registers double-buffered over the k-slices
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a[0][i] = *reinterpret_cast<const bf16x8*>(ap + i * 32 * LDS_S);
        b[0][i] = *reinterpret_cast<const bf16x8*>(bp + i * 32 * LDS_S);
    }
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks + 1 < NKS) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[nxt][i] = *reinterpret_cast<const bf16x8*>(ap + i * 32 * LDS_S + (ks + 1) * 16);
                b[nxt][i] = *reinterpret_cast<const bf16x8*>(bp + i * 32 * LDS_S + (ks + 1) * 16);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
}

#ifndef LBX16S_DEPTH
#define LBX16S_DEPTH 2                      // operand tiles in flight per workgroup (register ring)
#endif
#ifndef LBX16S_WAVES
#define LBX16S_WAVES (LBX16S_BK == 64 ? 2 : 3)   // waves per SIMD: 73 KB of LDS per workgroup at BKS = 64 (2 per CU), 40 KB at 32 (3)
#endif
__global__ __launch_bounds__(256, LBX16S_WAVES) void gemm16s_rows_kernel(RowsH A, RowsH Bw, RowsOutD Cd, unsigned short* __restrict__ C16,
                                                              float* __restrict__ P, long m_beg, long M, int K, int N, int epi,
                                                              const float* __restrict__ aux, int tiles_n, unsigned ntiles,
                                                              int k_per_split, const unsigned short* __restrict__ mask16, ReduceJobs rj) {
    extern __shared__ __attribute__((aligned(16))) char smem16s[];      // 4 tiles: 40 KB (BKS 32) or 72 KB (BKS 64, above the static limit)
    if (blockIdx.x < rj.total) {                                        // carried reduces (gemm_shared.h: ReduceJobs)
        if (blockIdx.y == 0) reduce_jobs_run(rj, blockIdx.x);
        return;
    }
    __bf16 (*As)[TILE_S] = reinterpret_cast<__bf16 (*)[TILE_S]>(smem16s);
    __bf16 (*Bs)[TILE_S] = reinterpret_cast<__bf16 (*)[TILE_S]>(smem16s + 2 * TILE_S * sizeof(__bf16));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned chunk = xcd_chunk_id(blockIdx.x - rj.total, ntiles);
    const int tn = chunk % tiles_n;
    const long m0 = m_beg + (long)(chunk / tiles_n) * BT;
    const int n0 = tn * BT;
    const int split = blockIdx.y;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);

    KInner16S la, lb;
    la.init(A, m0, M, tid, kbeg);
    lb.init(Bw, n0, N, tid, kbeg);

    f32x16 acc[2][2];
    zero_acc(acc);

    // register ring of depth LBX16S_DEPTH: while tile kt is multiplied out of LDS, tiles kt+1 .. kt+DEPTH-1 sit in registers
    // and the loads of tile kt+DEPTH are issued (16 VGPRs per tile in flight)
    constexpr int D = LBX16S_DEPTH;
    TileS ra[D], rb[D];
    const int nk = (kend - kbeg + BKS - 1) / BKS;
    auto fetch = [&](TileS& a, TileS& b, int t) {
        if (t + 1 < nk) { la.load<false>(a, kend); lb.load<false>(b, kend); }
        else { la.load<true>(a, kend); lb.load<true>(b, kend); }
    };
    if (nk > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < nk) fetch(ra[d], rb[d], d);
        la.store(As[0], ra[0]);
        lb.store(Bs[0], rb[0]);
    }
    __syncthreads();
    // step kt: LDS buffer kt & 1 holds tile kt; register slot (kt + 1) % D holds tile kt+1; tile kt+D goes into slot kt % D
#define LBX16S_STEP(SLOT)                                                                     \
    {                                                                                         \
        constexpr int NXT = ((SLOT) + 1) % D;                                                 \
        if (kt + D < nk) fetch(ra[SLOT], rb[SLOT], kt + D);                                   \
        mma_tile16s(As[kt & 1], Bs[kt & 1], wm, wn, lane, acc);                                \
        if (kt + 1 < nk) { la.store(As[(kt & 1) ^ 1], ra[NXT]); lb.store(Bs[(kt & 1) ^ 1], rb[NXT]); } \
        __syncthreads();                                                                      \
        ++kt;                                                                                 \
    }
    for (int kt = 0; kt < nk;) {
        LBX16S_STEP(0)
        if (D > 1 && kt < nk) LBX16S_STEP(1 % D)
        if (D > 2 && kt < nk) LBX16S_STEP(2 % D)
        if (D > 3 && kt < nk) LBX16S_STEP(3 % D)
    }
#undef LBX16S_STEP
    store_rows_tile<2, 2, true>(acc, m0, n0, wm, wn, lane, m_beg, M, N, epi, aux, Cd, P, split, 0ull, false, C16, mask16);
}

}  // namespace
#include "gemm16_dma.h"
#include "gemm16_pp.h"
#include "gemm16_pp_tn.h"
#include "gemm16_kres.h"
#include "gemm16_tn_kres.h"
namespace {

// ------------------------------------------------------------------------------------------------
// bf16-STORAGE wgrad (lidbox_gemm_bf16s_tn): P[split][K1][N] = A[Mslice, K1]^T . B[Mslice, N] with both operands bfloat16
// in HBM and the contraction index = the ROW (A = the bf16 shadow of the layer input, read through the implicit-row
// descriptor; B = the bf16 shadow of the output gradient).  Tiles go from global memory to LDS unchanged -- [64 contraction
// rows][128 columns] as 16-byte pieces along the row, so a wave-load covers 4 rows x 256 B = 8 whole 128-byte lines -- and
// the MFMA operands (lane -> column, 8 consecutive contraction rows) come out of LDS through the hardware transpose read
// ds_read_b64_tr_b16: a 16-lane group reads a [4 rows][16 columns] block as 4-column pieces and lane j receives column j's
// four values (measured in tools/micro/tr_read.hip).  No convert / transpose VALU work and half the L2 -> LDS bytes of
// gemm16_tn_kernel.  LDS row stride 320 bytes: the 32-lane service group of the transpose read touches 4 rows x 64 B, which
// lands on 4 disjoint quarters of the 256-byte bank row (320 = 64 mod 256).
// The bias gradient (column sums of B) is summed in fp32 from the bf16 shadow's values.
// ------------------------------------------------------------------------------------------------
#ifndef LBX16T_LDS
#define LBX16T_LDS 160                      // LDS row stride of the wgrad tiles, in bf16
#endif
constexpr int BKT = 64;                     // contraction rows of one LDS tile
constexpr int LDS_T = LBX16T_LDS;
constexpr int TILE_T = BKT * LDS_T;

struct TileT {
    u32x4 v[4];
};

// thread -> 16-byte piece pc = tid & 15 (8 columns) of contraction rows kr + 16 j, kr = tid >> 4, j = 0, 1, 2, ...: one
// sequence of stride 16 across tiles (a tile takes four of them), followed by a single incremental (utterance row, pointer)
// tracker -- adds only.
struct KOuterRowsS {
    const __bf16* p;
    long m, step, wrap;
    unsigned t, rpb;
    int pc, kr;

    __device__ __forceinline__ void init(const RowsH& r, long mbeg, int col0, int ncols, int tid) {
        pc = tid & 15;
        kr = tid >> 4;
        const int c = col0 + 8 * pc;
        const int cload = c < ncols ? c : 0;                    // clamped column piece (results never stored)
        m = mbeg + kr;
        rpb = r.batch == 1 ? 0xffffffffu : (unsigned)r.rpb;
        const unsigned b = r.batch == 1 ? 0u : (unsigned)m / rpb;
        t = (unsigned)m - (r.batch == 1 ? 0u : b * rpb);
        p = r.base + (long)b * r.bs + (long)t * r.rs + cload;
        step = 16 * r.rs;
        wrap = r.batch == 1 ? 0 : r.bs - (long)r.rpb * r.rs;
    }
    template <bool CHECK>
    __device__ __forceinline__ void load(TileT& x, long mend) {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x.v[j] = (!CHECK || m < mend) ? *reinterpret_cast<const u32x4*>(p) : z;
            m += 16;
            t += 16;
            p += step;
            while (t >= rpb) { t -= rpb; p += wrap; }
        }
    }
    __device__ __forceinline__ void store(__bf16* tile, const TileT& x) const {
        __bf16* d = tile + kr * LDS_T + 8 * pc;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(d + j * 16 * LDS_T) = x.v[j];
    }
};

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ bf16x8 tr_read8(const __bf16* lo_rows) {
    // rows r .. r+3 of the lane group's block, then rows r+4 .. r+7: 8 consecutive contraction indices of the lane's column
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lo_rows));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lo_rows + 4 * LDS_T));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// 64 x 64 per wave over one [64][128] tile pair: lane l = 16 g + j addresses the 4-column piece (j & 3) of row (j >> 2) in
// the block of columns 16 (g & 1) .. +15 and contraction rows 8 (g >> 1) .. ; it receives column l & 31, rows 8 (l >> 5) ..+7,
// which is the operand layout of v_mfma_f32_32x32x16_bf16.
__device__ __forceinline__ void mma_tile16t(const __bf16* As, const __bf16* Bs, int wm, int wn, int lane, f32x16 (&acc)[2][2]) {
    const int g = lane >> 4, j = lane & 15;
    const int off = ((j >> 2) + 8 * (g >> 1)) * LDS_T + 4 * (j & 3) + 16 * (g & 1);
    const __bf16* ap = As + off + wm * 64;
    const __bf16* bp = Bs + off + wn * 64;
    constexpr int NKS = BKT / 16;
    bf16x8 a[2][2], b[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a[0][i] = tr_read8(ap + i * 32);
        b[0][i] = tr_read8(bp + i * 32);
    }
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks + 1 < NKS) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[nxt][i] = tr_read8(ap + (ks + 1) * 16 * LDS_T + i * 32);
                b[nxt][i] = tr_read8(bp + (ks + 1) * 16 * LDS_T + i * 32);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][i], b[cur][jj], acc[i][jj], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
}

// fp32 sums of the 8 bf16 columns of a staged piece, over the thread's four rows
__device__ __forceinline__ void colsum8(float (&s)[8], const TileT& x) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned u = x.v[j][w];
            s[2 * w] += __builtin_bit_cast(float, u << 16);
            s[2 * w + 1] += __builtin_bit_cast(float, u & 0xffff0000u);
        }
}

__global__ __launch_bounds__(256, 2) void gemm16s_tn_kernel(RowsH A, RowsH Bd, float* __restrict__ P, float* __restrict__ Pc, long M,
                                                         int K1, int N, int tiles_n, int ntiles, long rows_per_split) {
    extern __shared__ __attribute__((aligned(16))) char smem16t[];       // 4 tiles of 64 x LDS_T bf16 (80 KB at stride 160)
    __bf16 (*As)[TILE_T] = reinterpret_cast<__bf16 (*)[TILE_T]>(smem16t);
    __bf16 (*Bs)[TILE_T] = reinterpret_cast<__bf16 (*)[TILE_T]>(smem16t + 2 * TILE_T * sizeof(__bf16));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // block -> (slice, tile) through the XCD-chunk remap: the tiles of one M slice run on ONE XCD (block % 8), so the slice's
    // panels are fetched by one L2 (round 3: 168 MB at the fabric per launch against 44 MB algorithmic, every panel in 4-8 L2s)
    const unsigned vb = xcd_chunk_id(blockIdx.x, gridDim.x);
    const int tile = (int)(vb % (unsigned)ntiles);
    const int split = (int)(vb / (unsigned)ntiles);
    const int tn = tile % tiles_n, tk = tile / tiles_n;
    const int i0 = tk * BT, n0 = tn * BT;
    const long mbeg = (long)split * rows_per_split;
    long mend = mbeg + rows_per_split;
    if (mend > M) mend = M;

    KOuterRowsS la, lb;
    la.init(A, mbeg, i0, K1, tid);
    lb.init(Bd, mbeg, n0, N, tid);

    f32x16 acc[2][2];
    zero_acc(acc);
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool do_csum = (Pc != nullptr) && tk == 0;

    TileT ra[2], rb[2];                                                // register ring of depth 2
    const int nk = (int)((mend - mbeg + BKT - 1) / BKT);
    auto fetch = [&](TileT& a, TileT& b, int t) {
        if (t + 1 < nk) { la.load<false>(a, mend); lb.load<false>(b, mend); }
        else { la.load<true>(a, mend); lb.load<true>(b, mend); }
    };
    auto stage = [&](int buf, const TileT& a, const TileT& b) {
        if (do_csum) colsum8(csum, b);                                 // every staged row exactly once
        la.store(As[buf], a);
        lb.store(Bs[buf], b);
    };
    if (nk > 0) {
        fetch(ra[0], rb[0], 0);
        if (nk > 1) fetch(ra[1], rb[1], 1);
        stage(0, ra[0], rb[0]);
    }
    __syncthreads();
#define LBX16T_STEP(PAR)                                                  \
    {                                                                     \
        if (kt + 2 < nk) fetch(ra[PAR], rb[PAR], kt + 2);                 \
        mma_tile16t(As[PAR], Bs[PAR], wm, wn, lane, acc);                 \
        if (kt + 1 < nk) stage((PAR) ^ 1, ra[(PAR) ^ 1], rb[(PAR) ^ 1]);  \
        __syncthreads();                                                  \
        ++kt;                                                             \
    }
    for (int kt = 0; kt < nk;) {
        LBX16T_STEP(0)
        if (kt < nk) LBX16T_STEP(1)
    }
#undef LBX16T_STEP
    float* Pd = P + (long)split * K1 * N;
    store_partial_blocks<2, 2>(Pd, acc, i0, n0, wm, wn, lane, K1, N);
    if (do_csum) {
        // threads with the same column piece: lanes pc + 16 {0..3} of every wave -- fixed-order butterfly, then the four
        // waves through LDS (free after the K loop's last barrier) in wave order
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            csum[c] += __shfl_xor(csum[c], 16, 64);
            csum[c] += __shfl_xor(csum[c], 32, 64);
        }
        float* red = reinterpret_cast<float*>(smem16t);               // [4 waves][128 columns]
        if (lane < 16) {
#pragma unroll
            for (int c = 0; c < 8; ++c) red[wave * BT + 8 * lane + c] = csum[c];
        }
        __syncthreads();
        if (tid < BT && n0 + tid < N)
            Pc[(long)split * N + n0 + tid] = (red[tid] + red[BT + tid]) + (red[2 * BT + tid] + red[3 * BT + tid]);
    }
}

// elementwise fp32 -> bf16 (round-to-nearest-even), 4 values per thread when aligned
__global__ void f32_to_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long n) {
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * i);
        *reinterpret_cast<bf16x4*>(dst + 4 * i) = to_bf16(v);
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = __builtin_bit_cast(unsigned short, (__bf16)src[i]);
}

// elementwise bf16 -> fp32 (exact), 4 values per thread when aligned
__global__ void bf16_to_f32_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst, long n) {
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const uint2 v = *reinterpret_cast<const uint2*>(src + 4 * i);
        f32x4 o;
        o[0] = __uint_as_float(v.x << 16);
        o[1] = __uint_as_float(v.x & 0xffff0000u);
        o[2] = __uint_as_float(v.y << 16);
        o[3] = __uint_as_float(v.y & 0xffff0000u);
        *reinterpret_cast<f32x4*>(dst + 4 * i) = o;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = __uint_as_float((unsigned)src[i] << 16);
}

// dst[c][r] = bf16(src[r][c]): 32 x 32 tiles through LDS (coalesced on both sides)
__global__ __launch_bounds__(256) void transpose_f32_to_bf16_kernel(const float* __restrict__ src, int R, int C, long ld_src,
                                                                    unsigned short* __restrict__ dst, long ld_dst) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        t[j][tx] = (r < R && c < C) ? src[(long)r * ld_src + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < C && r < R) dst[(long)c * ld_dst + r] = __builtin_bit_cast(unsigned short, (__bf16)t[tx][j]);
    }
}

// one launch for all bf16 weight shadows of a model: blocks [0, conv_blocks) convert the flat parameter vector elementwise,
// the others each handle one 64 x 64 tile of one listed [R][C] matrix of it: written transposed ([C][R]) or as it is, at
// the destination's own leading dimension (padded / stacked operand images).  16-byte loads and 8-byte stores where the
// matrix allows (`vec`: C, the source offset and ld_dst multiples of 4, dst 8-byte aligned, R a multiple of 4 when transposed --
// every image of the x-vector); round 5's 32 x 32 tiles with 2-byte stores took 11.6 us per step at 5.3 M elements.
constexpr int MAX_WT = 48, WT_TILE = 64;
struct WeightShadows {
    long src_off[MAX_WT];               // offset of the matrix inside the flat fp32 vector
    unsigned short* dst[MAX_WT];
    int R[MAX_WT], C[MAX_WT], ld_dst[MAX_WT];
    int tile_end[MAX_WT];               // running total of 64 x 64 tiles up to and including matrix i; bit 31 of R: transpose
    unsigned char vec[MAX_WT];
    int n;
};

__global__ __launch_bounds__(256) void refresh_bf16_weights_kernel(const float* __restrict__ flat, unsigned short* __restrict__ flat16,
                                                                   long n, int conv_blocks, WeightShadows w) {
    __shared__ float t[WT_TILE][WT_TILE + 1];
    if ((int)blockIdx.x < conv_blocks) {
        const long n4 = n >> 2;
        const long stride = (long)conv_blocks * blockDim.x;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
            *reinterpret_cast<bf16x4*>(flat16 + 4 * i) = to_bf16(*reinterpret_cast<const f32x4*>(flat + 4 * i));
        for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
            flat16[i] = __builtin_bit_cast(unsigned short, (__bf16)flat[i]);
        return;
    }
    int tile = (int)blockIdx.x - conv_blocks, m = 0;
    while (m + 1 < w.n && tile >= w.tile_end[m]) ++m;
    if (m > 0) tile -= w.tile_end[m - 1];
    const bool tr = w.R[m] < 0, vec = w.vec[m] != 0;
    const int R = w.R[m] & 0x7fffffff, C = w.C[m];
    const long ld = w.ld_dst[m];
    const float* src = flat + w.src_off[m];
    unsigned short* dst = w.dst[m];
    const int tiles_c = (C + WT_TILE - 1) / WT_TILE;
    const int c0 = (tile % tiles_c) * WT_TILE, r0 = (tile / tiles_c) * WT_TILE;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // 16 groups of four columns x 16 rows per pass
    auto bf = [](float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); };
#pragma unroll
    for (int j = ty; j < WT_TILE; j += 16) {
        const int r = r0 + j, c = c0 + 4 * tx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < R) {
            if (vec && c + 3 < C) v = *reinterpret_cast<const f32x4*>(src + (long)r * C + c);
            else
                for (int e = 0; e < 4; ++e) if (c + e < C) v[e] = src[(long)r * C + c + e];
            if (!tr) {
                if (vec && c + 3 < C) *reinterpret_cast<bf16x4*>(dst + (long)r * ld + c) = to_bf16(v);
                else
                    for (int e = 0; e < 4; ++e) if (c + e < C) dst[(long)r * ld + c + e] = bf(v[e]);
            }
        }
        if (tr)
            for (int e = 0; e < 4; ++e) t[j][4 * tx + e] = v[e];
    }
    if (!tr) return;
    __syncthreads();
#pragma unroll
    for (int j = ty; j < WT_TILE; j += 16) {                     // destination row c0 + j, four source rows r0 + 4 tx ...
        const int c = c0 + j, r = r0 + 4 * tx;
        if (c >= C) continue;
        f32x4 v;
        for (int e = 0; e < 4; ++e) v[e] = t[4 * tx + e][j];
        if (vec && r + 3 < R) *reinterpret_cast<bf16x4*>(dst + (long)c * ld + r) = to_bf16(v);
        else
            for (int e = 0; e < 4; ++e) if (r + e < R) dst[(long)c * ld + r + e] = bf(v[e]);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct Rows16Plan {
    int splits, k_per_split;
};

// One tile shape; small-M problems (Dense layers) are split along K until ~2 workgroups per CU (the resident
// count: ~190 VGPRs with two tiles in flight) exist.
Rows16Plan plan_rows16(long M, int N, int K, size_t ws_bytes, int BK = 32) {
    const long tiles = lbx_cdiv(M, BT) * lbx_cdiv(N, BT);
    Rows16Plan best{1, K};
    if (tiles >= 2 * NUM_CU) return best;
    long s = (2 * NUM_CU) / tiles;
    const long max_s = K / (2 * BK);                 // at least two K-steps per split
    if (s > max_s) s = max_s;
    if (s > 64) s = 64;
    if (const char* e = getenv("LIDBOX_GEMM16_SPLITS")) { const long v = atol(e); if (v >= 1) s = v < max_s ? v : (max_s > 1 ? max_s : 1); }   // tuning aid
    while (s > 1 && (size_t)s * M * N * sizeof(float) > ws_bytes) --s;
    if (s <= 1) return best;
    const int kps = (int)(lbx_cdiv(lbx_cdiv(K, s), BK) * BK);
    const int splits = (int)lbx_cdiv(K, kps);
    if (splits <= 1) return best;
    return Rows16Plan{splits, kps};
}

struct Tn16Plan {
    int splits;
    long rows_per_split;
};

Tn16Plan plan_tn16(long M, int K1, int N, int BK = 32) {
    const long tiles = lbx_cdiv(K1, BT) * lbx_cdiv(N, BT);
    long target = 2 * NUM_CU;
    // few output tiles = many M slices = a partial-sum volume (slices x K1 x N floats, written and re-read by the reduce) that
    // outweighs filling the second workgroup slot of every CU: measured on the storage kernel, frame1 (8 tiles) 58 -> 52 us and
    // frame4 (16 tiles) 25 -> 22 us with one round of 256 instead of 512 workgroups (profiles/r02_bf16_storage_wgrad.txt)
    if (BK == 64 && tiles <= 16) target = NUM_CU;
    if (const char* e = getenv("LIDBOX_GEMM16_TN_SLOTS")) { const long v = atol(e); if (v >= 1) target = v; }   // tuning aid
    long s = target / tiles;                         // whole rounds only: one workgroup too many costs a full round
    const long max_s = lbx_cdiv(M, 4 * BK);          // at least four K-steps per slice
    if (s > max_s) s = max_s;
    if (const char* e = getenv("LIDBOX_GEMM16_TN_SPLITS")) { const long v = atol(e); if (v >= 1) s = v; }                                 // tuning aid
    if (s < 1) s = 1;
    const long rps = lbx_cdiv(lbx_cdiv(M, s), BK) * BK;
    return Tn16Plan{(int)lbx_cdiv(M, rps), rps};
}

// the 256 x 256 ping-pong wgrad tile (gemm16_pp_tn.h): one workgroup per CU, so the slices are cut for ONE round of the chip.
// Shape part of the decision (the workspace query sees only this); the launch also needs descriptors it can address (below).
struct TnPpPlan {
    bool use;
    int splits;
    long rows_per_split;
};

TnPpPlan plan_tn16_pp(long M, int K1, int N) {
    TnPpPlan pl{false, 1, M};
    const long tiles = lbx_cdiv(K1, PPT_BT) * lbx_cdiv(N, PPT_BT);
    int mode = -1;                                   // LIDBOX_GEMM16_TN_PP = 0 | 1: never | whenever it can run (tuning aid)
    if (const char* e = getenv("LIDBOX_GEMM16_TN_PP")) mode = atoi(e);
    if (mode == 0 || tiles > NUM_CU || M >= (1L << 31) - PPT_BM) return pl;
    long s = NUM_CU / tiles;
    const long max_s = lbx_cdiv(M, 4 * PPT_BM);      // at least four K steps per slice
    if (s > max_s) s = max_s;
    if (const char* e = getenv("LIDBOX_GEMM16_TN_PP_SPLITS")) { const long v = atol(e); if (v >= 1) s = v; }
    if (s < 1) s = 1;
    pl.rows_per_split = lbx_cdiv(lbx_cdiv(M, s), PPT_BM) * PPT_BM;
    pl.splits = (int)lbx_cdiv(M, pl.rows_per_split);
    // worth it where the tile is full and the slices are long: the big conv layers (measured: profiles/r05_bf16_pp_wgrad.txt)
    const bool full = K1 % PPT_BT == 0 && N % PPT_BT == 0;
    pl.use = mode == 1 || (full && tiles >= 8 && tiles * pl.splits >= (3 * NUM_CU) / 4 && pl.rows_per_split >= 2048);
    return pl;
}

const char* const ALIGN_MSG =
    "bf16 path needs 16-byte aligned operands and K, N, leading dimensions and row strides that are multiples of 4";

template <bool B_KINNER>
int launch_rows16(const char* fn, lidbox_rows_t A, const float* Bm, long ldb, lidbox_rows_out_t Cd, int K, int N, int epi,
                  const float* aux, void* ws, size_t ws_bytes, hipStream_t st) {
    const long M = (long)A.batch * A.rows_per_batch;
    if (M == 0 || N == 0) return LIDBOX_OK;
    const lidbox_rows_t Cin{Cd.base, Cd.batch_stride, Cd.row_stride, Cd.batch, Cd.rows_per_batch};
    if (!(rows_aligned(A) && rows_aligned(Cin) && K % 4 == 0 && aligned16(Bm) && ldb % 4 == 0 &&
          (B_KINNER || N % 4 == 0) && aligned16(ws))) {
        lidbox_set_error("%s: invalid argument: %s", fn, ALIGN_MSG);
        return LIDBOX_E_INVALID;
    }
    const Rows16Plan pl = plan_rows16(M, N, K, ws ? ws_bytes : 0);
    const int tiles_n = (int)lbx_cdiv(N, BT);
    const RowsOutD Co{Cd.base, Cd.batch_stride, Cd.row_stride, Cd.batch, Cd.rows_per_batch};
    float* P = (float*)ws;
    // rows [m_beg, m_end) with a given split plan (+ its fixed-order reduce)
    auto launch_range = [&](long m_beg, long m_end, const Rows16Plan& q) -> int {
        const long msub = m_end - m_beg;
        const long ntiles = lbx_cdiv(msub, BT) * tiles_n;
        hipLaunchKernelGGL((gemm16_rows_kernel<B_KINNER>), dim3((unsigned)ntiles, (unsigned)q.splits), dim3(256), 0, st,
                           to_dev(A), Bm, ldb, Co, P, m_beg, m_end, K, N, epi, aux, tiles_n, (unsigned)ntiles, q.k_per_split);
        LBX_LAUNCH_OK();
        if (q.splits > 1) {
            long g = lbx_cdiv(msub * N, 256);
            if (g > 2048) g = 2048;
            hipLaunchKernelGGL(rows_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)P, q.splits, m_beg, msub,
                               N, Co, epi, aux);
            LBX_LAUNCH_OK();
        }
        return LIDBOX_OK;
    };
    // Tail quantisation (as in gemm.hip): two workgroups per CU = 512 slots, and the last partial round of a launch
    // runs at the pace of a full one.  When the tiles beyond the last whole round are few (<= a quarter round), the
    // launch covers the row prefix that is a whole number of rounds and the few remaining rows go to a second launch
    // split along K (each of its workgroups does 1/splits of a tile, so it costs a fraction of a round).
    static const bool no_tail_split = getenv("LIDBOX_GEMM16_NO_TAIL_SPLIT") != nullptr;      // A/B aid
    const long slots = 2 * NUM_CU;
    const long tiles = lbx_cdiv(M, BT) * tiles_n;
    if (!no_tail_split && pl.splits == 1 && tiles > slots && K >= LBX16_TAIL_MIN_K) {   // short K: the extra launch costs more than the round
        const long rounds = tiles / slots, rem_tiles = tiles - rounds * slots;
        const long main_tiles_m = rounds * slots / tiles_n;
        const long m_main = main_tiles_m * BT, m_rem = M - m_main;
        if (rounds >= LBX16_TAIL_MIN_ROUNDS && rem_tiles > 0 && rem_tiles <= slots / 4 && main_tiles_m >= 1 && m_rem > 0) {
            const Rows16Plan rp = plan_rows16(m_rem, N, K, ws ? ws_bytes : 0);
            if (rp.splits > 1) {
                if (int rc = launch_range(0, m_main, pl)) return rc;
                return launch_range(m_main, M, rp);
            }
        }
    }
    return launch_range(0, M, pl);
}

// ---- LDS-DMA variant (gemm16_dma.h): tile shape / ring depth per problem
struct Dma16Choice {
    int bm = 0, bn = 0, stages = 0;     // bm == 0: the register-staged kernel
    int splits = 1, k_per_split = 0;
};

// LIDBOX_GEMM16S_DMA: "0" off, "bm,bn,stages[,splits]" forces a variant (tuning aid, tools/gemm16_sweep.py); unset: policy
Dma16Choice choose_dma16(long M, int N, int K, size_t ws_bytes, bool writes_fp32 = false) {
    Dma16Choice c;
    int bm = 0, bn = 0, stg = 0, sp = 0;
    if (const char* e = getenv("LIDBOX_GEMM16S_DMA")) {
        if (sscanf(e, "%d,%d,%d,%d", &bm, &bn, &stg, &sp) < 3) return c;
        // bm == 256: the eight-wave ping-pong tile (gemm16_pp.h); its third number is SUB (1 | 2), not a ring depth
        const bool pp = bm == 256 && (bn == 128 || bn == 256) && (stg == 1 || stg == 2);
        if (!pp && !((bm == 64 || bm == 128) && (bn == 64 || bn == 128) && stg >= 2 && stg <= 4)) return c;
    } else {
        // (round 3's table: profiles/r03_bf16_dma_variants.txt -- deeper rings lose: the launches are latency-, not bandwidth-bound,
        // and every stage costs a resident workgroup)
        // Round 4, after the epilogues lost their predication and 64-bit addressing, every variant on every rows launch of the
        // step again (tools/bf16s_variants.py, us per call at bs 256 / 512, profiles/r04_bf16_variants.txt): 64 x 128 tiles on
        // two stages are the best or within 1-2 us of it on every launch (sum 354.8 / 626.4 against 365.2 / 655.2 for round
        // 3's table) except the longest one -- frame2's forward, K = 1536 on >= 3 tiles of 128 x 128 per CU -- which runs on
        // 128 x 128 LDS-DMA tiles (69.8 -> 59.7 us at bs 256; the register-staged kernel it used is now the slowest choice).
        // Smaller batches (same script at bs 128 / 64, r04_bf16_variants.txt): a launch whose 64 x 128 tiles fill the chip's
        // 3 x 256 slots just over once (792 tiles: 1.03 rounds) runs a nearly empty second round -- 64 x 64 tiles win there by
        // 10-15 % (128 x 128 for K >= 1536) -- and launches of less than half a round of 128 x 128 tiles, which round 3 left on
        // the register-staged kernel, take 64 x 64 tiles split along K (bs 64: 14.7 / 7.5 / 16.5 / 9.4 us against 26 / 24 / 26 / 24).
        // Round 5: the eight-wave ping-pong tile (gemm16_pp.h, 256 x 256, one workgroup per CU).  tools/bf16s_variants.py at bs 256 /
        // 512 (profiles/r05_bf16_pp_variants.txt): it wins wherever its tiles fill >= 3/4 of the CUs in every round and the
        // contraction is >= 512 deep -- frame2 forward 59 -> 49 us / 122 -> 98 us, frame2's dgrads 56 + 37 -> 38 + 27 us, frame3's
        // dgrad 38 -> 27 us -- and for K >= 1500 from half a round on (bs 512: frame5 dgrad 55 -> 48 us); it loses on short
        // contractions (frame1: K = 200, four steps under a 256 x 256 epilogue) and on launches of a quarter round (M = 8 448).
        const long t256 = lbx_cdiv(M, 256L) * lbx_cdiv((long)N, 256L);
        const long r256 = lbx_cdiv(t256, (long)NUM_CU);
        bool no_pp = getenv("LIDBOX_GEMM16S_NO_PP") != nullptr;                       // A/B aid
        if (const char* e = getenv("LIDBOX_GEMM16S_NO_PP_SHAPE")) {                   // A/B aid: "M,N,K" of one launch to keep off the tile
            long em = 0; int en = 0, ek = 0;
            if (sscanf(e, "%ld,%d,%d", &em, &en, &ek) == 3 && em == M && en == N && ek == K) no_pp = true;
        }
        // (a launch that writes fp32 -- frame5's forward, 4 or 6 bytes per element -- pays a 256 x 256 epilogue twice over
        // in a 1.5-round launch: bs 512 62 vs 57 us; it stays on the small tiles unless the contraction is long)
        if (!no_pp && !(writes_fp32 && K < 1024 && t256 > NUM_CU) &&
            ((K >= 512 && 4 * t256 >= 3 * r256 * NUM_CU) || (K >= 1500 && 2 * t256 >= r256 * NUM_CU))) {
            c.bm = 256; c.bn = 256; c.stages = 2;
            c.splits = 1;
            c.k_per_split = (int)(lbx_cdiv((long)K, (long)D16_BK) * D16_BK);
            return c;
        }
        const long t128 = lbx_cdiv(M, 128L) * lbx_cdiv((long)N, 128L);
        const long t64x128 = lbx_cdiv(M, 64L) * lbx_cdiv((long)N, 128L);
        const bool just_over_a_round = t64x128 > 3 * NUM_CU && 10 * t64x128 <= 12 * 3 * NUM_CU;
        if (t128 < NUM_CU / 2) { bm = 64; bn = 64; stg = 2; }
        else if (K >= 1536 && (t128 >= 3 * NUM_CU || just_over_a_round)) { bm = 128; bn = 128; stg = 2; }
        else if (just_over_a_round) { bm = 64; bn = 64; stg = 2; }
        else { bm = 64; bn = 128; stg = 2; }
    }
    c.bm = bm; c.bn = bn; c.stages = stg;
    const long tiles = lbx_cdiv(M, (long)bm) * lbx_cdiv((long)N, (long)bn);
    long s = sp > 0 ? sp : 1;
    if (sp <= 0 && tiles < 2 * NUM_CU) {                       // small-M problems: split along K until the chip is covered twice
        s = (2 * NUM_CU) / tiles;
        const long max_s = K / (2 * D16_BK);
        if (s > max_s) s = max_s;
        if (s > 64) s = 64;
    }
    while (s > 1 && (size_t)s * M * N * sizeof(float) > ws_bytes) --s;
    if (s < 1) s = 1;
    c.k_per_split = (int)(lbx_cdiv(lbx_cdiv((long)K, s), (long)D16_BK) * D16_BK);
    c.splits = (int)lbx_cdiv((long)K, (long)c.k_per_split);
    return c;
}

template <int BM, int BN, int STAGES, int OCC>
int launch_rows16s_dma_t(const Dma16Choice& dc, const RowsH& Ah, const RowsH& Bh, const RowsOutD& Co, unsigned short* S, float* P, long M,
                         int K, int N, int epi, const float* aux, const unsigned short* mask16, hipStream_t st, const ReduceJobs& rj) {
    constexpr size_t lds_bytes = (size_t)STAGES * (BM + BN) * D16_ROW_BYTES;
    static bool attr_set = false;
    if (lds_bytes > 65536 && !attr_set) {
        LBX_HIP(hipFuncSetAttribute((const void*)gemm16s_rows_dma_kernel<BM, BN, STAGES, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_bytes));
        attr_set = true;
    }
    const int tiles_n = (int)lbx_cdiv((long)N, (long)BN);
    const long ntiles = lbx_cdiv(M, (long)BM) * tiles_n;
    hipLaunchKernelGGL((gemm16s_rows_dma_kernel<BM, BN, STAGES, OCC>), dim3((unsigned)ntiles + rj.total, (unsigned)dc.splits), dim3(256), lds_bytes, st,
                       Ah, Bh, Co, S, P, 0L, M, K, N, epi, aux, tiles_n, (unsigned)ntiles, dc.k_per_split, mask16, rj);
    LBX_LAUNCH_OK();
    if (dc.splits > 1) {
        long g = lbx_cdiv(M * N, 256);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(rows_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)P, dc.splits, 0L, M, N, Co, epi, aux, S,
                           mask16);
        LBX_LAUNCH_OK();
    }
    return LIDBOX_OK;
}

// the eight-wave ping-pong tile (gemm16_pp.h): dc.bm == 256, dc.stages = SUB
template <int BN, int SUB>
int launch_rows16s_pp_t(const Dma16Choice& dc, const RowsH& Ah, const RowsH& Bh, const RowsOutD& Co, unsigned short* S, float* P, long M,
                        int K, int N, int epi, const float* aux, const unsigned short* mask16, hipStream_t st, const ReduceJobs& rj) {
    constexpr size_t lds_bytes = (size_t)pp_lds_bytes<BN>();
    static std::atomic<unsigned long long> attr_devs{0};
    int dev = 0;
    LBX_HIP(hipGetDevice(&dev));
    if (dev >= 64 || !(attr_devs.load() >> dev & 1ull)) {
        LBX_HIP(hipFuncSetAttribute((const void*)gemm16s_rows_pp_kernel<BN, SUB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_bytes));
        if (dev < 64) attr_devs.fetch_or(1ull << dev);
    }
    const int tiles_n = (int)lbx_cdiv((long)N, (long)BN);
    const long ntiles = lbx_cdiv(M, 256L) * tiles_n;
    hipLaunchKernelGGL((gemm16s_rows_pp_kernel<BN, SUB>), dim3((unsigned)ntiles + rj.total, (unsigned)dc.splits), dim3(512), lds_bytes, st,
                       Ah, Bh, Co, S, P, 0L, M, K, N, epi, aux, tiles_n, (unsigned)ntiles, dc.k_per_split, mask16, rj);
    LBX_LAUNCH_OK();
    if (dc.splits > 1) {
        long g = lbx_cdiv(M * N, 256);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(rows_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)P, dc.splits, 0L, M, N, Co, epi, aux, S,
                           mask16);
        LBX_LAUNCH_OK();
    }
    return LIDBOX_OK;
}

thread_local int g16_last_carried = 0;                 // jobs the calling thread's last lidbox_gemm_bf16s_nt_carry ran inside its GEMM launch
thread_local int g16_tn_last_kres = 0;                 // slices of the calling thread's last storage wgrad if the K1-resident kernel ran it, else 0
thread_local int g16_tn_last_pp = 0;                   // slices of the calling thread's last storage wgrad if the ping-pong tile ran it, else 0
thread_local int g16_last_pair = 0;                    // 1 if the calling thread's last lidbox_gemm_bf16s_nt_pair_carry ran both problems in one grid
thread_local int g16_last_variant[3] = {0, 0, 0};     // {bm, bn, stages} of the calling thread's last lidbox_gemm_bf16s_nt (0: register-staged)

// two problems in one grid of the 256 x 256 ping-pong tile (gemm16_pp.h: gemm16s_rows_pp2_kernel); p0's tiles lead
int launch_rows16s_pp2(const PpProblem& p0, const PpProblem& p1, hipStream_t st, const ReduceJobs& rj) {
    constexpr size_t lds_bytes = (size_t)pp_lds_bytes<256>();
    static std::atomic<unsigned long long> attr_devs{0};
    int dev = 0;
    LBX_HIP(hipGetDevice(&dev));
    if (dev >= 64 || !(attr_devs.load() >> dev & 1ull)) {
        LBX_HIP(hipFuncSetAttribute((const void*)gemm16s_rows_pp2_kernel<256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        if (dev < 64) attr_devs.fetch_or(1ull << dev);
    }
    hipLaunchKernelGGL((gemm16s_rows_pp2_kernel<256, 2>), dim3(p0.ntiles + p1.ntiles + rj.total), dim3(512), lds_bytes, st, p0, p1, rj);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

int launch_rows16s_dma(const Dma16Choice& dc, const RowsH& Ah, const RowsH& Bh, const RowsOutD& Co, unsigned short* S, float* P, long M, int K,
                       int N, int epi, const float* aux, const unsigned short* mask16, hipStream_t st, const ReduceJobs& rj) {
    if (dc.bm == 256) {
        if (dc.bn == 256 && dc.stages == 2) return launch_rows16s_pp_t<256, 2>(dc, Ah, Bh, Co, S, P, M, K, N, epi, aux, mask16, st, rj);
        if (dc.bn == 256 && dc.stages == 1) return launch_rows16s_pp_t<256, 1>(dc, Ah, Bh, Co, S, P, M, K, N, epi, aux, mask16, st, rj);
        if (dc.bn == 128 && dc.stages == 2) return launch_rows16s_pp_t<128, 2>(dc, Ah, Bh, Co, S, P, M, K, N, epi, aux, mask16, st, rj);
        if (dc.bn == 128 && dc.stages == 1) return launch_rows16s_pp_t<128, 1>(dc, Ah, Bh, Co, S, P, M, K, N, epi, aux, mask16, st, rj);
    }
#define LBX_D16(BM_, BN_, ST_, OCC_) \
    if (dc.bm == BM_ && dc.bn == BN_ && dc.stages == ST_) return launch_rows16s_dma_t<BM_, BN_, ST_, OCC_>(dc, Ah, Bh, Co, S, P, M, K, N, epi, aux, mask16, st, rj)
    LBX_D16(64, 64, 2, 5);
    LBX_D16(64, 64, 3, 3);
    LBX_D16(64, 64, 4, 2);
    LBX_D16(128, 64, 2, 3);
    LBX_D16(128, 64, 3, 2);
    LBX_D16(64, 128, 2, 3);
    LBX_D16(64, 128, 3, 2);
    LBX_D16(128, 128, 2, 2);
    LBX_D16(128, 128, 3, 1);
#undef LBX_D16
    lidbox_set_error("lidbox_gemm_bf16s_nt: no LDS-DMA instantiation for tile %d x %d, %d stages", dc.bm, dc.bn, dc.stages);
    return LIDBOX_E_INVALID;
}

// workgroups the carried reduces of a storage-GEMM launch share (gemm.hip: carry_cap -- same measurement)
inline long carry_cap16() {
    if (const char* e = getenv("LIDBOX_GEMM_CARRY_BLOCKS")) { const long v = atol(e); if (v >= 8) return v; }
    return 96;
}

// jobs / njobs: pending wgrad reduces this launch carries in its leading workgroups (the first kernel launched takes them)
int launch_rows16s(const char* fn, lidbox_rows_t A, const void* B16, long ldb, lidbox_rows_out_t Cd, void* C16, int K, int N,
                   int epi, const float* aux, void* ws, size_t ws_bytes, hipStream_t st, const unsigned short* mask16,
                   const ReduceJob* jobs = nullptr, int njobs = 0) {
    const long M = (long)A.batch * A.rows_per_batch;
    ReduceJobs rj;
    if (njobs > 0) rj = pack_carry(jobs, njobs, carry_cap16());
    if (M == 0 || N == 0) {
        if (rj.total) hipLaunchKernelGGL(splitk_reduce4_kernel, dim3(rj.total), dim3(256), 0, st, rj);
        return LIDBOX_OK;
    }
    const lidbox_rows_t Cin{Cd.base, Cd.batch_stride, Cd.row_stride, Cd.batch, Cd.rows_per_batch};
    const bool a_ok = aligned16(A.base) && A.row_stride % 8 == 0 && (A.batch == 1 || A.batch_stride % 8 == 0);
    if (!(a_ok && rows_aligned(Cin) && K % 8 == 0 && aligned16(B16) && ldb % 8 == 0 && aligned16(ws) &&
          (C16 == nullptr || (((uintptr_t)C16) & 1) == 0))) {
        lidbox_set_error("%s: invalid argument: bf16-storage operands need 16-byte aligned bases and K, ldb, row and batch "
                         "strides (in bf16 elements) that are multiples of 8", fn);
        return LIDBOX_E_INVALID;
    }
    const RowsOutD Co_{Cd.base, Cd.batch_stride, Cd.row_stride, Cd.batch, Cd.rows_per_batch};
    const RowsH Ah_{(const __bf16*)A.base, A.batch_stride, A.row_stride, A.batch, A.rows_per_batch};
    const RowsH Bh_{(const __bf16*)B16, 0, ldb, 1, 0};
    {
        // short contraction over overlapping windows of short utterances (frame1's forward): both operands resident in LDS
        // (gemm16_kres.h).  LIDBOX_GEMM16S_KRES=0 | 1: never | whenever it can run (tuning aid)
        {
            int mode = -1;
            if (const char* e = getenv("LIDBOX_GEMM16S_KRES")) mode = atoi(e);
            const int base_epi = epi & ~LIDBOX_EPI_MASK_BF16;
            const bool epi_ok = base_epi == LIDBOX_EPI_NONE || base_epi == LIDBOX_EPI_BIAS || base_epi == LIDBOX_EPI_BIAS_RELU || base_epi == LIDBOX_EPI_RELU;
            // vector stores: whole 8-column chunks on 16-byte (fp32) / 8-byte (shadow) boundaries
            const bool c_ok = ((Cd.batch == A.batch && Cd.rows_per_batch == A.rows_per_batch) || Cd.batch == 1) && N % 8 == 0 &&
                              Cd.row_stride % 4 == 0 && (Cd.batch == 1 || Cd.batch_stride % 4 == 0) && aligned16(Cd.base) &&
                              (((uintptr_t)C16) & 7) == 0;
            const int tiles_n = (int)lbx_cdiv(N, KRES_BN);
            const double a_span = ((double)(A.batch - 1) * (double)A.batch_stride + (double)A.rows_per_batch * (double)A.row_stride + K) * 2.0;
            const bool can = mode != 0 && rj.total == 0 && epi_ok && c_ok && mask16 == nullptr && kres_applies(A, K, N) && A.batch_stride >= 0 &&
                             a_span < 4.0e9 && tiles_n <= NUM_CU &&
                             ((double)(A.rows_per_batch - 1) * (double)A.row_stride + K) * 2.0 < 2.0e9;
            // worth it when the launch is a store stream: windows that overlap (row stride < K) and enough utterances to keep every
            // CU's workgroup busy for a few tiles
            if (can && (mode == 1 || (A.row_stride * 2 <= K && (long)A.batch * tiles_n >= 2 * NUM_CU))) {
                int dev = 0;
                LBX_HIP(hipGetDevice(&dev));
                static std::atomic<unsigned long long> kres_attr{0};
                if (dev < 64 && !((kres_attr.load() >> dev) & 1ull)) {
                    LBX_HIP(hipFuncSetAttribute((const void*)gemm16s_rows_kres_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, KRES_LDS_BYTES));
                    LBX_HIP(hipFuncSetAttribute((const void*)gemm16s_rows_kres_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, KRES_LDS_BYTES));
                    kres_attr.fetch_or(1ull << dev);
                }
                const int blocks = (int)lbx_cdiv(A.rows_per_batch, KRES_ROWS);
                const long nunits = (long)A.batch * blocks;
                long nstreams = NUM_CU / tiles_n;                               // one workgroup per CU (146 KB of LDS)
                if (nstreams > lbx_cdiv(nunits, (long)KRES_WAVES)) nstreams = lbx_cdiv(nunits, (long)KRES_WAVES);
                if (nstreams < 1) nstreams = 1;
                if (const char* e = getenv("LIDBOX_GEMM16S_KRES_STREAMS")) { const long q = atol(e); if (q >= 1) nstreams = q; }   // tuning aid
                g16_last_variant[0] = 1; g16_last_variant[1] = KRES_BN; g16_last_variant[2] = 1;      // {1, 64, 1}: the K-resident kernel
                if ((K + 15) / 16 == 13)                                        // frame1 of the x-vector: K = 200
                    hipLaunchKernelGGL(gemm16s_rows_kres_kernel<13>, dim3((unsigned)(tiles_n * nstreams)), dim3(64 * KRES_WAVES), KRES_LDS_BYTES, st, Ah_,
                                       Bh_, Co_, (unsigned short*)C16, K, N, base_epi, aux, tiles_n, (int)nstreams, blocks, nunits);
                else
                    hipLaunchKernelGGL(gemm16s_rows_kres_kernel<0>, dim3((unsigned)(tiles_n * nstreams)), dim3(64 * KRES_WAVES), KRES_LDS_BYTES, st, Ah_,
                                       Bh_, Co_, (unsigned short*)C16, K, N, base_epi, aux, tiles_n, (int)nstreams, blocks, nunits);
                LBX_LAUNCH_OK();
                return LIDBOX_OK;
            }
        }
        const Dma16Choice dc = choose_dma16(M, N, K, ws ? ws_bytes : 0, Cd.base != nullptr);
        // 32-bit byte offsets per lane inside the kernel: the operands' extents must fit
        const double a_ext = ((double)(A.batch - 1) * (double)A.batch_stride + (double)A.rows_per_batch * (double)A.row_stride + K) * 2.0;
        const double b_ext = ((double)N * (double)ldb + K) * 2.0;
        g16_last_variant[0] = g16_last_variant[1] = g16_last_variant[2] = 0;
        if (dc.bm != 0 && a_ext < 4.0e9 && b_ext < 4.0e9) {
            g16_last_variant[0] = dc.bm; g16_last_variant[1] = dc.bn; g16_last_variant[2] = dc.stages;
        }
        if (dc.bm != 0 && a_ext < 4.0e9 && b_ext < 4.0e9)
            return launch_rows16s_dma(dc, Ah_, Bh_, Co_, (unsigned short*)C16, (float*)ws, M, K, N, epi, aux, mask16, st, rj);
    }
    const Rows16Plan pl = plan_rows16(M, N, K, ws ? ws_bytes : 0, BKS);
    const size_t lds_bytes = 4 * (size_t)TILE_S * sizeof(__bf16);
    static bool lds_attr_set = false;
    if (lds_bytes > 65536 && !lds_attr_set) {
        LBX_HIP(hipFuncSetAttribute((const void*)gemm16s_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        lds_attr_set = true;
    }
    const int tiles_n = (int)lbx_cdiv(N, BT);
    const RowsOutD Co{Cd.base, Cd.batch_stride, Cd.row_stride, Cd.batch, Cd.rows_per_batch};
    const RowsH Ah{(const __bf16*)A.base, A.batch_stride, A.row_stride, A.batch, A.rows_per_batch};
    const RowsH Bh{(const __bf16*)B16, 0, ldb, 1, 0};
    float* P = (float*)ws;
    unsigned short* S = (unsigned short*)C16;
    auto launch_range = [&](long m_beg, long m_end, const Rows16Plan& q) -> int {
        const long msub = m_end - m_beg;
        const long ntiles = lbx_cdiv(msub, BT) * tiles_n;
        hipLaunchKernelGGL(gemm16s_rows_kernel, dim3((unsigned)ntiles + rj.total, (unsigned)q.splits), dim3(256), lds_bytes, st, Ah, Bh, Co, S, P,
                           m_beg, m_end, K, N, epi, aux, tiles_n, (unsigned)ntiles, q.k_per_split, mask16, rj);
        LBX_LAUNCH_OK();
        rj = ReduceJobs{};                           // carried by the first launch of the call
        if (q.splits > 1) {
            long g = lbx_cdiv(msub * N, 256);
            if (g > 2048) g = 2048;
            hipLaunchKernelGGL(rows_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)P, q.splits, m_beg, msub,
                               N, Co, epi, aux, S, mask16);
            LBX_LAUNCH_OK();
        }
        return LIDBOX_OK;
    };
    static const bool no_tail_split = getenv("LIDBOX_GEMM16_NO_TAIL_SPLIT") != nullptr;
    static const long slots_per_cu = getenv("LIDBOX_GEMM16S_SLOTS") ? atol(getenv("LIDBOX_GEMM16S_SLOTS")) : LBX16S_WAVES;   // tuning aid
    const long slots = slots_per_cu * NUM_CU;
    const long tiles = lbx_cdiv(M, BT) * tiles_n;
    if (!no_tail_split && pl.splits == 1 && tiles > slots && K >= LBX16_TAIL_MIN_K) {
        const long rounds = tiles / slots, rem_tiles = tiles - rounds * slots;
        const long main_tiles_m = rounds * slots / tiles_n;
        const long m_main = main_tiles_m * BT, m_rem = M - m_main;
        if (rounds >= LBX16_TAIL_MIN_ROUNDS && rem_tiles > 0 && rem_tiles <= slots / 4 && main_tiles_m >= 1 && m_rem > 0) {
            const Rows16Plan rp = plan_rows16(m_rem, N, K, ws ? ws_bytes : 0, BKS);
            if (rp.splits > 1) {
                if (int rc = launch_range(0, m_main, pl)) return rc;
                return launch_range(m_main, M, rp);
            }
        }
    }
    return launch_range(0, M, pl);
}

}  // namespace

extern "C" int lidbox_gemm_bf16s_nt(lidbox_rows_t A16, const void* B16, long ldb, lidbox_rows_out_t C, void* C16, int K,
                                    int N, int epilogue, const float* aux, void* workspace, size_t workspace_bytes,
                                    lidbox_stream_t stream) {
    return lidbox_gemm_bf16s_nt_carry(A16, B16, ldb, C, C16, K, N, epilogue, aux, workspace, workspace_bytes, nullptr, 0, stream);
}

// lidbox_gemm_bf16s_nt whose first launch also runs up to two pending wgrad reduces in its leading workgroups (lidbox_hip.h:
// "carried reduce"; LIDBOX_GEMM_NO_CARRY=1: as one launch of their own behind the GEMM)
extern "C" int lidbox_gemm_bf16s_nt_carry(lidbox_rows_t A16, const void* B16, long ldb, lidbox_rows_out_t C, void* C16, int K,
                                          int N, int epilogue, const float* aux, void* workspace, size_t workspace_bytes,
                                          const lidbox_reduce_job_t* jobs, int njobs, lidbox_stream_t stream) {
    static_assert(sizeof(lidbox_reduce_job_t) == sizeof(ReduceJob), "lidbox_reduce_job_t mirrors ReduceJob");
    ReduceJob js[MAX_CARRY];
    int m = 0;
    for (int i = 0; jobs && i < njobs; ++i) {
        if (jobs[i].nblocks == 0) continue;
        LBX_ARG(m < MAX_CARRY, "at most 2 non-empty jobs per call");
        memcpy(&js[m], jobs + i, sizeof(ReduceJob));
        LBX_ARG((const void*)js[m].P != workspace, "a job's slices live in this call's workspace");
        LBX_ARG(js[m].splits >= 0, "an optimizer-prepare job runs through lidbox_reduce_jobs_run only");
        ++m;
    }
    const bool carry = m > 0 && getenv("LIDBOX_GEMM_NO_CARRY") == nullptr;
    // LIDBOX_EPI_MASK_BF16: the ReLU mask source (aux) is bfloat16 data at C's element offsets; C.base == NULL: the result
    // exists only as the shadow C16 (no fp32 copy is written) -- not with the accumulating epilogues, which read C
    const bool mask16 = (epilogue & LIDBOX_EPI_MASK_BF16) != 0;
    const int epi = epilogue & ~LIDBOX_EPI_MASK_BF16;
    lidbox_rows_out_t Cv = C;
    if (!C.base) {
        LBX_ARG(C16 && epi != LIDBOX_EPI_ACCUM && epi != LIDBOX_EPI_ACCUM_RELU && epi != LIDBOX_EPI_ACCUM_RELU_MASK,
                "C.base == NULL needs the shadow C16 and a non-accumulating epilogue");
        Cv.base = (float*)C16;                                      // descriptor checks only
    }
    LBX_ARG(!mask16 || epi == LIDBOX_EPI_RELU_MASK || epi == LIDBOX_EPI_ACCUM_RELU_MASK, "LIDBOX_EPI_MASK_BF16 goes with a ReLU-mask epilogue");
    if (validate_rows_call(__func__, A16, (const float*)B16, ldb, Cv, K, N, epi, aux, K)) return LIDBOX_E_INVALID;
    int rc = launch_rows16s(__func__, A16, B16, ldb, C, C16, K, N, epi, mask16 ? nullptr : aux, workspace, workspace_bytes,
                            (hipStream_t)stream, mask16 ? (const unsigned short*)aux : nullptr, js, carry ? m : 0);
    if (rc) return rc;
    g16_last_carried = carry ? m : 0;
    if (m > 0 && !carry) {
        ReduceJobs all;
        for (int i = 0; i < m; ++i) { all.j[i] = js[i]; all.total += js[i].nblocks; }
        hipLaunchKernelGGL(splitk_reduce4_kernel, dim3(all.total), dim3(256), 0, (hipStream_t)stream, all);
        LBX_LAUNCH_OK();
    }
    return LIDBOX_OK;
}

extern "C" int lidbox_gemm_bf16s_last_carried(void) { return g16_last_carried; }

// Two independent lidbox_gemm_bf16s_nt problems (they may share A; their outputs must not overlap) as ONE launch when both would run on
// the 256 x 256 ping-pong tile without a K split -- the two row residues of a strided convolution's output-stationary dgrad, whose tile
// counts (1.92 + 1.55 rounds of 256 CUs at 512 utterances) pack better as one grid of 3.47 rounds; the deeper contraction leads.
// Otherwise: the two calls one after the other (the jobs ride with the first).  Same bits either way (the same tile code).
namespace {
struct Nt16Call {
    lidbox_rows_t A; const void* B; long ldb; lidbox_rows_out_t C; void* C16; int K, N, epilogue; const float* aux;
};
// the checks of lidbox_gemm_bf16s_nt_carry / launch_rows16s that decide whether a call may join a two-problem grid
bool pp2_problem(const Nt16Call& c, size_t ws_bytes, PpProblem* out) {
    const bool mask16 = (c.epilogue & LIDBOX_EPI_MASK_BF16) != 0;
    const int epi = c.epilogue & ~LIDBOX_EPI_MASK_BF16;
    const long M = (long)c.A.batch * c.A.rows_per_batch;
    if (M <= 0 || c.N <= 0 || c.K <= 0) return false;
    if (!c.C.base && (!c.C16 || epi == LIDBOX_EPI_ACCUM || epi == LIDBOX_EPI_ACCUM_RELU || epi == LIDBOX_EPI_ACCUM_RELU_MASK)) return false;
    if (mask16 && epi != LIDBOX_EPI_RELU_MASK && epi != LIDBOX_EPI_ACCUM_RELU_MASK) return false;
    lidbox_rows_out_t Cv = c.C;
    if (!c.C.base) Cv.base = (float*)c.C16;
    if (validate_rows_call("lidbox_gemm_bf16s_nt_pair_carry", c.A, (const float*)c.B, c.ldb, Cv, c.K, c.N, epi, c.aux, c.K)) return false;
    const lidbox_rows_t Cin{c.C.base, c.C.batch_stride, c.C.row_stride, c.C.batch, c.C.rows_per_batch};
    const bool a_ok = aligned16(c.A.base) && c.A.row_stride % 8 == 0 && (c.A.batch == 1 || c.A.batch_stride % 8 == 0);
    if (!(a_ok && rows_aligned(Cin) && c.K % 8 == 0 && aligned16(c.B) && c.ldb % 8 == 0 && (c.C16 == nullptr || (((uintptr_t)c.C16) & 1) == 0)))
        return false;
    if (!mask16 && kres_applies(c.A, c.K, c.N)) return false;                  // the K-resident forward kernel may take it
    const Dma16Choice dc = choose_dma16(M, c.N, c.K, ws_bytes, c.C.base != nullptr);
    if (!(dc.bm == 256 && dc.bn == 256 && dc.stages == 2 && dc.splits == 1)) return false;
    const double a_ext = ((double)(c.A.batch - 1) * (double)c.A.batch_stride + (double)c.A.rows_per_batch * (double)c.A.row_stride + c.K) * 2.0;
    const double b_ext = ((double)c.N * (double)c.ldb + c.K) * 2.0;
    if (!(a_ext < 4.0e9 && b_ext < 4.0e9)) return false;
    out->A = RowsH{(const __bf16*)c.A.base, c.A.batch_stride, c.A.row_stride, c.A.batch, c.A.rows_per_batch};
    out->Bw = RowsH{(const __bf16*)c.B, 0, c.ldb, 1, 0};
    out->Cd = RowsOutD{c.C.base, c.C.batch_stride, c.C.row_stride, c.C.batch, c.C.rows_per_batch};
    out->C16 = (unsigned short*)c.C16;
    out->P = nullptr;
    out->mask16 = mask16 ? (const unsigned short*)c.aux : nullptr;
    out->aux = mask16 ? nullptr : c.aux;
    out->M = M; out->K = c.K; out->N = c.N; out->epi = epi;
    out->tiles_n = (int)lbx_cdiv((long)c.N, 256L);
    out->ntiles = (unsigned)(lbx_cdiv(M, 256L) * out->tiles_n);
    return true;
}
}  // namespace

extern "C" int lidbox_gemm_bf16s_nt_pair_carry(lidbox_rows_t A0, const void* B0, long ldb0, lidbox_rows_out_t C0, void* C16_0, int K0, int N0,
                                               int epilogue0, const float* aux0, lidbox_rows_t A1, const void* B1, long ldb1,
                                               lidbox_rows_out_t C1, void* C16_1, int K1, int N1, int epilogue1, const float* aux1,
                                               void* workspace, size_t workspace_bytes, const lidbox_reduce_job_t* jobs, int njobs,
                                               lidbox_stream_t stream) {
    g16_last_pair = 0;
    const Nt16Call c0{A0, B0, ldb0, C0, C16_0, K0, N0, epilogue0, aux0}, c1{A1, B1, ldb1, C1, C16_1, K1, N1, epilogue1, aux1};
    static const bool off = getenv("LIDBOX_GEMM16S_NO_PAIR") != nullptr;                       // A/B aid
    PpProblem p0, p1;
    ReduceJob js[MAX_CARRY];
    int m = 0;
    bool jobs_ok = true;
    for (int i = 0; jobs && i < njobs; ++i) {
        if (jobs[i].nblocks == 0) continue;
        if (m >= MAX_CARRY) { jobs_ok = false; break; }
        memcpy(&js[m], jobs + i, sizeof(ReduceJob));
        if ((const void*)js[m].P == workspace || js[m].splits < 0) { jobs_ok = false; break; }
        ++m;
    }
    if (!off && jobs_ok && getenv("LIDBOX_GEMM_NO_CARRY") == nullptr && aligned16(workspace) &&
        pp2_problem(c0, workspace ? workspace_bytes : 0, &p0) && pp2_problem(c1, workspace ? workspace_bytes : 0, &p1)) {
        ReduceJobs rj;
        if (m > 0) rj = pack_carry(js, m, carry_cap16());
        const bool swap = p1.K > p0.K;                                           // long tiles first
        int rc = launch_rows16s_pp2(swap ? p1 : p0, swap ? p0 : p1, (hipStream_t)stream, rj);
        if (rc) return rc;
        g16_last_pair = 1;
        g16_last_carried = m;
        g16_last_variant[0] = 256; g16_last_variant[1] = 256; g16_last_variant[2] = 2;
        return LIDBOX_OK;
    }
    int rc = lidbox_gemm_bf16s_nt_carry(A0, B0, ldb0, C0, C16_0, K0, N0, epilogue0, aux0, workspace, workspace_bytes, jobs, njobs, stream);
    if (rc) return rc;
    const int carried = g16_last_carried;
    rc = lidbox_gemm_bf16s_nt_carry(A1, B1, ldb1, C1, C16_1, K1, N1, epilogue1, aux1, workspace, workspace_bytes, nullptr, 0, stream);
    g16_last_carried = carried;
    return rc;
}
extern "C" int lidbox_gemm_bf16s_last_pair(void) { return g16_last_pair; }
extern "C" int lidbox_gemm_bf16s_tn_last_pp(void) { return g16_tn_last_pp; }
extern "C" int lidbox_gemm_bf16s_tn_last_kres(void) { return g16_tn_last_kres; }

extern "C" int lidbox_gemm_bf16s_last_variant(int* out3) {
    LBX_ARG(out3, "out3 != NULL");
    out3[0] = g16_last_variant[0]; out3[1] = g16_last_variant[1]; out3[2] = g16_last_variant[2];
    return LIDBOX_OK;
}

extern "C" size_t lidbox_gemm_bf16s_tn_workspace(int M, int K1, int N) {
    if (M <= 0 || K1 <= 0 || N <= 0) return 0;
    const Tn16Plan pl = plan_tn16(M, K1, N, BKT);
    const TnPpPlan pp = plan_tn16_pp(M, K1, N);
    int splits = pp.use && pp.splits > pl.splits ? pp.splits : pl.splits;            // either kernel may run (descriptor checks at launch)
    const TnKresPlan kp = plan_tn16_kres(M, K1, N);
    if (kp.shape_ok && kp.max_slices > splits) splits = kp.max_slices;               // ... or the K1-resident one (gemm16_tn_kres.h)
    return ((size_t)splits * K1 * N + (size_t)splits * N) * sizeof(float);
}

extern "C" int lidbox_gemm_bf16s_tn(lidbox_rows_t A16, lidbox_rows_t B16, float* Cm, long ldc, int K1, int N, int accumulate,
                                    float* bias_grad, void* workspace, size_t workspace_bytes, lidbox_stream_t stream) {
    lidbox_reduce_job_t job;
    int rc = lidbox_gemm_bf16s_tn_partial(A16, B16, Cm, ldc, K1, N, accumulate, bias_grad, workspace, workspace_bytes, &job, stream);
    if (rc || job.nblocks == 0) return rc;
    ReduceJobs js;
    memcpy(&js.j[0], &job, sizeof(ReduceJob));
    js.total = js.j[0].nblocks;
    hipLaunchKernelGGL(splitk_reduce4_kernel, dim3(js.total), dim3(256), 0, (hipStream_t)stream, js);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

// the GEMM of lidbox_gemm_bf16s_tn without its reduce: *job describes the pending fixed-order slice sum (lidbox_hip.h:
// "carried reduce"); job->nblocks == 0: nothing pending (the scalar reduce of an unaligned output was launched here)
extern "C" int lidbox_gemm_bf16s_tn_partial(lidbox_rows_t A16, lidbox_rows_t B16, float* Cm, long ldc, int K1, int N, int accumulate,
                                            float* bias_grad, void* workspace, size_t workspace_bytes, lidbox_reduce_job_t* job,
                                            lidbox_stream_t stream) {
    LBX_ARG(job, "job != NULL");
    memset(job, 0, sizeof *job);
    if (check_rows(__func__, A16.base, A16.batch_stride, A16.row_stride, A16.batch, A16.rows_per_batch)) return LIDBOX_E_INVALID;
    if (check_rows(__func__, B16.base, B16.batch_stride, B16.row_stride, B16.batch, B16.rows_per_batch)) return LIDBOX_E_INVALID;
    LBX_ARG(Cm && K1 >= 1 && N >= 1 && ldc >= N, "C != NULL, K1, N >= 1, ldc >= N");
    const long M = (long)A16.batch * A16.rows_per_batch;
    LBX_ARG(M == (long)B16.batch * B16.rows_per_batch, "A and B row counts differ");
    LBX_ARG(M >= 1, "M >= 1");
    auto ok16 = [](const lidbox_rows_t& r) {
        return aligned16(r.base) && r.row_stride % 8 == 0 && (r.batch == 1 || r.batch_stride % 8 == 0);
    };
    // a row is read as whole 16-byte pieces: the last piece of a width that is not a multiple of 8 must still lie inside the
    // row's stride (a shadow padded to 8-element rows); what it holds beyond the width only reaches outputs never stored
    auto width_ok = [](const lidbox_rows_t& r, int w) { return w % 8 == 0 || r.row_stride >= (long)((w + 7) / 8) * 8; };
    LBX_ARG(ok16(A16) && ok16(B16) && width_ok(A16, K1) && width_ok(B16, N) && aligned16(workspace),
            "bf16-storage operands need 16-byte aligned bases, row and batch strides (in bf16 elements) that are multiples of 8, and "
            "K1 / N that are multiples of 8 or rows padded to one");
    Tn16Plan pl = plan_tn16(M, K1, N, BKT);
    TnPpPlan pp = plan_tn16_pp(M, K1, N);
    // the ping-pong tile addresses a row as a 32-bit byte offset from the tile's first column and walks both operands with one
    // (utterance, row) counter: same utterance length on both sides, everything within 4 GB of the base, whole 16-byte chunks
    auto span_ok = [&](const lidbox_rows_t& r, int w) {
        const double span = ((double)(r.batch - 1) * (double)r.batch_stride + (double)(r.rows_per_batch - 1) * (double)r.row_stride + w) * 2.0;
        return r.batch_stride >= 0 && r.row_stride >= 0 && span < 4.0e9;
    };
    if (pp.use && !(span_ok(A16, K1) && span_ok(B16, N) && K1 % 8 == 0 && N % 8 == 0 &&
                    ((A16.batch == 1 && B16.batch == 1) || (A16.batch == B16.batch && A16.rows_per_batch == B16.rows_per_batch))))
        pp.use = false;
    hipStream_t st = (hipStream_t)stream;
    // the K1-resident kernel (gemm16_tn_kres.h): short contraction over overlapping windows, the whole K1 extent in one workgroup's
    // accumulators.  It walks the input as one flattened sequence of rows: utterances a whole number of row strides apart, at least as
    // many of those as output rows, one stage's frames inside its LDS image; same utterance structure on both operands.
    g16_tn_last_kres = 0;
    {
        const TnKresPlan kp = plan_tn16_kres(M, K1, N);
        // (slices are ranges of utterances: with fewer utterances than half the slices the chip would be mostly idle)
        const bool desc_ok = kp.shape_ok && (kp.forced || 2 * A16.batch >= kp.max_slices) && A16.batch == B16.batch &&
                             A16.rows_per_batch == B16.rows_per_batch &&
                             A16.row_stride > 0 && (A16.batch == 1 || A16.batch_stride % A16.row_stride == 0) &&
                             (A16.batch == 1 || A16.batch_stride / A16.row_stride >= A16.rows_per_batch) &&
                             63 * A16.row_stride + K1 <= TKR_IMG_ELEMS &&
                             (63 + lbx_cdiv(K1, A16.row_stride)) * tkr_img_stride((int)A16.row_stride) <= TKR_IMG_BYTES && span_ok(A16, K1) &&
                             span_ok(B16, N);
        if (desc_ok) {
            const int batch = A16.batch;
            int slices = kp.max_slices < batch ? kp.max_slices : batch;
            const int ups = (int)lbx_cdiv(batch, slices);
            slices = (int)lbx_cdiv(batch, ups);
            const size_t need_k = ((size_t)slices * K1 * N + (size_t)slices * N) * sizeof(float);
            if (workspace && workspace_bytes >= need_k) {
                const int tiles_n = N / TKR_BN;
                const int Tq = batch == 1 ? A16.rows_per_batch : (int)(A16.batch_stride / A16.row_stride);
                const long a_elems = (long)(batch - 1) * A16.batch_stride + (long)(A16.rows_per_batch - 1) * A16.row_stride + K1;
                const long b_bytes = ((long)(batch - 1) * B16.batch_stride + (long)(B16.rows_per_batch - 1) * B16.row_stride + N) * 2;
                float* P = (float*)workspace;
                float* Pc = bias_grad ? P + (size_t)slices * K1 * N : nullptr;
                const RowsH Ah{(const __bf16*)A16.base, A16.batch_stride, A16.row_stride, A16.batch, A16.rows_per_batch};
                const RowsH Bh{(const __bf16*)B16.base, B16.batch_stride, B16.row_stride, B16.batch, B16.rows_per_batch};
                {
                    int dev = 0;
                    LBX_HIP(hipGetDevice(&dev));
                    static std::atomic<unsigned long long> attr_set{0};
                    if (dev >= 64 || !((attr_set.load() >> dev) & 1ull)) {
                        LBX_HIP(hipFuncSetAttribute((const void*)gemm16s_tn_kres_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TKR_LDS_BYTES));
                        if (dev < 64) attr_set.fetch_or(1ull << dev);
                    }
                }
                hipLaunchKernelGGL(gemm16s_tn_kres_kernel, dim3((unsigned)(tiles_n * slices)), dim3(512), TKR_LDS_BYTES, st, Ah, Bh, P, Pc, K1, N,
                                   tiles_n, ups, Tq, a_elems, b_bytes);
                LBX_LAUNCH_OK();
                g16_tn_last_kres = slices;
                g16_tn_last_pp = 0;
                const long n = (long)K1 * N;
                if (reduce_job_vec_ok(P, Pc, slices, n, N, Cm, ldc, bias_grad)) {
                    const ReduceJob j = make_reduce_job(P, Pc, slices, n, N, Cm, ldc, accumulate, bias_grad);
                    memcpy(job, &j, sizeof j);
                    return LIDBOX_OK;
                }
                launch_splitk_reduce((const float*)P, (const float*)Pc, slices, n, N, Cm, ldc, accumulate, bias_grad, st);
                LBX_LAUNCH_OK();
                return LIDBOX_OK;
            }
        }
    }
    if (pp.use) pl = Tn16Plan{pp.splits, pp.rows_per_split};
    g16_tn_last_pp = pp.use ? pp.splits : 0;
    const size_t need = ((size_t)pl.splits * K1 * N + (size_t)pl.splits * N) * sizeof(float);
    LBX_ARG(workspace && workspace_bytes >= need, "workspace too small (lidbox_gemm_bf16s_tn_workspace)");
    if (pp.use) {
        int dev = 0;
        LBX_HIP(hipGetDevice(&dev));
        static std::atomic<unsigned long long> attr_set{0};
        if (dev < 64 && !((attr_set.load() >> dev) & 1ull)) {
            LBX_HIP(hipFuncSetAttribute((const void*)gemm16s_tn_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PPT_LDS_BYTES));
            attr_set.fetch_or(1ull << dev);
        }
        const int tiles_n = (int)lbx_cdiv(N, PPT_BT);
        const int ntiles = (int)(lbx_cdiv(K1, PPT_BT) * tiles_n);
        float* P = (float*)workspace;
        float* Pc = bias_grad ? P + (size_t)pl.splits * K1 * N : nullptr;
        const RowsH Ah{(const __bf16*)A16.base, A16.batch_stride, A16.row_stride, A16.batch, A16.rows_per_batch};
        const RowsH Bh{(const __bf16*)B16.base, B16.batch_stride, B16.row_stride, B16.batch, B16.rows_per_batch};
        hipLaunchKernelGGL(gemm16s_tn_pp_kernel, dim3((unsigned)(ntiles * pl.splits)), dim3(512), PPT_LDS_BYTES, st, Ah, Bh, P, Pc, M, K1, N,
                           tiles_n, ntiles, pl.rows_per_split);
        LBX_LAUNCH_OK();
        const long n = (long)K1 * N;
        if (reduce_job_vec_ok(P, Pc, pl.splits, n, N, Cm, ldc, bias_grad)) {
            const ReduceJob j = make_reduce_job(P, Pc, pl.splits, n, N, Cm, ldc, accumulate, bias_grad);
            memcpy(job, &j, sizeof j);
            return LIDBOX_OK;
        }
        launch_splitk_reduce((const float*)P, (const float*)Pc, pl.splits, n, N, Cm, ldc, accumulate, bias_grad, st);
        LBX_LAUNCH_OK();
        return LIDBOX_OK;
    }
    const size_t lds_bytes = 4 * (size_t)TILE_T * sizeof(__bf16);
    static bool lds_attr_set = false;
    if (lds_bytes > 65536 && !lds_attr_set) {
        LBX_HIP(hipFuncSetAttribute((const void*)gemm16s_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        lds_attr_set = true;
    }
    const int tiles_n = (int)lbx_cdiv(N, BT);
    const int ntiles = (int)(lbx_cdiv(K1, BT) * tiles_n);
    float* P = (float*)workspace;
    float* Pc = bias_grad ? P + (size_t)pl.splits * K1 * N : nullptr;
    const RowsH Ah{(const __bf16*)A16.base, A16.batch_stride, A16.row_stride, A16.batch, A16.rows_per_batch};
    const RowsH Bh{(const __bf16*)B16.base, B16.batch_stride, B16.row_stride, B16.batch, B16.rows_per_batch};
    hipLaunchKernelGGL(gemm16s_tn_kernel, dim3((unsigned)(ntiles * pl.splits)), dim3(256), lds_bytes, st, Ah, Bh, P, Pc, M, K1, N,
                       tiles_n, ntiles, pl.rows_per_split);
    LBX_LAUNCH_OK();
    const long n = (long)K1 * N;
    if (reduce_job_vec_ok(P, Pc, pl.splits, n, N, Cm, ldc, bias_grad)) {
        const ReduceJob j = make_reduce_job(P, Pc, pl.splits, n, N, Cm, ldc, accumulate, bias_grad);
        memcpy(job, &j, sizeof j);
        return LIDBOX_OK;
    }
    launch_splitk_reduce((const float*)P, (const float*)Pc, pl.splits, n, N, Cm, ldc, accumulate, bias_grad, st);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_f32_to_bf16(const float* src, void* dst, long n, lidbox_stream_t stream) {
    LBX_ARG(src && dst && n >= 0, "src, dst != NULL");
    LBX_ARG(aligned16(src) && (((uintptr_t)dst) & 7) == 0, "src 16-byte, dst 8-byte aligned");
    if (n == 0) return LIDBOX_OK;
    long g = lbx_cdiv(n / 4 + 1, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, src, (unsigned short*)dst, n);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_bf16_to_f32(const void* src, float* dst, long n, lidbox_stream_t stream) {
    LBX_ARG(src && dst && n >= 0, "src, dst != NULL");
    LBX_ARG(aligned16(dst) && (((uintptr_t)src) & 7) == 0, "dst 16-byte, src 8-byte aligned");
    if (n == 0) return LIDBOX_OK;
    long g = lbx_cdiv(n / 4 + 1, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)src, dst, n);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_refresh_bf16_weights(const float* flat, void* flat16, long n, const lidbox_weight_shadow_t* mats, int nmats,
                                          lidbox_stream_t stream) {
    // flat16 == NULL: only the listed matrices (a model refreshes just the images it reads)
    LBX_ARG(flat && n >= 0 && nmats >= 0 && (nmats == 0 || mats), "flat != NULL; nmats >= 0");
    LBX_ARG(aligned16(flat) && (((uintptr_t)flat16) & 7) == 0, "flat 16-byte, flat16 8-byte aligned");
    for (int i = 0; i < nmats; ++i) {
        const lidbox_weight_shadow_t& m = mats[i];
        LBX_ARG(m.dst && m.rows >= 1 && m.cols >= 1 && m.offset >= 0 && m.offset + (long)m.rows * m.cols <= n &&
                m.ld_dst >= (m.transpose ? m.rows : m.cols) && m.ld_dst <= 0x7fffffffL,
                "every matrix lies inside the flat vector and ld_dst covers a destination row");
    }
    long cb = lbx_cdiv(n / 4 + 1, 256);
    if (cb > 1024) cb = 1024;
    // the conversion of the flat vector rides with the first chunk of (at most 48) matrices
    int done = 0;
    do {
        WeightShadows w{};
        int tiles = 0, k = 0;
        for (; k < MAX_WT && done + k < nmats; ++k) {
            const lidbox_weight_shadow_t& m = mats[done + k];
            w.src_off[k] = m.offset;
            w.dst[k] = (unsigned short*)m.dst;
            w.R[k] = m.rows | (m.transpose ? (int)0x80000000 : 0);
            w.C[k] = m.cols;
            w.ld_dst[k] = (int)m.ld_dst;
            w.vec[k] = (m.cols % 4 == 0 && m.offset % 4 == 0 && m.ld_dst % 4 == 0 && ((uintptr_t)m.dst & 7) == 0 &&
                        (!m.transpose || m.rows % 4 == 0)) ? 1 : 0;
            tiles += (int)(lbx_cdiv(m.rows, WT_TILE) * lbx_cdiv(m.cols, WT_TILE));
            w.tile_end[k] = tiles;
        }
        w.n = k;
        const long conv = done == 0 ? (n > 0 && flat16 ? cb : 0) : 0;
        if (conv + tiles > 0) {
            hipLaunchKernelGGL(refresh_bf16_weights_kernel, dim3((unsigned)(conv + tiles)), dim3(256), 0, (hipStream_t)stream, flat,
                               (unsigned short*)flat16, n, (int)conv, w);
            LBX_LAUNCH_OK();
        }
        done += k;
    } while (done < nmats);
    return LIDBOX_OK;
}

extern "C" int lidbox_transpose_f32_to_bf16(const float* src, int R, int C, long ld_src, void* dst, long ld_dst,
                                            lidbox_stream_t stream) {
    LBX_ARG(src && dst && R >= 0 && C >= 0 && ld_src >= C && ld_dst >= R, "src, dst != NULL; leading dimensions cover the rows");
    if (R == 0 || C == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(transpose_f32_to_bf16_kernel, dim3((unsigned)lbx_cdiv(C, 32), (unsigned)lbx_cdiv(R, 32)), dim3(256), 0,
                       (hipStream_t)stream, src, R, C, ld_src, (unsigned short*)dst, ld_dst);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

namespace {
}  // namespace

extern "C" size_t lidbox_gemm_bf16_rows_workspace(long M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const Rows16Plan pl = plan_rows16(M, N, K, (size_t)64 << 20);
    return pl.splits > 1 ? (size_t)pl.splits * M * N * sizeof(float) : 0;
}

extern "C" int lidbox_gemm_bf16_nn(lidbox_rows_t A, const float* Bm, long ldb, lidbox_rows_out_t C, int K, int N,
                                   int epilogue, const float* aux, void* workspace, size_t workspace_bytes,
                                   lidbox_stream_t stream) {
    if (validate_rows_call(__func__, A, Bm, ldb, C, K, N, epilogue, aux, N)) return LIDBOX_E_INVALID;
    return launch_rows16<false>(__func__, A, Bm, ldb, C, K, N, epilogue, aux, workspace, workspace_bytes,
                                (hipStream_t)stream);
}

extern "C" int lidbox_gemm_bf16_nt(lidbox_rows_t A, const float* Bm, long ldb, lidbox_rows_out_t C, int K, int N,
                                   int epilogue, const float* aux, void* workspace, size_t workspace_bytes,
                                   lidbox_stream_t stream) {
    if (validate_rows_call(__func__, A, Bm, ldb, C, K, N, epilogue, aux, K)) return LIDBOX_E_INVALID;
    return launch_rows16<true>(__func__, A, Bm, ldb, C, K, N, epilogue, aux, workspace, workspace_bytes,
                               (hipStream_t)stream);
}

extern "C" size_t lidbox_gemm_bf16_tn_workspace(int M, int K1, int N) {
    if (M <= 0 || K1 <= 0 || N <= 0) return 0;
    const Tn16Plan pl = plan_tn16(M, K1, N);
    return ((size_t)pl.splits * K1 * N + (size_t)pl.splits * N) * sizeof(float);
}

extern "C" int lidbox_gemm_bf16_tn(lidbox_rows_t A, lidbox_rows_t Bd, float* Cm, long ldc, int K1, int N,
                                   int accumulate, float* bias_grad, void* workspace, size_t workspace_bytes,
                                   lidbox_stream_t stream) {
    if (check_rows(__func__, A.base, A.batch_stride, A.row_stride, A.batch, A.rows_per_batch)) return LIDBOX_E_INVALID;
    if (check_rows(__func__, Bd.base, Bd.batch_stride, Bd.row_stride, Bd.batch, Bd.rows_per_batch)) return LIDBOX_E_INVALID;
    LBX_ARG(Cm && K1 >= 1 && N >= 1 && ldc >= N, "C != NULL, K1, N >= 1, ldc >= N");
    const long M = (long)A.batch * A.rows_per_batch;
    LBX_ARG(M == (long)Bd.batch * Bd.rows_per_batch, "A and B row counts differ");
    LBX_ARG(M >= 1, "M >= 1");
    LBX_ARG(rows_aligned(A) && rows_aligned(Bd) && K1 % 4 == 0 && N % 4 == 0 && aligned16(workspace), ALIGN_MSG);
    const Tn16Plan pl = plan_tn16(M, K1, N);
    const size_t need = ((size_t)pl.splits * K1 * N + (size_t)pl.splits * N) * sizeof(float);
    LBX_ARG(workspace && workspace_bytes >= need, "workspace too small (lidbox_gemm_bf16_tn_workspace)");
    hipStream_t st = (hipStream_t)stream;
    const int tiles_n = (int)lbx_cdiv(N, BT);
    const int ntiles = (int)(lbx_cdiv(K1, BT) * tiles_n);
    float* P = (float*)workspace;
    float* Pc = bias_grad ? P + (size_t)pl.splits * K1 * N : nullptr;
    hipLaunchKernelGGL(gemm16_tn_kernel, dim3((unsigned)(ntiles * pl.splits)), dim3(256), 0, st, to_dev(A), to_dev(Bd), P,
                       Pc, M, K1, N, tiles_n, ntiles, pl.rows_per_split);
    LBX_LAUNCH_OK();
    const long n = (long)K1 * N;
    launch_splitk_reduce((const float*)P, (const float*)Pc, pl.splits, n, N, Cm, ldc, accumulate, bias_grad, st);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}
