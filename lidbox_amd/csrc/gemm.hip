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
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD): 128x128 block tile, 4 waves as
//     2x2, each wave 64x64 = 2x2 MFMA blocks (64 accumulator VGPRs).
//   * LDS tiles are K-outer ([k][row]) for both operands so every MFMA operand fetch is one
//     conflict-free ds_read_b32 of 32 consecutive floats per half-wave.  Operands whose
//     contraction index is contiguous in HBM (A of NN/NT, B of NT) are transposed on the way
//     in (row stride = 128 + 8/(BK/4) keeps those scattered ds_write_b32 conflict-free);
//     the others are copied with ds_write_b128.
//   * global -> register prefetch of tile t+1 is issued before the MFMAs of tile t and
//     written to the other LDS buffer afterwards: one barrier per K step.
//   * block id -> tile uses the XCD-chunk remap so the n-tiles that share an A row panel run
//     on one XCD's L2; the weight matrix (<= 3 MB) is L2-resident everywhere.
//   * wgrad contracts over M = B*T (up to 50 688): split across workgroups into partial sums in
//     a workspace and reduced in a fixed order (deterministic; no float atomics).
// Roofline: MFMA fp32, 157.3 TFLOP/s.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128;

struct RowsD {
    const float* base;
    long bs, rs;
    int batch, rpb;
};

__device__ __forceinline__ long row_offset(const RowsD& r, long m) {
    if (r.batch == 1) return m * r.rs;
    const long b = m / r.rpb;
    return b * r.bs + (m - b * r.rpb) * r.rs;
}

template <int BK>
struct Cfg {
    static constexpr int F4_PER_ROW = BK / 4;                 // K-inner operand: float4 per tile row
    static constexpr int ROWS_PER_PASS = 256 / F4_PER_ROW;
    static constexpr int KI_PASSES = 128 / ROWS_PER_PASS;
    static constexpr int LDT = 128 + 8 / F4_PER_ROW;          // transposed-store row stride (floats)
    static constexpr int KO_PASSES = BK / 8;                  // K-outer operand: 8 k-rows per pass
    static constexpr int LDD = 128;                           // direct-copy row stride
    static constexpr int TILE_FLOATS = BK * LDT;              // per operand per buffer (max of both)
};

// ---- K-inner operand (contraction index contiguous in HBM): rows r0+p*ROWS_PER_PASS, float4 c4
template <int BK, bool ALIGNED>
struct KInnerLoader {
    using C = Cfg<BK>;
    float4 v[C::KI_PASSES];
    long off[C::KI_PASSES];
    bool rok[C::KI_PASSES];
    int c4, r0;

    __device__ __forceinline__ void init(const RowsD& rows, long row_base, long nrows, int tid) {
        c4 = tid % C::F4_PER_ROW;
        r0 = tid / C::F4_PER_ROW;
#pragma unroll
        for (int p = 0; p < C::KI_PASSES; ++p) {
            const long r = row_base + r0 + p * C::ROWS_PER_PASS;
            rok[p] = r < nrows;
            off[p] = rok[p] ? row_offset(rows, r) : 0;
        }
    }
    __device__ __forceinline__ void load(const float* base, int k0, int K) {
        const int k = k0 + c4 * 4;
#pragma unroll
        for (int p = 0; p < C::KI_PASSES; ++p) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rok[p]) {
                const float* src = base + off[p] + k;
                if (ALIGNED) {
                    if (k < K) x = *reinterpret_cast<const float4*>(src);
                } else {
                    if (k + 0 < K) x.x = src[0];
                    if (k + 1 < K) x.y = src[1];
                    if (k + 2 < K) x.z = src[2];
                    if (k + 3 < K) x.w = src[3];
                }
            }
            v[p] = x;
        }
    }
    __device__ __forceinline__ void store(float* tile) const {
#pragma unroll
        for (int p = 0; p < C::KI_PASSES; ++p) {
            float* d = tile + (c4 * 4) * C::LDT + r0 + p * C::ROWS_PER_PASS;
            d[0 * C::LDT] = v[p].x;
            d[1 * C::LDT] = v[p].y;
            d[2 * C::LDT] = v[p].z;
            d[3 * C::LDT] = v[p].w;
        }
    }
};

// ---- K-outer operand, plain matrix [Kdim][ld] (weights in NN): k-rows kk0+8p, float4 column c4
template <int BK, bool ALIGNED>
struct KOuterLoader {
    using C = Cfg<BK>;
    float4 v[C::KO_PASSES];
    int c4, kk0;

    __device__ __forceinline__ void init(int tid) {
        c4 = tid & 31;
        kk0 = tid >> 5;
    }
    // generic: per-k-row offsets supplied by the caller through a functor
    template <typename RowOff>
    __device__ __forceinline__ void load(const float* base, RowOff row_off, long k0, long Kdim, long col0,
                                         long ncols) {
        const long c = col0 + c4 * 4;
#pragma unroll
        for (int p = 0; p < C::KO_PASSES; ++p) {
            const long k = k0 + kk0 + 8 * p;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < Kdim) {
                const float* src = base + row_off(k) + c;
                if (ALIGNED) {
                    if (c < ncols) x = *reinterpret_cast<const float4*>(src);
                } else {
                    if (c + 0 < ncols) x.x = src[0];
                    if (c + 1 < ncols) x.y = src[1];
                    if (c + 2 < ncols) x.z = src[2];
                    if (c + 3 < ncols) x.w = src[3];
                }
            }
            v[p] = x;
        }
    }
    __device__ __forceinline__ void store(float* tile) const {
#pragma unroll
        for (int p = 0; p < C::KO_PASSES; ++p)
            *reinterpret_cast<float4*>(tile + (kk0 + 8 * p) * C::LDD + c4 * 4) = v[p];
    }
};

// 64x64 per wave: 2x2 blocks of v_mfma_f32_32x32x2_f32 over one BK-deep LDS tile pair
template <int BK, int LDA, int LDB>
__device__ __forceinline__ void mma_tile(const float* As, const float* Bs, int wm, int wn, int lane,
                                         f32x16 (&acc)[2][2]) {
    const int h = lane >> 5, l = lane & 31;
    const float* ap = As + h * LDA + wm * 64 + l;
    const float* bp = Bs + h * LDB + wn * 64 + l;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
        const float a0 = ap[kk * LDA], a1 = ap[kk * LDA + 32];
        const float b0 = bp[kk * LDB], b1 = bp[kk * LDB + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
}

struct RowsOutD {
    float* base;
    long bs, rs;
    int batch, rpb;
};

// ------------------------------------------------------------------------------------------------
// C[M,N] = epi(A[M,K] . B)    B_KINNER = false: B[K][N] (NN)   true: B[N][K] (NT)
// ------------------------------------------------------------------------------------------------
template <int BK, bool B_KINNER, bool ALIGNED>
__global__ __launch_bounds__(256) void gemm_rows_kernel(RowsD A, const float* __restrict__ Bm, long ldb,
                                                        RowsOutD Cd, long M, int K, int N, int epi,
                                                        const float* __restrict__ aux, int tiles_n,
                                                        unsigned nwg) {
    using C = Cfg<BK>;
    constexpr int LDB = B_KINNER ? C::LDT : C::LDD;
    __shared__ __attribute__((aligned(16))) float As[2][BK * C::LDT];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned chunk = xcd_chunk_id(blockIdx.x, nwg);
    const int tn = chunk % tiles_n;
    const long tm = chunk / tiles_n;
    const long m0 = tm * BM;
    const int n0 = tn * BN;

    KInnerLoader<BK, ALIGNED> la;
    la.init(A, m0, M, tid);
    KInnerLoader<BK, ALIGNED> lbi;
    KOuterLoader<BK, ALIGNED> lbo;
    RowsD Brows{Bm, 0, ldb, 1, 0};
    if (B_KINNER) lbi.init(Brows, n0, N, tid);
    else lbo.init(tid);
    auto b_off = [&](long k) { return k * ldb; };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (K + BK - 1) / BK;
    la.load(A.base, 0, K);
    if (B_KINNER) lbi.load(Bm, 0, K);
    else lbo.load(Bm, b_off, 0, K, n0, N);
    la.store(As[0]);
    if (B_KINNER) lbi.store(Bs[0]);
    else lbo.store(Bs[0]);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            la.load(A.base, (kt + 1) * BK, K);
            if (B_KINNER) lbi.load(Bm, (kt + 1) * BK, K);
            else lbo.load(Bm, b_off, (long)(kt + 1) * BK, K, n0, N);
        }
        mma_tile<BK, C::LDT, LDB>(As[cur], Bs[cur], wm, wn, lane, acc);
        if (more) {
            la.store(As[cur ^ 1]);
            if (B_KINNER) lbi.store(Bs[cur ^ 1]);
            else lbo.store(Bs[cur ^ 1]);
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5) of each block
    const int h = lane >> 5, l = lane & 31;
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
        const int col = n0 + wn * 64 + bj * 32 + l;
        if (col >= N) continue;
        const float bias = (epi == LIDBOX_EPI_BIAS || epi == LIDBOX_EPI_BIAS_RELU) ? aux[col] : 0.f;
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = m0 + wm * 64 + bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= M) continue;
                long off;
                if (Cd.batch == 1) off = row * Cd.rs;
                else { const long b = row / Cd.rpb; off = b * Cd.bs + (row - b * Cd.rpb) * Cd.rs; }
                float* dst = Cd.base + off + col;
                float v = acc[bi][bj][r];
                switch (epi) {
                    case LIDBOX_EPI_BIAS: v += bias; break;
                    case LIDBOX_EPI_BIAS_RELU: v = fmaxf(v + bias, 0.f); break;
                    case LIDBOX_EPI_RELU_MASK: v = aux[off + col] > 0.f ? v : 0.f; break;
                    case LIDBOX_EPI_ACCUM: v += *dst; break;
                    case LIDBOX_EPI_ACCUM_RELU_MASK: v = *dst + (aux[off + col] > 0.f ? v : 0.f); break;
                    default: break;
                }
                *dst = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad: P[split][K1][N] = A[Mslice, K1]^T . B[Mslice, N]
// ------------------------------------------------------------------------------------------------
template <int BK, bool ALIGNED>
__global__ __launch_bounds__(256) void gemm_tn_kernel(RowsD A, RowsD Bd, float* __restrict__ P, long M,
                                                      int K1, int N, int tiles_n, int ntiles, int splits,
                                                      long rows_per_split) {
    using C = Cfg<BK>;
    __shared__ __attribute__((aligned(16))) float As[2][BK * C::LDD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * C::LDD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // consecutive block ids = the splits of one tile pair -> spread over XCDs; tiles share nothing
    const int tile = blockIdx.x % ntiles;
    const int split = blockIdx.x / ntiles;
    const int tn = tile % tiles_n, tk = tile / tiles_n;
    const int i0 = tk * BM, n0 = tn * BN;
    const long mbeg = (long)split * rows_per_split;
    long mend = mbeg + rows_per_split;
    if (mend > M) mend = M;

    KOuterLoader<BK, ALIGNED> la, lb;
    la.init(tid);
    lb.init(tid);
    auto a_off = [&](long m) { return row_offset(A, m); };
    auto b_off = [&](long m) { return row_offset(Bd, m); };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (int)((mend - mbeg + BK - 1) / BK);
    if (nk > 0) {
        la.load(A.base, a_off, mbeg, mend, i0, K1);
        lb.load(Bd.base, b_off, mbeg, mend, n0, N);
        la.store(As[0]);
        lb.store(Bs[0]);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            la.load(A.base, a_off, mbeg + (long)(kt + 1) * BK, mend, i0, K1);
            lb.load(Bd.base, b_off, mbeg + (long)(kt + 1) * BK, mend, n0, N);
        }
        mma_tile<BK, C::LDD, C::LDD>(As[cur], Bs[cur], wm, wn, lane, acc);
        if (more) {
            la.store(As[cur ^ 1]);
            lb.store(Bs[cur ^ 1]);
        }
        __syncthreads();
    }
    float* Pd = P + (long)split * K1 * N;
    const int h = lane >> 5, l = lane & 31;
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
        const int col = n0 + wn * 64 + bj * 32 + l;
        if (col >= N) continue;
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < K1) Pd[(long)row * N + col] = acc[bi][bj][r];
            }
    }
}

// C[i] (+)= sum_s P[s][i], fixed order
__global__ void splitk_reduce_kernel(const float* __restrict__ P, int splits, long n, int K1, int N,
                                     float* __restrict__ Cm, long ldc, int accumulate) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += P[(long)k * n + i];
        const long row = i / N, col = i - row * N;
        float* d = Cm + row * ldc + col;
        *d = accumulate ? *d + s : s;
    }
}

// column sums: stage 1 partial[rs][N] over row slices, stage 2 fixed-order reduce
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
        for (long m = mbeg + g; m < mend; m += 4) s += A.base[row_offset(A, m) + c];
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

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

inline bool rows_aligned(const lidbox_rows_t& r) {
    return aligned16(r.base) && r.row_stride % 4 == 0 && (r.batch == 1 || r.batch_stride % 4 == 0);
}

inline RowsD to_dev(const lidbox_rows_t& r) { return RowsD{r.base, r.batch_stride, r.row_stride, r.batch, r.rows_per_batch}; }

constexpr int GEMM_BK = 16;

template <bool B_KINNER>
int launch_rows(lidbox_rows_t A, const float* Bm, long ldb, lidbox_rows_out_t Cd, int K, int N, int epi,
                const float* aux, hipStream_t st) {
    const long M = (long)A.batch * A.rows_per_batch;
    if (M == 0 || N == 0) return LIDBOX_OK;
    const int tiles_n = (int)lbx_cdiv(N, BN);
    const long tiles_m = lbx_cdiv(M, BM);
    const long nwg = tiles_m * tiles_n;
    // float4 paths: A rows and K multiple of 4; B: NN needs N%4 (columns), NT needs K%4 (rows)
    const bool al = rows_aligned(A) && K % 4 == 0 && aligned16(Bm) && ldb % 4 == 0 &&
                    (B_KINNER ? true : N % 4 == 0);
    RowsD Ad = to_dev(A);
    RowsOutD Co{Cd.base, Cd.batch_stride, Cd.row_stride, Cd.batch, Cd.rows_per_batch};
    if (al)
        hipLaunchKernelGGL((gemm_rows_kernel<GEMM_BK, B_KINNER, true>), dim3((unsigned)nwg), dim3(256), 0, st,
                           Ad, Bm, ldb, Co, M, K, N, epi, aux, tiles_n, (unsigned)nwg);
    else
        hipLaunchKernelGGL((gemm_rows_kernel<GEMM_BK, B_KINNER, false>), dim3((unsigned)nwg), dim3(256), 0, st,
                           Ad, Bm, ldb, Co, M, K, N, epi, aux, tiles_n, (unsigned)nwg);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

int check_rows(const char* fn, const void* base, long bs, long rs, int batch, int rpb) {
    if (!base || batch < 0 || rpb < 0 || rs < 0 || bs < 0) {
        lidbox_set_error("%s: invalid rows descriptor", fn);
        return LIDBOX_E_INVALID;
    }
    return LIDBOX_OK;
}

void tn_plan(long M, int K1, int N, int* splits, long* rows_per_split) {
    const long ntiles = lbx_cdiv(K1, BM) * lbx_cdiv(N, BN);
    long s = lbx_cdiv(1024, ntiles);                 // aim at ~4 workgroups per CU
    const long max_s = lbx_cdiv(M, 8 * GEMM_BK);     // at least 8 K-steps per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    long rps = lbx_cdiv(M, s);
    rps = lbx_cdiv(rps, GEMM_BK) * GEMM_BK;
    s = lbx_cdiv(M, rps);
    *splits = (int)s;
    *rows_per_split = rps;
}

}  // namespace

extern "C" int lidbox_gemm_nn(lidbox_rows_t A, const float* Bm, long ldb, lidbox_rows_out_t C, int K, int N,
                              int epilogue, const float* aux, lidbox_stream_t stream) {
    if (check_rows(__func__, A.base, A.batch_stride, A.row_stride, A.batch, A.rows_per_batch)) return LIDBOX_E_INVALID;
    if (check_rows(__func__, C.base, C.batch_stride, C.row_stride, C.batch, C.rows_per_batch)) return LIDBOX_E_INVALID;
    LBX_ARG(Bm && K >= 1 && N >= 0 && ldb >= N, "B != NULL, K >= 1, ldb >= N");
    LBX_ARG((long)A.batch * A.rows_per_batch == (long)C.batch * C.rows_per_batch, "A and C row counts differ");
    LBX_ARG(epilogue >= LIDBOX_EPI_NONE && epilogue <= LIDBOX_EPI_ACCUM_RELU_MASK, "epilogue");
    LBX_ARG(!(epilogue == LIDBOX_EPI_BIAS || epilogue == LIDBOX_EPI_BIAS_RELU || epilogue == LIDBOX_EPI_RELU_MASK ||
              epilogue == LIDBOX_EPI_ACCUM_RELU_MASK) || aux, "aux required by this epilogue");
    return launch_rows<false>(A, Bm, ldb, C, K, N, epilogue, aux, (hipStream_t)stream);
}

extern "C" int lidbox_gemm_nt(lidbox_rows_t A, const float* Bm, long ldb, lidbox_rows_out_t C, int K, int N,
                              int epilogue, const float* aux, lidbox_stream_t stream) {
    if (check_rows(__func__, A.base, A.batch_stride, A.row_stride, A.batch, A.rows_per_batch)) return LIDBOX_E_INVALID;
    if (check_rows(__func__, C.base, C.batch_stride, C.row_stride, C.batch, C.rows_per_batch)) return LIDBOX_E_INVALID;
    LBX_ARG(Bm && K >= 1 && N >= 0 && ldb >= K, "B != NULL, K >= 1, ldb >= K");
    LBX_ARG((long)A.batch * A.rows_per_batch == (long)C.batch * C.rows_per_batch, "A and C row counts differ");
    LBX_ARG(epilogue >= LIDBOX_EPI_NONE && epilogue <= LIDBOX_EPI_ACCUM_RELU_MASK, "epilogue");
    LBX_ARG(!(epilogue == LIDBOX_EPI_BIAS || epilogue == LIDBOX_EPI_BIAS_RELU || epilogue == LIDBOX_EPI_RELU_MASK ||
              epilogue == LIDBOX_EPI_ACCUM_RELU_MASK) || aux, "aux required by this epilogue");
    return launch_rows<true>(A, Bm, ldb, C, K, N, epilogue, aux, (hipStream_t)stream);
}

extern "C" size_t lidbox_gemm_tn_workspace(int M, int K1, int N) {
    if (M <= 0 || K1 <= 0 || N <= 0) return 0;
    int splits;
    long rps;
    tn_plan(M, K1, N, &splits, &rps);
    return (size_t)splits * K1 * N * sizeof(float);
}

extern "C" int lidbox_gemm_tn(lidbox_rows_t A, lidbox_rows_t Bd, float* Cm, long ldc, int K1, int N,
                              int accumulate, void* workspace, size_t workspace_bytes, lidbox_stream_t stream) {
    if (check_rows(__func__, A.base, A.batch_stride, A.row_stride, A.batch, A.rows_per_batch)) return LIDBOX_E_INVALID;
    if (check_rows(__func__, Bd.base, Bd.batch_stride, Bd.row_stride, Bd.batch, Bd.rows_per_batch)) return LIDBOX_E_INVALID;
    LBX_ARG(Cm && K1 >= 1 && N >= 1 && ldc >= N, "C != NULL, K1, N >= 1, ldc >= N");
    const long M = (long)A.batch * A.rows_per_batch;
    LBX_ARG(M == (long)Bd.batch * Bd.rows_per_batch, "A and B row counts differ");
    LBX_ARG(M >= 1 && M <= 0x7fffffffL, "1 <= M < 2^31");
    int splits;
    long rps;
    tn_plan(M, K1, N, &splits, &rps);
    const size_t need = (size_t)splits * K1 * N * sizeof(float);
    LBX_ARG(workspace && workspace_bytes >= need, "workspace too small (lidbox_gemm_tn_workspace)");
    hipStream_t st = (hipStream_t)stream;
    const int tiles_n = (int)lbx_cdiv(N, BN);
    const int ntiles = (int)(lbx_cdiv(K1, BM) * tiles_n);
    const bool al = rows_aligned(A) && rows_aligned(Bd) && K1 % 4 == 0 && N % 4 == 0;
    float* P = (float*)workspace;
    if (al)
        hipLaunchKernelGGL((gemm_tn_kernel<GEMM_BK, true>), dim3((unsigned)(ntiles * splits)), dim3(256), 0, st,
                           to_dev(A), to_dev(Bd), P, M, K1, N, tiles_n, ntiles, splits, rps);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<GEMM_BK, false>), dim3((unsigned)(ntiles * splits)), dim3(256), 0, st,
                           to_dev(A), to_dev(Bd), P, M, K1, N, tiles_n, ntiles, splits, rps);
    LBX_LAUNCH_OK();
    const long n = (long)K1 * N;
    long g = lbx_cdiv(n, 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, P, splits, n, K1, N, Cm, ldc,
                       accumulate);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
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
