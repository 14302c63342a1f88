// gemm16_tn_kres.h -- bf16-storage wgrad of a short-contraction Conv1D whose windows overlap (frame1 of the x-vector: k = 5 frames of
// 40 mel bands, stride 1: K1 = 200; reference lidbox/models/xvector.py:53 under tf.GradientTape), included by gemm_bf16.hip.
//
//   P[slice][K1][N] = sum over the slice's rows m of  X_window[m][K1]^T . dY[m][N]
//
// On the 128 x 128 tiles of gemm16s_tn_kernel this launch is the worst of the bf16 step (round 5: 60.7 us at 512 utterances, 0.137 of
// the MFMA peak): K1 = 200 pads to two 128-row tiles (1.28 x the work), each of which streams dY again, and every tile re-fetches its
// im2col rows although consecutive windows share 4 of their 5 frames.  Here the WHOLE K1 extent (<= 224 = seven 32-row MFMA blocks)
// stays resident in one workgroup's accumulators:
//   * a workgroup owns 128 output columns and a range of utterances; dY streams through it once (the launch's only large operand);
//   * the windows are not materialised: with the batch stride a multiple of the row stride, window m of the flattened input is the
//     K1 elements that start at element m * rs, so ONE copy of the utterances' frames in LDS (63 rs + K1 elements per 64-row stage,
//     5.4 KB) serves all 64 windows of a stage -- the transpose read (ds_read_b64_tr_b16) takes any row stride, here rs instead of a
//     tile's.  Rows of the flattened range that are no output row (the k - 1 pad rows between utterances) meet zero rows of dY;
//   * eight waves: wave w takes column block w & 3 (32 columns) and the row blocks 0-3 (w < 4) or 4-6 (w >= 4): no accumulator is
//     shared, a substep is 1 + 4 (or 3) transpose-read operands for 4 (3) MFMAs;
//   * register-staged like gemm16s_tn_kernel, with the loads as bounds-checked raw buffer loads issued three 64-row stages ahead into
//     four register sets (branch-free: hipcc's vmcnt stays exact) and one LDS-only barrier per stage (__syncthreads would drain the
//     prefetch); bias gradient = fp32 column sums of the staged dY pieces; raw fp32 slabs P[slice] summed by the caller's carried
//     reduce job (fixed order, no atomics).
// Measured at 512 utterances (profiles/r06_bf16_tn_kres_frame1.txt): 38.4-39.3 us against 56.0-56.7 us on the four-wave tiles (the
// ping-pong tile: 38.3, with 52 MB of slabs against 26 MB here).  Ablated: no MFMA loop 21 us, neither loads nor MFMAs 15 us (launch,
// 26 MB of slab stores, 26 x (3 ds_write_b128 + barrier)).  Tried and dropped, each parity-green: the bias gradient as an extra MFMA
// against an all-ones block in the upper waves instead of loader VALU (41.8 us); the upper four waves doing all of the staging beside
// the lower waves' MFMAs (44.9 us); a frames image in the input's own 80-byte rows (2-way conflicts on the A fetches: same time).
#pragma once

namespace {

constexpr int TKR_ROWS = 64;                                       // contraction rows per LDS stage
constexpr int TKR_BN = 128;                                        // output columns per workgroup
constexpr int TKR_LDB = 160;                                       // LDS row stride of the dY tile in bf16 (gemm16s_tn_kernel's)
constexpr int TKR_B_BYTES = TKR_ROWS * TKR_LDB * 2;                // 20 480
constexpr int TKR_IMG_ELEMS = 4096;                                // frames of one stage: 63 rs + K1 <= this (one 16-byte piece per thread)
constexpr int TKR_IMG_BYTES = 32768;                               // their LDS image, rows padded (tkr_img_stride)
constexpr int TKR_STAGE_BYTES = TKR_B_BYTES + TKR_IMG_BYTES;       // 53 248
constexpr int TKR_LDS_BYTES = 2 * TKR_STAGE_BYTES;                 // 106 496
constexpr int TKR_MAX_K1 = 224;
constexpr int TKR_PD = 4;                                          // stages of loads in flight (register sets); the step macro is unrolled by it

#ifndef LBX_TKR_ABLATE
#define LBX_TKR_ABLATE 0                                            // measurement builds only (wrong results): 1 no compute, 2 no loads, 4 no slab stores,
#endif                                                              // 8 no operand fetches after a stage's first substep

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_tkr;
typedef unsigned u32x4_tkr __attribute__((ext_vector_type(4)));

// LDS bytes between consecutive input rows (rs elements each) of the image: a 32-lane group of the transpose read touches 4 rows x 64
// bytes, which cover the 64 banks once when the rows start 64 bytes apart modulo 256 (the input's own 2 rs = 80 bytes put row 3 on
// row 0's banks: 2-way conflicts on every A fetch)
__host__ __device__ inline int tkr_img_stride(int rs) {
    int s = 2 * rs;
    while (s % 256 != 64) s += 8;
    return s;
}

// Workgroup barrier that publishes LDS writes only: __syncthreads() also waits for every global load in flight (s_waitcnt vmcnt(0)),
// which would drain the TKR_PD - 1 stages of prefetch at every stage
__device__ __forceinline__ void tkr_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// 8 consecutive contraction rows of the lane's column out of a [row][column] image with `stride` elements between rows
__device__ __forceinline__ bf16x8 tkr_tr8(const __bf16* p, const int stride) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_tkr*)(p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_tkr*)(p + 4 * stride));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(512) void gemm16s_tn_kres_kernel(RowsH A, RowsH Bd, float* __restrict__ P, float* __restrict__ Pc, int K1,
                                                              int N, int tiles_n, int ups, int Tq, long a_elems, long b_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem_tkr[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the block counts below steer scalar branches
    const int cb = wave & 3, rh = wave >> 2;
    // block -> (slice, column tile) through the XCD-chunk remap: the column tiles of a slice read the same rows of dY and the same
    // frames, they run on one XCD
    const unsigned vb = xcd_chunk_id(blockIdx.x, gridDim.x);
    const int tn = (int)(vb % (unsigned)tiles_n), slice = (int)(vb / (unsigned)tiles_n);
    const int n0 = tn * TKR_BN;
    const int b0 = slice * ups;
    const int b1 = min(b0 + ups, A.batch);
    const int rs = (int)A.rs;
    const long e0 = (long)b0 * Tq * rs;                             // first element of the slice in the flattened input
    const int nrows = (b1 - b0) * Tq;
    const int nst = (nrows + TKR_ROWS - 1) / TKR_ROWS;
    const int img_elems = 63 * rs + K1;                             // multiple of 8 (host)
    const int S = tkr_img_stride(rs);                               // LDS bytes per input row of the image
    // the thread's piece of the image: elements 8 tid .. of the stage's frames -> row (8 tid) / rs (pieces behind the stage's frames
    // carry zeros and land in the image's tail, below TKR_IMG_BYTES: 512 x 8 elements <= 4096 + a row)
    const int img_dst = min(((8 * tid) / rs) * S + ((8 * tid) % rs) * 2, TKR_IMG_BYTES - 16);
    const int nblk = (K1 + 31) >> 5;                                // 32-row blocks of the output (<= 7)
    const int nb = rh == 0 ? min(4, nblk) : max(0, nblk - 4);       // this wave's blocks: 4 rh .. 4 rh + nb - 1

    // ---- loader: two 16-byte pieces of dY per thread and stage (rows (tid >> 4) + 32 j, columns n0 + 8 (tid & 15) ..), one of the image
    //      (elements 8 tid ..), as RAW BUFFER LOADS: a piece that is no row of the slice (pad rows between utterances, rows behind the
    //      slice, stages behind the last one) gets an offset outside the descriptor and comes back as zeros from the bounds check --
    //      no branch, no select on the data, so that the TKR_PD stages in flight stay straight-line code whose vmcnt hipcc counts exactly
    const int pc = tid & 15;
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Bd.base), 0, (int)b_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(A.base), 0, (int)(a_elems * 2), 0x00020000);
    const unsigned bcol2 = (unsigned)(n0 + 8 * pc) * 2u;
    const unsigned uTq = (unsigned)Tq, urpb = (unsigned)Bd.rpb, nb1 = (unsigned)(b1 - b0);
    // TKR_PD register sets: the loads of a stage are issued TKR_PD - 1 stages before it is staged -- one stage's compute (~0.5 us of
    // MFMA / LDS work) is far shorter than an HBM round trip under load, and one workgroup per CU has nobody else to hide it
    u32x4 rb_[TKR_PD][2], ri_[TKR_PD];
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // Per piece: utterance of the slice, row inside it and byte offset of the NEXT stage to fetch, stepped by 64 rows per call with
    // adds and selects (utterances of >= 64 flattened rows wrap at most once per step; shorter ones -- tests -- take the divisions):
    // two waves per SIMD run the same phase, so every vector instruction of the loader is time the matrix pipe idles
    unsigned pu_[2], pt_[2], po_[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned lr = (unsigned)((tid >> 4) + 32 * j);
        pu_[j] = lr / uTq;
        pt_[j] = lr - pu_[j] * uTq;
        po_[j] = (unsigned)(((long)(b0 + (int)pu_[j]) * Bd.bs + (long)pt_[j] * Bd.rs) * 2) + bcol2;
    }
    const bool fast_wrap = uTq >= (unsigned)TKR_ROWS;                                   // kernel-uniform
    const unsigned step2 = (unsigned)(TKR_ROWS * Bd.rs * 2), wrap2 = (unsigned)((Bd.bs - (long)Tq * Bd.rs) * 2);
    unsigned io_ = (unsigned)((e0 + 8 * tid) * 2);                                      // image piece: byte offset of the next stage
    const unsigned istep2 = (unsigned)(TKR_ROWS * rs * 2);
    const bool img_lane = 8 * tid < img_elems;
    // (fetch is called for stages 0, 1, 2, ... in order: the trackers step with it)
    auto fetch = [&](int s, u32x4 (&rb)[2], u32x4& ri) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rB, (int)((pu_[j] < nb1 && pt_[j] < urpb) ? po_[j] : 0xfffffff0u), 0, 0);
            if (fast_wrap) {
                pt_[j] += TKR_ROWS;
                po_[j] += step2;
                const bool w = pt_[j] >= uTq;
                pt_[j] -= w ? uTq : 0u;
                pu_[j] += w ? 1u : 0u;
                po_[j] += w ? wrap2 : 0u;
            } else {
                const unsigned lr = (unsigned)((s + 1) * TKR_ROWS + (tid >> 4) + 32 * j);
                pu_[j] = lr / uTq;
                pt_[j] = lr - pu_[j] * uTq;
                po_[j] = (unsigned)(((long)(b0 + (int)pu_[j]) * Bd.bs + (long)pt_[j] * Bd.rs) * 2) + bcol2;
            }
        }
        ri = __builtin_amdgcn_raw_buffer_load_b128(rA, (int)((img_lane && s < nst) ? io_ : 0xfffffff0u), 0, 0);
        io_ += istep2;
    };
    auto stage = [&](int buf, const u32x4 (&rb)[2], const u32x4& ri) {
        char* sb = smem_tkr + buf * TKR_STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<u32x4*>(sb + (((tid >> 4) + 32 * j) * TKR_LDB + 8 * pc) * 2) = rb[j];
#pragma unroll
            for (int w = 0; w < 4; ++w) {                           // fp32 column sums of what is staged: the bias gradient
                const unsigned u = rb[j][w];
                csum[2 * w] += __builtin_bit_cast(float, u << 16);
                csum[2 * w + 1] += __builtin_bit_cast(float, u & 0xffff0000u);
            }
        }
        *reinterpret_cast<u32x4*>(sb + TKR_B_BYTES + img_dst) = ri;
    };

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;


    // lane l = 16 g + j addresses the 4-column piece (j & 3) of row (j >> 2) + 8 (g >> 1) in the block's columns 16 (g & 1) .. + 15; it
    // receives column l & 31, contraction rows 8 (l >> 5) .. + 7: the operand layout of v_mfma_f32_32x32x16_bf16
    const int g4 = lane >> 4, j4 = lane & 15;
    const int lrow = (j4 >> 2) + 8 * (g4 >> 1), lcol = 4 * (j4 & 3) + 16 * (g4 & 1);
    const int boff = lrow * TKR_LDB + lcol + 32 * cb;
    // element k1 of the window that starts at input row m sits in image row m + k1 / rs at column k1 % rs (a 4-element piece never
    // straddles a row: rs is a multiple of 4): per lane and block a fixed byte offset
    int aoffb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k1 = lcol + 32 * i + 128 * rh;
        aoffb[i] = (lrow + k1 / rs) * S + (k1 % rs) * 2;
    }


    // one stage: 4 substeps of 16 contraction rows, operands of substep ks + 1 fetched while the MFMAs of ks issue.  NB = the wave's
    // row blocks, a compile-time count per instantiation (a run-time bound would put a branch around every fetch and every MFMA)
    // one stage: 4 substeps of 16 contraction rows, operands of substep ks + 1 fetched while the MFMAs of ks issue.  NB = the wave's
    // row blocks, a compile-time count per instantiation (a run-time bound would put a branch around every fetch and every MFMA).
    // (The bias gradient as one more MFMA against an all-ones block in the upper waves, instead of the loader's column sums, was
    // measured slower: 41.8 vs 38.8 us at 512 utterances -- the launch is bound by its MFMA phases.)
    auto compute_nb = [&](int buf, auto nb_tag) {
        constexpr int NB = decltype(nb_tag)::value;
        const __bf16* Bs = reinterpret_cast<const __bf16*>(smem_tkr + buf * TKR_STAGE_BYTES);
        const char* Im = smem_tkr + buf * TKR_STAGE_BYTES + TKR_B_BYTES;
        bf16x8 a[2][NB], b[2];
        b[0] = tkr_tr8(Bs + boff, TKR_LDB);
#pragma unroll
        for (int i = 0; i < NB; ++i) a[0][i] = tkr_tr8(reinterpret_cast<const __bf16*>(Im + aoffb[i]), S / 2);
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int ks = 0; ks < TKR_ROWS / 16; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (LBX_TKR_ABLATE & 8) {                               // operands of the first substep for all four: the MFMA chain alone
                b[nxt] = b[cur];
#pragma unroll
                for (int i = 0; i < NB; ++i) a[nxt][i] = a[cur][i];
            } else if (ks + 1 < TKR_ROWS / 16) {
                b[nxt] = tkr_tr8(Bs + boff + (ks + 1) * 16 * TKR_LDB, TKR_LDB);
#pragma unroll
                for (int i = 0; i < NB; ++i) a[nxt][i] = tkr_tr8(reinterpret_cast<const __bf16*>(Im + aoffb[i] + (ks + 1) * 16 * S), S / 2);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][i], b[cur], acc[i], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto compute = [&](int buf) {
        if (LBX_TKR_ABLATE & 1) return;
        switch (nb) {                                               // wave-uniform (scalar) dispatch, once per stage
            case 4: compute_nb(buf, std::integral_constant<int, 4>{}); break;
            case 3: compute_nb(buf, std::integral_constant<int, 3>{}); break;
            case 2: compute_nb(buf, std::integral_constant<int, 2>{}); break;
            case 1: compute_nb(buf, std::integral_constant<int, 1>{}); break;
            default: break;
        }
    };
#pragma unroll
    for (int k = 0; k < TKR_PD; ++k) fetch(k, rb_[k], ri_[k]);
    if (nst > 0) stage(0, rb_[0], ri_[0]);
    __syncthreads();
    // stage s: the set of stage s + 1 (loads issued TKR_PD - 1 iterations ago) goes to the other LDS buffer FIRST -- its ds_writes then
    // run beside this stage's MFMAs -- , register set s % TKR_PD (staged in the previous iteration) takes the loads of stage
    // s + TKR_PD; one barrier per stage
#define LBX_TKR_STEP(U)                                                                              \
    {                                                                                                \
        if (s + 1 < nst) stage((s + 1) & 1, rb_[((U) + 1) % TKR_PD], ri_[((U) + 1) % TKR_PD]);     \
        if (!(LBX_TKR_ABLATE & 2)) fetch(s + TKR_PD, rb_[U], ri_[U]);                                \
        compute(s & 1);                                                                              \
        tkr_lds_barrier();                                                                           \
        ++s;                                                                                         \
    }
    for (int s = 0; s < nst;) {
        LBX_TKR_STEP(0)
        if (s < nst) LBX_TKR_STEP(1)
        if (s < nst) LBX_TKR_STEP(2)
        if (s < nst) LBX_TKR_STEP(3)
    }
#undef LBX_TKR_STEP

    // ---- epilogue: the wave's blocks -> P[slice] (rows < K1), then the column sums
    float* Pd = P + (long)slice * K1 * N;
    const int h = lane >> 5, col = n0 + 32 * cb + (lane & 31);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i >= nb || ((LBX_TKR_ABLATE & 4) && acc[i][0] != 12345.f)) continue;
        const int rbase = 128 * rh + 32 * i + 4 * h;
        if (rbase - 4 * h + 32 <= K1) {
            float* p = Pd + (long)rbase * N + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) p[(long)((r & 3) + 8 * (r >> 2)) * N] = acc[i][r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < K1) Pd[(long)row * N + col] = acc[i][r];
            }
        }
    }
    if (Pc) {
        // threads tid = pc + 16 i (i = 0 .. 31) hold partial sums of the same 8 columns: through LDS (free after the loop's last
        // barrier), summed in i order by one thread per column
        float* red = reinterpret_cast<float*>(smem_tkr);           // [32][128]
#pragma unroll
        for (int c = 0; c < 8; ++c) red[(tid >> 4) * TKR_BN + 8 * pc + c] = csum[c];
        __syncthreads();
        if (tid < TKR_BN) {
            float sum = 0.f;
            for (int i = 0; i < 32; ++i) sum += red[i * TKR_BN + tid];
            Pc[(long)slice * N + n0 + tid] = sum;
        }
    }
}

// Shape part of the decision (what the workspace query can see): K1 fits seven blocks, N in whole 128-column tiles, enough rows for
// the slab traffic (tiles_n x slices x K1 x 128 floats) to be small beside dY.  The descriptors are checked at launch.
struct TnKresPlan {
    bool shape_ok, forced;
    int max_slices;                              // slices = min(this, utterances): one round of the chip
};

inline TnKresPlan plan_tn16_kres(long M, int K1, int N) {
    TnKresPlan pl{false, false, 1};
    int mode = -1;                               // LIDBOX_GEMM16_TN_KRES = 0 | 1: never | whenever it can run (tuning / test aid)
    if (const char* e = getenv("LIDBOX_GEMM16_TN_KRES")) mode = atoi(e);
    if (mode == 0 || K1 > TKR_MAX_K1 || K1 % 8 != 0 || N % TKR_BN != 0 || N / TKR_BN > NUM_CU) return pl;
    pl.max_slices = NUM_CU / (N / TKR_BN);
    // measured inside the captured step (profiles/r06_bf16_tn_kres_frame1.txt): configs[4]'s shard (512 utterances, 101 k rows) 1.003 ->
    // 0.988 ms; at 256 utterances (51 k rows) the launch itself is 3 us faster but its 26 MB of slabs (the four-wave tiles: 13 MB) cost
    // the carried reduce more than that (0.628 -> 0.632 ms): from 64 k rows on
    pl.forced = mode == 1;
    pl.shape_ok = pl.forced || (K1 > 128 && M >= 65536);
    return pl;
}

}  // namespace
