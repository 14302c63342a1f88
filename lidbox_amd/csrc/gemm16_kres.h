// gemm16_kres.h -- short-contraction forward of the bf16-storage rows GEMM with BOTH operands resident in LDS (lidbox_gemm_bf16s_nt
// for the x-vector's first frame layer: reference xvector.py:53 `frame1` = Conv1D(512, 5, 1) over 40 mel bins, K = 200, under a
// bfloat16 compute policy), included by gemm_bf16.hip.
//
// Why: frame1's forward is 10.4 GFLOP per 256 utterances behind a 52 MB output -- a store stream, not a matrix problem.  On the
// generic tiles (64 x 128, K = 200 is four steps) every tile re-fetches its 50 KB weight panel and its im2col rows (the same input
// frame five times) through L2 -> LDS, and its epilogue waits behind them: 35 us at 256 utterances against ~12 us for the stores.
// Here a workgroup is persistent and its twelve waves (three per SIMD) are INDEPENDENT of each other after the prologue:
//   * the workgroup's 64-column WEIGHT PANEL ([64][K], 27 KB at K = 200) is loaded once and stays in LDS, rows at a stride of K
//     rounded up to 16 plus 8 elements (an odd number of 16-byte chunks: conflict-free ds_read_b128 operand fetches);
//   * a wave's unit of work is 32 ROWS OF ONE UTTERANCE: their frames are copied into the wave's own LDS image as they lie in memory
//     (one contiguous run of 31 * row_stride + K elements: 2.9 KB) and the implicit rows of the convolution -- row t = the K
//     elements from frame t on -- are read out of that image directly: the MFMA operand of row t, k slice s is the 16 bytes at
//     t * row_stride + 16 s, so the five-fold overlap of the windows costs nothing (row stride 80 B = 5 chunks, odd: conflict-free
//     as well);
//   * the image of the wave's next unit is fetched by LDS-DMA into a second buffer during the MFMAs of the current one and waited for
//     BEFORE the current unit's stores go out (loads and stores retire through one in-order counter: a wait behind the stores would
//     wait for them too); epilogue through the wave's own LDS strip as in pp_store_tile (bias, ReLU, 16-byte fp32 / bf16-shadow stores) with
//     add-only addressing (a unit never leaves its utterance);
//     no barrier after the panel load: the two waves of a SIMD drift apart, and one computes while the other stores.
// Workgroup w: column tile w % tiles_n, unit stream w / tiles_n; a stream's wave v takes units 12 stream + v, + 12 streams, ... with
// unit u = (utterance u / blocks, rows 32 (u % blocks) ..).
#pragma once

#include "gemm16_pp.h"

namespace {

#ifndef LBX_KRES_ABLATE
#define LBX_KRES_ABLATE 0                               // measurement builds only (wrong results): 1 no epilogue, 2 no MFMA loop, 4 no image DMA
#endif
constexpr int KRES_KMAX = 208;                        // contraction, rounded up to 16
constexpr int KRES_BN = 64;
constexpr int KRES_ROWS = 32;                         // rows of a unit
constexpr int KRES_A_BYTES = 3072;                    // one image: 32 rows' frames as in memory; two per wave
constexpr int KRES_B_BYTES = KRES_BN * (KRES_KMAX * 2 + 16);   // 27 648
constexpr int KRES_WAVES = 12;                        // three per SIMD
constexpr int KRES_STRIP_BYTES = 16 * PP_EPI_LDW * 4; // a 16-row strip: half of the wave's 32 x 64 block at a time (4 352)
constexpr int KRES_WAVE_BYTES = 2 * KRES_A_BYTES + KRES_STRIP_BYTES;             // 10 496
constexpr int KRES_LDS_BYTES = KRES_B_BYTES + KRES_WAVES * KRES_WAVE_BYTES;      // 153 600

// what the host checks before choosing this kernel
inline bool kres_applies(const lidbox_rows_t& A, int K, int N) {
    if (!(K % 8 == 0 && K >= 16 && K <= KRES_KMAX && A.batch >= 1 && A.rows_per_batch >= 1)) return false;
    if (A.row_stride < 8 || A.row_stride % 8 != 0 || (A.batch > 1 && A.batch_stride % 8 != 0)) return false;
    const int kp = (K + 15) / 16 * 16;
    // the last row's operand reads end inside the image
    return (long)(KRES_ROWS - 1) * A.row_stride * 2 + kp * 2 <= KRES_A_BYTES;
}

template <int NKS>                                    // k slices of 16 (compile-time: the loop is unrolled and its operand fetches run ahead); 0: K at run time
__global__ __launch_bounds__(64 * KRES_WAVES, 3) void gemm16s_rows_kres_kernel(RowsH A, RowsH Bw, RowsOutD Cd, unsigned short* __restrict__ C16,
                                                                    int K, int N, int epi, const float* __restrict__ aux, int tiles_n,
                                                                    int nstreams, int blocks, long nunits) {
    constexpr int NJ = 2;                                     // a wave's 32 x 64 block
    extern __shared__ __attribute__((aligned(16))) char smem16k[];
    char* const Bl = smem16k;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem16k);
    char* const Al = smem16k + KRES_B_BYTES + wv * KRES_WAVE_BYTES;
    const unsigned al0 = lds0 + (unsigned)(KRES_B_BYTES + wv * KRES_WAVE_BYTES);
    float* const strip = reinterpret_cast<float*>(Al + 2 * KRES_A_BYTES);
    const int ct = blockIdx.x % tiles_n, stream = blockIdx.x / tiles_n;
    const int n0 = ct * KRES_BN;
    const int nks = NKS ? NKS : (K + 15) / 16, kp = nks * 16;
    const bool half_tail = (K & 8) != 0;                      // the last k slice's upper half lies past K
    const int bstride = kp * 2 + 16;                          // bytes per weight row in LDS
    const int bchunks = bstride / 16;
    const int rs2 = (int)A.rs * 2;                            // bytes per input frame step
    const int utt_bytes = ((A.rpb - 1) * (int)A.rs + K) * 2;  // an utterance's frames, contiguous (a multiple of 16)

    // ---- once: zero the wave's images (bytes a copy does not reach are read by rows that are never stored, and stay finite), load the panel
    for (int i = lane; i < 2 * KRES_A_BYTES / 16; i += 64) reinterpret_cast<u32x4_t*>(Al)[i] = u32x4_t{0u, 0u, 0u, 0u};
    for (int p = tid; p < KRES_BN * bchunks; p += 64 * KRES_WAVES) {
        const int row = p / bchunks, c = p - row * bchunks;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (c * 8 < K && n0 + row < N) v = *reinterpret_cast<const u32x4_t*>(Bw.base + (long)(n0 + row) * Bw.rs + c * 8);
        reinterpret_cast<u32x4_t*>(Bl)[p] = v;
    }
    __syncthreads();

    // rows 32 blk .. of utterance b: the frames they read, 1 KB per DMA instruction; past the utterance's last frame: zeros
    const float* const abase = sk_uniform(reinterpret_cast<const float*>(A.base));
    auto issue_image = [&](long u, int buf) {
        const int b = (int)(u / blocks), blk = (int)(u - (long)b * blocks);
        const int first = KRES_ROWS * blk * rs2;
        const int avail = utt_bytes - first;                                       // > 0
        const int want = (KRES_ROWS - 1) * rs2 + K * 2;
        const char* src = reinterpret_cast<const char*>(abase) + (long)b * A.bs * 2 + first;
        for (int p = 0; p * 1024 < want; ++p) {
            const int off = p * 1024 + lane * 16;
            sk_dma_f(off < avail ? reinterpret_cast<const float*>(src + off) : g_sk_zero, al0 + (unsigned)(buf * KRES_A_BYTES + p * 1024));
        }
    };

    // operand addresses: A row (32 bi + lane % 32) of the image, B column (32 bj + lane % 32) of the panel, k slice s:
    // 16 bytes at + 32 s + 16 (lane / 32)
    const unsigned aoff = (unsigned)((lane & 31) * rs2 + 16 * (lane >> 5));
    const unsigned boff = (unsigned)((lane & 31) * bstride + 16 * (lane >> 5));
    const long ustep = (long)KRES_WAVES * nstreams;

    // ---- epilogue addressing: a unit never leaves its utterance, so row offsets are adds (pp_store_tile divides per chunk).  The strip
    // is written in the accumulator layout (lane = column) and read back row-wise: lane -> row (lane >> 3) + 8 c, columns 8 (lane & 7) ..+7
    const int h = lane >> 5, l = lane & 31;
    const int erow = lane >> 3, ecol = (lane & 7) * 8;
    const bool has_bias = epi == LIDBOX_EPI_BIAS || epi == LIDBOX_EPI_BIAS_RELU;
    const bool do_relu = epi == LIDBOX_EPI_BIAS_RELU || epi == LIDBOX_EPI_RELU;
    float bias[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bias[j] = (has_bias && n0 + j * 32 + l < N) ? aux[n0 + j * 32 + l] : 0.f;
    const bool col_ok = n0 + ecol < N;                         // N is a multiple of 8 (host): the whole chunk is inside
    const bool sh16 = (Cd.rs & 7) == 0 && (Cd.batch == 1 || (Cd.bs & 7) == 0) && ((((uintptr_t)C16) & 15) == 0);   // 16-byte shadow stores
    const long crs8 = 8 * Cd.rs;

    auto read_ops = [&](const char* ai, const char* bi_, int s, bf16x8& a, bf16x8 (&bb)[NJ]) {
        a = *reinterpret_cast<const bf16x8*>(ai + s * 32);
        if (half_tail && s == nks - 1 && lane >= 32) {         // k >= K: the next frame's values, not zeros -- drop them
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = (__bf16)0.f;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) bb[j] = *reinterpret_cast<const bf16x8*>(bi_ + (j * 32) * bstride + s * 32);
    };

    long u = (long)KRES_WAVES * stream + wv;
    if (u < nunits) issue_image(u, 0);
    sk_wait_vm<0>();
    int buf = 0;
    for (; u < nunits; u += ustep, buf ^= 1) {
        const int b = (int)(u / blocks), blk = (int)(u - (long)b * blocks);
        if (!(LBX_KRES_ABLATE & 4) && u + ustep < nunits) issue_image(u + ustep, buf ^ 1);   // the other buffer's last reader was the previous unit's loop
        f32x16 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const char* ai = Al + buf * KRES_A_BYTES + aoff;
        const char* bi_ = Bl + boff;
        __builtin_amdgcn_s_setprio(1);
        if (NKS) {
            // unrolled: the compiler moves the operand fetches of later slices ahead of the MFMAs of earlier ones
            bf16x8 a[NKS ? NKS : 1], bq[NKS ? NKS : 1][NJ];
#pragma unroll
            for (int s = 0; s < NKS; ++s) read_ops(ai, bi_, s, a[s], bq[s]);
#pragma unroll
            for (int s = 0; s < ((LBX_KRES_ABLATE & 2) ? 1 : NKS); ++s)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], bq[s][j], acc[j], 0, 0, 0);
        } else {
            // operands of slice s + 1 are fetched before the MFMAs of slice s are issued
            bf16x8 a0, a1, b0[NJ], b1[NJ];
            const int ns = (LBX_KRES_ABLATE & 2) ? 1 : nks;
            read_ops(ai, bi_, 0, a0, b0);
            for (int s = 0; s < ns; s += 2) {
                if (s + 1 < ns) read_ops(ai, bi_, s + 1, a1, b1);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0[j], acc[j], 0, 0, 0);
                if (s + 1 < ns) {
                    if (s + 2 < ns) read_ops(ai, bi_, s + 2, a0, b0);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1[j], acc[j], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        // the next image (it had the whole loop to land) and the previous unit's stores -- not this unit's, which go out below
        sk_wait_vm<0>();
        if (LBX_KRES_ABLATE & 1) {
            float sacc = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc += acc[0][r] + acc[1][r];
            if (sacc == 12345.f) C16[tid] = 1;
            continue;
        }
        const int t0 = KRES_ROWS * blk + erow;                 // this lane's first row inside the utterance
        const long rbase = (Cd.batch == 1 ? ((long)b * A.rpb + t0) * Cd.rs : (long)b * Cd.bs + (long)t0 * Cd.rs) + n0 + ecol;
#pragma unroll
        for (int half = 0; half < 2; ++half) {                 // accumulator registers 8 half .. +7 are the block's rows 16 half .. +15
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    strip[((r & 3) + 8 * (r >> 2) + 4 * h) * PP_EPI_LDW + j * 32 + l] = acc[j][8 * half + r] + bias[j];
            wave_lds_sync();
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float* src = strip + (erow + 8 * c) * PP_EPI_LDW + ecol;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
                float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (do_relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.f);
                }
                const int dr = 16 * half + 8 * c;              // row of the block, minus erow
                if (t0 + dr < A.rpb && col_ok) {
                    const long off = rbase + (2 * half + c) * crs8;
                    if (Cd.base) {
                        *reinterpret_cast<f32x4*>(Cd.base + off) = f32x4{x[0], x[1], x[2], x[3]};
                        *reinterpret_cast<f32x4*>(Cd.base + off + 4) = f32x4{x[4], x[5], x[6], x[7]};
                    }
                    if (C16) {
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (__bf16)x[e];
                        if (sh16) {
                            *reinterpret_cast<bf16x8*>(C16 + off) = o;
                        } else {
                            const u32x4_t w = __builtin_bit_cast(u32x4_t, o);
                            *reinterpret_cast<uint2*>(C16 + off) = make_uint2(w[0], w[1]);
                            *reinterpret_cast<uint2*>(C16 + off + 4) = make_uint2(w[2], w[3]);
                        }
                    }
                }
            }
            wave_lds_sync();
        }
    }
}

}  // namespace
