/*
 * lidbox_hip.h -- C ABI of liblidbox_hip.so, the MI355X (gfx950) implementation of the
 * lidbox feature + x-vector hot path.
 *
 * The reference (py-lidbox/lidbox) has no FFI: its operator surface is a set of Python
 * callables whose arithmetic is delegated to TensorFlow ops.  Each entry point below names
 * the reference callable (file:line relative to the reference checkout) whose arithmetic it
 * replaces; lidbox_amd/ (Python) mirrors the reference's names on top of these symbols and
 * INTEGRATION.md shows the ctypes binding a lidbox maintainer would add.
 *
 * Conventions
 *   - plain C types only: raw DEVICE pointers (unless a parameter says "host"), sizes, scalars,
 *     and a hipStream_t passed as void*;
 *   - the caller allocates every output and workspace; nothing here allocates per call
 *     (plan creation allocates small immutable device tables once);
 *   - every launch is stream-ordered and re-entrant: concurrent callers on different streams
 *     are safe, plans are immutable after creation;
 *   - return value: 0 = ok, negative = error (LIDBOX_E_*); message via lidbox_hip_last_error()
 *     (thread-local).  Nothing throws across the boundary;
 *   - all tensors are dense row-major float32 unless stated otherwise.
 */
#ifndef LIDBOX_HIP_H
#define LIDBOX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIDBOX_HIP_ABI_VERSION 1

#define LIDBOX_OK            0
#define LIDBOX_E_INVALID    -1   /* bad argument (shape, size, null pointer, unsupported value) */
#define LIDBOX_E_LAUNCH     -2   /* HIP runtime / launch failure */
#define LIDBOX_E_ALLOC      -3   /* device allocation failed (plan creation only) */

typedef void* lidbox_stream_t;                 /* hipStream_t */
typedef struct lidbox_feat_plan lidbox_feat_plan;

int         lidbox_hip_abi_version(void);
const char* lidbox_hip_last_error(void);

/* ------------------------------------------------------------------ host-side scalars/constants */

/* lidbox/features/audio.py:185-189  ms_to_frames: int32(float32(sr) * 1e-3f * float32(ms)) */
int lidbox_ms_to_frames(int sample_rate, int ms);

/* tf.signal.frame(pad_end=False) frame count used by tf.signal.stft (audio.py:229) */
int lidbox_num_frames(int num_samples, int frame_length, int frame_step);

/* lidbox/features/mel_ops.py:28-75  linear_to_mel_weight_matrix (non-endpoint _linspace :11-16),
 * float32 op order.  out_host: [num_spectrogram_bins * num_mel_bins] floats on the HOST. */
int lidbox_mel_weight_matrix(int num_mel_bins, int num_spectrogram_bins, int sample_rate,
                             float lower_edge_hertz, float upper_edge_hertz, float* out_host);

/* tf.signal.hann_window(L, periodic=True) as tf.signal.stft applies it; out_host: [L] HOST floats */
int lidbox_hann_window(int window_length, float* out_host);

/* ------------------------------------------------------------------ feature extraction (a2-a6, a11) */

enum {
    LIDBOX_FEAT_SPECTROGRAM = 0,   /* audio.py:219-230   |STFT|^power          -> [B,T,F]      */
    LIDBOX_FEAT_MEL         = 1,   /* audio.py:247-261   spectrogram . W_mel   -> [B,T,M]      */
    LIDBOX_FEAT_LOGMEL      = 2,   /* tf_utils.py:177-179 ln(mel + 1e-6)       -> [B,T,M]      */
    LIDBOX_FEAT_MFCC        = 3    /* tf_utils.py:180-185 DCT-II/sqrt(2M), [coef_begin:coef_end) -> [B,T,C] */
};

/* Immutable per-configuration tables (window, FFT twiddles, banded mel weights, DCT rows) on the
 * CURRENT device.  Mirrors the constants lidbox rebuilds inside every traced call
 * (audio.py:226-229, mel_ops.py:28-75, tf_utils.py:181-184). */
int  lidbox_feat_plan_create(int sample_rate, int frame_length, int frame_step, int fft_length,
                             float power, int num_mel_bins, float fmin, float fmax,
                             int coef_begin, int coef_end, lidbox_feat_plan** out_plan);
void lidbox_feat_plan_destroy(lidbox_feat_plan* plan);

/* number of output channels for `kind` (F, M, M, coef_end-coef_begin) */
int  lidbox_feat_plan_channels(const lidbox_feat_plan* plan, int kind);
/* 1 if (kind, N, sig_stride, pointer alignment) takes the fused single-kernel path */
int  lidbox_feat_plan_is_fused(const lidbox_feat_plan* plan, int kind, const float* signals,
                               long sig_stride);
/* bytes of caller-provided workspace the NON-fused path needs (0 when fused) */
size_t lidbox_extract_features_workspace(const lidbox_feat_plan* plan, int kind, int B, int N,
                                         const float* signals, long sig_stride);

/* lidbox/data/tf_utils.py:166-185 (spectrogram -> mel -> log -> MFCC stages of extract_features)
 * signals: [B, N] with row stride sig_stride (floats); out: [B, T, channels(kind)] with
 * out_batch_stride floats between utterances (0 = dense T*channels; a larger stride lets the
 * features land directly behind the causal zero rows of the first Conv1D's input buffer);
 * T = lidbox_num_frames(N, frame_length, frame_step). */
int lidbox_extract_features_fwd(const lidbox_feat_plan* plan, int kind, const float* signals,
                                int B, int N, long sig_stride, float* out, long out_batch_stride,
                                void* workspace, size_t workspace_bytes, lidbox_stream_t stream);
/* The same, and additionally out16 (may be NULL): a bfloat16 copy of the features (round-to-nearest-even) at the same element
 * offsets as out (so the same batch stride, counted in elements) -- the shadow the bf16-storage Conv1D path reads
 * (lidbox_gemm_bf16s_nt).  The fused log-mel kernel (the train step's) writes it from its own store stage when
 * out_batch_stride is a multiple of 4; other kinds and shapes get one conversion pass after their kernel.  out16 8-byte aligned. */
int lidbox_extract_features_fwd_shadow(const lidbox_feat_plan* plan, int kind, const float* signals,
                                       int B, int N, long sig_stride, float* out, long out_batch_stride, void* out16,
                                       void* workspace, size_t workspace_bytes, lidbox_stream_t stream);

/* The general form (round 6).  src_format: LIDBOX_SRC_F32 (signals = float [B, N]) or LIDBOX_SRC_PCM16 (signals = int16_t [B, N], mono,
 * read in place: value / 32768 as tf.audio.decode_wav gives it, lidbox/features/audio.py:17-23 -- bit-identical to lidbox_pcm16_to_f32
 * followed by the float call, at half the bytes read; needs the fused kernel, 8-byte aligned signals and sig_stride / frame_length /
 * frame_step multiples of 4, LIDBOX_E_INVALID otherwise).  sig_stride counts samples.  nonfinite (may be NULL): an int the kernels can
 * write -- device memory or pinned host memory -- set to 1 when any value the call wrote to out is NaN or +-Inf:
 * tf.debugging.assert_all_finite of tf_utils.py:168-194 without a pass over the output; the caller zeroes it and reads it (after
 * synchronising the stream) when it wants the answer. */
enum { LIDBOX_SRC_F32 = 0, LIDBOX_SRC_PCM16 = 1 };
int lidbox_extract_features_fwd_ex(const lidbox_feat_plan* plan, int kind, const void* signals, int src_format,
                                   int B, int N, long sig_stride, float* out, long out_batch_stride, void* out16,
                                   int* nonfinite, void* workspace, size_t workspace_bytes, lidbox_stream_t stream);
/* lidbox/features/audio.py:17-23 (read_wav -> decode_wav) + tf_utils.py:166-185 in one pass over 16-bit mono PCM */
int lidbox_extract_features_fwd_pcm16(const lidbox_feat_plan* plan, int kind, const int16_t* pcm, int B, int N, long sig_stride,
                                      float* out, long out_batch_stride, int* nonfinite, lidbox_stream_t stream);

/* ------------------------------------------------------------------ normalisation (a7-a10) */

/* lidbox/features/__init__.py:12-32  cmn / cmvn over the middle axis of x viewed as
 * [outer, R, inner] (axis=1 of [B,T,C]: outer=B, R=T, inner=C).  Population std of x,
 * divide_no_nan.  normalize_variance=0 -> cmn. */
int lidbox_cmvn_fwd(const float* x, long outer, long R, long inner, int normalize_variance,
                    float* out, lidbox_stream_t stream);
/* the same with `outer` slices x_outer_stride / out_outer_stride floats apart (>= R * inner); x == out (in place) is
 * allowed: the MFCC + CMVN front-end of cnn.py normalises the features where the first Conv1D reads them. */
int lidbox_cmvn_strided_fwd(const float* x, long outer, long R, long inner, long x_outer_stride,
                            int normalize_variance, float* out, long out_outer_stride, lidbox_stream_t stream);

/* lidbox/features/__init__.py:35-67  sliding branch of window_normalization on x [B,T,C]
 * (axis=1, T > window_len >= 2): REFLECT pad [w/2, w/2-1+(w&1)], per-window mean/std. */
int lidbox_window_norm_fwd(const float* x, int B, int T, int C, int window_len,
                           int normalize_variance, float* out, lidbox_stream_t stream);

/* min and max of n floats -> out2[0]=min, out2[1]=max (device).  First half of
 * features.feature_scaling (features/__init__.py:7-8, axis=None) and of audio.power_to_db's
 * batch-global max (audio.py:173).  scratch: >= 2*1024 floats (device). */
int lidbox_minmax(const float* x, long n, float* out2, float* scratch, lidbox_stream_t stream);

/* lidbox/features/__init__.py:5-9, axis=None: lo + (hi-lo)*divide_no_nan(x-min, max-min) */
int lidbox_feature_scaling_fwd(const float* x, long n, const float* minmax2, float lo, float hi,
                               float* out, lidbox_stream_t stream);
/* feature_scaling over ONE axis (reference lidbox/features/__init__.py:5-9 with axis = k): x viewed as [outer][R][inner],
 * out = lo + (hi - lo) * divide_no_nan(x - min_R, max_R - min_R) with the min / max over R per (outer, inner).  outer <= 65535. */
int lidbox_feature_scaling_axis_fwd(const float* x, long outer, long R, long inner, float lo, float hi, float* out,
                                    lidbox_stream_t stream);
/* audio.log10 (reference lidbox/features/audio.py:162-164): out = ln(x) / ln(10), elementwise, float32 op order. */
int lidbox_log10_fwd(const float* x, long n, float* out, lidbox_stream_t stream);

/* lidbox/features/audio.py:167-174 power_to_db: 20*(log10(max(amin,S)) - log10(max(amin,Smax))),
 * floored at (its own max) - top_db.  minmax2 = lidbox_minmax(S).  scratch as lidbox_minmax. */
int lidbox_power_to_db_fwd(const float* S, long n, const float* minmax2, float amin, float top_db,
                           float* out, lidbox_stream_t stream);

/* ------------------------------------------------------------------ GEMM family (a4, a12, a14, a16) */

/* Implicit-GEMM view of a Keras Conv1D(padding="causal") input (xvector.py:38-39, cnn.py:32-35)
 * and of Dense inputs.  A logical row m = (b, t), 0 <= b < batch, 0 <= t < rows_per_batch,
 * starts at  base + b*batch_stride + t*row_stride  (floats) and is `K` floats long.  Activations
 * are stored with (k-1) leading zero rows per utterance, so a causal window is a contiguous
 * K = k*C_in run and no im2col copy exists.  A plain [M,K] matrix is batch=1, rows_per_batch=M,
 * row_stride=ld. */
typedef struct {
    const float* base;
    long  batch_stride;
    long  row_stride;
    int   batch;
    int   rows_per_batch;
} lidbox_rows_t;

typedef struct {
    float* base;
    long  batch_stride;
    long  row_stride;
    int   batch;
    int   rows_per_batch;
} lidbox_rows_out_t;

enum {
    LIDBOX_EPI_NONE      = 0,
    LIDBOX_EPI_BIAS      = 1,   /* + bias[n]                                  (Dense, activation=None)  */
    LIDBOX_EPI_BIAS_RELU = 2,   /* relu(. + bias[n])                          (frame_layer/segment_layer) */
    LIDBOX_EPI_RELU_MASK = 3,   /* . * (mask[m,n] > 0)   backward through ReLU; mask has C's layout */
    LIDBOX_EPI_ACCUM     = 4,   /* C += .                                                          */
    LIDBOX_EPI_ACCUM_RELU_MASK = 5, /* C += . * (mask > 0)                                         */
    LIDBOX_EPI_ACCUM_RELU = 6,  /* C = relu(C + .)       last tap of a dilated Conv1D (bias added by the first tap) */
    LIDBOX_EPI_RELU      = 7,   /* relu(.)               Dense(use_bias=False, activation="relu"), clstm.py:35 */
    /* flag, lidbox_gemm_bf16s_nt only, OR-ed onto a *_RELU_MASK epilogue: the mask source `aux` is bfloat16 data at C's
     * element offsets (the bf16 shadow of the activation; only sign / zero-ness is looked at) */
    LIDBOX_EPI_MASK_BF16 = 0x100
};

/* Which decomposition the cost model picks (for profiling tools: it names the kernel instantiation a
 * launch will use).  kind 0 = nn, 1 = nt, 2 = tn (then K = K1).  out4 = {BM, BN, splits, k or rows per split}. */
int lidbox_gemm_plan_query(int kind, long M, int N, int K, size_t workspace_bytes, int* out4);

/* Waves per workgroup of the kernel lidbox_gemm_plan_query names: 4 = gemm_rows_kernel / gemm_tn_kernel, 8 =
 * gemm_rows8_kernel (the same tile worked by twice the waves; nn / nt only). */
int lidbox_gemm_plan_waves(int kind, long M, int N, int K, size_t workspace_bytes);

/* 1 when lidbox_gemm_nn / _nt (kind 0 / 1) or lidbox_gemm_tn (kind 2; K = K1) would run this shape on the persistent
 * stream-K kernels (csrc/gemm_sk.h: 128 x 128 tiles, LDS-DMA ring, in-launch fixed-order reduce) given 16-byte aligned
 * operands and a workspace of workspace_bytes.  The workspace needs no initialisation (the arrival counters kept in its
 * first 16 KiB carry a launch epoch); as before, two calls that may run concurrently must not share a workspace. */
int lidbox_gemm_plan_is_stream_k(int kind, long M, int N, int K, size_t workspace_bytes);

/* What the calling thread's most recent lidbox_gemm_nn / _nt / _tn call put on the stream (for profiling tools: it lets a
 * HIP-event bracket around the call be compared with rocprofv3's per-kernel averages).  out3 = {kernels of the
 * instantiation lidbox_gemm_plan_query names, GEMM kernels of another instantiation (the remainder of a tail split is
 * planned on its own), reduce kernels}. */
int lidbox_gemm_last_launches(int* out3);
/* Kernel family of that same call: 0 = register-staged kernels (gemm_rows_kernel / gemm_tn_kernel), 1 = the same
 * decomposition on the LDS-DMA operand path (gemm_rows_dma_kernel: 16-byte aligned operands), 2 = stream-K, 3 = the LDS-DMA
 * tile worked by eight waves (gemm_rows_dma8_kernel: nt launches planned as 128-row tiles with 8 waves). */
int lidbox_gemm_last_family(void);
/* Pieces per tile (2 .. 8) when lidbox_gemm_nn / _nt (kind 0 / 1) would stream the last partial round of tiles of this
 * shape along K inside the one launch (LDS-DMA family, 16-byte aligned operands, a workspace of workspace_bytes: 16 KiB of
 * arrival counters + one tile-sized slab per piece, no initialisation needed); 0 when every tile runs whole. */
int lidbox_gemm_plan_stream_tail(int kind, long M, int N, int K, size_t workspace_bytes);

/* Split-K workspace (bytes) that lets lidbox_gemm_nn / _nt fill the chip when M*N is small
 * (Dense layers at M = batch): partial sums are reduced in a fixed order with the epilogue
 * fused into the reduce.  0 = not needed.  Passing NULL/0 is always legal (no split). */
size_t lidbox_gemm_rows_workspace(long M, int N, int K);

/* C[M,N] = epi( A[M,K] . B )  -- forward of Conv1D/Dense (a12/a14), linear_to_mel (a4).
 * A: implicit rows (K contiguous).  B: [K,N] row-major, ldb (Keras kernel layout [k*C_in, C_out]).
 * C: implicit rows (N contiguous).  aux: bias[N] or mask (layout of C) or NULL. */
int lidbox_gemm_nn(lidbox_rows_t A, const float* Bm, long ldb, lidbox_rows_out_t C,
                   int K, int N, int epilogue, const float* aux,
                   void* workspace, size_t workspace_bytes, lidbox_stream_t stream);

/* C[M,N] = epi( A[M,K] . B^T ), B is [N,K] row-major (ldb): C[m,n] = sum_k A[m,k]*B[n,k].
 * dgrad of Conv1D/Dense: A = dY rows, B = W rows (Keras kernel [k*C_in, C_out] is exactly
 * [N = k*C_in][K = C_out]), C = dX rows. */
int lidbox_gemm_nt(lidbox_rows_t A, const float* Bm, long ldb, lidbox_rows_out_t C,
                   int K, int N, int epilogue, const float* aux,
                   void* workspace, size_t workspace_bytes, lidbox_stream_t stream);

/* C[K1,N] (ldc) = A[M,K1]^T . B[M,N]   -- wgrad: dW = col^T . dY, contraction over the M rows.
 * Split over M into partial sums reduced deterministically through `workspace`
 * (>= lidbox_gemm_tn_workspace() bytes).  accumulate != 0: C += result.
 * bias_grad (may be NULL): [N], receives the column sums of B (db = sum_m dY[m,:]) computed
 * from the B tiles the kernel already holds in LDS (same accumulate flag). */
size_t lidbox_gemm_tn_workspace(int M, int K1, int N);
int lidbox_gemm_tn(lidbox_rows_t A, lidbox_rows_t Bm, float* C, long ldc, int K1, int N,
                   int accumulate, float* bias_grad, void* workspace, size_t workspace_bytes,
                   lidbox_stream_t stream);

/* ---- carried reduce: a wgrad's fixed-order slice sum inside a LATER GEMM launch -----------------------------------------
 * lidbox_gemm_tn = a GEMM over M slices + a bandwidth-bound reduce launch that sums the slices in order.  The two halves
 * are also callable on their own, so that the reduce can run in the LEADING WORKGROUPS of the next MFMA-bound launch on the
 * same stream (the layer's dgrad) instead of between two launches; every element still adds its slices in the order
 * 0 .. splits-1, so the results are bit-identical to lidbox_gemm_tn whoever runs the job.
 *   lidbox_gemm_tn_partial  the GEMM only; *job describes the pending reduce (the workspace holds the slices and must stay
 *                           untouched until the job has run; job->nblocks == 0: nothing pending -- unaligned operands had
 *                           their scalar reduce launched by this call)
 *   lidbox_gemm_nt_carry    lidbox_gemm_nt + up to two pending `jobs` (NULL / empty ones: plain lidbox_gemm_nt): in the same
 *                           launch when that is an LDS-DMA tile launch (16-byte aligned operands; one leading workgroup per
 *                           CU, shared in proportion to the jobs' bytes), else as one launch of their own behind it;
 *                           LIDBOX_GEMM_NO_CARRY=1 forces the latter (A/B aid); `workspace` must not hold a job's slices
 *   lidbox_gemm_nt_tn_carry lidbox_gemm_nt_tn that carries one pending job of an EARLIER layer in its launch and, where
 *                           its own wgrad reduce cannot run inside its launches (the one-launch pair), hands that out as
 *                           *job_out for a later carry call (job_out == NULL: launched here)
 *   lidbox_reduce_jobs_run  up to two jobs as one launch of their own
 *   lidbox_gemm_last_carried  number of jobs the calling thread's most recent *_carry call ran inside its GEMM launch
 * Reference: the wgrad / dgrad of Conv1D and Dense, lidbox/models/xvector.py:38-43,53-64 under Keras' fit (keras_utils.py:198-203). */
typedef struct lidbox_reduce_job {
    const float* partials;        /* [splits][n] */
    const float* bias_partials;   /* [splits][N] or NULL */
    float* C;
    float* bias_grad;             /* or NULL */
    long n, ldc;                  /* n = K1 * N elements of C, rows of N at stride ldc */
    int splits, N, accumulate;
    unsigned nblocks;             /* workgroups of a stand-alone launch; 0 = nothing to do */
} lidbox_reduce_job_t;
int lidbox_gemm_tn_partial(lidbox_rows_t A, lidbox_rows_t Bm, float* C, long ldc, int K1, int N, int accumulate,
                           float* bias_grad, void* workspace, size_t workspace_bytes, lidbox_reduce_job_t* job,
                           lidbox_stream_t stream);
int lidbox_gemm_nt_carry(lidbox_rows_t A, const float* Bm, long ldb, lidbox_rows_out_t C, int K, int N, int epilogue,
                         const float* aux, void* workspace, size_t workspace_bytes, const lidbox_reduce_job_t* jobs, int njobs,
                         lidbox_stream_t stream);
int lidbox_gemm_nt_tn_carry(lidbox_rows_t dY, const float* W, long ldb, lidbox_rows_out_t dX, int Co, int N, int epilogue,
                            const float* aux, void* ws_nt, size_t ws_nt_bytes, lidbox_rows_t X, float* dW, long ldc, int K1,
                            int accumulate, float* bias_grad, void* ws_tn, size_t ws_tn_bytes,
                            const lidbox_reduce_job_t* jobs, int njobs, lidbox_reduce_job_t* job_out, lidbox_stream_t stream);
int lidbox_reduce_jobs_run(const lidbox_reduce_job_t* jobs, int njobs, lidbox_stream_t stream);
int lidbox_gemm_last_carried(void);
/* A zero fill as a job (splits = 0): `batch` runs of row_floats zeros at base, batch_stride floats apart (16-byte aligned,
 * multiples of 4).  Carried like a reduce -- the pad rows a strided Conv1D's accumulating dgrad groups do not all cover
 * (Keras Conv1D(strides), xvector.py:53-57 backward) are cleared inside the group-0 launch instead of by a fill launch. */
int lidbox_zero_job(float* base, long batch_stride, long row_floats, int batch, lidbox_reduce_job_t* job);

/* A layer's dgrad and wgrad in one call -- both read the output gradient dY [M, Co]:
 *     dX rows = epilogue(dY . W^T)        exactly lidbox_gemm_nt(dY, W, ldb, dX, Co, N, epilogue, aux, ws_nt, ...)
 *     dW [K1, Co] (ldc) (+)= X^T . dY     exactly lidbox_gemm_tn(X, dY, dW, ldc, K1, Co, accumulate, bias_grad, ws_tn, ...)
 * Small problems (the dense head: M = batch rows, both planned as 64 x 64 tiles that fit the chip at once) go out as ONE
 * kernel launch followed by their reduces; everything else is lidbox_gemm_tn_partial + lidbox_gemm_nt_carry: the wgrad GEMM,
 * then the dgrad launch with the wgrad's reduce in its leading workgroups.  Bit-identical to the two plain calls either way.
 * The two workspaces should not overlap (one launch: the two GEMMs run concurrently; two launches: the slices wait in
 * ws_tn while the dgrad uses ws_nt -- a shared workspace falls back to the plain sequence). */
/* 1 when lidbox_gemm_nt_tn would launch this pair (16-byte aligned operands) as one kernel (profiling tools). */
int lidbox_gemm_plan_is_pair(long M, int Co, int N, int K1, size_t ws_nt_bytes, size_t ws_tn_bytes);
int lidbox_gemm_nt_tn(lidbox_rows_t dY, const float* W, long ldb, lidbox_rows_out_t dX, int Co, int N, int epilogue,
                      const float* aux, void* ws_nt, size_t ws_nt_bytes, lidbox_rows_t X, float* dW, long ldc, int K1,
                      int accumulate, float* bias_grad, void* ws_tn, size_t ws_tn_bytes, lidbox_stream_t stream);

/* ---- bf16-compute variants (BASELINE config 5: "bf16 compute / fp32 master") -------------------
 * Same arguments, layouts, epilogues and determinism as lidbox_gemm_nn / _nt / _tn.  Every buffer
 * stays fp32 in HBM; both operands are rounded to bfloat16 (round-to-nearest-even) while they are
 * staged on chip, products accumulate in fp32 (v_mfma_f32_32x32x16_bf16), epilogues and the fused
 * bias gradient are fp32.  Result == fp32 GEMM of the bf16-rounded operands up to summation order.
 * Requires 16-byte aligned bases/workspace and K (K1), N, ldb, row and batch strides that are
 * multiples of 4 (LIDBOX_E_INVALID otherwise) -- true of every x-vector / CNN layer.  Workspaces
 * are sized by their own functions (the decompositions differ from the fp32 family's). */
size_t lidbox_gemm_bf16_rows_workspace(long M, int N, int K);
int lidbox_gemm_bf16_nn(lidbox_rows_t A, const float* Bm, long ldb, lidbox_rows_out_t C,
                        int K, int N, int epilogue, const float* aux,
                        void* workspace, size_t workspace_bytes, lidbox_stream_t stream);
int lidbox_gemm_bf16_nt(lidbox_rows_t A, const float* Bm, long ldb, lidbox_rows_out_t C,
                        int K, int N, int epilogue, const float* aux,
                        void* workspace, size_t workspace_bytes, lidbox_stream_t stream);
size_t lidbox_gemm_bf16_tn_workspace(int M, int K1, int N);
int lidbox_gemm_bf16_tn(lidbox_rows_t A, lidbox_rows_t Bm, float* C, long ldc, int K1, int N,
                        int accumulate, float* bias_grad, void* workspace, size_t workspace_bytes,
                        lidbox_stream_t stream);

/* ---- bf16-STORAGE variant (operands already bfloat16 in HBM) ------------------------------------
 * C[M,N] = epi( A16[M,K] . B16^T ): A16 = implicit rows of a bf16 buffer (A16.base points at bfloat16 data, strides in
 * bf16 ELEMENTS -- the bf16 "shadow" of an activation / gradient buffer, same element layout as the fp32 one), B16 =
 * [N,K] bf16 row-major (ldb in elements): a weight shadow (dgrad reads the Keras kernel [k*C_in, C_out] as is, forward
 * reads its transpose [C_out, k*C_in]).  C (fp32), epilogues, split decompositions and determinism as
 * lidbox_gemm_bf16_nt; C16 (may be NULL) receives the bf16 shadow of every finished C value at C's element offsets.
 * Numerically identical to lidbox_gemm_bf16_nn/_nt on the fp32 originals (rounding happens where the shadow is
 * written instead of where it is read).  Needs 16-byte aligned bases and K, ldb, row / batch strides % 8 == 0.
 * C.base may be NULL (with C16 given and a non-accumulating epilogue): the result then exists only as the shadow -- the
 * strides of C still describe the layout -- which halves to thirds the bytes an intermediate activation costs. */
int lidbox_gemm_bf16s_nt(lidbox_rows_t A16, const void* B16, long ldb, lidbox_rows_out_t C, void* C16,
                         int K, int N, int epilogue, const float* aux,
                         void* workspace, size_t workspace_bytes, lidbox_stream_t stream);
/* wgrad on bf16 storage: C[K1,N] (+)= A16^T . B16 with both operands implicit rows of bf16 buffers (bases point at
 * bfloat16 data, strides in bf16 elements): A16 = the shadow of the layer input (K1 = k*C_in contiguous elements per
 * row), B16 = the shadow of the output gradient.  C, accumulate, bias_grad, workspace and determinism as
 * lidbox_gemm_bf16_tn (xvector.py:38-43 under a bfloat16 policy); the bias gradient is the fp32 sum of the shadow's
 * values.  Needs 16-byte aligned bases, row / batch strides % 8 == 0, and K1 / N % 8 == 0 or rows padded to a multiple
 * of 8 elements (row_stride >= the width rounded up to 8: a 1500-channel gradient lives in a 1504-wide shadow). */
/* Kernel variant of the calling thread's most recent lidbox_gemm_bf16s_nt (profiling tools): out3 = {tile rows, tile columns,
 * LDS ring stages} of the LDS-DMA instantiation gemm16s_rows_dma_kernel; {256, tile columns, sub-steps} for the eight-wave
 * ping-pong tile (gemm16_pp.h); {1, 64, 1} for the K-resident short-contraction kernel (gemm16_kres.h: K <= 208 over windows of
 * utterances whose 32-row frame image fits 3 KB -- the x-vector's first frame layer; LIDBOX_GEMM16S_KRES=0 / 1 forces never /
 * whenever it can run); zeros for the register-staged 128 x 128 kernel. */
int lidbox_gemm_bf16s_last_variant(int* out3);
size_t lidbox_gemm_bf16s_tn_workspace(int M, int K1, int N);
int lidbox_gemm_bf16s_tn(lidbox_rows_t A16, lidbox_rows_t B16, float* C, long ldc, int K1, int N,
                         int accumulate, float* bias_grad, void* workspace, size_t workspace_bytes,
                         lidbox_stream_t stream);
/* The carried-reduce forms of the storage kernels (see lidbox_gemm_tn_partial / lidbox_gemm_nt_carry above: same contract,
 * same bit-identical sums; jobs of either family may be carried -- the dense head's wgrads run on the fp32 family):
 *   lidbox_gemm_bf16s_tn_partial   the wgrad GEMM only, *job = its pending fixed-order slice sum
 *   lidbox_gemm_bf16s_nt_carry     lidbox_gemm_bf16s_nt whose (first) launch runs up to two pending jobs in its leading workgroups
 *   lidbox_gemm_bf16s_last_carried jobs the calling thread's most recent lidbox_gemm_bf16s_nt_carry ran inside its launch */
int lidbox_gemm_bf16s_tn_partial(lidbox_rows_t A16, lidbox_rows_t B16, float* C, long ldc, int K1, int N, int accumulate,
                                 float* bias_grad, void* workspace, size_t workspace_bytes, lidbox_reduce_job_t* job,
                                 lidbox_stream_t stream);
int lidbox_gemm_bf16s_nt_carry(lidbox_rows_t A16, const void* B16, long ldb, lidbox_rows_out_t C, void* C16, int K, int N,
                               int epilogue, const float* aux, void* workspace, size_t workspace_bytes,
                               const lidbox_reduce_job_t* jobs, int njobs, lidbox_stream_t stream);
int lidbox_gemm_bf16s_last_carried(void);
/* Two independent lidbox_gemm_bf16s_nt problems as ONE launch (round 6).  Reference: the input gradient of a strided Conv1D
 * (xvector.py:40, kernel 3 / stride 2: tf.keras computes it as one conv_backprop_input); here it is one GEMM per row residue of the
 * output-stationary form -- 492 tiles 1 024 deep and 396 tiles 512 deep at 512 utterances: 1.92 + 1.55 rounds of 256 CUs as two
 * launches, 3.47 as one grid.  Both problems must be launches lidbox_gemm_bf16s_nt would run on the 256 x 256 ping-pong tile without a
 * K split (gemm16_pp.h); anything else, or LIDBOX_GEMM16S_NO_PAIR=1, runs the two calls one after the other (the jobs ride with the
 * first): same bits either way.  The outputs must not overlap; the operands may.  lidbox_gemm_bf16s_last_pair(): 1 if the calling
 * thread's most recent call ran as one grid. */
int lidbox_gemm_bf16s_nt_pair_carry(lidbox_rows_t A16_0, const void* B16_0, long ldb0, lidbox_rows_out_t C0, void* C16_0, int K0, int N0,
                                    int epilogue0, const float* aux0, lidbox_rows_t A16_1, const void* B16_1, long ldb1,
                                    lidbox_rows_out_t C1, void* C16_1, int K1, int N1, int epilogue1, const float* aux1,
                                    void* workspace, size_t workspace_bytes, const lidbox_reduce_job_t* jobs, int njobs,
                                    lidbox_stream_t stream);
int lidbox_gemm_bf16s_last_pair(void);
/* Kernel variant of the calling thread's most recent lidbox_gemm_bf16s_tn / _tn_partial (tests, profiling tools): the number of M
 * slices of the 256 x 256 eight-wave ping-pong tile (gemm16_pp_tn.h) if that ran, 0 for the four-wave 128 x 128 kernel.  The tile
 * is chosen for long slices of big layers (K1, N multiples of 256, >= 8 tiles, >= 2048 rows per slice: frame2's wgrad at 512
 * utterances); LIDBOX_GEMM16_TN_PP=0 / 1 in the environment forces never / whenever the operands allow (tuning aid). */
int lidbox_gemm_bf16s_tn_last_pp(void);
/* ... and the number of utterance slices of the K1-resident kernel (gemm16_tn_kres.h, round 6) if that ran, else 0: a contraction
 * of <= 224 (whole output rows of one workgroup), N a multiple of 128, overlapping or adjacent windows of a batched input whose batch
 * stride is a whole number of row strides -- frame1's wgrad (k = 5, 40 channels: K1 = 200).  LIDBOX_GEMM16_TN_KRES=0 / 1: never /
 * whenever the operands allow. */
int lidbox_gemm_bf16s_tn_last_kres(void);
/* All bf16 weight shadows of a model in ONE launch (once per train step, after the optimizer): flat16[i] = bf16(flat[i]) for
 * the n parameters, and for each listed row-major [rows][cols] matrix at flat + offset a bf16 image at dst with its own
 * leading dimension: transposed ([cols][rows]: a Keras Conv1D kernel [k*C_in][C_out] -> the K-inner [C_out][k*C_in] operand
 * the forward GEMM reads) or as it is (rows padded to 8 elements, or several taps' [C_in][C_out] blocks side by side: the
 * operand images dgrad reads).  More than 48 matrices take further launches.  flat16 == NULL: only the listed matrices
 * (a model whose GEMMs read images only; a kernel that IS read in place is then listed with dst inside the flat16 buffer). */
typedef struct {
    long  offset;       /* of the matrix inside flat, in floats */
    int   rows, cols;
    void* dst;          /* bf16 destination */
    long  ld_dst;       /* elements between destination rows */
    int   transpose;    /* 1: dst[c * ld_dst + r] = src[r][c]   0: dst[r * ld_dst + c] = src[r][c] */
} lidbox_weight_shadow_t;
int lidbox_refresh_bf16_weights(const float* flat, void* flat16, long n, const lidbox_weight_shadow_t* mats, int nmats,
                                lidbox_stream_t stream);
/* dst[i] = bf16(src[i]) (round-to-nearest-even), n elements; dst[c][r] = bf16(src[r][c]) for an R x C matrix */
int lidbox_f32_to_bf16(const float* src, void* dst, long n, lidbox_stream_t stream);
/* dst[i] = float(src[i]) (exact), n elements.  With lidbox_f32_to_bf16: the bf16 wire format of the data-parallel gradient
 * exchange (Trainer(grad_wire_dtype="bfloat16"): a bucket is rounded once, all-reduced as bf16, widened back to fp32 for
 * Adam).  New in this build: the reference is single-device (lidbox/models/keras_utils.py:191-203). */
int lidbox_bf16_to_f32(const void* src, float* dst, long n, lidbox_stream_t stream);
/* Measurement aid (no reference counterpart): one wave that runs for `microseconds` by the device's constant-rate wall clock.
 * A kernel of known duration -- bench.py brackets it with HIP events to measure what a bracket adds to a launch. */
int lidbox_calibration_spin(double microseconds, lidbox_stream_t stream);
int lidbox_transpose_f32_to_bf16(const float* src, int R, int C, long ld_src, void* dst, long ld_dst,
                                 lidbox_stream_t stream);

/* out[n] (+)= sum_m rows[m, n]  -- bias gradient; deterministic two-stage reduction through
 * `workspace` (>= lidbox_colsum_workspace() bytes). */
size_t lidbox_colsum_workspace(long M, int N);
int lidbox_colsum(lidbox_rows_t A, int N, float* out, int accumulate, void* workspace,
                  size_t workspace_bytes, lidbox_stream_t stream);

/* ------------------------------------------------------------------ signal steps ahead of the features (8f.3)
 * Ragged batches: utterance b = signals[starts[b] .. starts[b] + lengths[b]), `starts` / `lengths` = B int64
 * each in DEVICE memory (gaps between utterances are allowed, e.g. 16-byte alignment of every start).
 * Frames / chunks of all utterances are numbered consecutively; frame_offsets / chunk_offsets (B+1 int64, device,
 * CSR) give each utterance's first one.  All entry points are stream-ordered; data-dependent sizes come back as
 * counters for the caller to size the next buffer (the reference's eager tensors do the same). */

/* features/audio.py:17-23 (tf.audio.decode_wav + reduce_mean over the channel axis) for 16-bit PCM already in device memory:
 * pcm = frames x channels interleaved int16 (a ragged batch of utterances with the same channel count is one flat call),
 * out[f] = (sum_c pcm[f][c] / 32768) / channels in fp32 -- bit-identical to the reference's float32 arithmetic (the scaled
 * samples and their sums are exact, the division rounds once).  Halves the bytes an ingest pipeline moves over PCIe. */
int lidbox_pcm16_to_f32(const int16_t* pcm, long frames, int channels, float* out, lidbox_stream_t stream);

/* data/steps.py:586-588,604-614 in float32 like the reference's tf.cast chain:
 * out4 = {chunk_length, chunk_step, padded signal length, number of chunks} (host only) */
int lidbox_signal_chunk_plan(long num_samples, int sample_rate, int length_ms, int step_ms, int max_pad_ms,
                             long* out4);
/* features/audio.py:314-317: RMS of every non-overlapping frame of frame_len samples (the tail is dropped):
 * frame_rms[frame_offsets[b] + f] */
int lidbox_frame_rms(const float* signals, const int64_t* starts, const int64_t* frame_offsets, int B,
                     long total_frames, int frame_len, float* frame_rms, lidbox_stream_t stream);
/* features/audio.py:318-326: per utterance threshold = strength * max(min_rms_threshold, mean frame RMS);
 * decisions = rms > threshold (uint8 0/1); runs of 0 shorter than min_non_speech_frames are set to 1
 * (invert_too_short_consecutive_false, :289-297).  Also returns slots[f] = number of speech frames before f in
 * its utterance and counts[b] = speech frames of utterance b (inputs of lidbox_apply_vad).  thresholds: [B]. */
int lidbox_vad_decisions(const float* frame_rms, const int64_t* frame_offsets, int B, long total_frames,
                         float strength, float min_rms_threshold, int min_non_speech_frames,
                         uint8_t* decisions, int32_t* slots, int32_t* counts, float* thresholds,
                         lidbox_stream_t stream);
/* slots / counts (as above) for decisions that came from elsewhere (data/steps.py:191-198 takes them as given) */
int lidbox_vad_scan(const uint8_t* decisions, const int64_t* frame_offsets, int B, int32_t* slots, int32_t* counts,
                    lidbox_stream_t stream);
/* data/steps.py:191-198 (also features/audio.py:351-353): out[out_starts[b] + slot*frame_len ..] = speech frames */
int lidbox_apply_vad(const float* signals, const int64_t* starts, const int64_t* frame_offsets,
                     const uint8_t* decisions, const int32_t* slots, const int64_t* out_starts, int B,
                     long total_frames, int frame_len, float* out, lidbox_stream_t stream);
/* data/steps.py:611-614: out [total_chunks, chunk_len]; chunk c of utterance b = samples [c*step, c*step + len),
 * zero past the utterance's end (the bounded padding of :611-612 is part of the plan that sized chunk_offsets) */
int lidbox_signal_chunks(const float* signals, const int64_t* starts, const int64_t* lengths,
                         const int64_t* chunk_offsets, int B, long total_chunks, int chunk_len, int chunk_step,
                         float* out, lidbox_stream_t stream);
/* features/audio.py:57-59: out = 10^(dBFS/20) * x / max|x| per utterance (same ragged layout as signals) */
int lidbox_peak_normalize(const float* signals, const int64_t* starts, const int64_t* lengths, int B,
                          float dBFS, float* out, lidbox_stream_t stream);
/* same result; the caller states the longest utterance and whether every start is a multiple of 4 samples (both known on
 * the host): utterances of up to 65536 samples are then normalised from registers with a single read of the signal */
int lidbox_peak_normalize_max(const float* signals, const int64_t* starts, const int64_t* lengths, int B,
                              float dBFS, long max_length, int aligned16, float* out, lidbox_stream_t stream);
/* features/audio.py:266-270: out_rms[b] = sqrt(mean(x^2)) per utterance */
int lidbox_signal_rms(const float* signals, const int64_t* starts, const int64_t* lengths, int B,
                      float* out_rms, lidbox_stream_t stream);
/* features/audio.py:128-148 on B dense pairs [B, N]; snr_db [B]; three outputs [B, N] */
int lidbox_snr_mixer(const float* clean, const float* noise, const float* snr_db, int B, long N,
                     float* clean_norm, float* noise_new, float* noisy, lidbox_stream_t stream);

/* lidbox/util.py:41-57 merge_chunk_predictions with the default stack_and_average: x [rows, D] sorted so that
 * the rows of segment s are segment_offsets[s] .. segment_offsets[s+1] (int64, device); out [num_segments, D]. */
int lidbox_segment_mean(const float* x, const int64_t* segment_offsets, int num_segments, int D, float* out,
                        lidbox_stream_t stream);

/* ------------------------------------------------------------------ pooling / losses / optimiser */

/* lidbox/models/xvector.py:25-35 GlobalMeanStddevPooling1D: x [B,T,C] (batch_stride, row_stride
 * in floats) -> out [B, 2C] = (mean, sqrt(clip(var, 1e-10, FLT_MAX))), two-pass population var. */
int lidbox_stats_pool_fwd(const float* x, int B, int T, int C, long batch_stride, long row_stride,
                          float* out, lidbox_stream_t stream);
/* backward; dx has x's layout; relu_mask != 0 additionally multiplies by (x > 0) (x is the
 * post-ReLU frame5 output, so this is the ReLU backward of the producing layer). */
int lidbox_stats_pool_bwd(const float* x, const float* pooled, const float* dout, int B, int T, int C,
                          long batch_stride, long row_stride, int relu_mask, float* dx,
                          lidbox_stream_t stream);
/* as lidbox_stats_pool_bwd; additionally writes bf16(dx) (round-to-nearest-even) to the shadow buffer dx16 with its own
 * batch / row strides in bf16 elements (row_stride16 >= C: a shadow padded to 8-element rows feeds lidbox_gemm_bf16s_tn);
 * dx may be NULL: only the shadow is written */
int lidbox_stats_pool_bwd_shadow(const float* x, const float* pooled, const float* dout, int B, int T, int C,
                                 long batch_stride, long row_stride, int relu_mask, float* dx,
                                 void* dx16, long batch_stride16, long row_stride16, lidbox_stream_t stream);
/* The statistics pooling of the bf16 policy's all-shadow mode: x16 = the bfloat16 shadow of the last frame layer's output
 * ([B][T][C] at batch / row strides bs / rs in bf16 elements, e.g. 1500 channels in 1504-wide rows), which that layer's GEMM then
 * writes INSTEAD of an fp32 copy.  Same fp32 arithmetic as lidbox_stats_pool_fwd / _bwd_shadow on the shadow's values
 * (xvector.py:30-35); the backward writes only the gradient's shadow dx16.  T <= 40 (the register kernels), C and strides
 * multiples of 4, 8-byte aligned shadows; anything else is refused. */
int lidbox_stats_pool_fwd_bf16(const void* x16, int B, int T, int C, long bs, long rs, float* out, lidbox_stream_t stream);
int lidbox_stats_pool_bwd_bf16(const void* x16, const float* pooled, const float* dout, int B, int T, int C, long bs, long rs,
                               int relu_mask, void* dx16, long bs16, long rs16, lidbox_stream_t stream);
/* Keras GlobalAveragePooling1D (cnn.py:37) */
int lidbox_avg_pool_fwd(const float* x, int B, int T, int C, long batch_stride, long row_stride,
                        float* out, lidbox_stream_t stream);
int lidbox_avg_pool_bwd(const float* x, const float* dout, int B, int T, int C, long batch_stride,
                        long row_stride, int relu_mask, float* dx, lidbox_stream_t stream);

/* lidbox/models/clstm.py:36-42 frequency_attention as used by xvector_freq_attention.py:29, on `rows` = B*T
 * frames of C channels (dense [rows, C]); d_f <= 64 bins of C/d_f consecutive channels.
 * fwd: F_out [rows, d_f] = softmax(logits) (may alias logits); Hw[r, c] = H[r, c] * F_out[r, c / (C/d_f)].
 * bwd: dlogits = F * (dF - sum_f F*dF), dF[r, f] = sum_{c in bin f} dHw*H;  dH = dHw * F[bin]
 *      (times (H > 0) when relu_mask != 0: H is the post-ReLU output of the last frame layer).  The
 *      contribution through the Dense layers that produced the logits is added by the caller. */
int lidbox_freq_attention_fwd(const float* H, const float* logits, long rows, int C, int d_f,
                              float* F_out, float* Hw, lidbox_stream_t stream);
int lidbox_freq_attention_bwd(const float* H, const float* F, const float* dHw, long rows, int C, int d_f,
                              int relu_mask, float* dlogits, float* dH, lidbox_stream_t stream);

/* tf.nn.log_softmax (xvector.py:65) over rows of z [B,N] */
int lidbox_log_softmax_fwd(const float* z, int B, int N, float* logp, lidbox_stream_t stream);
/* Keras SparseCategoricalCrossentropy(from_logits=True) on log-softmax outputs, mean reduction
 * (keras_utils.py:141-147): loss_out[0] = mean_b(-logp[b, y_b]); dz = (exp(logp) - onehot) * scale
 * (scale = 1/global_batch).  dz may be NULL (evaluation). */
int lidbox_nll_fwd_bwd(const float* logp, const int32_t* labels, int B, int N, float scale,
                       float* loss_out, float* dz, lidbox_stream_t stream);

/* tf.nn.softmax over rows of z [B,N]: the output layer of a model built with output_activation="softmax"
 * (lidbox/models/cnn.py:43-44: getattr(tf.nn, output_activation)) */
int lidbox_softmax_fwd(const float* z, int B, int N, float* out, lidbox_stream_t stream);
/* Keras SparseCategoricalCrossentropy(from_logits=False) on softmax outputs, mean reduction (keras_utils.py:141-147 with such a
 * model): from the logits z, p = softmax(z) (written to probs when not NULL), q = clip(p, 1e-7, 1 - 1e-7), loss_out[0] =
 * mean_b(log sum_j q_bj - log q_b,y); dz (may be NULL) = d(loss)/dz * B * scale through the clip (a clipped probability passes
 * no gradient) and the softmax; labels outside [0,N): NaN loss, zero gradient row. */
int lidbox_softmax_nll_fwd_bwd(const float* z, const int32_t* labels, int B, int N, float scale, float* probs,
                               float* loss_out, float* dz, lidbox_stream_t stream);

/* Output layer + loss of a classifier and their backward in two small launches (train step, few classes): logp =
 * log_softmax(h W + b) with h [B,K] the previous layer's output, W [K,N] a Keras Dense kernel (xvector.py:62-65); loss_out[0]
 * and dz as lidbox_nll_fwd_bwd (scale = 1/global_batch; labels outside [0,N): NaN loss, zero gradient row); dW [K,N] =
 * h^T dz, db [N] = column sums of dz, dh [B,K] = dz W^T, multiplied by (h > 0) when relu_mask != 0 (dh may be NULL).
 * fp32 FMA arithmetic, fixed summation orders (deterministic, no atomics).  lidbox_softmax_head_supported(K, N): N <= 32. */
size_t lidbox_softmax_head_workspace(int B, int K, int N);
int lidbox_softmax_head_supported(int K, int N);
int lidbox_softmax_head_fwd_bwd(const float* h, const float* W, const float* bias, const int32_t* labels, int B, int K, int N,
                                float scale, int relu_mask, float* logp, float* loss_out, float* dW, float* db, float* dh,
                                void* workspace, size_t workspace_bytes, lidbox_stream_t stream);

/* tf.math.l2_normalize(axis=1) forward / backward on [B,D] */
int lidbox_l2_normalize_fwd(const float* x, int B, int D, float* out, lidbox_stream_t stream);
int lidbox_l2_normalize_bwd(const float* x, const float* dout, int B, int D, float* dx,
                            lidbox_stream_t stream);

/* lidbox/losses.py:25-49 SparseAngularProximity.call: per-example loss [B] from L2-normalised
 * z [B,D] (D >= N), sparse labels; dz (may be NULL) = d(mean loss * scale*B)/dz, acos' clamped. */
int lidbox_ap_loss_fwd_bwd(const float* z, const int32_t* labels, int B, int D, int N,
                           float delta_weight, float scale, float* loss_per_example, float* dz,
                           lidbox_stream_t stream);
/* The angular-proximity head of a train step in one launch (D <= 4096): zn = l2_normalize(x) (may be NULL), the per-example loss
 * of lidbox_ap_loss_fwd_bwd on zn, dx (may be NULL) = the loss gradient taken back through the normalisation
 * (lidbox_l2_normalize_bwd), scores (may be NULL) [B,N] = -acos(zn[:, :N]) (SparseAngularProximity.predict, losses.py:51-52).
 * Bit-identical to the four separate calls.  Replaces: lidbox/losses.py:25-52 behind an L2-normalised output (ap_lstm.py:42). */
int lidbox_ap_head_fwd_bwd(const float* x, const int32_t* labels, int B, int D, int N, float delta_weight, float scale,
                           float* zn, float* loss_per_example, float* dx, float* scores, lidbox_stream_t stream);

/* lidbox/metrics.py:51-71 AverageDetectionCost.update_state with sparse labels: counters
 * tp, fn [N,Th]; fp_pairs, tn_pairs [N,N,Th] (float32, accumulated in place). */
int lidbox_cavg_update(const float* scores, const int32_t* labels, int B, int N,
                       const float* thresholds, int Th, float* tp, float* fn, float* fp_pairs,
                       float* tn_pairs, lidbox_stream_t stream);
/* lidbox/metrics.py:73-103 result(): out[0] = min over thresholds of C_avg; c_avg_out [Th] optional */
int lidbox_cavg_result(const float* tp, const float* fn, const float* fp_pairs, const float* tn_pairs,
                       int N, int Th, float C_miss, float C_fa, float P_tar, float* c_avg_out,
                       float* out, lidbox_stream_t stream);

/* tf.keras.optimizers.Adam dense update (keras_utils.py:137-140; epsilon 1e-7):
 *   t = ++state[0];  lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  m,v EMA;  p -= lr_t*m/(sqrt(v)+eps)
 * `state` is a 16-byte device block {int64 step; float lr_t; float lr_now} owned by the caller and
 * advanced ON THE DEVICE, so a captured hipGraph replays with the right bias correction.  lr_now > 0 replaces `lr` for
 * the step (tf.keras.optimizers.schedules.*, keras_utils.py:137-139: the host writes the schedule's value there before
 * the step, also ahead of a graph replay); zero-initialised state = constant `lr`.
 * grad_scale multiplies g first (1/world_size after an all-reduce(sum)). */
int lidbox_adam_step(float* param, const float* grad, float* m, float* v, long n, float lr,
                     float beta1, float beta2, float eps, float grad_scale, void* state,
                     lidbox_stream_t stream);
/* The same step in two halves: lidbox_adam_prepare_job describes its scalar half (advance the counter, publish lr_t) as a
 * job for lidbox_reduce_jobs_run -- the launch that finishes the step's last wgrad then prepares the optimizer too (GEMM
 * launches do not carry this kind: *_carry calls reject it) -- and lidbox_adam_apply is the elementwise update with the lr_t
 * found in `state`.  lidbox_adam_step == the job on its own + lidbox_adam_apply, bit for bit. */
/* tf.keras.optimizers.SGD / RMSprop dense updates (keras_utils.py:137-140: the config names the optimizer class), state = the
 * 16-byte device state of lidbox_adam_step {int64 step, float lr_t, float lr_now} (step advanced, lr_now honoured).
 * SGD: velocity (n floats, may be NULL when momentum == 0): v = momentum v - lr g; w += nesterov ? momentum v - lr g : v.
 * RMSprop: rms = rho rms + (1 - rho) g^2; centered: mean_grad = rho mean_grad + (1 - rho) g, denom = rms - mean_grad^2;
 * momentum == 0: w -= lr g / (sqrt(denom) + epsilon); momentum > 0: mom = momentum mom + lr g / sqrt(denom + epsilon), w -= mom
 * (TensorFlow 2.3's two forms).  g = grad * grad_scale. */
int lidbox_sgd_step(float* param, const float* grad, float* velocity, long n, float lr, float momentum, int nesterov,
                    float grad_scale, void* state, lidbox_stream_t stream);
int lidbox_rmsprop_step(float* param, const float* grad, float* rms, float* mean_grad, float* mom, long n, float lr, float rho,
                        float momentum, float epsilon, int centered, float grad_scale, void* state, lidbox_stream_t stream);

int lidbox_adam_prepare_job(void* state, float lr, float beta1, float beta2, lidbox_reduce_job_t* job);
int lidbox_adam_apply(float* param, const float* grad, float* m, float* v, long n, float beta1, float beta2, float eps,
                      float grad_scale, const void* state, lidbox_stream_t stream);

/* ------------------------------------------------------------------ BatchNormalization (f1: xvector_2d.py:36,43)
 * tf.keras.layers.BatchNormalization(axis=-1) over x viewed as [R rows, C channels] (dense).  Training: batch mean and
 * population variance -> mean_out / invstd_out (kept for backward), scale = gamma * invstd, shift = beta - mean * scale,
 * and (moving_* != NULL) moving = moving * momentum + batch * (1 - momentum).  Inference: the same constants from the
 * moving statistics.  lidbox_bn_apply writes y = x * scale + shift through a rows descriptor (the last front-end layer
 * lands behind the causal pad rows of the first Conv1D's input).  Deterministic (fixed-order partial sums in float64).
 * workspace: lidbox_bn_workspace(R, C) bytes, 8-byte aligned. */
size_t lidbox_bn_workspace(long R, int C);
int lidbox_bn_train_stats(const float* x, long R, int C, const float* gamma, const float* beta, float eps,
                          float momentum, float* moving_mean, float* moving_var, float* mean_out, float* invstd_out,
                          float* scale_out, float* shift_out, void* workspace, size_t workspace_bytes,
                          lidbox_stream_t stream);
int lidbox_bn_infer_consts(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                           float eps, int C, float* scale_out, float* shift_out, lidbox_stream_t stream);
int lidbox_bn_apply(const float* x, long R, int C, const float* scale, const float* shift, lidbox_rows_out_t y,
                    lidbox_stream_t stream);
/* backward of the training-mode normalisation: dgamma = sum dy * xhat, dbeta = sum dy,
 * dx = gamma * invstd * (dy - mean(dy) - xhat * mean(dy * xhat)); relu_mask != 0 multiplies dx by (x > 0): x is then the
 * output of the Conv2D's ReLU and dx the gradient in front of it.  dy through a rows descriptor, x / dx dense. */
int lidbox_bn_bwd(const float* x, lidbox_rows_t dy, long R, int C, const float* mean, const float* invstd,
                  const float* gamma, int relu_mask, float* dgamma, float* dbeta, float* dx, void* workspace,
                  size_t workspace_bytes, lidbox_stream_t stream);

/* out[0] = mean of n floats (one workgroup, fixed order).  Keras reduces the per-example losses of a batch this way
 * (losses.py:38 returns per-example values; keras_utils.py:141-149 compiles the mean). */
int lidbox_mean(const float* x, long n, float* out, lidbox_stream_t stream);

/* out[b, n] = -acos(z[b, n]) for the first N of D columns: SparseAngularProximity.predict (losses.py:44-47), the
 * scores the C_avg metric sees. */
int lidbox_neg_acos(const float* z, long B, int D, int N, float* out, lidbox_stream_t stream);

/* Keras Dropout (element-wise; FrameLayer2D(dropout_rate=...), xvector_2d.py:37-46) in place on the rows x[r][0..C) of an
 * implicit-row descriptor: zero with probability `rate`, kept values scaled by 1/(1-rate).  The mask is a counter-based hash
 * of (seed, *step_counter, r, c) -- the same call on the gradient rows regenerates the forward mask. */
int lidbox_dropout_rows(lidbox_rows_out_t x, int C, float rate, unsigned long long seed, const void* step_counter,
                        lidbox_stream_t stream);

/* Keras SpatialDropout1D (xvector.py:50-51, cnn.py:29-30) in place on x [B, T, C] (batch stride in floats): whole
 * channels of an utterance are zeroed with probability `rate`, kept ones scaled by 1/(1-rate).  The mask is a
 * counter-based hash of (seed, *step_counter, b, c); step_counter is a DEVICE int64 (NULL = 0), e.g. the Adam step,
 * so a replayed hipGraph draws a fresh mask per step.  mask_out: optional [B, C] copy of the applied factors. */
int lidbox_spatial_dropout(float* x, int B, int T, int C, long batch_stride, float rate,
                           unsigned long long seed, const void* step_counter, float* mask_out,
                           lidbox_stream_t stream);

/* stream-ordered pitched device-to-device copy / zero fill (hipMemcpy2DAsync / hipMemset2DAsync: memcpy / memset
 * nodes under graph capture): what moves a dense [B, T, C] batch behind the causal pad rows of a layer input. */
int lidbox_copy_2d(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes,
                   size_t height, lidbox_stream_t stream);
int lidbox_zero_2d(void* dst, size_t pitch, size_t width_bytes, size_t height, lidbox_stream_t stream);

/* x *= alpha over n floats (stream-ordered): turns an all-reduce(sum) of the replicas' BatchNormalization running
 * statistics into their mean */
int lidbox_scale(float* x, long n, float alpha, lidbox_stream_t stream);

/* fill n floats with value (stream-ordered) */
int lidbox_fill(float* x, long n, float value, lidbox_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LIDBOX_HIP_H */
