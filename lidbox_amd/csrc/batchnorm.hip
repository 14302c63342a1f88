// batchnorm.hip -- tf.keras.layers.BatchNormalization(axis=-1) forward / backward over activations viewed as
// [R rows, C channels] (gfx950).
//
// Replaces (reference file:line):
//   lidbox/models/xvector_2d.py:36,43      FrameLayer2D: Conv2D(activation="relu") -> BatchNormalization -> (Dropout)
// Keras defaults restated: momentum 0.99, epsilon 1e-3, gamma 1 / beta 0, moving_mean 0 / moving_variance 1;
// training normalises with the batch mean and the POPULATION variance of the batch and moves the running statistics by
// (1 - momentum) towards them -- the running VARIANCE towards the Bessel-corrected batch variance var * n / (n - 1): the
// layer's 4-D input with axis = -1 takes tf.keras' fused path, which keeps the fused kernel's unbiased estimate for the
// running average (`_bessels_correction_test_only` is True by default) while normalising with the population variance;
// inference uses the running statistics.
//
// All four kernels are HBM-bound streams over x (and dy): column sums are accumulated in FLOAT64 per thread (one pass gives
// mean and E[x^2] without cancellation trouble), partials [slices][C] are combined in a fixed order (deterministic, no
// atomics), the normalisation itself is y = x * scale[c] + shift[c] with the per-channel constants precomputed.
// Roofline: HBM (one read of x for the statistics, one read + one write for the apply; backward reads x and dy twice).
#include "common.h"

namespace {

struct RowMap {            // row r of a [R, C] view -> element offset (r / rpb) * bs + (r % rpb) * rs
    long bs, rs;
    int rpb;
};

__device__ __forceinline__ long map_row(const RowMap& m, long r) {
    if (m.rpb <= 0) return r * m.rs;
    const long b = r / m.rpb;
    return b * m.bs + (r - b * m.rpb) * m.rs;
}

constexpr int BN_COLS = 64;       // columns per workgroup; 4 row groups of 64 threads

// stage 1 of the column reductions.  MODE 0: s0 = sum x, s1 = sum x^2.  MODE 1: s0 = sum dy, s1 = sum dy * xhat with
// xhat = (x - mean) * invstd.  partial [2][C][slices] doubles (a channel's slices are contiguous: stage 2 reads them coalesced).
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_stage1(const float* __restrict__ x, const float* __restrict__ dy,
                                                        RowMap dmap, long R, int C, long rows_per_slice,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        double* __restrict__ partial) {
    __shared__ double red[2][256];
    const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * BN_COLS + col;
    const long r0 = (long)blockIdx.y * rows_per_slice;
    long r1 = r0 + rows_per_slice;
    if (r1 > R) r1 = R;
    double s0 = 0.0, s1 = 0.0;
    if (c < C) {
        const float mu = MODE == 1 ? mean[c] : 0.f, is = MODE == 1 ? invstd[c] : 0.f;
        for (long r = r0 + g; r < r1; r += 4) {
            const float xv = x[r * C + c];
            if (MODE == 0) {
                s0 += (double)xv;
                s1 += (double)xv * (double)xv;
            } else {
                const float d = dy[map_row(dmap, r) + c];
                s0 += (double)d;
                s1 += (double)d * (double)((xv - mu) * is);
            }
        }
    }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    if (g == 0 && c < C) {
        const long slices = gridDim.y;
        partial[(long)c * slices + blockIdx.y] = red[0][col] + red[0][col + 64] + red[0][col + 128] + red[0][col + 192];
        partial[((long)C + c) * slices + blockIdx.y] = red[1][col] + red[1][col + 64] + red[1][col + 128] + red[1][col + 192];
    }
}

// Sum of one channel's slice partials by one workgroup of 256 threads: thread t adds slices t, t + 256, ... in order, then a
// fixed-shape tree over the 256 thread sums -- the same association for every launch (deterministic), and the <= 1 024 slices of
// a channel are read as one contiguous run instead of by a single thread (the first version's serial loop cost 230 us per call).
__device__ __forceinline__ void bn_channel_sums(const double* __restrict__ partial, int slices, int C, int c, double& s0, double& s1) {
    __shared__ double red[2][256];
    const int t = threadIdx.x;
    double a0 = 0.0, a1 = 0.0;
    for (int k = t; k < slices; k += 256) {
        a0 += partial[(long)c * slices + k];
        a1 += partial[((long)C + c) * slices + k];
    }
    red[0][t] = a0;
    red[1][t] = a1;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (t < h) {
            red[0][t] += red[0][t + h];
            red[1][t] += red[1][t + h];
        }
        __syncthreads();
    }
    s0 = red[0][0];
    s1 = red[1][0];
}

// stage 2 (training statistics), one workgroup per channel: batch mean, population variance, invstd, the per-channel
// scale / shift of the apply kernel, and the moving-statistics update
__global__ __launch_bounds__(256) void bn_stats_stage2(const double* __restrict__ partial, int slices, long R, int C,
                                const float* __restrict__ gamma,
                                const float* __restrict__ beta, float eps, float momentum, float* __restrict__ moving_mean,
                                float* __restrict__ moving_var, float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x;
    double s0, s1;
    bn_channel_sums(partial, slices, C, c, s0, s1);
    if (threadIdx.x != 0) return;
    const double mu = s0 / (double)R;
    double var = s1 / (double)R - mu * mu;
    if (var < 0.0) var = 0.0;
    const float muf = (float)mu, varf = (float)var;
    const float is = 1.0f / sqrtf(varf + eps);
    mean_out[c] = muf;
    invstd_out[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - muf * sc;
    if (moving_mean) {
        moving_mean[c] = moving_mean[c] * momentum + muf * (1.f - momentum);
        const float unbiased = R > 1 ? (float)(var * ((double)R / (double)(R - 1))) : varf;
        moving_var[c] = moving_var[c] * momentum + unbiased * (1.f - momentum);
    }
}

__global__ void bn_infer_consts(const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ moving_mean, const float* __restrict__ moving_var, float eps,
                                int C, float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(moving_var[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - moving_mean[c] * sc;
}

// y[row(r)][c] = x[r][c] * scale[c] + shift[c]; C4 = C / 4 float4 columns when VEC
template <bool VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, long R, int C, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float* __restrict__ y, RowMap ymap) {
    const long per_row = VEC ? C / 4 : C;
    const long total = R * per_row;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / per_row;
        const int j = (int)(i - r * per_row);
        if (VEC) {
            const float4 xv = reinterpret_cast<const float4*>(x)[i];
            const float4 sc = reinterpret_cast<const float4*>(scale)[j], sh = reinterpret_cast<const float4*>(shift)[j];
            float4 o;
            o.x = fmaf(xv.x, sc.x, sh.x); o.y = fmaf(xv.y, sc.y, sh.y);
            o.z = fmaf(xv.z, sc.z, sh.z); o.w = fmaf(xv.w, sc.w, sh.w);
            *reinterpret_cast<float4*>(y + map_row(ymap, r) + 4 * j) = o;
        } else {
            y[map_row(ymap, r) + j] = fmaf(x[i], scale[j], shift[j]);
        }
    }
}

// stage 2 of backward, one workgroup per channel: dgamma = sum dy * xhat, dbeta = sum dy (fixed order), plus the per-channel
// constants of dx = gamma * invstd * (dy - dbeta / R - xhat * dgamma / R)
__global__ __launch_bounds__(256) void bn_bwd_stage2(const double* __restrict__ partial, int slices, long R, int C,
                              const float* __restrict__ gamma,
                              const float* __restrict__ invstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                              float* __restrict__ k_dy, float* __restrict__ k_mean_dy, float* __restrict__ k_mean_dyx) {
    const int c = blockIdx.x;
    double s0, s1;
    bn_channel_sums(partial, slices, C, c, s0, s1);
    if (threadIdx.x != 0) return;
    dbeta[c] = (float)s0;
    dgamma[c] = (float)s1;
    k_dy[c] = gamma[c] * invstd[c];
    k_mean_dy[c] = (float)(s0 / (double)R);
    k_mean_dyx[c] = (float)(s1 / (double)R);
}

// dx[r][c] = k_dy * (dy - mean_dy - xhat * mean_dyx), times (x > 0) when the normalised tensor is a ReLU output
// (Conv2D(activation="relu") feeds the BatchNormalization in FrameLayer2D): dx is then the gradient before the ReLU.
template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, RowMap dmap,
                                                           long R, int C, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ k_dy,
                                                           const float* __restrict__ k_mean_dy,
                                                           const float* __restrict__ k_mean_dyx, int relu_mask,
                                                           float* __restrict__ dx) {
    const long per_row = VEC ? C / 4 : C;
    const long total = R * per_row;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / per_row;
        const int j = (int)(i - r * per_row);
        if (VEC) {
            const float4 xv = reinterpret_cast<const float4*>(x)[i];
            const float4 dv = *reinterpret_cast<const float4*>(dy + map_row(dmap, r) + 4 * j);
            const float4 mu = reinterpret_cast<const float4*>(mean)[j], is = reinterpret_cast<const float4*>(invstd)[j];
            const float4 kd = reinterpret_cast<const float4*>(k_dy)[j], km = reinterpret_cast<const float4*>(k_mean_dy)[j];
            const float4 kx = reinterpret_cast<const float4*>(k_mean_dyx)[j];
            float4 g;
#define LBX_BN1(c)                                                                   \
            g.c = kd.c * (dv.c - km.c - (xv.c - mu.c) * is.c * kx.c);                \
            if (relu_mask && !(xv.c > 0.f)) g.c = 0.f;
            LBX_BN1(x) LBX_BN1(y) LBX_BN1(z) LBX_BN1(w)
#undef LBX_BN1
            reinterpret_cast<float4*>(dx)[i] = g;
        } else {
            const float xv = x[i];
            const float xh = (xv - mean[j]) * invstd[j];
            float g = k_dy[j] * (dy[map_row(dmap, r) + j] - k_mean_dy[j] - xh * k_mean_dyx[j]);
            if (relu_mask && !(xv > 0.f)) g = 0.f;
            dx[i] = g;
        }
    }
}

inline int bn_slices(long R) {
    long s = R / 256;
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    return (int)s;
}

inline unsigned bn_grid(long n) {
    long g = lbx_cdiv(n, 256);
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" size_t lidbox_bn_workspace(long R, int C) {
    if (R < 0 || C < 1) return 0;
    return (size_t)bn_slices(R) * 2 * (size_t)C * sizeof(double) + 3 * (size_t)C * sizeof(float);
}

extern "C" int lidbox_bn_train_stats(const float* x, long R, int C, const float* gamma, const float* beta, float eps,
                                     float momentum, float* moving_mean, float* moving_var, float* mean_out,
                                     float* invstd_out, float* scale_out, float* shift_out, void* workspace,
                                     size_t workspace_bytes, lidbox_stream_t stream) {
    LBX_ARG(x && gamma && beta && mean_out && invstd_out && scale_out && shift_out, "pointers != NULL");
    LBX_ARG(R >= 1 && C >= 1, "R >= 1, C >= 1");
    LBX_ARG((moving_mean == nullptr) == (moving_var == nullptr), "moving_mean and moving_var come together");
    LBX_ARG(eps > 0.f && momentum >= 0.f && momentum <= 1.f, "eps > 0, 0 <= momentum <= 1");
    LBX_ARG(workspace && workspace_bytes >= lidbox_bn_workspace(R, C) && (((uintptr_t)workspace) & 7) == 0,
            "workspace >= lidbox_bn_workspace(R, C) bytes, 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int slices = bn_slices(R);
    const long rps = lbx_cdiv(R, slices);
    double* partial = (double*)workspace;
    RowMap none{0, 0, 0};
    hipLaunchKernelGGL(bn_reduce_stage1<0>, dim3((unsigned)lbx_cdiv(C, BN_COLS), (unsigned)slices), dim3(256), 0, st, x,
                       (const float*)nullptr, none, R, C, rps, (const float*)nullptr, (const float*)nullptr, partial);
    LBX_LAUNCH_OK();
    hipLaunchKernelGGL(bn_stats_stage2, dim3((unsigned)C), dim3(256), 0, st, partial, slices, R, C, gamma, beta,
                       eps, momentum, moving_mean, moving_var, mean_out, invstd_out, scale_out, shift_out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_bn_infer_consts(const float* gamma, const float* beta, const float* moving_mean,
                                      const float* moving_var, float eps, int C, float* scale_out, float* shift_out,
                                      lidbox_stream_t stream) {
    LBX_ARG(gamma && beta && moving_mean && moving_var && scale_out && shift_out && C >= 1 && eps > 0.f, "pointers != NULL; C >= 1");
    hipLaunchKernelGGL(bn_infer_consts, dim3((unsigned)lbx_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, gamma, beta,
                       moving_mean, moving_var, eps, C, scale_out, shift_out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_bn_apply(const float* x, long R, int C, const float* scale, const float* shift, lidbox_rows_out_t y,
                               lidbox_stream_t stream) {
    LBX_ARG(x && scale && shift && y.base && R >= 0 && C >= 1, "pointers != NULL; C >= 1");
    LBX_ARG((long)y.batch * y.rows_per_batch == R, "y describes R rows");
    if (R == 0) return LIDBOX_OK;
    RowMap ym{y.batch_stride, y.row_stride, y.batch == 1 ? 0 : y.rows_per_batch};
    const bool vec = C % 4 == 0 && (((uintptr_t)x | (uintptr_t)y.base | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0 &&
                     y.row_stride % 4 == 0 && (y.batch == 1 || y.batch_stride % 4 == 0);
    if (vec)
        hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(bn_grid(R * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, R, C, scale,
                           shift, y.base, ym);
    else
        hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(bn_grid(R * C)), dim3(256), 0, (hipStream_t)stream, x, R, C, scale,
                           shift, y.base, ym);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_bn_bwd(const float* x, lidbox_rows_t dy, long R, int C, const float* mean, const float* invstd,
                             const float* gamma, int relu_mask, float* dgamma, float* dbeta, float* dx, void* workspace,
                             size_t workspace_bytes, lidbox_stream_t stream) {
    LBX_ARG(x && dy.base && mean && invstd && gamma && dgamma && dbeta && dx, "pointers != NULL");
    LBX_ARG(R >= 1 && C >= 1 && (long)dy.batch * dy.rows_per_batch == R, "R >= 1, C >= 1, dy describes R rows");
    LBX_ARG(workspace && workspace_bytes >= lidbox_bn_workspace(R, C) && (((uintptr_t)workspace) & 7) == 0,
            "workspace >= lidbox_bn_workspace(R, C) bytes, 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int slices = bn_slices(R);
    const long rps = lbx_cdiv(R, slices);
    double* partial = (double*)workspace;
    float* consts = (float*)(partial + (size_t)slices * 2 * C);
    RowMap dm{dy.batch_stride, dy.row_stride, dy.batch == 1 ? 0 : dy.rows_per_batch};
    hipLaunchKernelGGL(bn_reduce_stage1<1>, dim3((unsigned)lbx_cdiv(C, BN_COLS), (unsigned)slices), dim3(256), 0, st, x,
                       dy.base, dm, R, C, rps, mean, invstd, partial);
    LBX_LAUNCH_OK();
    hipLaunchKernelGGL(bn_bwd_stage2, dim3((unsigned)C), dim3(256), 0, st, partial, slices, R, C, gamma, invstd,
                       dgamma, dbeta, consts, consts + C, consts + 2 * C);
    LBX_LAUNCH_OK();
    // the three constant rows start C floats apart inside the workspace: float4 access needs C % 4 == 0 and 16-byte aligned bases
    const bool vec = C % 4 == 0 && (((uintptr_t)x | (uintptr_t)dy.base | (uintptr_t)dx | (uintptr_t)mean | (uintptr_t)invstd |
                                     (uintptr_t)consts) & 15) == 0 && dy.row_stride % 4 == 0 && (dy.batch == 1 || dy.batch_stride % 4 == 0);
    if (vec)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(bn_grid(R * (C / 4))), dim3(256), 0, st, x, dy.base, dm, R, C, mean, invstd,
                           consts, consts + C, consts + 2 * C, relu_mask, dx);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(bn_grid(R * C)), dim3(256), 0, st, x, dy.base, dm, R, C, mean, invstd,
                           consts, consts + C, consts + 2 * C, relu_mask, dx);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}
