// norm.hip -- CMN/CMVN, sliding-window normalisation, min-max scaling, power_to_db (gfx950).
//
// Replaces (reference file:line):
//   lidbox/features/__init__.py:12-32   cmn / cmvn       (reduce_mean, reduce_std, divide_no_nan)
//   lidbox/features/__init__.py:35-67   window_normalization (REFLECT pad + tf.signal.frame)
//   lidbox/features/__init__.py:5-9     feature_scaling
//   lidbox/features/audio.py:162-174    log10, power_to_db
// All HBM-bound and tiny next to the signal read (31 KB/utt vs 128 KB/utt); they are written
// for coalesced access along the innermost axis and two-pass (mean, then centred variance)
// statistics, the structure of tf.math.reduce_std.
#include <float.h>

#include "common.h"

namespace {

// x viewed as [outer, R, inner]; one workgroup per (outer, column tile of cw columns);
// 256 threads = cw columns x (256/cw) row groups.
// x and out may be the same buffer (every element is read and written by the same thread, the write comes last);
// xs / os: floats between consecutive `outer` slices (>= R * inner).
__global__ __launch_bounds__(256) void cmvn_kernel(const float* x, long outer, long R, long inner, long xs, long os,
                                                   int cw, int normalize_variance,
                                                   float* out) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int col = tid % cw, g = tid / cw, ng = 256 / cw;
    const long c = (long)blockIdx.x * cw + col;
    const bool active = c < inner;
    // grid.y is capped at 65 535: a workgroup walks the outer slices o = blockIdx.y, + gridDim.y, ... (block-uniform trip count)
    for (long o = blockIdx.y; o < outer; o += gridDim.y) {
        const float* xp = x + o * xs + c;
        float* op = out + o * os + c;

        float s = 0.f;
        if (active)
            for (long r = g; r < R; r += ng) s += xp[r * inner];
        __syncthreads();                                 // the previous slice's readers of red[] are done
        red[tid] = s;
        __syncthreads();
        for (int h = ng / 2; h > 0; h >>= 1) {
            if (g < h) red[tid] += red[tid + h * cw];
            __syncthreads();
        }
        const float mean = red[col] / (float)R;
        __syncthreads();

        float sd = 1.f;
        if (normalize_variance) {
            float v = 0.f;
            if (active)
                for (long r = g; r < R; r += ng) {
                    const float d = xp[r * inner] - mean;
                    v = fmaf(d, d, v);
                }
            red[tid] = v;
            __syncthreads();
            for (int h = ng / 2; h > 0; h >>= 1) {
                if (g < h) red[tid] += red[tid + h * cw];
                __syncthreads();
            }
            sd = sqrtf(red[col] / (float)R);             // population std of x
        }
        if (active)
            for (long r = g; r < R; r += ng) {
                const float d = xp[r * inner] - mean;
                op[r * inner] = normalize_variance ? (sd != 0.f ? d / sd : 0.f) : d;   // divide_no_nan
            }
    }
}

// numpy-'reflect' index into [0, T)
__device__ __forceinline__ int reflect_idx(int i, int T) {
    if (i < 0) i = -i;
    if (i >= T) i = 2 * (T - 1) - i;
    return i;
}

// Sliding window statistics.  Window t covers padded rows t..t+w-1, padded row i <-> x row reflect(i - w/2).
// One thread per (utterance, channel) walks the time axis with running sums of (x - x0) and (x - x0)^2 in FLOAT64
// (one value enters and one leaves per step), so the kernel is O(T) per channel instead of the reference's O(T*w)
// materialised windows (features/__init__.py:61-66).  The variance comes from the one-pass identity, which in
// float64 on float32 data shifted by the channel's first sample agrees with tf.math.reduce_std's two-pass result
// to float32 rounding (the tests compare with the float64 two-pass oracle at 1e-3 like every other row).
// Consecutive threads are consecutive channels: every access is a coalesced row segment.
__global__ __launch_bounds__(256) void window_norm_kernel(const float* __restrict__ x, int B, int T, int C, int w,
                                                          int normalize_variance, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * C) return;
    const int c = (int)(i % C);
    const long b = i / C;
    const float* xb = x + b * (long)T * C + c;
    float* ob = out + b * (long)T * C + c;
    const int left = w / 2;
    const float x0 = xb[0];
    double s = 0.0, q = 0.0;
    for (int j = 0; j < w; ++j) {
        const double v = (double)(xb[(long)reflect_idx(j - left, T) * C] - x0);
        s += v;
        q += v * v;
    }
    const double inv_w = 1.0 / (double)w;
    // 8 time steps per iteration: their 24 loads do not depend on the running sums and are issued first
    for (int t0 = 0; t0 < T; t0 += 8) {
        float xc[8], xa[8], xe[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = min(t0 + u, T - 1);
            xc[u] = xb[(long)t * C];
            xa[u] = xb[(long)reflect_idx(t - left, T) * C];
            xe[u] = xb[(long)reflect_idx(min(t + w - left, 2 * (T - 1)), T) * C];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = t0 + u;
            if (t >= T) break;
            const double mean = s * inv_w;
            const float d = (float)((double)(xc[u] - x0) - mean);
            if (!normalize_variance) {
                ob[(long)t * C] = d;
            } else {
                const double var = q * inv_w - mean * mean;
                const float sd = var > 0.0 ? (float)sqrt(var) : 0.f;
                ob[(long)t * C] = sd != 0.f ? d / sd : 0.f;                   // divide_no_nan
            }
            const double a = (double)(xa[u] - x0), e = (double)(xe[u] - x0);
            s += e - a;                                                        // window of step t + 1
            q += e * e - a * a;
        }
    }
}

constexpr int MM_MAX_WG = 1024;

// tf.reduce_min / reduce_max propagate NaN (reference features/__init__.py:7-8 via tf.math.reduce_*); fminf / fmaxf drop it
__device__ __forceinline__ float nan_min(float a, float b) { return (a != a) ? a : ((b != b) ? b : fminf(a, b)); }
__device__ __forceinline__ float nan_max(float a, float b) { return (a != a) ? a : ((b != b) ? b : fmaxf(a, b)); }
__device__ __forceinline__ float wave_nan_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = nan_min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_nan_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = nan_max(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(256) void minmax_stage1(const float* __restrict__ x, long n,
                                                     float* __restrict__ scratch) {
    __shared__ float smin[4], smax[4];
    float mn = FLT_MAX, mx = -FLT_MAX;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        mn = nan_min(mn, v);
        mx = nan_max(mx, v);
    }
    mn = wave_nan_min(mn);
    mx = wave_nan_max(mx);
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        scratch[blockIdx.x] = nan_min(nan_min(smin[0], smin[1]), nan_min(smin[2], smin[3]));
        scratch[MM_MAX_WG + blockIdx.x] = nan_max(nan_max(smax[0], smax[1]), nan_max(smax[2], smax[3]));
    }
}

__global__ __launch_bounds__(256) void minmax_stage2(const float* __restrict__ scratch, int nwg,
                                                     float* __restrict__ out2) {
    __shared__ float smin[4], smax[4];
    float mn = FLT_MAX, mx = -FLT_MAX;
    for (int i = threadIdx.x; i < nwg; i += 256) {
        mn = nan_min(mn, scratch[i]);
        mx = nan_max(mx, scratch[MM_MAX_WG + i]);
    }
    mn = wave_nan_min(mn);
    mx = wave_nan_max(mx);
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out2[0] = nan_min(nan_min(smin[0], smin[1]), nan_min(smin[2], smin[3]));
        out2[1] = nan_max(nan_max(smax[0], smax[1]), nan_max(smax[2], smax[3]));
    }
}

__global__ void feature_scaling_kernel(const float* __restrict__ x, long n,
                                       const float* __restrict__ mm, float lo, float hi,
                                       float* __restrict__ out) {
    const float mn = mm[0], range = mm[1] - mm[0];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float q = range != 0.f ? (x[i] - mn) / range : 0.f;       // divide_no_nan
        out[i] = lo + (hi - lo) * q;
    }
}

__device__ __forceinline__ float log10_tf(float v) { return logf(v) / logf(10.0f); }   // audio.py:164

// audio.log10 (audio.py:162-164) as its own op: ln(x) / ln(10), elementwise
__global__ void log10_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = log10_tf(x[i]);
}

// feature_scaling over ONE axis (features/__init__.py:5-9 with axis = k): x viewed as [outer][R][inner], min / max over R for
// every (outer, inner), out = lo + (hi - lo) * divide_no_nan(x - min, max - min).  The decomposition of cmvn_kernel: a
// workgroup owns cw consecutive inner positions of one outer index, 256 / cw row groups share the R rows.
__global__ __launch_bounds__(256) void axis_scaling_kernel(const float* __restrict__ x, long outer, long R, long inner, int cw, float lo, float hi,
                                                           float* __restrict__ out) {
    __shared__ float rmin[256], rmax[256];
    const int tid = threadIdx.x;
    const int col = tid % cw, g = tid / cw, ng = 256 / cw;
    const long c = (long)blockIdx.x * cw + col;
    const bool active = c < inner;
    // grid.y strides over the outer index (a grid dimension holds at most 65 535 workgroups)
    for (long o = blockIdx.y; o < outer; o += gridDim.y) {
        const float* xp = x + o * R * inner + c;
        float* op = out + o * R * inner + c;
        float mn = INFINITY, mx = -INFINITY;
        if (active)
            for (long r = g; r < R; r += ng) {
                const float v = xp[r * inner];
                mn = nan_min(mn, v);
                mx = nan_max(mx, v);
            }
        rmin[tid] = mn;
        rmax[tid] = mx;
        __syncthreads();
        for (int h = ng / 2; h > 0; h >>= 1) {
            if (g < h) {
                rmin[tid] = nan_min(rmin[tid], rmin[tid + h * cw]);
                rmax[tid] = nan_max(rmax[tid], rmax[tid + h * cw]);
            }
            __syncthreads();
        }
        const float lo_x = rmin[col], range = rmax[col] - rmin[col];
        if (active)
            for (long r = g; r < R; r += ng) {
                const float q = range != 0.f ? (xp[r * inner] - lo_x) / range : 0.f;      // divide_no_nan
                op[r * inner] = lo + (hi - lo) * q;
            }
        __syncthreads();                                   // rmin / rmax are rewritten by the next outer index
    }
}

__global__ void power_to_db_kernel(const float* __restrict__ S, long n, const float* __restrict__ mm,
                                   float amin, float top_db, float* __restrict__ out) {
    const float ref = log10_tf(fmaxf(amin, mm[1]));
    // the batch-global max of the dB spectrogram is 20*(ref - ref) = 0 exactly, so the floor
    // reduce_max(db) - top_db (audio.py:174) is -top_db
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float db = 20.0f * (log10_tf(fmaxf(amin, S[i])) - ref);
        out[i] = fmaxf(db, 0.0f - top_db);
    }
}

inline unsigned ew_grid(long n) {
    long g = lbx_cdiv(n, 256);
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int lidbox_cmvn_fwd(const float* x, long outer, long R, long inner, int normalize_variance,
                               float* out, lidbox_stream_t stream) {
    return lidbox_cmvn_strided_fwd(x, outer, R, inner, R * inner, normalize_variance, out, R * inner, stream);
}

extern "C" int lidbox_cmvn_strided_fwd(const float* x, long outer, long R, long inner, long x_outer_stride,
                                       int normalize_variance, float* out, long out_outer_stride,
                                       lidbox_stream_t stream) {
    LBX_ARG(x && out, "x, out != NULL");
    LBX_ARG(outer >= 0 && R >= 0 && inner >= 0, "non-negative shape");
    LBX_ARG(x_outer_stride >= R * inner && out_outer_stride >= R * inner, "outer strides >= R * inner");
    if (outer == 0 || R == 0 || inner == 0) return LIDBOX_OK;
    int cw = 64;
    while (cw > 1 && cw / 2 >= inner) cw /= 2;
    dim3 grid((unsigned)lbx_cdiv(inner, cw), (unsigned)(outer < 65535 ? outer : 65535));
    hipLaunchKernelGGL(cmvn_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, outer, R, inner, x_outer_stride,
                       out_outer_stride, cw, normalize_variance, out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_window_norm_fwd(const float* x, int B, int T, int C, int window_len,
                                      int normalize_variance, float* out, lidbox_stream_t stream) {
    LBX_ARG(x && out, "x, out != NULL");
    LBX_ARG(B >= 0 && T >= 0 && C >= 0, "non-negative shape");
    LBX_ARG(window_len >= 2 && window_len < T, "2 <= window_len < T (the sliding branch; use cmvn otherwise)");
    const long total = (long)B * T * C;
    if (total == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(window_norm_kernel, dim3((unsigned)lbx_cdiv((long)B * C, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, B, T, C, window_len, normalize_variance, out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_minmax(const float* x, long n, float* out2, float* scratch, lidbox_stream_t stream) {
    LBX_ARG(x && out2 && scratch && n > 0, "x, out2, scratch != NULL, n > 0");
    long g = lbx_cdiv(n, 256 * 8);
    const int nwg = (int)(g < 1 ? 1 : (g > MM_MAX_WG ? MM_MAX_WG : g));
    hipLaunchKernelGGL(minmax_stage1, dim3(nwg), dim3(256), 0, (hipStream_t)stream, x, n, scratch);
    LBX_LAUNCH_OK();
    hipLaunchKernelGGL(minmax_stage2, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, nwg, out2);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_feature_scaling_fwd(const float* x, long n, const float* minmax2, float lo, float hi,
                                          float* out, lidbox_stream_t stream) {
    LBX_ARG(x && out && minmax2 && n >= 0, "x, out, minmax2 != NULL");
    if (n == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(feature_scaling_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n,
                       minmax2, lo, hi, out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_feature_scaling_axis_fwd(const float* x, long outer, long R, long inner, float lo, float hi, float* out,
                                               lidbox_stream_t stream) {
    LBX_ARG(x && out, "x, out != NULL");
    LBX_ARG(outer >= 0 && R >= 0 && inner >= 0, "non-negative shape");
    if (outer == 0 || R == 0 || inner == 0) return LIDBOX_OK;
    int cw = 64;
    while (cw > 1 && cw / 2 >= inner) cw /= 2;
    hipLaunchKernelGGL(axis_scaling_kernel, dim3((unsigned)lbx_cdiv(inner, cw), (unsigned)(outer < 65535 ? outer : 65535)), dim3(256), 0, (hipStream_t)stream, x, outer, R, inner,
                       cw, lo, hi, out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_log10_fwd(const float* x, long n, float* out, lidbox_stream_t stream) {
    LBX_ARG(x && out && n >= 0, "x, out != NULL");
    if (n == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(log10_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n, out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_power_to_db_fwd(const float* S, long n, const float* minmax2, float amin, float top_db,
                                      float* out, lidbox_stream_t stream) {
    LBX_ARG(S && out && minmax2 && n >= 0, "S, out, minmax2 != NULL");
    if (n == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(power_to_db_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, S, n,
                       minmax2, amin, top_db, out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}
