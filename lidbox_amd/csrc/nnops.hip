// nnops.hip -- pooling, log-softmax/NLL, L2-normalise, angular-proximity loss, C_avg counters,
// Keras-Adam and fill kernels (gfx950).
//
// Replaces (reference file:line):
//   lidbox/models/xvector.py:25-35   GlobalMeanStddevPooling1D
//   lidbox/models/cnn.py:37          GlobalAveragePooling1D
//   lidbox/models/xvector.py:65      tf.nn.log_softmax
//   lidbox/models/keras_utils.py:141-147  SparseCategoricalCrossentropy(from_logits) + Adam(eps 1e-7)
//   lidbox/losses.py:25-52           SparseAngularProximity
//   lidbox/metrics.py:51-103         AverageDetectionCost update_state / result
// Reductions over time run inside one workgroup (time is at most a few hundred frames);
// row-wise ops give each row to one wave64 and reduce with wavefront shuffles.
#include <atomic>
#include <float.h>
#include <stdint.h>

#include <string.h>
#include "common.h"

namespace {

// bfloat16 bits of x, round-to-nearest-even (what v_cvt_pk_bf16_f32 / the GEMM epilogue shadows produce)
__device__ __forceinline__ unsigned short bf16_bits(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }

constexpr float STDDEV_SQRT_MIN_CLIP = 1e-10f;    // xvector.py:22

// grid (ceil(C / (16*V)), B); 256 threads = 16 channel groups of V channels x 16 time groups.  Two passes over the
// utterance's [T, C] block (the second one hits L2): mean, then mean((x - mean)^2) -- the reference's two-pass
// population variance (xvector.py:31-33).  The 16 time groups are combined through LDS in a fixed order.
template <bool STATS, int V>
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ x, int T, int C,
                                                       long bs, long rs, float* __restrict__ out) {
    __shared__ float red[16][16 * V + 1];
    const int tid = threadIdx.x, cg = tid & 15, g = tid >> 4;
    const int c = (blockIdx.x * 16 + cg) * V;
    const int b = blockIdx.y;
    const bool active = c < C;
    const float* xp = x + (long)b * bs + c;
    float s[V];
#pragma unroll
    for (int v = 0; v < V; ++v) s[v] = 0.f;
    if (active)
        for (int t = g; t < T; t += 16) {
            if (V == 4) {
                const float4 q = *reinterpret_cast<const float4*>(xp + (long)t * rs);
                s[0] += q.x; s[1] += q.y; s[2] += q.z; s[3] += q.w;
            } else {
#pragma unroll
                for (int v = 0; v < V; ++v) s[v] += xp[(long)t * rs + v];
            }
        }
#pragma unroll
    for (int v = 0; v < V; ++v) red[g][cg * V + v] = s[v];
    __syncthreads();
    float mean[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) m += red[k][cg * V + v];
        mean[v] = m / (float)T;
    }
    if (!STATS) {
        if (active && g == 0)
#pragma unroll
            for (int v = 0; v < V; ++v) out[(long)b * C + c + v] = mean[v];
        return;
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < V; ++v) s[v] = 0.f;
    if (active)
        for (int t = g; t < T; t += 16) {
            float xv[V];
            if (V == 4) {
                const float4 q = *reinterpret_cast<const float4*>(xp + (long)t * rs);
                xv[0] = q.x; xv[1] = q.y; xv[2] = q.z; xv[3] = q.w;
            } else {
#pragma unroll
                for (int v = 0; v < V; ++v) xv[v] = xp[(long)t * rs + v];
            }
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float d = xv[v] - mean[v];
                s[v] = fmaf(d, d, s[v]);
            }
        }
#pragma unroll
    for (int v = 0; v < V; ++v) red[g][cg * V + v] = s[v];
    __syncthreads();
    if (active && g == 0) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float var = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) var += red[k][cg * V + v];
            var /= (float)T;
            out[(long)b * 2 * C + c + v] = mean[v];
            out[(long)b * 2 * C + C + c + v] = sqrtf(fminf(fmaxf(var, STDDEV_SQRT_MIN_CLIP), FLT_MAX));
        }
    }
}

// Short utterances (T <= TMAX <= 40: the x-vector pools 33 frames): one wave per (utterance, 256 channels), a lane owns 4
// consecutive channels and keeps all T rows of them in registers -- every load is issued before the first use, the second
// pass of the two-pass variance reads registers instead of memory, and there is no LDS or barrier.  Rows t >= T re-read row
// T-1 (no control flow between the loads) and are left out of the sums.  Sums run over t in order (deterministic).
// four consecutive channels as fp32: a 16-byte load of floats, or an 8-byte load of bfloat16 (the bf16 policy's shadow of the
// last frame layer's output: a bf16 value IS the fp32 value with a zero low half)
__device__ __forceinline__ float4 pool_load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 pool_load4(const unsigned short* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}

template <bool STATS, int TMAX, typename XT>
__global__ __launch_bounds__(64) void pool_fwd_reg_kernel(const XT* __restrict__ x, int T, int C, long bs, long rs,
                                                          float* __restrict__ out) {
    const int c = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (c >= C) return;
    const long b = blockIdx.y;
    const XT* xp = x + b * bs + c;
    float4 v[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) v[t] = pool_load4(xp + (long)(t < T ? t : T - 1) * rs);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const float w = t < T ? 1.f : 0.f;
        s.x = fmaf(w, v[t].x, s.x); s.y = fmaf(w, v[t].y, s.y); s.z = fmaf(w, v[t].z, s.z); s.w = fmaf(w, v[t].w, s.w);
    }
    const float invT = 1.f / (float)T;
    const float4 mean = make_float4(s.x / (float)T, s.y / (float)T, s.z / (float)T, s.w / (float)T);
    if (!STATS) {
        *reinterpret_cast<float4*>(out + b * C + c) = mean;
        return;
    }
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const float w = t < T ? 1.f : 0.f;
        const float dx = v[t].x - mean.x, dy = v[t].y - mean.y, dz = v[t].z - mean.z, dw = v[t].w - mean.w;
        q.x = fmaf(w * dx, dx, q.x); q.y = fmaf(w * dy, dy, q.y); q.z = fmaf(w * dz, dz, q.z); q.w = fmaf(w * dw, dw, q.w);
    }
    (void)invT;
    auto sd = [&](float var) { return sqrtf(fminf(fmaxf(var / (float)T, STDDEV_SQRT_MIN_CLIP), FLT_MAX)); };
    *reinterpret_cast<float4*>(out + b * 2 * C + c) = mean;
    *reinterpret_cast<float4*>(out + b * 2 * C + C + c) = make_float4(sd(q.x), sd(q.y), sd(q.z), sd(q.w));
}

template <bool STATS, int TMAX, typename XT>
void launch_pool_fwd_reg(const XT* x, int B, int T, int C, long bs, long rs, float* out, hipStream_t st) {
    hipLaunchKernelGGL((pool_fwd_reg_kernel<STATS, TMAX, XT>), dim3((unsigned)lbx_cdiv(C, 256), (unsigned)B), dim3(64), 0, st, x, T, C,
                       bs, rs, out);
}

// the register kernel for every T it covers (1 .. 40)
template <bool STATS, typename XT>
void launch_pool_fwd_short(const XT* x, int B, int T, int C, long bs, long rs, float* out, hipStream_t st) {
    if (T <= 8) launch_pool_fwd_reg<STATS, 8>(x, B, T, C, bs, rs, out, st);
    else if (T <= 16) launch_pool_fwd_reg<STATS, 16>(x, B, T, C, bs, rs, out, st);
    else if (T <= 24) launch_pool_fwd_reg<STATS, 24>(x, B, T, C, bs, rs, out, st);
    else if (T <= 32) launch_pool_fwd_reg<STATS, 32>(x, B, T, C, bs, rs, out, st);
    else if (T <= 36) launch_pool_fwd_reg<STATS, 36>(x, B, T, C, bs, rs, out, st);
    else launch_pool_fwd_reg<STATS, 40>(x, B, T, C, bs, rs, out, st);
}

template <bool STATS>
void launch_pool_fwd(const float* x, int B, int T, int C, long bs, long rs, float* out, hipStream_t st) {
    const bool vec = C % 4 == 0 && bs % 4 == 0 && rs % 4 == 0 && (((uintptr_t)x) & 15) == 0;
    if (vec && T >= 1 && T <= 40 && (((uintptr_t)out) & 15) == 0) {
        launch_pool_fwd_short<STATS>(x, B, T, C, bs, rs, out, st);
        return;
    }
    if (vec)
        hipLaunchKernelGGL((pool_fwd_kernel<STATS, 4>), dim3((unsigned)lbx_cdiv(C, 64), (unsigned)B), dim3(256), 0, st, x, T,
                           C, bs, rs, out);
    else
        hipLaunchKernelGGL((pool_fwd_kernel<STATS, 1>), dim3((unsigned)lbx_cdiv(C, 16), (unsigned)B), dim3(256), 0, st, x, T,
                           C, bs, rs, out);
}

// grid (ceil(C / (64*V)), B, time splits), one wave per block: a lane owns V consecutive channels of one utterance, loads
// their pooling statistics once and streams its share of the T rows: dx = a + k * (x - mean), a = dmean / T,
// k = 2 dvar / T  (no per-element index division, 16-byte accesses when C, the strides and the bases allow V = 4)
template <bool STATS, int V>
__global__ __launch_bounds__(64) void pool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ pooled,
                                                      const float* __restrict__ dout, int T, int C, long bs, long rs,
                                                      int relu_mask, float* __restrict__ dx, unsigned short* __restrict__ dx16,
                                                      long bs16, long rs16) {
    const int c = (blockIdx.x * 64 + threadIdx.x) * V;
    if (c >= C) return;
    const long b = blockIdx.y;
    const float invT = 1.f / (float)T;
    float a[V], k[V], mean[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        if (STATS) {
            const float sd = pooled[b * 2 * C + C + c + v], dsd = dout[b * 2 * C + C + c + v];
            // clip_by_value passes gradient only inside [1e-10, max]; sd == sqrt(1e-10) <=> clipped
            const float dvar = sd > 1.0000001e-5f ? dsd / (2.f * sd) : 0.f;
            mean[v] = pooled[b * 2 * C + c + v];
            a[v] = dout[b * 2 * C + c + v] * invT;
            k[v] = dvar * 2.f * invT;
        } else {
            mean[v] = 0.f;
            a[v] = dout[b * C + c + v] * invT;
            k[v] = 0.f;
        }
    }
    const float* xp = x + b * bs + c;
    float* dp = dx + b * bs + c;
    for (int t = blockIdx.z; t < T; t += gridDim.z) {
        float xv[V], g[V];
        if (V == 4) {
            const float4 q = *reinterpret_cast<const float4*>(xp + (long)t * rs);
            xv[0] = q.x; xv[1] = q.y; xv[2] = q.z; xv[3] = q.w;
        } else {
#pragma unroll
            for (int v = 0; v < V; ++v) xv[v] = xp[(long)t * rs + v];
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            g[v] = STATS ? fmaf(k[v], xv[v] - mean[v], a[v]) : a[v];
            if (relu_mask && !(xv[v] > 0.f)) g[v] = 0.f;
        }
        if (dx == nullptr) {
            // shadow only
        } else if (V == 4) {
            *reinterpret_cast<float4*>(dp + (long)t * rs) = make_float4(g[0], g[1], g[2], g[3]);
        } else {
#pragma unroll
            for (int v = 0; v < V; ++v) dp[(long)t * rs + v] = g[v];
        }
        if (dx16) {
#pragma unroll
            for (int v = 0; v < V; ++v) dx16[b * bs16 + (long)t * rs16 + c + v] = bf16_bits(g[v]);
        }
    }
}

// 16-byte variant with the loads batched: a wave handles R consecutive rows of (utterance, 256 channels); the statistics come
// in as four float4 and all R row loads are issued before the first store (rows >= T re-read row T-1, only their store is
// predicated), so the wave pays one memory round trip instead of one per row.
template <bool STATS, int R, typename XT>
__global__ __launch_bounds__(64) void pool_bwd_rows_kernel(const XT* __restrict__ x, const float* __restrict__ pooled,
                                                           const float* __restrict__ dout, int T, int C, long bs, long rs,
                                                           int relu_mask, float* __restrict__ dx, unsigned short* __restrict__ dx16,
                                                           long bs16, long rs16) {
    const int c = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (c >= C) return;
    const long b = blockIdx.y;
    const int t0 = blockIdx.z * R;
    const XT* xp = x + b * bs + c;
    float4 v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = pool_load4(xp + (long)(t0 + i < T ? t0 + i : T - 1) * rs);
    const float invT = 1.f / (float)T;
    float a[4], k[4], mean[4];
    if (STATS) {
        const float4 mn = *reinterpret_cast<const float4*>(pooled + b * 2 * C + c);
        const float4 sd = *reinterpret_cast<const float4*>(pooled + b * 2 * C + C + c);
        const float4 dm = *reinterpret_cast<const float4*>(dout + b * 2 * C + c);
        const float4 ds = *reinterpret_cast<const float4*>(dout + b * 2 * C + C + c);
        const float sdv[4] = {sd.x, sd.y, sd.z, sd.w}, dsv[4] = {ds.x, ds.y, ds.z, ds.w};
        const float mnv[4] = {mn.x, mn.y, mn.z, mn.w}, dmv[4] = {dm.x, dm.y, dm.z, dm.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // clip_by_value passes gradient only inside [1e-10, max]; sd == sqrt(1e-10) <=> clipped
            const float dvar = sdv[j] > 1.0000001e-5f ? dsv[j] / (2.f * sdv[j]) : 0.f;
            mean[j] = mnv[j];
            a[j] = dmv[j] * invT;
            k[j] = dvar * 2.f * invT;
        }
    } else {
        const float4 dm = *reinterpret_cast<const float4*>(dout + b * C + c);
        const float dmv[4] = {dm.x, dm.y, dm.z, dm.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { mean[j] = 0.f; a[j] = dmv[j] * invT; k[j] = 0.f; }
    }
    float* dp = dx + b * bs + c;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const float xv[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g[j] = STATS ? fmaf(k[j], xv[j] - mean[j], a[j]) : a[j];
            if (relu_mask && !(xv[j] > 0.f)) g[j] = 0.f;
        }
        if (t0 + i < T) {
            if (dx) *reinterpret_cast<float4*>(dp + (long)(t0 + i) * rs) = make_float4(g[0], g[1], g[2], g[3]);
            if (dx16) {                                          // bf16 shadow (8-byte store: c, the strides and the base are multiples of 4)
                const unsigned lo = bf16_bits(g[0]) | ((unsigned)bf16_bits(g[1]) << 16);
                const unsigned hi = bf16_bits(g[2]) | ((unsigned)bf16_bits(g[3]) << 16);
                *reinterpret_cast<uint2*>(dx16 + b * bs16 + (long)(t0 + i) * rs16 + c) = make_uint2(lo, hi);
            }
        }
    }
}

template <bool STATS>
void launch_pool_bwd(const float* x, const float* pooled, const float* dout, int B, int T, int C, long bs, long rs,
                     int relu_mask, float* dx, hipStream_t st, unsigned short* dx16 = nullptr, long bs16 = 0, long rs16 = 0) {
    const bool vec = C % 4 == 0 && bs % 4 == 0 && rs % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)dx)) & 15) == 0 &&
                     (dx16 == nullptr || (bs16 % 4 == 0 && rs16 % 4 == 0 && (((uintptr_t)dx16) & 7) == 0));
    // short utterances only (the x-vector pools 33 frames): at T = 99 (the CNN) the row loop below measured faster
    if (vec && T >= 1 && T <= 48 && ((((uintptr_t)pooled) | ((uintptr_t)dout)) & 15) == 0) {
        constexpr int R = 12;
        dim3 grid((unsigned)lbx_cdiv(C, 256), (unsigned)B, (unsigned)lbx_cdiv(T, R));
        hipLaunchKernelGGL((pool_bwd_rows_kernel<STATS, R, float>), grid, dim3(64), 0, st, x, pooled, dout, T, C, bs, rs, relu_mask, dx, dx16, bs16, rs16);
        return;
    }
    const int V = vec ? 4 : 1;
    unsigned zs = (unsigned)(T < 8 ? T : 8);                 // time splits: more waves for the small-batch case
    dim3 grid((unsigned)lbx_cdiv(C, 64 * V), (unsigned)B, zs);
    if (vec)
        hipLaunchKernelGGL((pool_bwd_kernel<STATS, 4>), grid, dim3(64), 0, st, x, pooled, dout, T, C, bs, rs, relu_mask, dx, dx16, bs16, rs16);
    else
        hipLaunchKernelGGL((pool_bwd_kernel<STATS, 1>), grid, dim3(64), 0, st, x, pooled, dout, T, C, bs, rs, relu_mask, dx, dx16, bs16, rs16);
}

// one wave per row
__global__ __launch_bounds__(256) void log_softmax_kernel(const float* __restrict__ z, int B, int N,
                                                          float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* zr = z + (long)row * N;
    float m = -FLT_MAX;
    for (int n = lane; n < N; n += 64) m = fmaxf(m, zr[n]);
    m = wave_max(m);
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s += expf(zr[n] - m);
    s = wave_sum(s);
    const float ls = logf(s);
    for (int n = lane; n < N; n += 64) out[(long)row * N + n] = (zr[n] - m) - ls;
}

// single workgroup: mean NLL + gradient wrt the logits feeding log_softmax
__global__ __launch_bounds__(256) void nll_kernel(const float* __restrict__ logp,
                                                  const int32_t* __restrict__ labels, int B, int N,
                                                  float scale, float* __restrict__ loss_out,
                                                  float* __restrict__ dz) {
    __shared__ float red[4];
    float part = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const int y = labels[b];
        // Keras applies softmax cross-entropy to the log-probabilities themselves:
        // -log_softmax(logp)[y]; logsumexp(logp) is 0 up to rounding and is kept for fidelity
        float m = -FLT_MAX;
        for (int n = 0; n < N; ++n) m = fmaxf(m, logp[(long)b * N + n]);
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += expf(logp[(long)b * N + n] - m);
        const float lse = m + logf(s);
        // a label outside [0, N) (TF: an error on CPU, NaN on GPU): NaN loss, zero gradient row, never indexed with
        const bool ok = y >= 0 && y < N;
        part += ok ? lse - logp[(long)b * N + y] : NAN;
        if (dz)
            for (int n = 0; n < N; ++n)
                dz[(long)b * N + n] = ok ? (expf(logp[(long)b * N + n] - lse) - (n == y ? 1.f : 0.f)) * scale : 0.f;
    }
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) loss_out[0] = (red[0] + red[1] + red[2] + red[3]) / (float)B;
}

// tf.nn.softmax over rows (cnn.py:43-44 with output_activation="softmax"), one wave per row
__global__ __launch_bounds__(256) void softmax_kernel(const float* __restrict__ z, int B, int N, float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* zr = z + (long)row * N;
    float m = -FLT_MAX;
    for (int n = lane; n < N; n += 64) m = fmaxf(m, zr[n]);
    m = wave_max(m);
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s += expf(zr[n] - m);
    s = wave_sum(s);
    for (int n = lane; n < N; n += 64) out[(long)row * N + n] = expf(zr[n] - m) / s;
}

// Keras SparseCategoricalCrossentropy(from_logits=False) on softmax outputs (keras_utils.py:141-142 with a model built with
// output_activation="softmax"): p = softmax(z); q = clip(p, 1e-7, 1 - 1e-7); the backend then takes
// sparse_softmax_cross_entropy_with_logits(labels, log q), i.e. loss = log(sum_j q_j) - log q_y (the sum is 1 up to the
// clipping); dz = the gradient with respect to the logits z through clip and softmax (a clipped probability passes no
// gradient).  Single workgroup like nll_kernel; probs (may be NULL) receives p.
constexpr float KERAS_EPSILON = 1e-7f;
__global__ __launch_bounds__(256) void softmax_nll_kernel(const float* __restrict__ z, const int32_t* __restrict__ labels, int B, int N,
                                                          float scale, float* __restrict__ probs, float* __restrict__ loss_out,
                                                          float* __restrict__ dz) {
    __shared__ float red[4];
    float part = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* zr = z + (long)b * N;
        const int y = labels[b];
        const bool ok = y >= 0 && y < N;
        float m = -FLT_MAX;
        for (int n = 0; n < N; ++n) m = fmaxf(m, zr[n]);
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += expf(zr[n] - m);
        float sq = 0.f, qy = 1.f, dot = 0.f;                        // sum of clipped probabilities, q_y, sum_j p_j g_j
        for (int n = 0; n < N; ++n) {
            const float p = expf(zr[n] - m) / s;
            if (probs) probs[(long)b * N + n] = p;
            sq += fminf(fmaxf(p, KERAS_EPSILON), 1.f - KERAS_EPSILON);
        }
        for (int n = 0; n < N; ++n) {
            const float p = expf(zr[n] - m) / s;
            const float q = fminf(fmaxf(p, KERAS_EPSILON), 1.f - KERAS_EPSILON);
            const bool open = p > KERAS_EPSILON && p < 1.f - KERAS_EPSILON;        // clip_by_value passes the gradient inside only
            if (n == y) qy = q;
            const float g = open ? (1.f / sq - (n == y ? 1.f / q : 0.f)) : 0.f;
            dot += p * g;
        }
        part += ok ? logf(sq) - logf(qy) : NAN;
        if (dz)
            for (int n = 0; n < N; ++n) {
                const float p = expf(zr[n] - m) / s;
                const float q = fminf(fmaxf(p, KERAS_EPSILON), 1.f - KERAS_EPSILON);
                const bool open = p > KERAS_EPSILON && p < 1.f - KERAS_EPSILON;
                const float g = open ? (1.f / sq - (n == y ? 1.f / q : 0.f)) : 0.f;
                dz[(long)b * N + n] = ok ? p * (g - dot) * scale : 0.f;
            }
    }
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) loss_out[0] = (red[0] + red[1] + red[2] + red[3]) / (float)B;
}

// ------------------------------------------------------------------------------------------------
// Output layer + loss of a classifier in TWO small launches (train step only) instead of nine: z = h W + b (Dense(N),
// xvector.py:62-64), logp = log_softmax(z) (xvector.py:65), Keras SparseCategoricalCrossentropy(from_logits=True) on logp
// (keras_utils.py:141-147) and the whole backward of those three: dW = h^T dz, db = sum dz, dh = (dz W^T) (* (h > 0)).
// For few classes (N <= 32) the launches this replaces (GEMM + split reduce, log-softmax, NLL, wgrad + reduce, dgrad +
// reduce) are latency, not work: 0.5 MFLOP per utterance row.
//   rows kernel : a wave shares a row's dot products (fp32 FMA, fixed shuffle butterfly); the row-local softmax, loss,
//                 dz and dh follow in registers; dz [B,N] and the per-row losses go to the workspace.
//   wgrad kernel: a workgroup owns 16 consecutive k of one class n; 16 lanes per output each add the rows r = s, s+16, ...
//                 and one lane adds the 16 partial sums in lane order; the last workgroup does db and the mean loss
//                 (a wave per quantity, fixed butterfly).
// No atomics, fixed summation orders: deterministic.
// ------------------------------------------------------------------------------------------------
// row k of W [K][N] into NP registers (zero beyond N / outside K)
template <int NP>
__device__ __forceinline__ void load_w_row(float (&w)[NP], const float* __restrict__ W, int k, int N, bool in, bool vec) {
    if (vec) {
#pragma unroll
        for (int q = 0; q < NP / 4; ++q) {
            const float4 v = in ? *reinterpret_cast<const float4*>(W + (long)k * NP + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int n = 0; n < NP; ++n) w[n] = (in && n < N) ? W[(long)k * N + n] : 0.f;
    }
}

template <int NP>
__global__ __launch_bounds__(256) void softmax_head_rows_kernel(const float* __restrict__ h, const float* __restrict__ W,
                                                                const float* __restrict__ bias, const int32_t* __restrict__ labels,
                                                                int B, int K, int N, float scale, int relu_mask,
                                                                float* __restrict__ logp, float* __restrict__ dh,
                                                                float* __restrict__ dzbuf, float* __restrict__ lossrow) {
    // one wave per row, four rows per workgroup; lane l owns k = l, l + 64, ...  Loads are issued U k-values at a time
    // before their FMAs: written one k per iteration the loop paid a memory round trip per k (35 us for K = 512).
    constexpr int U = NP <= 4 ? 8 : (NP <= 8 ? 4 : (NP <= 16 ? 2 : 1));
    const bool wvec = N == NP && ((((uintptr_t)W) & 15) == 0);       // rows of W are NP aligned floats: 16-byte loads
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B) return;                                           // wave-uniform
    const float* hr = h + r * K;
    const int y = labels[r];                                      // issued with the first batch of loads, used after the reduce
    float bv[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) bv[n] = n < N ? bias[n] : 0.f;
    float z[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) z[n] = 0.f;
    float hv[U], wv[U][NP];                                       // the last trip's values stay live for the dh pass
    for (int k0 = lane; k0 < K; k0 += 64 * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 64 * u;
            const bool in = k < K;
            hv[u] = in ? hr[k] : 0.f;
            load_w_row<NP>(wv[u], W, k, N, in, wvec);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int n = 0; n < NP; ++n) z[n] = fmaf(hv[u], wv[u][n], z[n]);
    }
#pragma unroll
    for (int n = 0; n < NP; ++n) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) z[n] += __shfl_xor(z[n], o, 64);     // fixed butterfly over the row's wave
    }
    // every lane now holds the row's logits
    float m = -FLT_MAX;
#pragma unroll
    for (int n = 0; n < NP; ++n)
        if (n < N) { z[n] += bv[n]; m = fmaxf(m, z[n]); }
    float se = 0.f;
#pragma unroll
    for (int n = 0; n < NP; ++n)
        if (n < N) se += expf(z[n] - m);
    const float lse = m + logf(se);
    // Keras applies softmax cross-entropy to the log-probabilities themselves (nll_kernel): logsumexp(logp) is 0 up to rounding
    float m2 = -FLT_MAX;
#pragma unroll
    for (int n = 0; n < NP; ++n)
        if (n < N) { z[n] -= lse; m2 = fmaxf(m2, z[n]); }          // z = logp from here on
    float s2 = 0.f;
#pragma unroll
    for (int n = 0; n < NP; ++n)
        if (n < N) s2 += expf(z[n] - m2);
    const float lse2 = m2 + logf(s2);
    const bool ok = y >= 0 && y < N;                              // outside [0, N): NaN loss, zero gradient row (nll_kernel)
    float lp_y = 0.f;
    float dz[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        dz[n] = 0.f;
        if (n < N) {
            if (n == y) lp_y = z[n];
            if (ok) dz[n] = (expf(z[n] - lse2) - (n == y ? 1.f : 0.f)) * scale;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int n = 0; n < NP; ++n)
            if (n < N) { logp[r * N + n] = z[n]; dzbuf[r * N + n] = dz[n]; }
        lossrow[r] = ok ? lse2 - lp_y : NAN;
    }
    if (dh && lane < K) {
        auto emit = [&](int k0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + 64 * u;
                float acc = 0.f;
#pragma unroll
                for (int n = 0; n < NP; ++n) acc = fmaf(dz[n], wv[u][n], acc);
                if (relu_mask && !(hv[u] > 0.f)) acc = 0.f;
                if (k < K) dh[r * K + k] = acc;
            }
        };
        // this lane's last trip first -- the registers still hold it -- then the earlier ones, loaded again
        const int klast = lane + ((K - 1 - lane) / (64 * U)) * (64 * U);
        emit(klast);
        for (int k0 = lane; k0 < klast; k0 += 64 * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + 64 * u;                         // < K: a full trip
                hv[u] = hr[k];
                load_w_row<NP>(wv[u], W, k, N, true, wvec);
            }
            emit(k0);
        }
    }
}

__global__ __launch_bounds__(256) void softmax_head_wgrad_kernel(const float* __restrict__ h, const float* __restrict__ dzbuf,
                                                                 const float* __restrict__ lossrow, int B, int K, int N,
                                                                 float* __restrict__ dW, float* __restrict__ db,
                                                                 float* __restrict__ loss_out) {
    __shared__ float red[16][17];
    const int tid = threadIdx.x;
    const int kblocks = (K + 15) / 16;
    if ((int)blockIdx.x == kblocks * N) {
        // db[n] and the mean loss: wave w takes the quantities w, w + 4, ...; lane l adds rows l, l + 64, ..., then a fixed
        // butterfly adds the 64 lanes
        const int lane = tid & 63;
        for (int q = tid >> 6; q <= N; q += 4) {
            float acc = 0.f;
            for (int r = lane; r < B; r += 64) acc += q < N ? dzbuf[(long)r * N + q] : lossrow[r];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
            if (lane == 0) {
                if (q < N) db[q] = acc;
                else loss_out[0] = acc / (float)B;
            }
        }
        return;
    }
    const int n = blockIdx.x / kblocks, kl = tid & 15, sub = tid >> 4;
    const int k = (blockIdx.x - n * kblocks) * 16 + kl;
    const int kc = k < K ? k : 0;
    float acc = 0.f;
    for (int r0 = sub; r0 < B; r0 += 16 * 8) {                    // eight rows' loads in flight
        float hv[8], dv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + 16 * u;
            const bool in = r < B;
            hv[u] = in ? h[(long)r * K + kc] : 0.f;
            dv[u] = in ? dzbuf[(long)r * N + n] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(hv[u], dv[u], acc);
    }
    red[kl][sub] = acc;
    __syncthreads();
    if (sub == 0 && k < K) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) s += red[kl][t];
        dW[(long)k * N + n] = s;
    }
}

constexpr float L2_EPS = 1e-12f;   // tf.math.l2_normalize default epsilon

__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, int B, int D,
                                                         float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* xr = x + (long)row * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s = fmaf(xr[d], xr[d], s);
    s = wave_sum(s);
    const float inv = rsqrtf(fmaxf(s, L2_EPS));
    for (int d = lane; d < D; d += 64) out[(long)row * D + d] = xr[d] * inv;
}

__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ dout, int B, int D,
                                                         float* __restrict__ dx) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* xr = x + (long)row * D;
    const float* gr = dout + (long)row * D;
    float s = 0.f, dot = 0.f;
    for (int d = lane; d < D; d += 64) {
        s = fmaf(xr[d], xr[d], s);
        dot = fmaf(xr[d], gr[d], dot);
    }
    s = wave_sum(s);
    dot = wave_sum(dot);
    const bool clipped = s < L2_EPS;
    const float inv = rsqrtf(fmaxf(s, L2_EPS));
    const float k = clipped ? 0.f : dot * inv * inv * inv;   // y (y . g) / |x|
    for (int d = lane; d < D; d += 64) dx[(long)row * D + d] = gr[d] * inv - xr[d] * k;
}

constexpr float AP_ACOS_CLAMP = 1e-6f;   // floor of 1 - x^2 in d acos/dx (TF would return inf at |x| = 1)

// one wave per example
__global__ __launch_bounds__(256) void ap_loss_kernel(const float* __restrict__ z,
                                                      const int32_t* __restrict__ labels, int B, int D,
                                                      int N, float delta, float scale,
                                                      float* __restrict__ loss, float* __restrict__ dz) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* zr = z + (long)row * D;
    const int y = labels[row];
    if (y < 0 || y >= N) {                                 // label outside the N language vectors: NaN loss, zero gradient row
        if (lane == 0) loss[row] = NAN;
        if (dz)
            for (int d = lane; d < D; d += 64) dz[(long)row * D + d] = 0.f;
        return;
    }
    const float th_y = acosf(zr[y]);                       // losses.py:31-33 (theta_l)
    float L = 0.f, dsum = 0.f;
    for (int n = lane; n < N; n += 64) {
        if (n == y) continue;
        const float th = acosf(zr[n]);
        const float s = 1.f / (1.f + expf(-delta * (th_y - th)));      // losses.py:35-36
        L += s;
        const float ds = delta * s * (1.f - s);
        dsum += ds;
        if (dz) {
            const float x = zr[n];
            dz[(long)row * D + n] = ds * rsqrtf(fmaxf(1.f - x * x, AP_ACOS_CLAMP)) * scale;   // -ds * acos'
        }
    }
    L = wave_sum(L);
    dsum = wave_sum(dsum);
    if (lane == 0) {
        loss[row] = L;
        if (dz) {
            const float x = zr[y];
            dz[(long)row * D + y] = -dsum * rsqrtf(fmaxf(1.f - x * x, AP_ACOS_CLAMP)) * scale;
        }
    }
    if (dz)
        for (int d = N + lane; d < D; d += 64) dz[(long)row * D + d] = 0.f;
}

// The angular-proximity head of a train step in ONE launch, one wave per example (D <= AP_HEAD_MAX_D): L2 normalisation
// (tf.math.l2_normalize, what ap_lstm.py:42 puts in front of the loss), SparseAngularProximity.call (losses.py:25-40) with its
// gradient, the gradient through the normalisation, and the scores predict() hands to the metrics (losses.py:51-52).  Every
// value is computed by the expressions, in the order, of l2norm_fwd_kernel -> ap_loss_kernel -> l2norm_bwd_kernel ->
// neg_acos_kernel (the separate entry points stay: the API leaves use them) -- bit-identical results, four launches fewer.
constexpr int AP_HEAD_MAX_D = 4096;
__global__ __launch_bounds__(256) void ap_head_kernel(const float* __restrict__ x, const int32_t* __restrict__ labels, int B, int D, int N,
                                                      float delta, float scale, float* __restrict__ zn_out, float* __restrict__ loss,
                                                      float* __restrict__ dx, float* __restrict__ scores) {
    extern __shared__ float s_ap[];               // [4 waves][2][D]: the wave's normalised row and its gradient
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    float* zr = s_ap + (threadIdx.x >> 6) * 2 * D;
    float* gz = zr + D;
    const float* xr = x + (long)row * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s = fmaf(xr[d], xr[d], s);
    s = wave_sum(s);
    const float inv = rsqrtf(fmaxf(s, L2_EPS));
    for (int d = lane; d < D; d += 64) {
        const float z = xr[d] * inv;
        zr[d] = z;
        gz[d] = 0.f;
        if (zn_out) zn_out[(long)row * D + d] = z;
    }
    wave_lds_sync();
    if (scores)
        for (int n = lane; n < N; n += 64) scores[(long)row * N + n] = -acosf(zr[n]);
    const int y = labels[row];
    if (y < 0 || y >= N) {                                 // label outside the N language vectors: NaN loss, zero gradient row
        if (lane == 0) loss[row] = NAN;
    } else {
        const float th_y = acosf(zr[y]);
        float L = 0.f, dsum = 0.f;
        for (int n = lane; n < N; n += 64) {
            if (n == y) continue;
            const float zv = zr[n];
            const float th = acosf(zv);
            const float sg = 1.f / (1.f + expf(-delta * (th_y - th)));
            L += sg;
            const float ds = delta * sg * (1.f - sg);
            dsum += ds;
            gz[n] = ds * rsqrtf(fmaxf(1.f - zv * zv, AP_ACOS_CLAMP)) * scale;
        }
        L = wave_sum(L);
        dsum = wave_sum(dsum);
        if (lane == 0) {
            loss[row] = L;
            const float zv = zr[y];
            gz[y] = -dsum * rsqrtf(fmaxf(1.f - zv * zv, AP_ACOS_CLAMP)) * scale;
        }
    }
    wave_lds_sync();
    if (dx) {
        float dot = 0.f;
        for (int d = lane; d < D; d += 64) dot = fmaf(xr[d], gz[d], dot);
        dot = wave_sum(dot);
        const bool clipped = s < L2_EPS;
        const float k = clipped ? 0.f : dot * inv * inv * inv;
        for (int d = lane; d < D; d += 64) dx[(long)row * D + d] = gz[d] * inv - xr[d] * k;
    }
}

// C_avg counters (reference metrics.py:51-71).  grid (N scored classes m, ceil(Th / 64), label chunks); 1 024 threads = 16 waves x 64
// thresholds.  LDS pos[l][th] counts the examples with label l whose score for class m is >= threshold.  The batch is cut into units of
// 16 ... 64 examples; wave w takes units w, w + 16, ...: it loads a unit's scores and labels with ONE load per lane and walks them through
// v_readlane (no memory round trip inside the walk -- round 4's kernel made one dependent load per example: 203 us at 512 x 100 x 100),
// counting with ds_add_u32 (lane = threshold = column: conflict-free inside a wave; integer counters: exact in any order, so all waves
// share ONE copy).  An iteration of the walk is a chain of ~11 dependent scalar / vector instructions, ~90 cycles for a wave that is
// alone on its SIMD: round 5's four waves with 128 iterations each spent 5.5 us of the kernel's 13 there at 512 examples; sixteen waves
// with 32 iterations each overlap four chains per SIMD (round 6).  The workgroup is the only writer of cells [chunk, m, th-tile]: no
// global atomics, exact results.
constexpr int CAVG_WAVES = 16, CAVG_BATCH = 8, CAVG_LCH = CAVG_WAVES * CAVG_BATCH;        // 128 labels per chunk
__global__ __launch_bounds__(1024) void cavg_update_kernel(const float* __restrict__ scores,
                                                           const int32_t* __restrict__ labels, int B, int N,
                                                           const float* __restrict__ thresholds, int Th,
                                                           int lch, int unit, float* __restrict__ tp,
                                                           float* __restrict__ fn, float* __restrict__ fp,
                                                           float* __restrict__ tn) {
    extern __shared__ unsigned s_cnt[];       // pos[lch][64] then cnt[lch]: integer counters (ds_add_u32)
    unsigned* pos = s_cnt;
    unsigned* cnt = s_cnt + lch * 64;
    const int m = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int th = blockIdx.y * 64 + lane;
    const float thr = th < Th ? thresholds[th] : 0.f;
    // cell (label l0 + l, class m, threshold th) of the positive / negative counter pairs: (tp, fn) on the diagonal, (fp, tn) off it
    auto cells = [&](int lab, float*& pa, float*& pb) {
        const bool diag = lab == m;
        const long cell = diag ? (long)m * Th + th : ((long)lab * N + m) * Th + th;
        pa = (diag ? tp : fp) + cell;                                          // metrics.py:63-66 / :68-71
        pb = (diag ? fn : tn) + cell;
    };
    for (int l0 = blockIdx.z * lch; l0 < N; l0 += gridDim.z * lch) {          // label chunks: across grid.z first, then in turn
        const int nl = min(lch, N - l0);
        // the counters this thread will update (its wave's labels wv, wv + 16, ...): every load goes out NOW, ahead of the LDS
        // clear and the walk -- the write-out at the end then adds and stores without a round trip to HBM of its own
        float va[CAVG_BATCH], vb[CAVG_BATCH];
#pragma unroll
        for (int u = 0; u < CAVG_BATCH; ++u) {
            const int l = wv + u * CAVG_WAVES;
            va[u] = vb[u] = 0.f;
            if (th < Th && l < nl) {
                float *pa, *pb;
                cells(l0 + l, pa, pb);
                va[u] = *pa;
                vb[u] = *pb;
            }
        }
        // first unit of this wave: its scores / labels are in flight during the clear as well
        int b0 = wv * unit;
        bool in = lane < unit && b0 + lane < B;
        float sv = in ? scores[(long)(b0 + lane) * N + m] : 0.f;
        int yv = in ? labels[b0 + lane] - l0 : -1;
        {
            uint4* z = reinterpret_cast<uint4*>(pos);
            for (int i = threadIdx.x; i < lch * 16; i += 1024) z[i] = make_uint4(0u, 0u, 0u, 0u);
            for (int l = threadIdx.x; l < nl; l += 1024) cnt[l] = 0u;
        }
        __syncthreads();
        unsigned* mine = pos + lane;
        for (; b0 < B; b0 += CAVG_WAVES * unit) {
            const int b1 = b0 + CAVG_WAVES * unit;                     // the wave's next unit: loaded before this one is walked
            in = lane < unit && b1 + lane < B;
            const float sv1 = in ? scores[(long)(b1 + lane) * N + m] : 0.f;
            const int yv1 = in ? labels[b1 + lane] - l0 : -1;
            if (yv < 0 || yv >= nl) yv = -1;
            if (yv >= 0) atomicAdd(&cnt[yv], 1u);                     // examples per label (all thresholds share it)
            // branch-free walk: an example that does not count (label outside the chunk, score below the threshold, lane past the
            // batch: yv = -1) adds 0 to row 0 -- with branches an iteration cost ~145 cycles of exec-mask and scalar-branch traffic
#pragma unroll 8
            for (int j = 0; j < unit; ++j) {
                const int y = __builtin_amdgcn_readlane(yv, j);       // wave-uniform
                const float sj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sv), j));
                const unsigned one = (y >= 0 && sj >= thr) ? 1u : 0u;  // metrics.py:60
                atomicAdd(mine + max(y, 0) * 64, one);
            }
            sv = sv1;
            yv = yv1;
        }
        __syncthreads();
        if (th < Th) {
            // every sum first, then every store: loads and stores retire through ONE in-order counter (vmcnt) on this chip, so a
            // store issued between two uses of prefetched values makes the second use wait for the store's round trip (the
            // first version of this loop: 25 dependent store round trips, 23 us for the kernel)
#pragma unroll
            for (int u = 0; u < CAVG_BATCH; ++u) {
                const int l = wv + u * CAVG_WAVES;
                if (l < nl) {
                    const unsigned pu = pos[l * 64 + lane];
                    va[u] += (float)pu;
                    vb[u] += (float)(cnt[l] - pu);                                     // s < thr  (:61)
                }
            }
#pragma unroll
            for (int u = 0; u < CAVG_BATCH; ++u) {
                const int l = wv + u * CAVG_WAVES;
                if (l < nl) {
                    float *pa, *pb;
                    cells(l0 + l, pa, pb);
                    *pa = va[u];
                    *pb = vb[u];
                }
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ float div_no_nan(float a, float b) { return b != 0.f ? a / b : 0.f; }

// One workgroup (the result is a min over thresholds of sums over languages: a few million divisions, once per evaluation).
// G = 1024 / Th groups of Th threads: group g computes the inner false-alarm sums of languages g, g + G, ... into LDS
// (`lds_pf` = N x Th floats; 0 when that does not fit: every thread then walks all languages itself), then thread th adds them
// up in language order -- the same order, hence the same bits, with or without the LDS pass.
__global__ __launch_bounds__(1024) void cavg_result_kernel(const float* __restrict__ tp,
                                                           const float* __restrict__ fn,
                                                           const float* __restrict__ fp,
                                                           const float* __restrict__ tn, int N, int Th,
                                                           float C_miss, float C_fa, float P_tar, int lds_pf,
                                                           float* __restrict__ c_avg_out,
                                                           float* __restrict__ out) {
    extern __shared__ float pf_l[];
    __shared__ float red[16];
    auto inner_of = [&](int l, int th) {
        float inner = 0.f;
        for (int m = 0; m < N; ++m) {
            const float f = fp[((long)l * N + m) * Th + th], t = tn[((long)l * N + m) * Th + th];
            inner += div_no_nan(f, f + t);                                      // metrics.py:89-95
        }
        return div_no_nan(inner, (float)(N - 1));
    };
    if (lds_pf) {
        const int G = 1024 / Th, g = threadIdx.x / Th, th = threadIdx.x % Th;
        if (g < G)
            for (int l = g; l < N; l += G) pf_l[l * Th + th] = inner_of(l, th);
        __syncthreads();
    }
    float best = FLT_MAX;
    for (int th = threadIdx.x; th < Th; th += 1024) {
        float pm = 0.f, pf = 0.f;
        for (int l = 0; l < N; ++l) {
            const float a = fn[(long)l * Th + th], b = tp[(long)l * Th + th];
            pm += div_no_nan(a, a + b);                                         // metrics.py:80-85
            pf += lds_pf ? pf_l[l * Th + th] : inner_of(l, th);
        }
        pm /= (float)N;
        pf /= (float)N;
        const float c = C_miss * P_tar * pm + C_fa * (1.f - P_tar) * pf;        // :98
        if (c_avg_out) c_avg_out[th] = c;
        best = fminf(best, c);
    }
    best = wave_min(best);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = red[0];
        for (int w = 1; w < 16; ++w) m = fminf(m, red[w]);
        out[0] = m;                                                             // :103
    }
}

struct AdamState {
    long long step;
    float lr_t;
    float lr_now;            // any non-zero BIT PATTERN: |lr_now| is the learning rate of this step (a schedule, written by the host
                             // before the step: a captured graph replays with it; a scheduled rate of exactly 0 is written as -0.0f);
                             // all bits zero (a freshly zeroed state): no schedule, the `lr` argument
};

// one thread: advance the step and publish the bias-corrected learning rate for this step.
// (Round 4 folded this launch into adam_kernel -- every workgroup derives lr_t itself, the last one to finish, found by an
// arrival ticket in the state, publishes t -- and measured adam_kernel at 35.8 us instead of 20.5 + 4.9: 2 048 workgroups
// arriving on ONE counter serialise at ~12 ns per atomic (MI355X_MICROARCH.md "fanin").  Not kept.)
__global__ void adam_prepare_kernel(AdamState* st, float lr, float b1, float b2) {
    const long long t = ++st->step;
    const double c = sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t));
    const float base = __float_as_uint(st->lr_now) != 0u ? fabsf(st->lr_now) : lr;
    st->lr_t = (float)((double)base * c);
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, const AdamState* __restrict__ st, float b1,
                            float b2, float eps, float gscale) {
    const float lr_t = st->lr_t;
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
#define LBX_ADAM1(c)                                              \
    {                                                             \
        const float gr = gg.c * gscale;                           \
        mm.c = b1 * mm.c + (1.f - b1) * gr;                       \
        vv.c = b2 * vv.c + (1.f - b2) * gr * gr;                  \
        pp.c = pp.c - lr_t * mm.c / (sqrtf(vv.c) + eps);          \
    }
        LBX_ADAM1(x) LBX_ADAM1(y) LBX_ADAM1(z) LBX_ADAM1(w)
#undef LBX_ADAM1
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // tail
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gr = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gr;
        const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
    }
}

// Optimizers other than Adam that a config may name (keras_utils.py:137-140: getattr(tf.keras.optimizers, cls)(**kwargs)).
// The step counter and the rate of the step live in the same 16-byte device state as Adam's {int64 step, float lr_t, float
// lr_now}: lr_t = the scheduled rate (lr_now, when a schedule wrote one) or lr, no bias correction.
__global__ void opt_prepare_kernel(AdamState* st, float lr) {
    ++st->step;
    st->lr_t = __float_as_uint(st->lr_now) != 0u ? fabsf(st->lr_now) : lr;
}

// tf.keras.optimizers.SGD: momentum == 0: w -= lr g;  else v = momentum v - lr g;  w += nesterov ? momentum v - lr g : v
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ vel, long n,
                           const AdamState* __restrict__ st, float momentum, int nesterov, float gscale) {
    const float lr = st->lr_t;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gr = g[i] * gscale;
        if (vel) {
            const float v = momentum * vel[i] - lr * gr;
            vel[i] = v;
            p[i] += nesterov ? momentum * v - lr * gr : v;
        } else {
            p[i] -= lr * gr;
        }
    }
}

// tf.keras.optimizers.RMSprop (TF 2.3 rmsprop.py): rms = rho rms + (1 - rho) g^2; centered: mg = rho mg + (1 - rho) g,
// denom = rms - mg^2.  momentum == 0: w -= lr g / (sqrt(denom) + eps);  momentum > 0 (the fused training op): mom = momentum
// mom + lr g / sqrt(denom + eps); w -= mom -- epsilon sits inside the root there, as in TensorFlow's kernels.
__global__ void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ rms, float* __restrict__ mg,
                               float* __restrict__ mom, long n, const AdamState* __restrict__ st, float rho, float momentum, float eps,
                               float gscale) {
    const float lr = st->lr_t;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gr = g[i] * gscale;
        const float r = rho * rms[i] + (1.f - rho) * gr * gr;
        rms[i] = r;
        float denom = r;
        if (mg) {
            const float a = rho * mg[i] + (1.f - rho) * gr;
            mg[i] = a;
            denom = r - a * a;
        }
        if (mom) {
            const float m = momentum * mom[i] + lr * gr / sqrtf(denom + eps);
            mom[i] = m;
            p[i] -= m;
        } else {
            p[i] -= lr * gr / (sqrtf(denom) + eps);
        }
    }
}

// mean of n floats, one workgroup (fixed summation order: deterministic)
__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) / (float)n;
}

// out[b, n] = -acos(z[b, n]), n < N of the D columns (the scores SparseAngularProximity.predict hands to the metrics,
// losses.py:44-47)
__global__ void neg_acos_kernel(const float* __restrict__ z, long B, int D, int N, float* __restrict__ out) {
    const long total = B * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / N;
        const int n = (int)(i - b * N);
        out[i] = -acosf(z[b * D + n]);
    }
}

// counter-based uniform in [0, 1): splitmix64 finaliser over (seed, step, utterance, channel)
__device__ __forceinline__ float hash_uniform(unsigned long long seed, unsigned long long step, unsigned b, unsigned c) {
    unsigned long long v = seed + 0x9E3779B97F4A7C15ull * (step + 1) + (((unsigned long long)b << 32) | c);
    v ^= v >> 30; v *= 0xBF58476D1CE4E5B9ull;
    v ^= v >> 27; v *= 0x94D049BB133111EBull;
    v ^= v >> 31;
    return (float)(v >> 40) * (1.0f / 16777216.0f);
}

// Keras SpatialDropout1D on x [B, T, C] (row stride C, batch stride bs) in place: a channel of an utterance is zeroed
// with probability `rate` for all T frames, the kept ones are scaled by 1 / (1 - rate).  The draw is a function of
// (seed, *step, b, c); `step` is a device counter (the Adam step), so a replayed hipGraph draws a fresh mask every step.
__global__ __launch_bounds__(256) void spatial_dropout_kernel(float* __restrict__ x, int T, int C, long bs, float rate,
                                                              unsigned long long seed, const long long* __restrict__ step,
                                                              float* __restrict__ mask_out) {
    const int b = blockIdx.y;
    const unsigned long long stp = step ? (unsigned long long)*step : 0ull;
    const float keep_scale = 1.f / (1.f - rate);
    float* xb = x + (long)b * bs;
    const long total = (long)T * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float m = hash_uniform(seed, stp, (unsigned)b, (unsigned)c) >= rate ? keep_scale : 0.f;
        xb[i] *= m;
        if (mask_out && i < C) mask_out[(long)b * C + c] = m;
    }
}

struct RowsOutD {
    float* base;
    long bs, rs;
    int batch, rpb;
};
__device__ __forceinline__ long row_offset(const RowsOutD& r, unsigned m) {
    if (r.batch == 1) return (long)m * r.rs;
    const unsigned b = m / (unsigned)r.rpb;
    return (long)b * r.bs + (long)(m - b * (unsigned)r.rpb) * r.rs;
}

// Keras Dropout (element-wise) in place on rows x[r][0..C) addressed through an implicit-row descriptor: an element is zeroed
// with probability `rate`, the kept ones scaled by 1 / (1 - rate).  The draw is a function of (seed, *step, r, c) only, so the
// SAME call on the gradient regenerates the forward mask (FrameLayer2D's dropout, xvector_2d.py:37-46: y = dropout(bn(x))
// forward, d(bn out) = mask * d(y) backward).
__global__ __launch_bounds__(256) void dropout_rows_kernel(RowsOutD X, long R, int C, float rate, unsigned long long seed,
                                                           const long long* __restrict__ step) {
    const unsigned long long stp = step ? (unsigned long long)*step : 0ull;
    const float keep_scale = 1.f / (1.f - rate);
    const long total = R * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        float* p = X.base + row_offset(X, (unsigned)r) + c;
        *p *= hash_uniform(seed, stp, (unsigned)r, (unsigned)c) >= rate ? keep_scale : 0.f;
    }
}

__global__ void scale_kernel(float* __restrict__ x, long n, float alpha) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= alpha;
}

__global__ void fill_kernel(float* __restrict__ x, long n, float value) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        x[i] = value;
}

inline unsigned ew_grid(long n) {
    long g = lbx_cdiv(n, 256);
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int lidbox_stats_pool_fwd(const float* x, int B, int T, int C, long bs, long rs, float* out,
                                     lidbox_stream_t stream) {
    LBX_ARG(x && out && B >= 0 && T >= 1 && C >= 1, "x, out != NULL; T, C >= 1");
    if (B == 0) return LIDBOX_OK;
    LBX_ARG(B <= 65535, "B <= 65535");
    launch_pool_fwd<true>(x, B, T, C, bs, rs, out, (hipStream_t)stream);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_avg_pool_fwd(const float* x, int B, int T, int C, long bs, long rs, float* out,
                                   lidbox_stream_t stream) {
    LBX_ARG(x && out && B >= 0 && T >= 1 && C >= 1, "x, out != NULL; T, C >= 1");
    if (B == 0) return LIDBOX_OK;
    LBX_ARG(B <= 65535, "B <= 65535");
    launch_pool_fwd<false>(x, B, T, C, bs, rs, out, (hipStream_t)stream);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_stats_pool_bwd(const float* x, const float* pooled, const float* dout, int B, int T,
                                     int C, long bs, long rs, int relu_mask, float* dx,
                                     lidbox_stream_t stream) {
    LBX_ARG(x && pooled && dout && dx && T >= 1 && C >= 1, "pointers != NULL; T, C >= 1");
    const long total = (long)B * T * C;
    if (total == 0) return LIDBOX_OK;
    LBX_ARG(B <= 65535, "B <= 65535");
    launch_pool_bwd<true>(x, pooled, dout, B, T, C, bs, rs, relu_mask, dx, (hipStream_t)stream);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_stats_pool_bwd_shadow(const float* x, const float* pooled, const float* dout, int B, int T, int C, long bs,
                                            long rs, int relu_mask, float* dx, void* dx16, long bs16, long rs16,
                                            lidbox_stream_t stream) {
    LBX_ARG(x && pooled && dout && dx16 && T >= 1 && C >= 1, "x, pooled, dout, dx16 != NULL (dx may be NULL: shadow only); T, C >= 1");
    LBX_ARG(rs16 >= C && (B <= 1 || bs16 >= (long)(T - 1) * rs16 + C) && (((uintptr_t)dx16) & 1) == 0, "shadow strides cover the rows");
    const long total = (long)B * T * C;
    if (total == 0) return LIDBOX_OK;
    LBX_ARG(B <= 65535, "B <= 65535");
    launch_pool_bwd<true>(x, pooled, dout, B, T, C, bs, rs, relu_mask, dx, (hipStream_t)stream, (unsigned short*)dx16, bs16, rs16);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

// The two pooling passes over the bf16 shadow of the last frame layer's output (the bf16 policy's all-shadow mode: that layer then
// writes no fp32 copy at all).  Same kernels, same fp32 arithmetic, on the shadow's values; short utterances only (the register
// kernels: T <= 40), channel counts and strides that are multiples of 4, 8-byte aligned shadows.
extern "C" int lidbox_stats_pool_fwd_bf16(const void* x16, int B, int T, int C, long bs, long rs, float* out, lidbox_stream_t stream) {
    LBX_ARG(x16 && out && B >= 0 && T >= 1 && C >= 1, "x16, out != NULL; T, C >= 1");
    LBX_ARG(T <= 40 && C % 4 == 0 && bs % 4 == 0 && rs % 4 == 0 && rs >= C && (((uintptr_t)x16) & 7) == 0 && (((uintptr_t)out) & 15) == 0,
            "the bf16 pooling kernels take T <= 40, C and strides that are multiples of 4, an 8-byte aligned shadow and a 16-byte aligned output");
    if (B == 0) return LIDBOX_OK;
    LBX_ARG(B <= 65535, "B <= 65535");
    launch_pool_fwd_short<true>((const unsigned short*)x16, B, T, C, bs, rs, out, (hipStream_t)stream);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_stats_pool_bwd_bf16(const void* x16, const float* pooled, const float* dout, int B, int T, int C, long bs, long rs,
                                          int relu_mask, void* dx16, long bs16, long rs16, lidbox_stream_t stream) {
    LBX_ARG(x16 && pooled && dout && dx16 && T >= 1 && C >= 1, "x16, pooled, dout, dx16 != NULL; T, C >= 1");
    LBX_ARG(T <= 40 && C % 4 == 0 && bs % 4 == 0 && rs % 4 == 0 && rs >= C && bs16 % 4 == 0 && rs16 % 4 == 0 && rs16 >= C &&
                ((((uintptr_t)x16) | ((uintptr_t)dx16)) & 7) == 0 && ((((uintptr_t)pooled) | ((uintptr_t)dout)) & 15) == 0,
            "the bf16 pooling kernels take T <= 40, C and strides that are multiples of 4, 8-byte aligned shadows and 16-byte aligned statistics");
    if ((long)B * T * C == 0) return LIDBOX_OK;
    LBX_ARG(B <= 65535, "B <= 65535");
    constexpr int R = 12;
    dim3 grid((unsigned)lbx_cdiv(C, 256), (unsigned)B, (unsigned)lbx_cdiv(T, R));
    hipLaunchKernelGGL((pool_bwd_rows_kernel<true, R, unsigned short>), grid, dim3(64), 0, (hipStream_t)stream, (const unsigned short*)x16,
                       pooled, dout, T, C, bs, rs, relu_mask, (float*)nullptr, (unsigned short*)dx16, bs16, rs16);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_avg_pool_bwd(const float* x, const float* dout, int B, int T, int C, long bs, long rs,
                                   int relu_mask, float* dx, lidbox_stream_t stream) {
    LBX_ARG(x && dout && dx && T >= 1 && C >= 1, "pointers != NULL; T, C >= 1");
    const long total = (long)B * T * C;
    if (total == 0) return LIDBOX_OK;
    LBX_ARG(B <= 65535, "B <= 65535");
    launch_pool_bwd<false>(x, nullptr, dout, B, T, C, bs, rs, relu_mask, dx, (hipStream_t)stream);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_log_softmax_fwd(const float* z, int B, int N, float* logp, lidbox_stream_t stream) {
    LBX_ARG(z && logp && B >= 0 && N >= 1, "z, logp != NULL; N >= 1");
    if (B == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(log_softmax_kernel, dim3((unsigned)lbx_cdiv(B, 4)), dim3(256), 0,
                       (hipStream_t)stream, z, B, N, logp);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_nll_fwd_bwd(const float* logp, const int32_t* labels, int B, int N, float scale,
                                  float* loss_out, float* dz, lidbox_stream_t stream) {
    LBX_ARG(logp && labels && loss_out && B >= 1 && N >= 1, "logp, labels, loss_out != NULL; B, N >= 1");
    hipLaunchKernelGGL(nll_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logp, labels, B, N, scale,
                       loss_out, dz);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" size_t lidbox_softmax_head_workspace(int B, int K, int N) {
    if (B <= 0 || K <= 0 || N <= 0) return 0;
    return ((size_t)B * N + (size_t)B) * sizeof(float);          // dz [B,N], per-row losses [B]
}

extern "C" int lidbox_softmax_head_supported(int K, int N) { return N >= 1 && N <= 32 && K >= 1; }

extern "C" int lidbox_softmax_head_fwd_bwd(const float* h, const float* W, const float* bias, const int32_t* labels, int B, int K,
                                           int N, float scale, int relu_mask, float* logp, float* loss_out, float* dW, float* db,
                                           float* dh, void* workspace, size_t workspace_bytes, lidbox_stream_t stream) {
    LBX_ARG(h && W && bias && labels && logp && loss_out && dW && db && B >= 1, "h, W, bias, labels, logp, loss, dW, db != NULL; B >= 1");
    LBX_ARG(lidbox_softmax_head_supported(K, N), "1 <= N <= 32 classes (lidbox_softmax_head_supported)");
    LBX_ARG(workspace && (((uintptr_t)workspace) & 3) == 0 && workspace_bytes >= lidbox_softmax_head_workspace(B, K, N),
            "workspace too small (lidbox_softmax_head_workspace)");
    float* dzbuf = (float*)workspace;
    float* lossrow = dzbuf + (size_t)B * N;
    hipStream_t st = (hipStream_t)stream;
    const unsigned G = (unsigned)lbx_cdiv(B, 4);
#define LBX_SH(NP)                                                                                                       \
    hipLaunchKernelGGL(softmax_head_rows_kernel<NP>, dim3(G), dim3(256), 0, st, h, W, bias, labels, B, K, N, scale, relu_mask, \
                       logp, dh, dzbuf, lossrow)
    if (N <= 4) LBX_SH(4);
    else if (N <= 8) LBX_SH(8);
    else if (N <= 16) LBX_SH(16);
    else LBX_SH(32);
#undef LBX_SH
    LBX_LAUNCH_OK();
    hipLaunchKernelGGL(softmax_head_wgrad_kernel, dim3((unsigned)(lbx_cdiv(K, 16) * N + 1)), dim3(256), 0, st, h, (const float*)dzbuf,
                       (const float*)lossrow, B, K, N, dW, db, loss_out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_l2_normalize_fwd(const float* x, int B, int D, float* out, lidbox_stream_t stream) {
    LBX_ARG(x && out && B >= 0 && D >= 1, "x, out != NULL; D >= 1");
    if (B == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)lbx_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream,
                       x, B, D, out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_l2_normalize_bwd(const float* x, const float* dout, int B, int D, float* dx,
                                       lidbox_stream_t stream) {
    LBX_ARG(x && dout && dx && B >= 0 && D >= 1, "x, dout, dx != NULL; D >= 1");
    if (B == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)lbx_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream,
                       x, dout, B, D, dx);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_ap_loss_fwd_bwd(const float* z, const int32_t* labels, int B, int D, int N,
                                      float delta_weight, float scale, float* loss_per_example, float* dz,
                                      lidbox_stream_t stream) {
    LBX_ARG(z && labels && loss_per_example, "z, labels, loss != NULL");
    LBX_ARG(N >= 1 && D >= N, "N >= 1 and D >= N (losses.py:14-15)");
    LBX_ARG(delta_weight > 0.f, "delta_weight > 0 (losses.py:16)");
    if (B == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(ap_loss_kernel, dim3((unsigned)lbx_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream,
                       z, labels, B, D, N, delta_weight, scale, loss_per_example, dz);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_ap_head_fwd_bwd(const float* x, const int32_t* labels, int B, int D, int N, float delta_weight, float scale,
                                      float* zn, float* loss_per_example, float* dx, float* scores, lidbox_stream_t stream) {
    LBX_ARG(x && labels && loss_per_example, "x, labels, loss != NULL");
    LBX_ARG(N >= 1 && D >= N, "N >= 1 and D >= N (losses.py:14-15)");
    LBX_ARG(delta_weight > 0.f, "delta_weight > 0 (losses.py:16)");
    LBX_ARG(D <= AP_HEAD_MAX_D, "D <= 4096 (wider rows: the separate entry points)");
    if (B == 0) return LIDBOX_OK;
    const size_t lds = (size_t)8 * D * sizeof(float);         // 128 KB at D = 4096: above the 64 KB a kernel may use without asking
    if (lds > 65536) {
        static std::atomic<unsigned long long> attr_devs{0};
        int dev = 0;
        LBX_HIP(hipGetDevice(&dev));
        if (dev >= 64 || !(attr_devs.load() >> dev & 1ull)) {
            LBX_HIP(hipFuncSetAttribute((const void*)ap_head_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (dev < 64) attr_devs.fetch_or(1ull << dev);
        }
    }
    hipLaunchKernelGGL(ap_head_kernel, dim3((unsigned)lbx_cdiv(B, 4)), dim3(256), lds, (hipStream_t)stream, x, labels,
                       B, D, N, delta_weight, scale, zn, loss_per_example, dx, scores);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_cavg_update(const float* scores, const int32_t* labels, int B, int N,
                                  const float* thresholds, int Th, float* tp, float* fn, float* fp_pairs,
                                  float* tn_pairs, lidbox_stream_t stream) {
    LBX_ARG(scores && labels && thresholds && tp && fn && fp_pairs && tn_pairs, "pointers != NULL");
    LBX_ARG(N >= 2 && Th >= 1, "N >= 2 (metrics.py:20), Th >= 1");
    if (B == 0) return LIDBOX_OK;
    // label chunk held in LDS (128 labels: 32.5 KB).  A chunk per workgroup along grid.z while the grid stays within one workgroup
    // of 16 waves per CU; every workgroup still walks the whole batch (examples outside its chunk add 0).  tools/cavg_time.py, 512
    // examples x 100 thresholds: 50 classes 13.1 (round 5) -> 7.2 (one chunk) -> 6.8 us (two); 100 classes 18.9 -> 10.5 (one) / 11.8 (two)
    const int thb = (int)lbx_cdiv(Th, 64);
    int nz = (int)(256 / ((long)N * thb));
    if (const char* e = getenv("LIDBOX_CAVG_NZ")) { const int v = atoi(e); if (v >= 1) nz = v; }      // tuning aid
    int lch = (int)lbx_cdiv(N, nz < 1 ? 1 : nz);
    lch = (lch + CAVG_WAVES - 1) / CAVG_WAVES * CAVG_WAVES;
    if (lch > CAVG_LCH) lch = CAVG_LCH;
    nz = (int)lbx_cdiv(N, lch);
    if (nz > 64) nz = 64;
    // examples per unit of the walk: 16 waves x unit covers the batch in one pass up to 1 024 examples
    int unit = 16;
    while (unit < 64 && (long)CAVG_WAVES * unit < B) unit <<= 1;
    if (const char* e = getenv("LIDBOX_CAVG_UNIT")) { const int v = atoi(e); if (v == 8 || v == 16 || v == 32 || v == 64) unit = v; }
    const size_t lds = ((size_t)lch * 64 + lch) * sizeof(unsigned);
    hipLaunchKernelGGL(cavg_update_kernel, dim3(N, (unsigned)thb, (unsigned)nz), dim3(1024), lds,
                       (hipStream_t)stream, scores, labels, B, N, thresholds, Th, lch, unit, tp, fn, fp_pairs,
                       tn_pairs);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_cavg_result(const float* tp, const float* fn, const float* fp_pairs,
                                  const float* tn_pairs, int N, int Th, float C_miss, float C_fa, float P_tar,
                                  float* c_avg_out, float* out, lidbox_stream_t stream) {
    LBX_ARG(tp && fn && fp_pairs && tn_pairs && out, "pointers != NULL");
    LBX_ARG(N >= 2 && Th >= 1, "N >= 2, Th >= 1");
    const size_t lds = (size_t)N * Th * sizeof(float);
    const int lds_pf = Th <= 512 && lds <= 48 * 1024;               // at least two language groups, inside the default LDS limit
    hipLaunchKernelGGL(cavg_result_kernel, dim3(1), dim3(1024), lds_pf ? lds : 0, (hipStream_t)stream, tp, fn, fp_pairs,
                       tn_pairs, N, Th, C_miss, C_fa, P_tar, lds_pf, c_avg_out, out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_softmax_fwd(const float* z, int B, int N, float* out, lidbox_stream_t stream) {
    LBX_ARG(z && out && B >= 0 && N >= 1, "z, out != NULL; N >= 1");
    if (B == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(softmax_kernel, dim3((unsigned)lbx_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, z, B, N, out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_softmax_nll_fwd_bwd(const float* z, const int32_t* labels, int B, int N, float scale, float* probs,
                                          float* loss_out, float* dz, lidbox_stream_t stream) {
    LBX_ARG(z && labels && loss_out && B >= 1 && N >= 1, "z, labels, loss_out != NULL; B, N >= 1");
    hipLaunchKernelGGL(softmax_nll_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, z, labels, B, N, scale, probs, loss_out, dz);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_sgd_step(float* param, const float* grad, float* velocity, long n, float lr, float momentum, int nesterov,
                               float grad_scale, void* state, lidbox_stream_t stream) {
    LBX_ARG(param && grad && state && n >= 0, "param, grad, state != NULL");
    LBX_ARG(momentum >= 0.f && (momentum == 0.f || velocity), "momentum >= 0; momentum > 0 needs the velocity buffer");
    hipLaunchKernelGGL(opt_prepare_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (AdamState*)state, lr);
    LBX_LAUNCH_OK();
    if (n == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(sgd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, param, grad, momentum > 0.f ? velocity : nullptr, n,
                       (const AdamState*)state, momentum, nesterov, grad_scale);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_rmsprop_step(float* param, const float* grad, float* rms, float* mean_grad, float* mom, long n, float lr, float rho,
                                   float momentum, float epsilon, int centered, float grad_scale, void* state, lidbox_stream_t stream) {
    LBX_ARG(param && grad && rms && state && n >= 0, "param, grad, rms, state != NULL");
    LBX_ARG((!centered || mean_grad) && (momentum == 0.f || mom) && momentum >= 0.f, "centered needs mean_grad, momentum > 0 needs mom");
    hipLaunchKernelGGL(opt_prepare_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (AdamState*)state, lr);
    LBX_LAUNCH_OK();
    if (n == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(rmsprop_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, param, grad, rms, centered ? mean_grad : nullptr,
                       momentum > 0.f ? mom : nullptr, n, (const AdamState*)state, rho, momentum, epsilon, grad_scale);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_adam_step(float* param, const float* grad, float* m, float* v, long n, float lr,
                                float beta1, float beta2, float eps, float grad_scale, void* state,
                                lidbox_stream_t stream) {
    LBX_ARG(param && grad && m && v && state && n >= 0, "pointers != NULL");
    LBX_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v | (uintptr_t)state) & 15) == 0,
            "param/grad/m/v/state must be 16-byte aligned");
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (AdamState*)state, lr,
                       beta1, beta2);
    LBX_LAUNCH_OK();
    if (n == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       m, v, n, (const AdamState*)state, beta1, beta2, eps, grad_scale);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

// The optimizer step in two halves, so that its scalar half can ride in the launch that finishes the step's last wgrad
// (lidbox_reduce_jobs_run): lidbox_adam_prepare_job describes "advance the step counter, publish lr_t" as a job (what
// adam_prepare_kernel does), lidbox_adam_apply is the elementwise update with the lr_t it finds in the state.
// lidbox_adam_step == the job run on its own + lidbox_adam_apply.
extern "C" int lidbox_adam_prepare_job(void* state, float lr, float beta1, float beta2, lidbox_reduce_job_t* job) {
    LBX_ARG(state && job && (((uintptr_t)state) & 15) == 0, "state, job != NULL; state 16-byte aligned");
    memset(job, 0, sizeof *job);
    unsigned ulr, ub1, ub2;
    memcpy(&ulr, &lr, 4); memcpy(&ub1, &beta1, 4); memcpy(&ub2, &beta2, 4);
    job->C = (float*)state;
    job->n = (long)(((unsigned long)ub1 << 32) | ulr);
    job->ldc = (long)ub2;
    job->splits = -1;
    job->nblocks = 1;
    return LIDBOX_OK;
}

extern "C" int lidbox_adam_apply(float* param, const float* grad, float* m, float* v, long n, float beta1, float beta2, float eps,
                                 float grad_scale, const void* state, lidbox_stream_t stream) {
    LBX_ARG(param && grad && m && v && state && n >= 0, "pointers != NULL");
    LBX_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v | (uintptr_t)state) & 15) == 0,
            "param/grad/m/v/state must be 16-byte aligned");
    if (n == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       m, v, n, (const AdamState*)state, beta1, beta2, eps, grad_scale);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_mean(const float* x, long n, float* out, lidbox_stream_t stream) {
    LBX_ARG(x && out && n >= 1, "x, out != NULL; n >= 1");
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_neg_acos(const float* z, long B, int D, int N, float* out, lidbox_stream_t stream) {
    LBX_ARG(z && out && B >= 0 && N >= 1 && D >= N, "z, out != NULL; 1 <= N <= D");
    if (B == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(neg_acos_kernel, dim3(ew_grid(B * N)), dim3(256), 0, (hipStream_t)stream, z, B, D, N, out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_spatial_dropout(float* x, int B, int T, int C, long batch_stride, float rate,
                                      unsigned long long seed, const void* step_counter, float* mask_out,
                                      lidbox_stream_t stream) {
    LBX_ARG(x && B >= 0 && T >= 0 && C >= 1, "x != NULL; C >= 1");
    LBX_ARG(rate >= 0.f && rate < 1.f, "0 <= rate < 1");
    LBX_ARG(batch_stride >= (long)T * C, "batch_stride >= T * C");
    LBX_ARG(B <= 65535, "B <= 65535");
    if (B == 0 || T == 0 || rate == 0.f) return LIDBOX_OK;
    long gx = lbx_cdiv((long)T * C, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(spatial_dropout_kernel, dim3((unsigned)gx, (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       x, T, C, batch_stride, rate, seed, (const long long*)step_counter, mask_out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_dropout_rows(lidbox_rows_out_t x, int C, float rate, unsigned long long seed, const void* step_counter,
                                   lidbox_stream_t stream) {
    LBX_ARG(x.base && C >= 1 && rate >= 0.f && rate < 1.f, "x != NULL, C >= 1, 0 <= rate < 1");
    const long R = (long)x.batch * x.rows_per_batch;
    LBX_ARG(R >= 0 && R <= 0x7fffffffL, "row count");
    if (R == 0 || rate == 0.f) return LIDBOX_OK;
    RowsOutD X{x.base, x.batch_stride, x.row_stride, x.batch, x.rows_per_batch};
    long g = lbx_cdiv(R * C, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(dropout_rows_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, X, R, C, rate, seed,
                       (const long long*)step_counter);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_copy_2d(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes,
                              size_t height, lidbox_stream_t stream) {
    LBX_ARG(dst && src && dst_pitch >= width_bytes && src_pitch >= width_bytes, "dst, src != NULL; pitches >= width");
    if (width_bytes == 0 || height == 0) return LIDBOX_OK;
    LBX_HIP(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, height, hipMemcpyDeviceToDevice,
                             (hipStream_t)stream));
    return LIDBOX_OK;
}

extern "C" int lidbox_zero_2d(void* dst, size_t pitch, size_t width_bytes, size_t height, lidbox_stream_t stream) {
    LBX_ARG(dst && pitch >= width_bytes, "dst != NULL; pitch >= width");
    if (width_bytes == 0 || height == 0) return LIDBOX_OK;
    LBX_HIP(hipMemset2DAsync(dst, pitch, 0, width_bytes, height, (hipStream_t)stream));
    return LIDBOX_OK;
}

// one wave that runs for a given time by the constant-rate wall clock: a kernel of KNOWN duration, for calibrating what a
// HIP-event bracket adds to a launch (bench.py: KernelTimer)
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}

extern "C" int lidbox_calibration_spin(double microseconds, lidbox_stream_t stream) {
    LBX_ARG(microseconds >= 0.0 && microseconds <= 1.0e6, "0 <= microseconds <= 1e6");
    int dev = 0, khz = 0;
    LBX_HIP(hipGetDevice(&dev));
    LBX_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    LBX_ARG(khz > 0, "the device reports no wall-clock rate");
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)(microseconds * 1e-3 * khz));
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_scale(float* x, long n, float alpha, lidbox_stream_t stream) {
    LBX_ARG(x && n >= 0, "x != NULL");
    if (n == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(scale_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n, alpha);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_fill(float* x, long n, float value, lidbox_stream_t stream) {
    LBX_ARG(x && n >= 0, "x != NULL");
    if (n == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n, value);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}
