// signal.hip -- the signal steps that sit immediately before the feature kernel, gfx950.
//
// Replaces (reference file:line):
//   lidbox/features/audio.py:57-59     peak_normalize
//   lidbox/features/audio.py:266-270   root_mean_square
//   lidbox/features/audio.py:275-297   run_length_encoding + invert_too_short_consecutive_false
//   lidbox/features/audio.py:308-329   framewise_rms_energy_vad_decisions
//   lidbox/features/audio.py:337-353   remove_silence  (= decisions + apply)
//   lidbox/data/steps.py:191-198       apply_vad: keep the frames marked as speech
//   lidbox/data/steps.py:600-614       create_signal_chunks: fixed-length chunks with bounded zero padding
//   lidbox/features/audio.py:128-148   snr_mixer
//   lidbox/util.py:41-57               merge_chunk_predictions (stack_and_average of each parent's chunk rows)
//
// The reference maps these over a tf.data.Dataset one variable-length signal at a time.  Here a whole
// RAGGED batch goes through each kernel: utterance b is signals[starts[b] .. starts[b] + lengths[b]) (int64
// arrays in device memory; starts may leave gaps, e.g. to keep every utterance 16-byte aligned), frames /
// chunks of all utterances are numbered consecutively (frame_offsets / chunk_offsets, CSR with B+1 entries)
// and a frame finds its utterance by binary search over those offsets.  Every kernel
// reads each sample once (snr_mixer re-reads its utterance from L2) -- all HBM-bound; sizes that depend on
// the data (speech frames per utterance) are returned as counters, the host sizes the next buffer from them
// exactly as the reference's eager tensors do.  No atomics on floats: results are deterministic.
#include <stdint.h>
#include <stdlib.h>

#include "common.h"

namespace {

// index of the utterance that owns global item g: largest b with off[b] <= g  (off has B+1 entries)
__device__ __forceinline__ int owner_of(const int64_t* __restrict__ off, int B, int64_t g) {
    int lo = 0, hi = B;                      // invariant: off[lo] <= g < off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    const int tid = threadIdx.x;
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float block_max_256(float v, float* red) {
    const int tid = threadIdx.x;
    v = wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---- frame RMS: 8 lanes per frame, 32 frames per workgroup ------------------------------------
__global__ __launch_bounds__(256) void frame_rms_kernel(const float* __restrict__ signals,
                                                        const int64_t* __restrict__ starts,
                                                        const int64_t* __restrict__ frame_offsets, int B,
                                                        int64_t total_frames, int L, float* __restrict__ out) {
    const int64_t gf = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int q = threadIdx.x & 7;
    const bool valid = gf < total_frames;
    float acc = 0.f;
    if (valid) {
        const int b = owner_of(frame_offsets, B, gf);
        const float* src = signals + starts[b] + (gf - frame_offsets[b]) * L;
        if ((L & 3) == 0 && (((uintptr_t)src) & 15) == 0) {
            for (int i = q * 4; i < L; i += 32) {
                const float4 v = *reinterpret_cast<const float4*>(src + i);
                acc = fmaf(v.x, v.x, acc);
                acc = fmaf(v.y, v.y, acc);
                acc = fmaf(v.z, v.z, acc);
                acc = fmaf(v.w, v.w, acc);
            }
        } else {
            for (int i = q; i < L; i += 8) acc = fmaf(src[i], src[i], acc);
        }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (valid && q == 0) out[gf] = sqrtf(acc / (float)L);
}

// ---- per-utterance threshold = strength * max(min_rms_threshold, mean(frame rms)) ------------
__global__ __launch_bounds__(256) void vad_threshold_kernel(const float* __restrict__ frame_rms,
                                                            const int64_t* __restrict__ frame_offsets, float strength,
                                                            float min_rms_threshold, float* __restrict__ thr) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int64_t f0 = frame_offsets[b], nf = frame_offsets[b + 1] - f0;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < nf; i += 256) s += frame_rms[f0 + i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) thr[b] = nf > 0 ? strength * fmaxf(min_rms_threshold, s / (float)nf) : 0.f;
}

// ---- decisions: rms > threshold, then runs of non-speech shorter than min_len become speech ---
// A frame's run length is (non-speech frames before it) + 1 + (after it), each side counted up to min_len
// (a capped side already proves the run is long enough), so the lookaround is bounded and needs no scan.
__global__ __launch_bounds__(256) void vad_decide_kernel(const float* __restrict__ frame_rms,
                                                         const int64_t* __restrict__ frame_offsets, int B,
                                                         int64_t total_frames, const float* __restrict__ thr,
                                                         int min_len, uint8_t* __restrict__ decisions) {
    const int64_t gf = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gf >= total_frames) return;
    const int b = owner_of(frame_offsets, B, gf);
    const int64_t f0 = frame_offsets[b], f1 = frame_offsets[b + 1];
    const float t = thr[b];
    bool speech = frame_rms[gf] > t;
    if (!speech && min_len > 0) {
        int left = 0, right = 0;
        for (int64_t j = gf - 1; j >= f0 && left < min_len && !(frame_rms[j] > t); --j) ++left;
        for (int64_t j = gf + 1; j < f1 && right < min_len && !(frame_rms[j] > t); ++j) ++right;
        if (left < min_len && right < min_len && left + 1 + right < min_len) speech = true;
    }
    decisions[gf] = speech ? 1 : 0;
}

// ---- exclusive prefix of the decisions inside each utterance (output slot of every speech frame) + totals
__global__ __launch_bounds__(256) void vad_scan_kernel(const uint8_t* __restrict__ decisions,
                                                       const int64_t* __restrict__ frame_offsets,
                                                       int32_t* __restrict__ slots, int32_t* __restrict__ counts) {
    __shared__ int wsum[4];
    const int b = blockIdx.x;
    const int64_t f0 = frame_offsets[b], nf = frame_offsets[b + 1] - f0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (int64_t base = 0; base < nf; base += 256) {
        const int64_t i = base + threadIdx.x;
        const bool d = i < nf && decisions[f0 + i] != 0;
        const unsigned long long m = __ballot(d);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        if (i < nf) slots[f0 + i] = carry + wbase + before;
        carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
    if (threadIdx.x == 0) counts[b] = carry;
}

// ---- apply_vad: copy the speech frames to their slots (8 lanes per frame) ----------------------
__global__ __launch_bounds__(256) void apply_vad_kernel(const float* __restrict__ signals,
                                                        const int64_t* __restrict__ starts,
                                                        const int64_t* __restrict__ frame_offsets,
                                                        const uint8_t* __restrict__ decisions,
                                                        const int32_t* __restrict__ slots,
                                                        const int64_t* __restrict__ out_starts, int B,
                                                        int64_t total_frames, int L, float* __restrict__ out) {
    const int64_t gf = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int q = threadIdx.x & 7;
    if (gf >= total_frames || !decisions[gf]) return;
    const int b = owner_of(frame_offsets, B, gf);
    const float* src = signals + starts[b] + (gf - frame_offsets[b]) * L;
    float* dst = out + out_starts[b] + (int64_t)slots[gf] * L;
    if ((L & 3) == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
        for (int i = q * 4; i < L; i += 32)
            *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(src + i);
    } else {
        for (int i = q; i < L; i += 8) dst[i] = src[i];
    }
}

// ---- create_signal_chunks: chunk row c of utterance b = samples [c*S, c*S + L), zeros past the end ----
// grid (ceil(L / 1024), total_chunks)
__global__ __launch_bounds__(256) void signal_chunks_kernel(const float* __restrict__ signals,
                                                            const int64_t* __restrict__ starts,
                                                            const int64_t* __restrict__ lengths,
                                                            const int64_t* __restrict__ chunk_offsets, int B,
                                                            int64_t c0, int L, int S, float* __restrict__ out) {
    const int64_t gc = c0 + blockIdx.y;
    const int b = owner_of(chunk_offsets, B, gc);
    const int64_t s0 = starts[b], n = lengths[b];
    const int64_t start = (gc - chunk_offsets[b]) * (int64_t)S;
    const float* src = signals + s0 + start;
    float* dst = out + gc * (int64_t)L;
    const int i = blockIdx.x * 1024 + threadIdx.x * 4;
    if (i >= L) return;
    if (i + 3 < L && start + i + 3 < n && ((((uintptr_t)(src + i)) | ((uintptr_t)(dst + i))) & 15) == 0) {
        *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(src + i);
    } else {
        for (int k = i; k < i + 4 && k < L; ++k) dst[k] = (start + k < n) ? src[k] : 0.f;
    }
}

// ---- peak_normalize: out = 10^(dBFS/20) * (x / max|x|), one workgroup per utterance ------------
__global__ __launch_bounds__(256) void peak_normalize_kernel(const float* __restrict__ signals,
                                                             const int64_t* __restrict__ starts,
                                                             const int64_t* __restrict__ lengths, float level,
                                                             float* __restrict__ out) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int64_t s0 = starts[b], n = lengths[b];
    const float* x = signals + s0;
    float m = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(x[i]));
    m = block_max_256(m, red);
    for (int64_t i = threadIdx.x; i < n; i += 256) out[s0 + i] = level * (x[i] / m);
}

// ---- sum of squares per utterance (root_mean_square) ------------------------------------------
__global__ __launch_bounds__(256) void signal_rms_kernel(const float* __restrict__ signals,
                                                         const int64_t* __restrict__ starts,
                                                         const int64_t* __restrict__ lengths,
                                                         float* __restrict__ out) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int64_t s0 = starts[b], n = lengths[b];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s = fmaf(signals[s0 + i], signals[s0 + i], s);
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[b] = sqrtf(s / (float)n);
}

// ---- snr_mixer on dense [B, N] pairs: one workgroup per pair, three passes (the 2nd and 3rd hit L2) ----
__device__ __forceinline__ float sumsq_scaled(const float* __restrict__ x, int64_t n, float scale, bool vec) {
    float s = 0.f;
    if (vec) {
        for (int64_t i = (int64_t)threadIdx.x * 4; i < n; i += 1024) {
            float4 v = *reinterpret_cast<const float4*>(x + i);
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
        }
    } else {
        for (int64_t i = threadIdx.x; i < n; i += 256) { const float v = scale * x[i]; s = fmaf(v, v, s); }
    }
    return s;
}

__global__ __launch_bounds__(256) void snr_mixer_kernel(const float* __restrict__ clean,
                                                        const float* __restrict__ noise,
                                                        const float* __restrict__ snr_db, int64_t N,
                                                        float* __restrict__ clean_norm, float* __restrict__ noise_new,
                                                        float* __restrict__ noisy) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* c = clean + (int64_t)b * N;
    const float* z = noise + (int64_t)b * N;
    const bool vec = (N & 3) == 0 && ((((uintptr_t)clean) | ((uintptr_t)noise) | ((uintptr_t)clean_norm) |
                                       ((uintptr_t)noise_new) | ((uintptr_t)noisy)) & 15) == 0;
    const float lvl25 = __powf(10.0f, -25.0f / 20.0f);
    const float fn = (float)N;
    // audio.py:134-139: normalise both to -25 dBFS, then measure the RMS of the normalised signals
    const float sc = lvl25 / sqrtf(block_sum_256(sumsq_scaled(c, N, 1.f, vec), red) / fn);
    const float sz = lvl25 / sqrtf(block_sum_256(sumsq_scaled(z, N, 1.f, vec), red) / fn);
    const float rmsclean = sqrtf(block_sum_256(sumsq_scaled(c, N, sc, vec), red) / fn);
    const float rmsnoise = sqrtf(block_sum_256(sumsq_scaled(z, N, sz, vec), red) / fn);
    const float level = __powf(10.0f, snr_db[b] / 20.0f);                      // :143
    const float noisescalar = sqrtf(rmsclean / level / rmsnoise);              // :144
    float* oc = clean_norm + (int64_t)b * N;
    float* on = noise_new + (int64_t)b * N;
    float* om = noisy + (int64_t)b * N;
    if (vec) {
        for (int64_t i = (int64_t)threadIdx.x * 4; i < N; i += 1024) {
            float4 a = *reinterpret_cast<const float4*>(c + i), w = *reinterpret_cast<const float4*>(z + i);
            a.x *= sc; a.y *= sc; a.z *= sc; a.w *= sc;
            w.x = noisescalar * (sz * w.x); w.y = noisescalar * (sz * w.y);
            w.z = noisescalar * (sz * w.z); w.w = noisescalar * (sz * w.w);
            *reinterpret_cast<float4*>(oc + i) = a;
            *reinterpret_cast<float4*>(on + i) = w;
            *reinterpret_cast<float4*>(om + i) = make_float4(a.x + w.x, a.y + w.y, a.z + w.z, a.w + w.w);
        }
    } else {
        for (int64_t i = threadIdx.x; i < N; i += 256) {
            const float a = sc * c[i], w = noisescalar * (sz * z[i]);
            oc[i] = a; on[i] = w; om[i] = a + w;
        }
    }
}

// ---- snr_mixer, register-resident: the pair (up to 1024 * 4 * NV samples each) is read ONCE into registers;
//      both RMS rounds and the three outputs come from there (5 N bytes of memory traffic instead of 9 N).
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
    const int tid = threadIdx.x;
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w];
    return s;
}

template <int NV>
__global__ __launch_bounds__(1024) void snr_mixer_reg_kernel(const float* __restrict__ clean,
                                                             const float* __restrict__ noise,
                                                             const float* __restrict__ snr_db, int64_t N,
                                                             float* __restrict__ clean_norm,
                                                             float* __restrict__ noise_new, float* __restrict__ noisy) {
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* c = clean + (int64_t)b * N;
    const float* z = noise + (int64_t)b * N;
    float4 cv[NV], zv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int64_t o = ((int64_t)i * 1024 + tid) * 4;
        const bool in = o < N;                                  // N % 4 == 0 on this path
        cv[i] = in ? *reinterpret_cast<const float4*>(c + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        zv[i] = in ? *reinterpret_cast<const float4*>(z + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto sumsq = [&](const float4 (&v)[NV], float scale) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float x = scale * v[i].x, y = scale * v[i].y, u = scale * v[i].z, w = scale * v[i].w;
            s = fmaf(x, x, s); s = fmaf(y, y, s); s = fmaf(u, u, s); s = fmaf(w, w, s);
        }
        return s;
    };
    const float lvl25 = __powf(10.0f, -25.0f / 20.0f);
    const float fn = (float)N;
    const float sc = lvl25 / sqrtf(block_sum_1024(sumsq(cv, 1.f), red) / fn);          // audio.py:134-135
    const float sz = lvl25 / sqrtf(block_sum_1024(sumsq(zv, 1.f), red) / fn);          // :138-139
    const float rmsclean = sqrtf(block_sum_1024(sumsq(cv, sc), red) / fn);
    const float rmsnoise = sqrtf(block_sum_1024(sumsq(zv, sz), red) / fn);
    const float level = __powf(10.0f, snr_db[b] / 20.0f);                              // :143
    const float noisescalar = sqrtf(rmsclean / level / rmsnoise);                      // :144
    float* oc = clean_norm + (int64_t)b * N;
    float* on = noise_new + (int64_t)b * N;
    float* om = noisy + (int64_t)b * N;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int64_t o = ((int64_t)i * 1024 + tid) * 4;
        if (o >= N) continue;
        float4 a = cv[i], w = zv[i];
        a.x *= sc; a.y *= sc; a.z *= sc; a.w *= sc;
        w.x = noisescalar * (sz * w.x); w.y = noisescalar * (sz * w.y);
        w.z = noisescalar * (sz * w.z); w.w = noisescalar * (sz * w.w);
        *reinterpret_cast<float4*>(oc + o) = a;
        *reinterpret_cast<float4*>(on + o) = w;
        *reinterpret_cast<float4*>(om + o) = make_float4(a.x + w.x, a.y + w.y, a.z + w.z, a.w + w.w);
    }
}

// ---- peak_normalize, register-resident (utterances of up to 1024 * 4 * NV samples, 16-byte aligned starts): one read
template <int NV>
__global__ __launch_bounds__(1024) void peak_normalize_reg_kernel(const float* __restrict__ signals,
                                                                  const int64_t* __restrict__ starts,
                                                                  const int64_t* __restrict__ lengths, float level,
                                                                  float* __restrict__ out) {
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t s0 = starts[b], n = lengths[b];
    const float* x = signals + s0;
    float4 v[NV];
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int64_t o = ((int64_t)i * 1024 + tid) * 4;
        if (o + 3 < n) {
            v[i] = *reinterpret_cast<const float4*>(x + o);
        } else {
            v[i].x = o + 0 < n ? x[o + 0] : 0.f;
            v[i].y = o + 1 < n ? x[o + 1] : 0.f;
            v[i].z = o + 2 < n ? x[o + 2] : 0.f;
            v[i].w = 0.f;
        }
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
    }
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    float* o_ = out + s0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int64_t o = ((int64_t)i * 1024 + tid) * 4;
        const float4 r = make_float4(level * (v[i].x / m), level * (v[i].y / m), level * (v[i].z / m), level * (v[i].w / m));
        if (o + 3 < n) {
            *reinterpret_cast<float4*>(o_ + o) = r;
        } else {
            if (o + 0 < n) o_[o + 0] = r.x;
            if (o + 1 < n) o_[o + 1] = r.y;
            if (o + 2 < n) o_[o + 2] = r.z;
        }
    }
}

// ---- mean of consecutive row groups: out[s, :] = mean(x[seg[s] .. seg[s+1], :])  (util.py:41-57) ----
__global__ __launch_bounds__(256) void segment_mean_kernel(const float* __restrict__ x,
                                                           const int64_t* __restrict__ seg, int D,
                                                           float* __restrict__ out) {
    const int s = blockIdx.y;
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    const int64_t r0 = seg[s], r1 = seg[s + 1];
    float acc = 0.f;
    for (int64_t r = r0; r < r1; ++r) acc += x[r * D + d];
    out[(int64_t)s * D + d] = acc / (float)(r1 - r0);
}

// ---- 16-bit PCM ingest: out[f] = mean over channels of pcm[f][c] / 32768  (features/audio.py:17-23) ----
// tf.audio.decode_wav scales int16 by 2^-15 (exact in fp32), reduce_mean sums the channels of a frame (exact: a few 16-bit values)
// and divides by their number once.  Mono with 8 frames per thread: one 16-byte load, two 16-byte stores.
__global__ __launch_bounds__(256) void pcm16_to_f32_kernel(const int16_t* __restrict__ pcm, int64_t frames, int channels,
                                                           float* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (channels == 1) {
        const bool vec = ((((uintptr_t)pcm) | ((uintptr_t)out)) & 15) == 0;
        const int64_t n8 = vec ? frames >> 3 : 0;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
            const int4 v = *reinterpret_cast<const int4*>(pcm + 8 * i);
            const int w[4] = {v.x, v.y, v.z, v.w};
            float x[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x[2 * j] = (float)(short)(w[j] & 0xffff) * (1.0f / 32768.0f);
                x[2 * j + 1] = (float)(w[j] >> 16) * (1.0f / 32768.0f);
            }
            const float4 lo = make_float4(x[0], x[1], x[2], x[3]), hi = make_float4(x[4], x[5], x[6], x[7]);
            *reinterpret_cast<float4*>(out + 8 * i) = lo;
            *reinterpret_cast<float4*>(out + 8 * i + 4) = hi;
        }
        for (int64_t i = (n8 << 3) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < frames; i += stride)
            out[i] = (float)pcm[i] * (1.0f / 32768.0f);
        return;
    }
    const float nch = (float)channels;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < frames; i += stride) {
        const int16_t* p = pcm + i * channels;
        float acc = 0.f;
        for (int c = 0; c < channels; ++c) acc += (float)p[c] * (1.0f / 32768.0f);
        out[i] = acc / nch;
    }
}

}  // namespace

extern "C" int lidbox_pcm16_to_f32(const int16_t* pcm, long frames, int channels, float* out, lidbox_stream_t stream) {
    LBX_ARG(frames >= 0 && channels >= 1 && channels <= 256, "frames >= 0, 1 <= channels <= 256");
    if (frames == 0) return LIDBOX_OK;
    LBX_ARG(pcm && out, "pcm, out != NULL");
    long g = lbx_cdiv(channels == 1 ? lbx_cdiv(frames, 8) : frames, 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(pcm16_to_f32_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, pcm, (int64_t)frames, channels, out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_signal_chunk_plan(long num_samples, int sample_rate, int length_ms, int step_ms,
                                        int max_pad_ms, long* out4) {
    LBX_ARG(out4 && num_samples >= 0 && sample_rate > 0, "out4 != NULL, num_samples >= 0, sample_rate > 0");
    // float32, like the tf.constant / tf.cast chain of steps.py:586-588, 604-606
    const float sr = (float)sample_rate;
    const float len_s = (float)(1e-3 * (double)length_ms), step_s = (float)(1e-3 * (double)step_ms);
    const float pad_s = (float)(1e-3 * (double)max_pad_ms);
    volatile float fl = sr * len_s, fs = sr * step_s, fp = sr * pad_s;
    const long L = (long)(int)fl, S = (long)(int)fs, P = (long)(int)fp;
    LBX_ARG(L >= 1 && S >= 1, "chunk length and step must be at least one sample");
    long n = num_samples;
    // Python-style floor division for the (possibly negative) numerator of steps.py:607
    auto floordiv = [](long a, long b) { long q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; };
    long full = 1 + floordiv(n - L, S);
    if (full < 0) full = 0;
    const long last = n - full * S;                                            // :610
    if (last < L && L <= last + P) n += L - last;                              // :611-612
    const long chunks = n >= L ? 1 + (n - L) / S : 0;                          // tf.signal.frame, :614
    out4[0] = L; out4[1] = S; out4[2] = n; out4[3] = chunks;
    return LIDBOX_OK;
}

extern "C" int lidbox_frame_rms(const float* signals, const int64_t* starts, const int64_t* frame_offsets, int B,
                                long total_frames, int frame_len, float* frame_rms, lidbox_stream_t stream) {
    LBX_ARG(signals && starts && frame_offsets && frame_rms, "pointers != NULL");
    LBX_ARG(B >= 0 && total_frames >= 0 && frame_len >= 1, "B, total_frames >= 0, frame_len >= 1");
    if (B == 0 || total_frames == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(frame_rms_kernel, dim3((unsigned)lbx_cdiv(total_frames, 32)), dim3(256), 0, (hipStream_t)stream,
                       signals, starts, frame_offsets, B, (int64_t)total_frames, frame_len, frame_rms);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_vad_decisions(const float* frame_rms, const int64_t* frame_offsets, int B, long total_frames,
                                    float strength, float min_rms_threshold, int min_non_speech_frames,
                                    uint8_t* decisions, int32_t* slots, int32_t* counts, float* thresholds,
                                    lidbox_stream_t stream) {
    LBX_ARG(frame_rms && frame_offsets && decisions && slots && counts && thresholds, "pointers != NULL");
    LBX_ARG(B >= 0 && total_frames >= 0 && min_non_speech_frames >= 0, "B, total_frames, min_non_speech_frames >= 0");
    if (B == 0) return LIDBOX_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(vad_threshold_kernel, dim3(B), dim3(256), 0, st, frame_rms, frame_offsets, strength,
                       min_rms_threshold, thresholds);
    LBX_LAUNCH_OK();
    if (total_frames > 0) {
        hipLaunchKernelGGL(vad_decide_kernel, dim3((unsigned)lbx_cdiv(total_frames, 256)), dim3(256), 0, st, frame_rms,
                           frame_offsets, B, (int64_t)total_frames, thresholds, min_non_speech_frames, decisions);
        LBX_LAUNCH_OK();
    }
    hipLaunchKernelGGL(vad_scan_kernel, dim3(B), dim3(256), 0, st, decisions, frame_offsets, slots, counts);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_vad_scan(const uint8_t* decisions, const int64_t* frame_offsets, int B, int32_t* slots,
                               int32_t* counts, lidbox_stream_t stream) {
    LBX_ARG(decisions && frame_offsets && slots && counts && B >= 0, "pointers != NULL, B >= 0");
    if (B == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(vad_scan_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, decisions, frame_offsets, slots,
                       counts);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_apply_vad(const float* signals, const int64_t* starts, const int64_t* frame_offsets,
                                const uint8_t* decisions, const int32_t* slots, const int64_t* out_starts, int B,
                                long total_frames, int frame_len, float* out, lidbox_stream_t stream) {
    LBX_ARG(signals && starts && frame_offsets && decisions && slots && out_starts, "pointers != NULL");
    LBX_ARG(B >= 0 && total_frames >= 0 && frame_len >= 1, "B, total_frames >= 0, frame_len >= 1");
    if (B == 0 || total_frames == 0) return LIDBOX_OK;
    LBX_ARG(out, "out != NULL");
    hipLaunchKernelGGL(apply_vad_kernel, dim3((unsigned)lbx_cdiv(total_frames, 32)), dim3(256), 0, (hipStream_t)stream,
                       signals, starts, frame_offsets, decisions, slots, out_starts, B, (int64_t)total_frames,
                       frame_len, out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_signal_chunks(const float* signals, const int64_t* starts, const int64_t* lengths,
                                    const int64_t* chunk_offsets, int B, long total_chunks, int chunk_len,
                                    int chunk_step, float* out, lidbox_stream_t stream) {
    LBX_ARG(signals && starts && lengths && chunk_offsets, "pointers != NULL");
    LBX_ARG(B >= 0 && total_chunks >= 0 && chunk_len >= 1 && chunk_step >= 1, "sizes");
    if (B == 0 || total_chunks == 0) return LIDBOX_OK;
    LBX_ARG(out, "out != NULL");
    hipStream_t st = (hipStream_t)stream;
    const unsigned gx = (unsigned)lbx_cdiv(chunk_len, 1024);
    for (long c0 = 0; c0 < total_chunks; c0 += 65535) {          // grid.y is limited to 65535 workgroups
        const long nc = total_chunks - c0 < 65535 ? total_chunks - c0 : 65535;
        hipLaunchKernelGGL(signal_chunks_kernel, dim3(gx, (unsigned)nc), dim3(256), 0, st, signals, starts, lengths,
                           chunk_offsets, B, (int64_t)c0, chunk_len, chunk_step, out);
        LBX_LAUNCH_OK();
    }
    return LIDBOX_OK;
}

extern "C" int lidbox_peak_normalize_max(const float* signals, const int64_t* starts, const int64_t* lengths, int B,
                                         float dBFS, long max_length, int aligned16, float* out,
                                         lidbox_stream_t stream) {
    LBX_ARG(signals && starts && lengths && out && B >= 0 && max_length >= 0, "pointers != NULL, B, max_length >= 0");
    if (B == 0) return LIDBOX_OK;
    // every utterance fits the registers of one 1024-thread workgroup and starts on a 16-byte boundary: one read
    if (aligned16 && max_length <= 1024L * 4 * 16 && ((((uintptr_t)signals) | ((uintptr_t)out)) & 15) == 0) {
        hipStream_t st = (hipStream_t)stream;
        const float level = powf(10.0f, dBFS / 20.0f);
        if (max_length <= 1024L * 4 * 4)
            hipLaunchKernelGGL(peak_normalize_reg_kernel<4>, dim3(B), dim3(1024), 0, st, signals, starts, lengths, level, out);
        else if (max_length <= 1024L * 4 * 8)
            hipLaunchKernelGGL(peak_normalize_reg_kernel<8>, dim3(B), dim3(1024), 0, st, signals, starts, lengths, level, out);
        else
            hipLaunchKernelGGL(peak_normalize_reg_kernel<16>, dim3(B), dim3(1024), 0, st, signals, starts, lengths, level, out);
        LBX_LAUNCH_OK();
        return LIDBOX_OK;
    }
    return lidbox_peak_normalize(signals, starts, lengths, B, dBFS, out, stream);
}

extern "C" int lidbox_peak_normalize(const float* signals, const int64_t* starts, const int64_t* lengths, int B,
                                     float dBFS, float* out, lidbox_stream_t stream) {
    LBX_ARG(signals && starts && lengths && out && B >= 0, "pointers != NULL, B >= 0");
    if (B == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(peak_normalize_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, signals, starts, lengths,
                       powf(10.0f, dBFS / 20.0f), out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_signal_rms(const float* signals, const int64_t* starts, const int64_t* lengths, int B,
                                 float* out_rms, lidbox_stream_t stream) {
    LBX_ARG(signals && starts && lengths && out_rms && B >= 0, "pointers != NULL, B >= 0");
    if (B == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(signal_rms_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, signals, starts, lengths, out_rms);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_snr_mixer(const float* clean, const float* noise, const float* snr_db, int B, long N,
                                float* clean_norm, float* noise_new, float* noisy, lidbox_stream_t stream) {
    LBX_ARG(clean && noise && snr_db && clean_norm && noise_new && noisy, "pointers != NULL");
    LBX_ARG(B >= 0 && N >= 1, "B >= 0, N >= 1");
    if (B == 0) return LIDBOX_OK;
    const bool vec = N % 4 == 0 && ((((uintptr_t)clean) | ((uintptr_t)noise) | ((uintptr_t)clean_norm) |
                                     ((uintptr_t)noise_new) | ((uintptr_t)noisy)) & 15) == 0;
    static const bool no_reg = getenv("LIDBOX_SNR_NO_REG") != nullptr;            // A/B aid
    if (vec && !no_reg && N <= 1024L * 4 * 8) {
        // whole pair in registers: 4 or 8 float4 per thread and signal (16 would spill at the 128-VGPR limit of a
        // 1024-thread workgroup); longer pairs take the three-pass kernel
        hipStream_t st = (hipStream_t)stream;
        if (N <= 1024L * 4 * 4)
            hipLaunchKernelGGL(snr_mixer_reg_kernel<4>, dim3(B), dim3(1024), 0, st, clean, noise, snr_db, (int64_t)N,
                               clean_norm, noise_new, noisy);
        else
            hipLaunchKernelGGL(snr_mixer_reg_kernel<8>, dim3(B), dim3(1024), 0, st, clean, noise, snr_db, (int64_t)N,
                               clean_norm, noise_new, noisy);
        LBX_LAUNCH_OK();
        return LIDBOX_OK;
    }
    hipLaunchKernelGGL(snr_mixer_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, clean, noise, snr_db, (int64_t)N,
                       clean_norm, noise_new, noisy);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_segment_mean(const float* x, const int64_t* segment_offsets, int num_segments, int D, float* out,
                                   lidbox_stream_t stream) {
    LBX_ARG(x && segment_offsets && out && num_segments >= 0 && D >= 1, "pointers != NULL, num_segments >= 0, D >= 1");
    LBX_ARG(num_segments <= 65535, "at most 65535 segments per call");
    if (num_segments == 0) return LIDBOX_OK;
    hipLaunchKernelGGL(segment_mean_kernel, dim3((unsigned)lbx_cdiv(D, 256), (unsigned)num_segments), dim3(256), 0,
                       (hipStream_t)stream, x, segment_offsets, D, out);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}
