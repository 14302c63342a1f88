// attention.hip -- frequency attention of the x-vector variant (gfx950).
//
// Replaces (reference file:line):
//   lidbox/models/clstm.py:31-42                 frequency_attention: softmax over d_f bins, bin-wise scaling
//   lidbox/models/xvector_freq_attention.py:29   its use between frame5 and stats pooling
// The two bias-free Dense layers around it (clstm.py:35-36) are lidbox_gemm_nn / _nt / _tn calls.
//
// One wave64 per frame (row of C channels): the d_f <= 64 logits live one per lane, softmax and its
// backward reduce with wavefront shuffles, the bin weights are handed to the channel loop through 256 B
// of LDS per wave.  Both kernels stream each row once: HBM-bound, 2*C*4 (+ d_f*8) bytes per frame forward,
// 3*C*4 backward.  Deterministic (no atomics).
#include <float.h>

#include "common.h"

namespace {

constexpr int MAX_BINS = 64;
constexpr int MAX_C_BWD = 4096;      // backward stages H and dHw rows in LDS: 4 waves * 2 * C floats (+ tables) <= 160 KiB

__device__ __forceinline__ void build_bin_table(unsigned short* s_bin, int C, int cb) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) s_bin[c] = (unsigned short)(c / cb);
    __syncthreads();
}

// rows x C, 4 waves per workgroup, grid-stride over rows
__global__ __launch_bounds__(256) void freq_attention_fwd_kernel(const float* __restrict__ H,
                                                                 const float* logits, long rows, int C,
                                                                 int d_f, float* F_out,
                                                                 float* __restrict__ Hw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_F = reinterpret_cast<float*>(smem);                               // [4][64]
    unsigned short* s_bin = reinterpret_cast<unsigned short*>(smem + 4 * MAX_BINS * 4);
    build_bin_table(s_bin, C, C / d_f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* wF = s_F + wave * MAX_BINS;
    const bool vec = (C & 3) == 0;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float z = lane < d_f ? logits[row * d_f + lane] : -FLT_MAX;
        const float m = wave_max(z);
        const float e = lane < d_f ? __expf(z - m) : 0.f;
        const float f = e / wave_sum(e);
        wave_lds_sync();                                   // previous row's readers are done
        if (lane < d_f) {
            F_out[row * d_f + lane] = f;
            wF[lane] = f;
        }
        wave_lds_sync();
        const float* h = H + row * C;
        float* o = Hw + row * C;
        if (vec) {
            for (int c = lane * 4; c < C; c += 256) {
                float4 v = *reinterpret_cast<const float4*>(h + c);
                v.x *= wF[s_bin[c]];
                v.y *= wF[s_bin[c + 1]];
                v.z *= wF[s_bin[c + 2]];
                v.w *= wF[s_bin[c + 3]];
                *reinterpret_cast<float4*>(o + c) = v;
            }
        } else {
            for (int c = lane; c < C; c += 64) o[c] = h[c] * wF[s_bin[c]];
        }
    }
}

__global__ __launch_bounds__(256) void freq_attention_bwd_kernel(const float* __restrict__ H,
                                                                 const float* __restrict__ F,
                                                                 const float* __restrict__ dHw, long rows, int C,
                                                                 int d_f, int relu_mask,
                                                                 float* __restrict__ dlogits,
                                                                 float* __restrict__ dH) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_F = reinterpret_cast<float*>(smem);                               // [4][64]
    float* s_rows = s_F + 4 * MAX_BINS;                                        // [4][2][C]
    unsigned short* s_bin = reinterpret_cast<unsigned short*>(s_rows + 8 * C);
    const int cb = C / d_f;
    build_bin_table(s_bin, C, cb);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* wF = s_F + wave * MAX_BINS;
    float* wH = s_rows + (long)wave * 2 * C;
    float* wD = wH + C;
    const bool vec = (C & 3) == 0;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float* h = H + row * C;
        const float* d = dHw + row * C;
        wave_lds_sync();                                   // previous row's readers are done
        if (vec) {
            for (int c = lane * 4; c < C; c += 256) {
                *reinterpret_cast<float4*>(wH + c) = *reinterpret_cast<const float4*>(h + c);
                *reinterpret_cast<float4*>(wD + c) = *reinterpret_cast<const float4*>(d + c);
            }
        } else {
            for (int c = lane; c < C; c += 64) { wH[c] = h[c]; wD[c] = d[c]; }
        }
        const float f = lane < d_f ? F[row * d_f + lane] : 0.f;
        if (lane < d_f) wF[lane] = f;
        wave_lds_sync();
        // dF[bin] = sum over the bin's channels of dHw * H   (lane = bin; stride cb across lanes)
        float dF = 0.f;
        if (lane < d_f) {
            const float* ph = wH + lane * cb;
            const float* pd = wD + lane * cb;
            for (int i = 0; i < cb; ++i) dF = fmaf(pd[i], ph[i], dF);
        }
        const float s = wave_sum(f * dF);                  // inactive lanes contribute 0
        if (lane < d_f) dlogits[row * d_f + lane] = f * (dF - s);
        float* o = dH + row * C;
        for (int c = lane; c < C; c += 64) {
            const float v = wD[c] * wF[s_bin[c]];
            o[c] = (relu_mask && !(wH[c] > 0.f)) ? 0.f : v;
        }
    }
}

int check_attention(const char* fn, long rows, int C, int d_f) {
    if (rows < 0 || C < 1 || d_f < 1 || d_f > MAX_BINS || C % d_f != 0 || C > 65535) {
        lidbox_set_error("%s: invalid argument: rows >= 0, 1 <= d_f <= 64, C %% d_f == 0, C <= 65535", fn);
        return LIDBOX_E_INVALID;
    }
    return LIDBOX_OK;
}

unsigned rows_grid(long rows) {
    long g = lbx_cdiv(rows, 4);
    if (g > 256 * 8) g = 256 * 8;
    return (unsigned)g;
}

}  // namespace

extern "C" int lidbox_freq_attention_fwd(const float* H, const float* logits, long rows, int C, int d_f,
                                         float* F_out, float* Hw, lidbox_stream_t stream) {
    LBX_ARG(H && logits && F_out && Hw, "H, logits, F_out, Hw != NULL");
    if (check_attention(__func__, rows, C, d_f)) return LIDBOX_E_INVALID;
    if (rows == 0) return LIDBOX_OK;
    const size_t lds = 4 * MAX_BINS * 4 + (size_t)C * 2;
    hipLaunchKernelGGL(freq_attention_fwd_kernel, dim3(rows_grid(rows)), dim3(256), lds, (hipStream_t)stream, H,
                       logits, rows, C, d_f, F_out, Hw);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

extern "C" int lidbox_freq_attention_bwd(const float* H, const float* F, const float* dHw, long rows, int C,
                                         int d_f, int relu_mask, float* dlogits, float* dH,
                                         lidbox_stream_t stream) {
    LBX_ARG(H && F && dHw && dlogits && dH, "H, F, dHw, dlogits, dH != NULL");
    if (check_attention(__func__, rows, C, d_f)) return LIDBOX_E_INVALID;
    LBX_ARG(C <= MAX_C_BWD, "C <= 4096");
    if (rows == 0) return LIDBOX_OK;
    const size_t lds = 4 * MAX_BINS * 4 + (size_t)8 * C * 4 + (size_t)C * 2;
    if (lds > 64 * 1024) {      // above the default dynamic-LDS limit: raise it once (a CU has 160 KiB)
        static const hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void*>(freq_attention_bwd_kernel),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        LBX_HIP(raised);
    }
    hipLaunchKernelGGL(freq_attention_bwd_kernel, dim3(rows_grid(rows)), dim3(256), lds, (hipStream_t)stream, H, F,
                       dHw, rows, C, d_f, relu_mask, dlogits, dH);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}
