// features.hip -- waveform -> |STFT|^p -> mel -> log -> MFCC for gfx950.
//
// Replaces (reference file:line, relative to the lidbox checkout):
//   lidbox/features/audio.py:185-189   ms_to_frames
//   lidbox/features/audio.py:219-230   spectrograms  (tf.signal.stft + abs + pow)
//   lidbox/features/mel_ops.py:11-75   linear_to_mel_weight_matrix (non-endpoint _linspace)
//   lidbox/features/audio.py:247-261   linear_to_mel (tensordot)
//   lidbox/data/tf_utils.py:172-185    spectrogram -> mel -> ln(x+1e-6) -> MFCC stages
//
// Fused fast path (fft_length == 512, frame_length <= 512): ONE kernel reads each sample once
// from HBM and writes only the final [B,T,C] features.
//   * 8 lanes per frame, 8 frames per wave64; a wave owns 8 consecutive frames of one utterance.
//   * the 512-point real FFT is a 256-point complex FFT of the packed signal, done as
//     16 x 16: an in-register radix-16 DFT, one exchange through LDS (two half passes so a
//     wave needs 9 KiB), twiddle, second in-register radix-16 DFT.
//   * lane q ends up holding columns k1 = q and 16-q, so both members of every conjugate pair
//     (k, 256-k) needed by the real-FFT untangling live in the same lane: no shuffles.
//   * the mel filterbank is banded (each FFT bin feeds <= 2 triangles; 464 non-zeros of
//     257x40): a per-band CSR dot product from LDS replaces the dense GEMM, which would
//     otherwise make this kernel compute-bound at the fp32 vector rate.
//   * ln(x + 1e-6) and the DCT-II rows for MFCC are epilogues; outputs are staged through
//     LDS so the global stores are contiguous runs.
// Roofline: HBM.  Algorithmic bytes per utterance (16 kHz x 2 s, 40 mel): 32000*4 read +
// 198*40*4 written = 159 680 B.
//
// Generic path (any fft_length / frame_length): plain DFT against a twiddle table, one
// workgroup per frame, spectrogram through a caller-provided workspace.  Correct, not fast.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <vector>

#include <type_traits>

#include "common.h"

// ------------------------------------------------------------------------------------------------
// host-side constants
// ------------------------------------------------------------------------------------------------
extern "C" int lidbox_ms_to_frames(int sample_rate, int ms) {
    // float32, left to right, truncation (audio.py:189)
    volatile float a = (float)sample_rate * 1e-3f;
    volatile float b = a * (float)ms;
    return (int)b;
}

extern "C" int lidbox_num_frames(int num_samples, int frame_length, int frame_step) {
    if (frame_length <= 0 || frame_step <= 0 || num_samples < frame_length) return 0;
    return 1 + (num_samples - frame_length) / frame_step;
}

extern "C" int lidbox_hann_window(int L, float* out) {
    LBX_ARG(L >= 1 && out, "window_length >= 1 and out != NULL");
    if (L == 1) { out[0] = 1.0f; return LIDBOX_OK; }
    // tf.signal.hann_window(periodic=True): denominator L + even - 1, float32 arithmetic
    const int even = 1 - (L % 2);
    const float n = (float)(L + even - 1);
    const float two_pi = (float)(2.0 * M_PI);
    for (int i = 0; i < L; ++i) {
        const float arg = two_pi * (float)i / n;
        out[i] = 0.5f - 0.5f * (float)cos((double)arg);
    }
    return LIDBOX_OK;
}

static inline float hz_to_mel_f32(float hz) {   // mel_ops.py:23-25
    return 1127.0f * logf(1.0f + hz / 700.0f);
}

extern "C" int lidbox_mel_weight_matrix(int M, int F, int sample_rate, float lo_hz, float hi_hz,
                                        float* W) {
    LBX_ARG(M >= 1 && F >= 2 && W, "num_mel_bins >= 1, num_spectrogram_bins >= 2, out != NULL");
    const float nyq = (float)sample_rate / 2.0f;                              // mel_ops.py:39
    const float mel_lo = hz_to_mel_f32(lo_hz), mel_hi = hz_to_mel_f32(hi_hz);
    std::vector<float> edges(M + 2);
    for (int j = 0; j < M + 2; ++j)                                          // _linspace, :11-16
        edges[j] = mel_lo + (mel_hi - mel_lo) * (float)j / (float)(M + 2);
    for (int m = 0; m < M; ++m) W[m] = 0.0f;                                  // DC row, :74-75
    for (int i = 1; i < F; ++i) {
        const float hz = 0.0f + (nyq - 0.0f) * (float)i / (float)F;          // non-endpoint
        const float mel = hz_to_mel_f32(hz);
        for (int m = 0; m < M; ++m) {
            const float lower = (mel - edges[m]) / (edges[m + 1] - edges[m]);         // :64-65
            const float upper = (edges[m + 2] - mel) / (edges[m + 2] - edges[m + 1]); // :66-67
            float w = fminf(lower, upper);
            W[(size_t)i * M + m] = w > 0.0f ? w : 0.0f;                               // :70-71
        }
    }
    return LIDBOX_OK;
}

// ------------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------------
constexpr int BLUESTEIN_MAX_M2 = 16384;     // complex points of one 128 KB LDS buffer
struct lidbox_feat_plan {
    int sample_rate, L, S, nfft, F, M, coef_begin, coef_end, ncoef, nnz;
    float power;
    bool fused_ok;           // nfft == 512 && L <= 512 && tables fit LDS
    int device;
    // device tables
    float*  d_win512;        // [512] 0.5*hann (zero beyond L)           (fused)
    float*  d_win512_pcm;    // [512] the same / 32768: 16-bit PCM sources read in place (tf.audio.decode_wav's scale, audio.py:17-23)
    float2* d_tw256;         // [16 k1][16 n2]  W256^(n2*k1)             (fused)
    float2* d_tw512;         // [256]           W512^k                   (fused)
    float*  d_win;           // [L] hann                                 (generic)
    float2* d_twN;           // [nfft] e^{-2 pi i j / nfft}              (generic)
    int*    d_mel_start;     // [M]
    int*    d_mel_cnt;       // [M]
    int*    d_mel_off;       // [M]
    float*  d_mel_w;         // [nnz]
    int*    d_seg_meta;      // [3][64]: band (-1 = unused lane), first bin, index-in-band | segments-of-band << 8   (fused)
    float*  d_seg_w;         // [seg_len][64] weights of lane's segment, zero padded                             (fused)
    int     seg_len, seg_steps;   // bins per segment; shuffle steps that combine a band's segments
    bool    seg_ok;          // the bands split into <= 64 segments of <= 32 bins
    float*  d_dct;           // [M][ncoef]
    // Bluestein (chirp-z) path for fft_length that is not a power of two (bs_m2 == 0: not available, direct DFT)
    int     bs_m2;           // convolution length: the power of two >= min(L, nfft) + F - 1 (<= BLUESTEIN_MAX_M2)
    float2* d_bs_chirp;      // [nfft]  w_n = e^{-i pi n^2 / nfft}
    float2* d_bs_bhat;       // [bs_m2] FFT of the wrapped conjugate chirp, scaled by 1 / bs_m2
    float2* d_bs_tw;         // [bs_m2] e^{-2 pi i j / bs_m2}
    void*   d_block;         // single allocation backing all of the above
};

extern "C" void lidbox_feat_plan_destroy(lidbox_feat_plan* p) {
    if (!p) return;
    if (p->d_block) (void)hipFree(p->d_block);
    delete p;
}

extern "C" int lidbox_feat_plan_create(int sample_rate, int L, int S, int nfft, float power,
                                       int M, float fmin, float fmax, int coef_begin, int coef_end,
                                       lidbox_feat_plan** out) {
    LBX_ARG(out, "out_plan != NULL");
    LBX_ARG(sample_rate > 0 && L >= 1 && S >= 1 && nfft >= 1, "sample_rate, frame_length, frame_step, fft_length > 0");
    LBX_ARG(nfft <= 16384, "fft_length <= 16384");
    LBX_ARG(M >= 1 && M <= 1024, "1 <= num_mel_bins <= 1024");
    LBX_ARG(power > 0.0f, "power > 0");
    if (coef_begin < 0) coef_begin = 0;
    if (coef_end > M) coef_end = M;
    if (coef_end < coef_begin) coef_end = coef_begin;      // empty MFCC slice: only the MFCC kind is unavailable on this plan

    lidbox_feat_plan* p = new lidbox_feat_plan();
    memset(p, 0, sizeof(*p));
    p->sample_rate = sample_rate; p->L = L; p->S = S; p->nfft = nfft; p->F = nfft / 2 + 1;
    p->M = M; p->coef_begin = coef_begin; p->coef_end = coef_end; p->ncoef = coef_end - coef_begin;
    p->power = power;
    LBX_HIP(hipGetDevice(&p->device));
    const int F = p->F;
    const int Leff = L < nfft ? L : nfft;       // tf.signal.rfft crops frames longer than fft_length

    // mel matrix (float32 op order) -> banded CSR
    std::vector<float> W((size_t)F * M);
    int rc = lidbox_mel_weight_matrix(M, F, sample_rate, fmin, fmax, W.data());
    if (rc) { delete p; return rc; }
    std::vector<int> start(M), cnt(M), off(M);
    std::vector<float> wts;
    for (int m = 0; m < M; ++m) {
        int lo = -1, hi = -1;
        for (int i = 0; i < F; ++i)
            if (W[(size_t)i * M + m] != 0.0f) { if (lo < 0) lo = i; hi = i; }
        start[m] = lo < 0 ? 0 : lo;
        cnt[m] = lo < 0 ? 0 : hi - lo + 1;
        off[m] = (int)wts.size();
        for (int i = 0; i < cnt[m]; ++i) wts.push_back(W[(size_t)(start[m] + i) * M + m]);
    }
    p->nnz = (int)wts.size();
    if (wts.empty()) wts.push_back(0.0f);

    // Segmented form for the fused kernel: every band is cut into ceil(cnt / seg_len) runs of consecutive bins,
    // one wave lane per run (all 8 frames of a tile), with the smallest seg_len that fits the 64 lanes.
    std::vector<int> seg_meta(3 * 64, 0);
    std::vector<float> seg_w;
    p->seg_ok = false;
    for (int sl = 1; sl <= 32 && !p->seg_ok; ++sl) {
        int total = 0, max_ns = 1;
        for (int m = 0; m < M; ++m) {
            const int ns = cnt[m] > 0 ? (cnt[m] + sl - 1) / sl : 1;
            total += ns;
            if (ns > max_ns) max_ns = ns;
        }
        if (total > 64) continue;
        p->seg_ok = true;
        p->seg_len = sl;
        p->seg_steps = 0;
        while ((1 << p->seg_steps) < max_ns) ++p->seg_steps;
        seg_w.assign((size_t)sl * 64, 0.0f);
        for (int i = 0; i < 64; ++i) seg_meta[i] = -1;
        int lane = 0;
        for (int m = 0; m < M; ++m) {
            const int ns = cnt[m] > 0 ? (cnt[m] + sl - 1) / sl : 1;
            for (int i = 0; i < ns; ++i, ++lane) {
                const int b0 = start[m] + i * sl;
                seg_meta[lane] = m;
                seg_meta[64 + lane] = b0;
                seg_meta[128 + lane] = i | (ns << 8);
                for (int j = 0; j < sl && i * sl + j < cnt[m]; ++j) seg_w[(size_t)j * 64 + lane] = wts[off[m] + i * sl + j];
            }
        }
    }
    if (seg_w.empty()) seg_w.assign(64, 0.0f);

    // window
    std::vector<float> win(L), win512(516, 0.0f);      // [512..515] stay zero: the load target of masked lanes
    lidbox_hann_window(L, win.data());
    for (int i = 0; i < Leff && i < 512; ++i) win512[i] = 0.5f * win[i];
    std::vector<float> win512_pcm(512);
    for (int i = 0; i < 512; ++i) win512_pcm[i] = win512[i] * (1.0f / 32768.0f);      // exact: a power of two
    // twiddles (double -> float)
    std::vector<float2> tw256(256), tw512(256), twN(nfft);
    for (int k1 = 0; k1 < 16; ++k1)
        for (int n2 = 0; n2 < 16; ++n2) {
            const double a = -2.0 * M_PI * (double)(n2 * k1) / 256.0;
            tw256[k1 * 16 + n2] = make_float2((float)cos(a), (float)sin(a));
        }
    for (int k = 0; k < 256; ++k) {
        const double a = -2.0 * M_PI * (double)k / 512.0;
        tw512[k] = make_float2((float)cos(a), (float)sin(a));
    }
    for (int j = 0; j < nfft; ++j) {
        const double a = -2.0 * M_PI * (double)j / (double)nfft;
        twN[j] = make_float2((float)cos(a), (float)sin(a));
    }
    // Bluestein tables (fft_length not a power of two): X_k = w_k sum_n (x_n w_n) conj(w)_{k-n}, w_n = e^{-i pi n^2 / N} -- a circular
    // convolution done with power-of-two FFTs.  Only the bins k < F = N / 2 + 1 of a frame that is zero from Leff on are wanted, so the
    // chirp is needed at k - n in (-Leff, F) and the convolution length is the power of two >= Leff + F - 1 (not 2 N - 1): 16 384 points
    // (the largest one LDS buffer holds) cover every fft_length <= 10 922 whatever the frame, and all of them up to the plan's 16 384
    // for frames of <= 8 192 samples.  n^2 is reduced mod 2 N in integers (exact phase).
    int m2 = 0;
    std::vector<float2> bs_chirp(1), bs_bhat(1), bs_tw(1);
    if (nfft >= 3 && (nfft & (nfft - 1)) != 0 && Leff + p->F - 1 <= BLUESTEIN_MAX_M2) {
        m2 = 1;
        while (m2 < Leff + p->F - 1) m2 <<= 1;
        bs_chirp.resize(nfft);
        std::vector<double> br(m2, 0.0), bi(m2, 0.0);
        for (int n = 0; n < nfft; ++n) {
            const long q = ((long)n * n) % (2L * nfft);
            const double a = M_PI * (double)q / (double)nfft;
            bs_chirp[n] = make_float2((float)cos(a), (float)-sin(a));
            if (n < p->F) { br[n] = cos(a); bi[n] = sin(a); }                       // conj(w_n) at k - n = n >= 0 ...
            if (n > 0 && n < Leff) { br[m2 - n] = cos(a); bi[m2 - n] = sin(a); }    // ... and at k - n = -n (the chirp is even)
        }
        // iterative radix-2 FFT of b in float64 on the host (once per plan)
        for (int i = 1, j = 0; i < m2; ++i) {
            int bit = m2 >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) { std::swap(br[i], br[j]); std::swap(bi[i], bi[j]); }
        }
        for (int len = 2; len <= m2; len <<= 1) {
            const double ang = -2.0 * M_PI / (double)len;
            for (int i = 0; i < m2; i += len)
                for (int k = 0; k < len / 2; ++k) {
                    const double wr = cos(ang * k), wi = sin(ang * k);
                    const double ur = br[i + k], ui = bi[i + k];
                    const double vr = br[i + k + len / 2] * wr - bi[i + k + len / 2] * wi;
                    const double vi = br[i + k + len / 2] * wi + bi[i + k + len / 2] * wr;
                    br[i + k] = ur + vr; bi[i + k] = ui + vi;
                    br[i + k + len / 2] = ur - vr; bi[i + k + len / 2] = ui - vi;
                }
        }
        bs_bhat.resize(m2);
        bs_tw.resize(m2);
        for (int j = 0; j < m2; ++j) {
            bs_bhat[j] = make_float2((float)(br[j] / m2), (float)(bi[j] / m2));
            const double a = -2.0 * M_PI * (double)j / (double)m2;
            bs_tw[j] = make_float2((float)cos(a), (float)sin(a));
        }
    }
    p->bs_m2 = m2;
    // DCT rows: c_k = sqrt(2/M) sum_n x_n cos(pi k (2n+1) / (2M)), k in [coef_begin, coef_end)
    std::vector<float> dct((size_t)M * p->ncoef + (p->ncoef == 0 ? 1 : 0));
    for (int n = 0; n < M; ++n)
        for (int c = 0; c < p->ncoef; ++c) {
            const int k = coef_begin + c;
            dct[(size_t)n * p->ncoef + c] =
                (float)(2.0 * cos(M_PI * (double)k * (2.0 * n + 1.0) / (2.0 * M)) / sqrt(2.0 * M));
        }

    // one device block, 256-byte aligned pieces
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o_win512 = 0;
    size_t o_tw256 = o_win512 + al(516 * 4);
    size_t o_tw512 = o_tw256 + al(256 * 8);
    size_t o_win = o_tw512 + al(256 * 8);
    size_t o_twN = o_win + al((size_t)L * 4);
    size_t o_ms = o_twN + al((size_t)nfft * 8);
    size_t o_mc = o_ms + al((size_t)M * 4);
    size_t o_mo = o_mc + al((size_t)M * 4);
    size_t o_mw = o_mo + al((size_t)M * 4);
    size_t o_dct = o_mw + al(wts.size() * 4);
    size_t o_sm = o_dct + al(dct.size() * 4);
    size_t o_sw = o_sm + al(seg_meta.size() * 4);
    size_t o_bc = o_sw + al(seg_w.size() * 4);
    size_t o_bh = o_bc + al(bs_chirp.size() * 8);
    size_t o_bt = o_bh + al(bs_bhat.size() * 8);
    size_t o_wp = o_bt + al(bs_tw.size() * 8);
    size_t total = o_wp + al(512 * 4);
    char* blk = nullptr;
    if (hipMalloc((void**)&blk, total) != hipSuccess) {
        lidbox_set_error("lidbox_feat_plan_create: hipMalloc(%zu) failed", total);
        delete p;
        return LIDBOX_E_ALLOC;
    }
    p->d_block = blk;
#define LBX_UP(off, vec, bytes)                                                             \
    if (hipMemcpy(blk + (off), (vec), (bytes), hipMemcpyHostToDevice) != hipSuccess) {      \
        lidbox_set_error("lidbox_feat_plan_create: hipMemcpy failed");                      \
        lidbox_feat_plan_destroy(p);                                                        \
        return LIDBOX_E_LAUNCH;                                                             \
    }
    LBX_UP(o_win512, win512.data(), 516 * 4);
    LBX_UP(o_tw256, tw256.data(), 256 * 8);
    LBX_UP(o_tw512, tw512.data(), 256 * 8);
    LBX_UP(o_win, win.data(), (size_t)L * 4);
    LBX_UP(o_twN, twN.data(), (size_t)nfft * 8);
    LBX_UP(o_ms, start.data(), (size_t)M * 4);
    LBX_UP(o_mc, cnt.data(), (size_t)M * 4);
    LBX_UP(o_mo, off.data(), (size_t)M * 4);
    LBX_UP(o_mw, wts.data(), wts.size() * 4);
    LBX_UP(o_dct, dct.data(), dct.size() * 4);
    LBX_UP(o_sm, seg_meta.data(), seg_meta.size() * 4);
    LBX_UP(o_sw, seg_w.data(), seg_w.size() * 4);
    LBX_UP(o_bc, bs_chirp.data(), bs_chirp.size() * 8);
    LBX_UP(o_bh, bs_bhat.data(), bs_bhat.size() * 8);
    LBX_UP(o_bt, bs_tw.data(), bs_tw.size() * 8);
    LBX_UP(o_wp, win512_pcm.data(), 512 * 4);
#undef LBX_UP
    p->d_win512 = (float*)(blk + o_win512);
    p->d_tw256 = (float2*)(blk + o_tw256);
    p->d_tw512 = (float2*)(blk + o_tw512);
    p->d_win = (float*)(blk + o_win);
    p->d_twN = (float2*)(blk + o_twN);
    p->d_mel_start = (int*)(blk + o_ms);
    p->d_mel_cnt = (int*)(blk + o_mc);
    p->d_mel_off = (int*)(blk + o_mo);
    p->d_mel_w = (float*)(blk + o_mw);
    p->d_dct = (float*)(blk + o_dct);
    p->d_seg_meta = (int*)(blk + o_sm);
    p->d_seg_w = (float*)(blk + o_sw);
    p->d_bs_chirp = (float2*)(blk + o_bc);
    p->d_bs_bhat = (float2*)(blk + o_bh);
    p->d_bs_tw = (float2*)(blk + o_bt);
    p->d_win512_pcm = (float*)(blk + o_wp);

    // fused path limits: LDS tables must leave room for >= 2 workgroups per CU
    p->fused_ok = (nfft == 512) && (L <= 512) && (M <= 64) && (p->nnz <= 1024) &&
                  ((size_t)M * p->ncoef <= 1024);
    *out = p;
    return LIDBOX_OK;
}

extern "C" int lidbox_feat_plan_channels(const lidbox_feat_plan* p, int kind) {
    if (!p) return -1;
    switch (kind) {
        case LIDBOX_FEAT_SPECTROGRAM: return p->F;
        case LIDBOX_FEAT_MEL:
        case LIDBOX_FEAT_LOGMEL: return p->M;
        case LIDBOX_FEAT_MFCC: return p->ncoef;
    }
    return -1;
}

// ------------------------------------------------------------------------------------------------
// fused kernel
// ------------------------------------------------------------------------------------------------
namespace {

#ifndef LBX_FEAT_NO_HOIST
#define LBX_FEAT_NO_HOIST 1                  // keep the untangle's per-lane addresses out of loop-invariant registers
#endif
#ifndef LBX_FEAT_SEGMEL
#define LBX_FEAT_SEGMEL 1                   // 0: always use the per-band CSR mel loop (A/B aid)
#endif
#ifndef LBX_FEAT_FAST_INTERIOR
#define LBX_FEAT_FAST_INTERIOR 1            // 0: always take the guarded sample-load path (A/B aid)
#endif
#ifndef LBX_FEAT_WAVES
#define LBX_FEAT_WAVES 3                    // waves per SIMD the register allocator aims for
#endif
constexpr float LOG_EPS = 1e-6f;           // tf_utils.py:178
constexpr int EXCH_ROW = 144;              // bytes: 8 x (2 complex) + 16 pad  -> conflict-free b128
constexpr int EXCH_FRAME = 8 * EXCH_ROW;   // 1152 B per frame per half pass
constexpr int P_STRIDE = 264;              // floats per frame in the power buffer (>= 257, = 8 mod 32)
constexpr int PT_ROWS = 257 + (256 >> 3) + 1; // transposed power buffer: row(bin) = bin + bin/8 (one skew row every 8 bins)
constexpr int WAVE_SCRATCH = PT_ROWS * 32;     // 9280 B >= 8 * EXCH_FRAME = 9216 B >= 8*264*4 (the power buffers alias the exchange)

struct FusedArgs {
    const float* signals;
    long sig_stride;
    int B, N, T, L, S;
    float power_half;           // power / 2 (only used when power != 2)
    int M, ncoef, nnz;
    int seg_len, seg_steps;     // segmented mel (SEGMEL kernels)
    int dct_runs, dct_len;      // MFCC under SEGMEL: runs per coefficient (ncoef * dct_runs <= 64), bands per run
    const int* seg_meta;
    const float* seg_w;
    const float* win512;
    const float2* tw256;
    const float2* tw512;
    const int* mel_start;
    const int* mel_cnt;
    const int* mel_off;
    const float* mel_w;
    const float* dct;
    float* out;
    long out_bs;                // floats between consecutive utterances in out
    unsigned short* out16;      // SHADOW instantiations: bfloat16 copy of the output (round-to-nearest-even) at the same ELEMENT offsets as out
    int* nonfinite;             // may be NULL: set to 1 when a stored feature value is not finite (a plain store: every writer writes the same
                                // value, and the word may live in pinned host memory)
    int tiles_per_utt;          // ceil(T / 8)
    long ntiles;                // B * tiles_per_utt
    int iters;                  // tiles per wave
    int tiles_per_wg;           // consecutive tiles one workgroup owns (its waves take them round-robin)
    unsigned nwg;
#if defined(LBX_FEAT_TIMING) || defined(LBX_FEAT_TIMELINE)
    long long* stamps;          // [nwg*NW][16] s_memtime samples of each wave's LBX_FEAT_TIMING-th tile, 12 = kernel entry, 13 = tables staged,
                                // 14 = wave exit, 15 = s_memrealtime at exit (debug builds only)
#endif
};

// Debug builds.  -DLBX_FEAT_TIMING=k: per-phase s_memtime stamps of every wave's k-th tile (scheduling fences around each stamp:
// the phase split is approximate and the build is slower).  -DLBX_FEAT_TIMELINE=k: only the kernel-level stamps and the k-th
// tile's first and last, no fences (tools/feat_timeline.py).
#ifdef LBX_FEAT_TIMELINE
#define LBX_FEAT_TIMING LBX_FEAT_TIMELINE
#define LBX_STAMP_FENCE() do { } while (0)
#define LBX_STAMP_ON(i) ((i) == 0 || (i) == 8)
#else
#define LBX_STAMP_FENCE() __builtin_amdgcn_sched_barrier(0)
#define LBX_STAMP_ON(i) true
#endif
#ifdef LBX_FEAT_TIMING
#define LBX_STAMP(i)                                                                              \
    do {                                                                                          \
        LBX_STAMP_FENCE();                                                                        \
        if (LBX_STAMP_ON(i) && it == LBX_FEAT_TIMING && lane == 0 && a.stamps)                    \
            a.stamps[((long)blockIdx.x * NW + wave) * 16 + (i)] = (long long)__builtin_amdgcn_s_memtime(); \
        LBX_STAMP_FENCE();                                                                        \
    } while (0)
#define LBX_STAMP_K(i)                                                                            \
    do {                                                                                          \
        LBX_STAMP_FENCE();                                                                        \
        if ((threadIdx.x & 63) == 0 && a.stamps)                                                  \
            a.stamps[((long)blockIdx.x * NW + (threadIdx.x >> 6)) * 16 + (i)] =                   \
                ((i) == 15 || (i) == 11) ? (long long)__builtin_amdgcn_s_memrealtime() : (long long)__builtin_amdgcn_s_memtime(); \
        LBX_STAMP_FENCE();                                                                        \
    } while (0)
#else
#define LBX_STAMP(i) do { } while (0)
#define LBX_STAMP_K(i) do { } while (0)
#endif

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 w) {
    return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
}

// forward 4-point DFT in place: (x0,x1,x2,x3) -> (X0,X1,X2,X3)
__device__ __forceinline__ void dft4(float2& x0, float2& x1, float2& x2, float2& x3) {
    const float2 t0 = cadd(x0, x2), t1 = csub(x0, x2), t2 = cadd(x1, x3), t3 = csub(x1, x3);
    x0 = cadd(t0, t2);
    x2 = csub(t0, t2);
    x1 = make_float2(t1.x + t3.y, t1.y - t3.x);   // t1 - i*t3
    x3 = make_float2(t1.x - t3.y, t1.y + t3.x);   // t1 + i*t3
}

// the same with x3 = 0 on input (frames shorter than the transform: the window is zero there)
__device__ __forceinline__ void dft4_z3(float2& x0, float2& x1, float2& x2, float2& x3) {
    const float2 t0 = cadd(x0, x2), t1 = csub(x0, x2), x1c = x1;
    x0 = cadd(t0, x1c);
    x2 = csub(t0, x1c);
    x1 = make_float2(t1.x + x1c.y, t1.y - x1c.x);   // t1 - i*x1
    x3 = make_float2(t1.x - x1c.y, t1.y + x1c.x);   // t1 + i*x1
}

// register that holds X[k] after dft16 (digit-reversed)
#define R16(k) (4 * ((k) & 3) + ((k) >> 2))

// forward 16-point DFT in place; input v[n] natural order, output X[k] in v[R16(k)].  NIN (13 or 16): inputs v[NIN ...] are zero and
// are never read (pass 1 of a frame of <= 32 * NIN samples: 25 ms at 16 kHz is 400 <= 416).
template <int NIN = 16>
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (12 + b < NIN) dft4(v[b], v[4 + b], v[8 + b], v[12 + b]);
        else dft4_z3(v[b], v[4 + b], v[8 + b], v[12 + b]);
    }
    // twiddle Y[b][c] (held in v[4c+b]) by W16^(b*c)
    v[4 * 1 + 1] = cmul(v[4 * 1 + 1], make_float2(C1, -S1));                              // W^1
    v[4 * 2 + 1] = make_float2((v[9].x + v[9].y) * H, (v[9].y - v[9].x) * H);             // W^2
    v[4 * 3 + 1] = cmul(v[4 * 3 + 1], make_float2(S1, -C1));                              // W^3
    v[4 * 1 + 2] = make_float2((v[6].x + v[6].y) * H, (v[6].y - v[6].x) * H);             // W^2
    v[4 * 2 + 2] = make_float2(v[10].y, -v[10].x);                                        // W^4 = -i
    v[4 * 3 + 2] = make_float2((v[14].y - v[14].x) * H, -(v[14].x + v[14].y) * H);        // W^6
    v[4 * 1 + 3] = cmul(v[4 * 1 + 3], make_float2(S1, -C1));                              // W^3
    v[4 * 2 + 3] = make_float2((v[11].y - v[11].x) * H, -(v[11].x + v[11].y) * H);        // W^6
    v[4 * 3 + 3] = cmul(v[4 * 3 + 3], make_float2(-C1, S1));                              // W^9
#pragma unroll
    for (int c = 0; c < 4; ++c) dft4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
}

// One conjugate pair of the real-FFT untangling.  zk = Z'[k], zm = Z'[256-k] (Z' already carries
// the 1/2 through the window), w = W512^k.  Returns |X[k]|^2 and |X[256-k]|^2.
__device__ __forceinline__ void untangle(float2 zk, float2 zm, float2 w, float& pk, float& pm) {
    const float er = zk.x + zm.x, ei = zk.y - zm.y;       // e = zk + conj(zm)
    const float orr = zk.y + zm.y, oi = zm.x - zk.x;      // o = (zk - conj(zm)) / i
    const float wr = w.x * orr - w.y * oi, wi = w.x * oi + w.y * orr;
    const float ar = er + wr, ai = ei + wi, br = er - wr, bi = ei - wi;
    pk = ar * ar + ai * ai;
    pm = br * br + bi * bi;
}

// size of the mel tables in LDS (floats): CSR form [3M + nnz] or segmented form [3*64 + seg_len*64]
__host__ __device__ inline int mel_table_floats(bool segmel, int M, int nnz, int seg_len) {
    return segmel ? 3 * 64 + seg_len * 64 : 3 * M + nnz;
}


// value of lane + d (d < 64; callers only use it where lane + d <= 63).  __shfl_down derives the lane index itself; hipcc hoists that
// out of the tile loop as one more live register (and spills it at the streaming kernel's cap): the caller's own copy is used instead.
__device__ __forceinline__ float lane_down(const float v, const int d, const int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane + d) & 63) << 2, __builtin_bit_cast(int, v)));
}

// ---- tile pieces shared by fused_feat512_kernel and feat512_stream_kernel -------------------------------------------------------
// Steps 2-6 of a tile.  za / zb: the windowed packed samples of lane q (n2 = 2q / 2q + 1, n1 = 0..15) of frame slot f.  Leaves
// |X[bin]|^power of the wave's 8 frames in its power buffer (wbuf: [8][P_STRIDE], or TRANSPOSED [PT_ROWS][8] with one skew row per
// 8 bins, so that the mel lanes -- one per run of bins -- read their 32-byte rows from different banks).  The exchange uses the same
// bytes.  stamp(i): phase stamps of the timing builds.
template <bool POW2, bool TRANSPOSED, int NIN = 16, typename Stamp>
__device__ __forceinline__ void fft512_power_tile(float2 (&za)[16], float2 (&zb)[16], char* wbuf, const int q, const int f,
                                                  const float2* s_tw256, const float2* s_tw512, const float power_half, Stamp&& stamp) {
    float* s_P = reinterpret_cast<float*>(wbuf);
    // ---- 2. pass 1: DFT16 over n1 for both n2 -> A[n2][k1] in reg R16(k1)
    dft16<NIN>(za);
    dft16<NIN>(zb);
    stamp(2);
    // ---- 3. twiddle by W256^(n2*k1)
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {
        const float4 w = *reinterpret_cast<const float4*>(&s_tw256[k1 * 16 + 2 * q]);
        za[R16(k1)] = cmul(za[R16(k1)], make_float2(w.x, w.y));
        zb[R16(k1)] = cmul(zb[R16(k1)], make_float2(w.z, w.w));
    }
    stamp(3);
    // ---- 4. exchange through LDS in two half passes; lane q ends with
    //         ua = A[0..15][k1 = q], ub = A[0..15][k1 = (16 - q) % 16, or 8 for q = 0]
    float2 ua[16], ub[16];
    char* ex = wbuf + f * EXCH_FRAME;
    wave_lds_sync();                                 // previous tile's readers are done
#ifdef LBX_ABL_NOEXCH
#pragma unroll
    for (int j = 0; j < 16; ++j) { ua[j] = za[j]; ub[j] = zb[j]; }
    (void)ex;
#else
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1)
        *reinterpret_cast<float4*>(ex + k1 * EXCH_ROW + q * 16) =
            make_float4(za[R16(k1)].x, za[R16(k1)].y, zb[R16(k1)].x, zb[R16(k1)].y);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(ex + q * EXCH_ROW + j * 16);
        ua[2 * j] = make_float2(v.x, v.y);
        ua[2 * j + 1] = make_float2(v.z, v.w);
    }
    wave_lds_sync();
#pragma unroll
    for (int k1 = 8; k1 < 16; ++k1)
        *reinterpret_cast<float4*>(ex + (k1 - 8) * EXCH_ROW + q * 16) =
            make_float4(za[R16(k1)].x, za[R16(k1)].y, zb[R16(k1)].x, zb[R16(k1)].y);
    wave_lds_sync();
    const int row2 = (q == 0) ? 0 : 8 - q;           // k1 = 8 for q = 0, else 16 - q
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(ex + row2 * EXCH_ROW + j * 16);
        ub[2 * j] = make_float2(v.x, v.y);
        ub[2 * j + 1] = make_float2(v.z, v.w);
    }
#endif
    stamp(4);
    // ---- 5. pass 2: DFT16 over n2 -> Z'[k1 + 16*k2] in reg R16(k2)
    dft16(ua);
    dft16(ub);
    wave_lds_sync();                                 // exchange reads done before P overwrites
    stamp(5);
    // ---- 6. untangle conjugate pairs, |.|^2, into the per-frame power buffer.
    //   q != 0 : slot s pairs ua[k2=s] (bin q+16s) with ub[k2=15-s] (bin 256-q-16s)
    // The per-lane bins / LDS addresses of this section depend only on q, so the compiler hoists them out of
    // the tile loop as ~15 live registers -- and, at the register cap, spills them to scratch and
    // reloads them here every tile (15 dependent scratch loads in the busiest phase of the kernel).  An opaque
    // copy of q keeps their (cheap) computation inside the loop instead.
    int qv = q;
#if LBX_FEAT_NO_HOIST
    asm volatile("" : "+v"(qv));
#endif
    // Power-buffer addresses: within slots 0-7 and within slots 8-15 the two bins of a slot move by +-16 per slot, and so do their rows
    // of the transposed layout (bin + bin / 8 moves by 18: 16 s is a multiple of 8), so four per-lane bases and immediate offsets
    // replace the per-store address arithmetic (round 4's census: 140 integer instructions per tile).
    auto P_addr = [&](int bin) -> float* { return TRANSPOSED ? s_P + (bin + (bin >> 3)) * 8 + f : s_P + f * P_STRIDE + bin; };
    constexpr int PSTEP = TRANSPOSED ? 18 * 8 : 16;        // floats between the bins of consecutive slots
    const bool q0 = (qv == 0);
    // Lane 0 holds the two self-paired columns (k1 = 0 in ua, k1 = 8 in ub).  Its pairs are laid over the general
    // pattern (ua[s] with ub[15-s]) so that only ONE operand of a slot differs: slots 0-7 pair ub[s] with ub[15-s]
    // (bins 8 + 16 s and 248 - 16 s), slots 8-15 pair ua[s] with ua[16-s] (bins 16 s and 256 - 16 s; s = 8 pairs bin 128
    // with itself), slot 16 pairs ua[0] with itself (bins 0 and 256): 33 selects per tile instead of 65.
    const int qlo = q0 ? 8 : qv;
    float* const lo1 = P_addr(qlo);                  // slot s < 8: bin qlo + 16 s
    float* const hi1 = P_addr(256 - qlo - 112);      //             bin 256 - qlo - 16 s = (this) + 16 (7 - s)
    float* const lo2 = P_addr(qv + 128);             // slot 8 <= s < 16: bin qv + 16 s
    float* const hi2 = P_addr(16 - qv);              //             bin 256 - qv - 16 s = (this) + 16 (15 - s)
    const float2* const twlo = s_tw512 + qlo;
    const float2* const twhi = s_tw512 + qv;
#pragma unroll
    for (int s = 0; s < 17; ++s) {
        float2 zk, zm, w;
        float *pk_at, *pm_at;
        if (s < 8) {
            const float2 g = ua[R16(s)], l0 = ub[R16(s)];
            zk = make_float2(q0 ? l0.x : g.x, q0 ? l0.y : g.y);
            zm = ub[R16(15 - s)];
            w = twlo[16 * s];
            pk_at = lo1 + PSTEP * s;
            pm_at = hi1 + PSTEP * (7 - s);
        } else if (s < 16) {
            zk = ua[R16(s)];
            const float2 g = ub[R16(15 - s)], l0 = ua[R16((16 - s) & 15)];
            zm = make_float2(q0 ? l0.x : g.x, q0 ? l0.y : g.y);
            w = twhi[16 * s];
            pk_at = lo2 + PSTEP * (s - 8);
            pm_at = hi2 + PSTEP * (15 - s);
        } else {
            zk = ua[R16(0)];
            zm = ua[R16(0)];
            w = s_tw512[0];
            pk_at = P_addr(0);
            pm_at = P_addr(256);
        }
        if (s < 16 || q0) {
            float pk, pm;
            untangle(zk, zm, w, pk, pm);
            if (!POW2) {
                pk = __powf(pk, power_half);
                pm = __powf(pm, power_half);
            }
            *pk_at = pk;
            *pm_at = pm;
        }
    }
    wave_lds_sync();
}

// ln(x) for x >= LOG_EPS (mel energies are >= 0, so x + 1e-6 is never denormal): v_log_f32 (log2, 1 ulp) times ln 2 -- what __logf does
// minus its denormal rescue (a compare, a scale and a select per value)
__device__ __forceinline__ float ln_pos(const float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }

// Step 7, segmented banded mel: lane = one run of <= seg_len consecutive bins of one band, for all 8 frames of the tile (two
// 16-byte reads per bin of the transposed power buffer); runs are zero-padded to seg_len so the loop is uniform.  A band's partial
// sums sit in consecutive lanes and are combined in a fixed order with wave shuffles; the band's first lane finishes (log) and
// stages: [8][M] for mel / log-mel, [M][8] for MFCC (what the DCT runs read).  s_stage may alias s_P: every read of the power
// buffer is issued before the first staging write of the same wave.
template <int KIND>
__device__ __forceinline__ void segmel_tile(const int lane, const int seg_len, const int seg_steps, const int M, const int* s_segmeta,
                                            const float* s_segw, const float* s_P, float* s_stage) {
    const int sband = s_segmeta[lane], bin0 = s_segmeta[64 + lane], sinfo = s_segmeta[128 + lane];
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    // chunks of 4 bins: the 12 LDS reads of a chunk are issued before its 32 FMAs; one LDS round trip per chunk instead of one
    // per bin.  A row of the transposed buffer is 32 bytes = two ds_read_b128, and the LDS serves a b128 wave access in groups of 16
    // lanes that cover the 64 banks once only when they hit 16 different 16-byte slots mod 256 bytes: with every lane on the FIRST half
    // of its row a group can reach 8 of the 16 slots (2-way conflicts by construction).  Odd lanes therefore read the second half first
    // -- each group holds 8 even and 8 odd lanes -- and carry frames 4-7 in acc[0..3], 0-3 in acc[4..7] until the swap below.  (Round 5
    // measured this neutral on a kernel that was not LDS-bound, profiles/r05_feature_mel_halfswap_ab.txt; LBX_FEAT_MEL_HALFSWAP=0: off.)
#ifndef LBX_FEAT_MEL_HALFSWAP
#define LBX_FEAT_MEL_HALFSWAP 1
#endif
    const int half0 = (LBX_FEAT_MEL_HALFSWAP && (lane & 1)) ? 4 : 0, half1 = 4 - half0;
    for (int j0 = 0; j0 < seg_len; j0 += 4) {
        float w[4];
        float4 p0[4], p1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u;
            w[u] = (j < seg_len) ? s_segw[min(j, seg_len - 1) * 64 + lane] : 0.f;
            const int bin = min(bin0 + j, 256);
            const float* row = s_P + (bin + (bin >> 3)) * 8;
            p0[u] = *reinterpret_cast<const float4*>(row + half0);
            p1[u] = *reinterpret_cast<const float4*>(row + half1);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] = fmaf(p0[u].x, w[u], acc[0]); acc[1] = fmaf(p0[u].y, w[u], acc[1]);
            acc[2] = fmaf(p0[u].z, w[u], acc[2]); acc[3] = fmaf(p0[u].w, w[u], acc[3]);
            acc[4] = fmaf(p1[u].x, w[u], acc[4]); acc[5] = fmaf(p1[u].y, w[u], acc[5]);
            acc[6] = fmaf(p1[u].z, w[u], acc[6]); acc[7] = fmaf(p1[u].w, w[u], acc[7]);
        }
    }
    if (LBX_FEAT_MEL_HALFSWAP) {
        const bool sw = (lane & 1) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float lo = acc[i], hi = acc[4 + i];
            acc[i] = sw ? hi : lo;
            acc[4 + i] = sw ? lo : hi;
        }
    }
    const int sidx = sinfo & 255, sns = sinfo >> 8;
    for (int st = 0; st < seg_steps; ++st) {
        const int d = 1 << st;
        const bool take = sidx + d < sns;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float t = lane_down(acc[i], d, lane);
            acc[i] += take ? t : 0.f;
        }
    }
    if (sband >= 0 && sidx == 0) {
        if (KIND == LIDBOX_FEAT_MFCC) {
            *reinterpret_cast<float4*>(s_stage + sband * 8) = make_float4(
                ln_pos(acc[0] + LOG_EPS), ln_pos(acc[1] + LOG_EPS), ln_pos(acc[2] + LOG_EPS), ln_pos(acc[3] + LOG_EPS));
            *reinterpret_cast<float4*>(s_stage + sband * 8 + 4) = make_float4(
                ln_pos(acc[4] + LOG_EPS), ln_pos(acc[5] + LOG_EPS), ln_pos(acc[6] + LOG_EPS), ln_pos(acc[7] + LOG_EPS));
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                s_stage[i * M + sband] = (KIND != LIDBOX_FEAT_MEL) ? ln_pos(acc[i] + LOG_EPS) : acc[i];
        }
    }
}

// DCT-II rows of the MFCC kind, same scheme as the mel runs: lane = (coefficient c, run of dct_len bands), all 8 frames at once from
// the [M][8] log-mel tile; a coefficient's dct_runs partial sums sit in consecutive lanes and are combined in a fixed order.
// DCT_REGS: the lane's weights are in wd[] (dct_len <= 8) instead of the LDS table.  Result tile [8][ncoef] at s_coef.
template <bool DCT_REGS, int CH = 4>
__device__ __forceinline__ void segdct_tile(const int lane, const int dct_runs, const int dct_len, const int ncoef, const int M,
                                            const float (&wd)[8], const float* s_dct, const float* s_stage, float* s_coef) {
    static_assert(!DCT_REGS || CH == 4, "register weights come in two groups of four");
    const int c = lane / dct_runs, run = lane - c * dct_runs;
    const bool on = c < ncoef;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    const int jend = DCT_REGS ? 8 : dct_len;
#pragma unroll 2
    for (int j0 = 0; j0 < jend; j0 += CH) {
        if (DCT_REGS && j0 >= dct_len) break;
        float w[CH];
        float4 p0[CH], p1[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int n = run * dct_len + j0 + u;
            const bool in = on && j0 + u < dct_len && n < M;
            const int nn = in ? n : 0;
            if (DCT_REGS) w[u] = j0 == 0 ? wd[u] : wd[4 + u];      // zero where the run has no band
            else w[u] = in ? s_dct[nn * ncoef + c] : 0.f;
            p0[u] = *reinterpret_cast<const float4*>(s_stage + nn * 8);
            p1[u] = *reinterpret_cast<const float4*>(s_stage + nn * 8 + 4);
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            acc[0] = fmaf(p0[u].x, w[u], acc[0]); acc[1] = fmaf(p0[u].y, w[u], acc[1]);
            acc[2] = fmaf(p0[u].z, w[u], acc[2]); acc[3] = fmaf(p0[u].w, w[u], acc[3]);
            acc[4] = fmaf(p1[u].x, w[u], acc[4]); acc[5] = fmaf(p1[u].y, w[u], acc[5]);
            acc[6] = fmaf(p1[u].z, w[u], acc[6]); acc[7] = fmaf(p1[u].w, w[u], acc[7]);
        }
    }
    for (int d = 1; d < dct_runs; d <<= 1) {
        const bool take = run + d < dct_runs;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float t = lane_down(acc[i], d, lane);
            acc[i] += take ? t : 0.f;
        }
    }
    if (on && run == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s_coef[i * ncoef + c] = acc[i];
    }
}

// The staged [8][M] tile (total = valid frames x M floats) out to dst: two float4 per lane where the tile allows it; d16 (SHADOW):
// the bf16 copy the first Conv1D of the bf16-storage path reads (gemm_bf16.hip), written here instead of by a conversion launch over
// the whole feature tensor.  Returns bad, turned NaN if a stored value is not finite (v * 0 is 0 or NaN: the sum cannot overflow).
template <bool SHADOW>
__device__ __forceinline__ float store_mel_tile(const int lane, const int total, float* dst, unsigned short* d16, const float* s_stage, float bad) {
    if ((total & 3) == 0 && (((uintptr_t)dst) & 15) == 0) {
        // 8 x M <= 512 floats (M <= 64 on this path); both reads are issued before the stores
        const int i0 = 4 * lane, i1 = 256 + 4 * lane;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (i0 < total) v0 = *reinterpret_cast<const float4*>(s_stage + i0);
        if (i1 < total) v1 = *reinterpret_cast<const float4*>(s_stage + i1);
        if (i0 < total) *reinterpret_cast<float4*>(dst + i0) = v0;
        if (i1 < total) *reinterpret_cast<float4*>(dst + i1) = v1;
        if constexpr (SHADOW) {
            typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
            const bf4 h0 = {(__bf16)v0.x, (__bf16)v0.y, (__bf16)v0.z, (__bf16)v0.w}, h1 = {(__bf16)v1.x, (__bf16)v1.y, (__bf16)v1.z, (__bf16)v1.w};
            if (i0 < total) *reinterpret_cast<bf4*>(d16 + i0) = h0;
            if (i1 < total) *reinterpret_cast<bf4*>(d16 + i1) = h1;
        }
        bad = fmaf(v0.x, 0.f, bad); bad = fmaf(v0.y, 0.f, bad); bad = fmaf(v0.z, 0.f, bad); bad = fmaf(v0.w, 0.f, bad);
        bad = fmaf(v1.x, 0.f, bad); bad = fmaf(v1.y, 0.f, bad); bad = fmaf(v1.z, 0.f, bad); bad = fmaf(v1.w, 0.f, bad);
    } else {
        for (int i = lane; i < total; i += 64) {
            const float v = s_stage[i];
            dst[i] = v;
            if constexpr (SHADOW) d16[i] = __builtin_bit_cast(unsigned short, (__bf16)v);
            bad = fmaf(v, 0.f, bad);
        }
    }
    return bad;
}

// The round-1 shape: 4-wave workgroups, three per CU, every workgroup stages its own copy of the tables, a wave loads its tile's
// samples with plain (guarded) loads at the top of the tile.  Since round 6 this kernel only serves what feat512_stream_kernel does
// not take: signals that are not 16-byte aligned (VEC4 = false), plans whose bands do not split into 64 runs (SEGMEL = false) or
// whose tables leave the streaming shape fewer than 8 waves.
template <int KIND, bool VEC4, bool POW2, bool SEGMEL>
__global__ __launch_bounds__(256, LBX_FEAT_WAVES) void fused_feat512_kernel(const FusedArgs a) {
    constexpr int NW = 4, NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ---- LDS carve: tables, then one scratch block per wave
    float* s_win = reinterpret_cast<float*>(smem);                    // 2048 B
    float2* s_tw256 = reinterpret_cast<float2*>(smem + 2048);         // 2048 B
    float2* s_tw512 = reinterpret_cast<float2*>(smem + 4096);         // 2048 B
    int* s_mstart = reinterpret_cast<int*>(smem + 6144);             // CSR: start / cnt / off [M], weights [nnz]
    int* s_mcnt = s_mstart + a.M;
    int* s_moff = s_mcnt + a.M;
    float* s_mw = reinterpret_cast<float*>(s_moff + a.M);
    int* s_segmeta = reinterpret_cast<int*>(smem + 6144);            // SEGMEL: [3][64], weights [seg_len][64]
    float* s_segw = reinterpret_cast<float*>(s_segmeta + 192);
    const int mel_floats = mel_table_floats(SEGMEL, a.M, a.nnz, a.seg_len);
    float* s_dct = reinterpret_cast<float*>(smem + 6144) + mel_floats;
    // MFCC under SEGMEL with <= 8 bands per run: a lane's DCT weights live in registers, no table in LDS (the table would
    // push a workgroup past a third of the CU's LDS: 2 instead of 3 workgroups per CU)
    const bool dct_regs = KIND == LIDBOX_FEAT_MFCC && SEGMEL && a.dct_len <= 8;
    const int table_floats = 1536 + mel_floats + ((KIND == LIDBOX_FEAT_MFCC && !dct_regs) ? a.M * a.ncoef : 0);
    const int table_bytes = (table_floats * 4 + 15) & ~15;
    // MFCC under SEGMEL keeps its 8 x ncoef result tile in the (by then dead) power buffer: 3 workgroups per CU still fit
    const int stage_floats = (KIND == LIDBOX_FEAT_SPECTROGRAM) ? 0 : 8 * a.M + ((KIND == LIDBOX_FEAT_MFCC && !SEGMEL) ? 8 * a.ncoef : 0);
    const int wave_bytes = WAVE_SCRATCH + ((stage_floats * 4 + 15) & ~15);

    const int tid = threadIdx.x;
    LBX_STAMP_K(11);
    LBX_STAMP_K(12);
    for (int i = tid; i < 512; i += NT) s_win[i] = a.win512[i];
    if (tid < 256) {
        s_tw256[tid] = a.tw256[tid];
        s_tw512[tid] = a.tw512[tid];
    }
    if (KIND != LIDBOX_FEAT_SPECTROGRAM) {
        if (SEGMEL) {
            if (tid < 192) s_segmeta[tid] = a.seg_meta[tid];
            for (int i = tid; i < a.seg_len * 64; i += NT) s_segw[i] = a.seg_w[i];
        } else {
            for (int i = tid; i < a.M; i += NT) {
                s_mstart[i] = a.mel_start[i];
                s_mcnt[i] = a.mel_cnt[i];
                s_moff[i] = a.mel_off[i];
            }
            for (int i = tid; i < a.nnz; i += NT) s_mw[i] = a.mel_w[i];
        }
        if (KIND == LIDBOX_FEAT_MFCC)
            if (!dct_regs)
                for (int i = tid; i < a.M * a.ncoef; i += NT) s_dct[i] = a.dct[i];
    }
    __syncthreads();
    LBX_STAMP_K(13);

    const int lane = tid & 63, wave = tid >> 6;
    const int q = lane & 7;          // lane within the frame
    const int f = lane >> 3;         // frame slot within the wave
    float wd[8];                     // DCT weights of this lane's (coefficient, run)
#pragma unroll
    for (int u = 0; u < 8; ++u) wd[u] = 0.f;
    if (dct_regs) {
        const int c = lane / a.dct_runs, run = lane - c * a.dct_runs;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int n = run * a.dct_len + u;
            if (c < a.ncoef && u < a.dct_len && n < a.M) wd[u] = a.dct[n * a.ncoef + c];
        }
    }
    char* wbuf = smem + table_bytes + wave * wave_bytes;
    float* s_P = reinterpret_cast<float*>(wbuf);                         // [8][P_STRIDE] or (SEGMEL) [PT_ROWS][8]; aliases exchange
    float* s_stage = reinterpret_cast<float*>(wbuf + WAVE_SCRATCH);      // [8][M] (+ [8][ncoef])
    float bad = 0.f;                                                     // 0, or NaN once a stored value was not finite

    const unsigned chunk = xcd_chunk_id(blockIdx.x, a.nwg);
    const long tile0 = (long)chunk * a.tiles_per_wg;

    for (int it = 0; it < a.iters; ++it) {
        const int local = it * NW + wave;
        const long tile = tile0 + local;
        if (local >= a.tiles_per_wg || tile >= a.ntiles) break;          // wave-uniform
        const int b = (int)(tile / a.tiles_per_utt);
        const int t0 = (int)(tile - (long)b * a.tiles_per_utt) * 8;
        const int t = t0 + f;
        const bool valid = t < a.T;
        const float* src = a.signals + (long)b * a.sig_stride + (long)t * a.S;

        // ---- 1. load + window.  lane q holds n2 = 2q (za) and 2q+1 (zb), n1 = 0..15:
        //         packed sample n = 16*n1 + n2  <->  reals 32*n1 + 4q .. +3
        //         The loads are issued in two batches of 8 with NO control flow around them (a branch
        //         per load makes the compiler wait for each one before issuing the next: 13 serialized
        //         HBM round trips per tile): lanes whose samples lie outside the frame / utterance read
        //         four zeros instead.
        LBX_STAMP(0);
        float2 za[16], zb[16];
        // Interior tiles (every read of the tile's 8 frames, up to sample 511 of the last one, stays inside the
        // utterance -- wave-uniform test; 24 of 25 tiles at 2 s): one base address per lane and immediate offsets,
        // no per-load address selects and no window guards: the samples after a frame are the utterance's own
        // later samples and the window table is zero there.  The guarded path below handles an utterance's last tile.
        // (Round 2 also measured skipping the always-zero samples 416..511 of frames <= 416 samples behind a wave-uniform branch:
        // slower for log-mel, 28.7 vs 28.0 us at B = 256, neutral for MFCC; removed.)
        const bool interior = VEC4 && LBX_FEAT_FAST_INTERIOR && (long)(t0 + 7) * a.S + 512 <= a.N;
        if (interior) {
            const float* base = src + 4 * q;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float4 x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const float4*>(base + 32 * (8 * half + j));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n1 = 8 * half + j;
                    const float4 w = *reinterpret_cast<const float4*>(s_win + 32 * n1 + 4 * q);
                    za[n1] = make_float2(x[j].x * w.x, x[j].y * w.y);
                    zb[n1] = make_float2(x[j].z * w.z, x[j].w * w.w);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float4 x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int idx = 32 * (8 * half + j) + 4 * q;
                    // masked lanes read four zeros that sit behind the window table (finite whatever the signal holds),
                    // so the products below need no guards and the lane masks die here
                    const float* zero4 = a.win512 + 512;
                    if (VEC4) {
                        x[j] = *reinterpret_cast<const float4*>((valid && idx < a.L) ? src + idx : zero4);
                    } else {
                        const float* p0 = (valid && idx + 0 < a.L) ? src + idx + 0 : zero4;
                        const float* p1 = (valid && idx + 1 < a.L) ? src + idx + 1 : zero4;
                        const float* p2 = (valid && idx + 2 < a.L) ? src + idx + 2 : zero4;
                        const float* p3 = (valid && idx + 3 < a.L) ? src + idx + 3 : zero4;
                        x[j] = make_float4(*p0, *p1, *p2, *p3);
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n1 = 8 * half + j;
                    const float4 w = *reinterpret_cast<const float4*>(s_win + 32 * n1 + 4 * q);      // zero beyond L
                    za[n1] = make_float2(x[j].x * w.x, x[j].y * w.y);
                    zb[n1] = make_float2(x[j].z * w.z, x[j].w * w.w);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        LBX_STAMP(1);
        // ---- 2.-6. FFT, untangling, |.|^power into the wave's power buffer
        fft512_power_tile<POW2, SEGMEL>(za, zb, wbuf, q, f, s_tw256, s_tw512, a.power_half, [&](int i) { LBX_STAMP(i); });
        LBX_STAMP(6);

        float* Pf = s_P + f * P_STRIDE;
        const int nvalid = min(8, a.T - t0);             // frames of this tile inside the utterance
        if (KIND == LIDBOX_FEAT_SPECTROGRAM) {
            float* dst = a.out + (long)b * a.out_bs + (long)t0 * 257;
            for (int ff = 0; ff < nvalid; ++ff) {
#pragma unroll
                for (int k0 = 0; k0 < 320; k0 += 64) {
                    const int k = k0 + lane;
                    if (k < 257) {
                        const float v = s_P[ff * P_STRIDE + k];
                        dst[ff * 257 + k] = v;
                        bad = fmaf(v, 0.f, bad);
                    }
                }
            }
        } else {
            if (SEGMEL) {
                segmel_tile<KIND>(lane, a.seg_len, a.seg_steps, a.M, s_segmeta, s_segw, s_P, s_stage);
            } else {
                // ---- 7. banded mel: lane (f, q) owns bands q, q+8, ...
                for (int m = q; m < a.M; m += 8) {
                    const int st = s_mstart[m], cn = s_mcnt[m];
                    const float* wv = s_mw + s_moff[m];
                    float acc = 0.f;
                    for (int j = 0; j < cn; ++j) acc = fmaf(Pf[st + j], wv[j], acc);
                    if (KIND != LIDBOX_FEAT_MEL) acc = __logf(acc + LOG_EPS);
                    s_stage[f * a.M + m] = acc;
                }
            }
            wave_lds_sync();
            LBX_STAMP(7);
            if (KIND == LIDBOX_FEAT_MFCC) {
                float* s_coef = SEGMEL ? s_P : s_stage + 8 * a.M;
                if (SEGMEL) {
                    if (dct_regs) segdct_tile<true>(lane, a.dct_runs, a.dct_len, a.ncoef, a.M, wd, s_dct, s_stage, s_coef);
                    else segdct_tile<false>(lane, a.dct_runs, a.dct_len, a.ncoef, a.M, wd, s_dct, s_stage, s_coef);
                } else {
                    for (int c = q; c < a.ncoef; c += 8) {
                        float acc = 0.f;
                        for (int n = 0; n < a.M; ++n) acc = fmaf(s_stage[f * a.M + n], s_dct[n * a.ncoef + c], acc);
                        s_coef[f * a.ncoef + c] = acc;
                    }
                }
                wave_lds_sync();
                float* dst = a.out + (long)b * a.out_bs + (long)t0 * a.ncoef;
                const int total = nvalid * a.ncoef;
                for (int i = lane; i < total; i += 64) {
                    const float v = s_coef[i];
                    dst[i] = v;
                    bad = fmaf(v, 0.f, bad);
                }
            } else {
                float* dst = a.out + (long)b * a.out_bs + (long)t0 * a.M;
                bad = store_mel_tile<false>(lane, nvalid * a.M, dst, nullptr, s_stage, bad);
            }
        }
        LBX_STAMP(8);
    }
    if (a.nonfinite && bad != bad) *a.nonfinite = 1;
    LBX_STAMP_K(14);
    LBX_STAMP_K(15);
}

// ------------------------------------------------------------------------------------------------
// streaming kernel (round 6): the fused tile as a software pipeline
// ------------------------------------------------------------------------------------------------
// Same tile arithmetic as fused_feat512_kernel (bit-identical results), different schedule.  Round 6's per-wave timeline of that
// kernel at 256 utterances (tools/feat_timeline.py, profiles/r06_feature_timeline.txt): tables staged 1.6 us after entry, the first
// tile's samples requested only then; first tile 9.7 us median / 16.4 us p90 (every wave of the machine waits for HBM, then all of
// them hit the LDS exchange together), second tile 7.2 us; waves leave between 15.8 and 25.5 us because 25 tiles per CU on 14 waves
// is 2 + 2 + ... + 1 with the two-tile waves piled on two of the four SIMDs.  Per tile the LDS pipe is busy ~830 cycles (the
// exchange's 16 ds_write_b128 alone 208) and a SIMD ~3.7 k issue cycles: two co-critical resources that only overlap across waves.
//   * ONE persistent workgroup per CU, as many waves as the LDS holds (16 = four per SIMD at <= 128 registers: the staging tile
//     aliases the dead power buffer, 9 280 B per wave + one copy of the tables).
//   * A wave's samples arrive by RAW BUFFER LOADS through a per-utterance descriptor whose num_records ends the utterance: reads
//     behind it return 0 from the bounds check, so there is no guarded path, no address select and no exec branch (what sank round
//     4's prefetch, tools/patches/feat_tile_prefetch.patch), and hipcc counts the loads itself.  The loads of the wave's FIRST tile
//     are issued before the tables are staged; the loads of tile i + 1 are issued behind the untangling of tile i, into the FFT
//     registers that died there, and land while mel / log / store run.
//   * Tiles are handed out dynamically (one LDS counter per workgroup): a wave asks for its next tile at the top of the current one,
//     so whichever SIMD is ahead takes the remainder.  A tile's result does not depend on the wave that computes it.
//   * SRC16: the samples are 16-bit PCM read in place (8 bytes per lane and n1 instead of 16, converted in registers; the 1 / 32768 of
//     tf.audio.decode_wav is folded into the window table: a power of two, so bit-identical to lidbox_pcm16_to_f32 -> this kernel).
//   * The store stage folds "any value not finite" of what it writes into *nonfinite (tf.debugging.assert_all_finite of
//     tf_utils.py:168-194 without a pass over the output).
typedef unsigned u32x4_s __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_s __attribute__((ext_vector_type(2)));

template <bool SRC16> struct StreamRegs { typedef u32x4_s type; };
template <> struct StreamRegs<true> { typedef u32x2_s type; };

__device__ __forceinline__ unsigned wave_uniform(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

template <int KIND, bool POW2, bool SHADOW, bool SRC16, int NL>
__global__ __launch_bounds__(1024) void feat512_stream_kernel(const FusedArgs a) {
    typedef typename StreamRegs<SRC16>::type xreg_t;
    constexpr int ESZ = SRC16 ? 2 : 4;                                  // bytes per source sample
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NW = (int)(blockDim.x >> 6);
    // ---- LDS carve: tables (+ DCT rows for MFCC), the tile counter, then one scratch block per wave
    float* s_win = reinterpret_cast<float*>(smem);
    float2* s_tw256 = reinterpret_cast<float2*>(smem + 2048);
    float2* s_tw512 = reinterpret_cast<float2*>(smem + 4096);
    int* s_segmeta = reinterpret_cast<int*>(smem + 6144);
    float* s_segw = reinterpret_cast<float*>(s_segmeta + 192);
    const int mel_floats = (KIND == LIDBOX_FEAT_SPECTROGRAM) ? 0 : mel_table_floats(true, a.M, a.nnz, a.seg_len);
    float* s_dct = reinterpret_cast<float*>(smem + 6144) + mel_floats;
    const int table_floats = 1536 + mel_floats + (KIND == LIDBOX_FEAT_MFCC ? a.M * a.ncoef : 0);
    int* s_next = reinterpret_cast<int*>(smem + table_floats * 4);
    const int table_bytes = (table_floats * 4 + 4 + 15) & ~15;

    const int tid = threadIdx.x;
    const int lane0 = tid & 63;
    const int wave = (int)wave_uniform((unsigned)tid >> 6);
    LBX_STAMP_K(11);
    LBX_STAMP_K(12);

    const unsigned chunk = xcd_chunk_id(blockIdx.x, a.nwg);
    const unsigned tile0 = chunk * (unsigned)a.tiles_per_wg;
    const unsigned ntl = min((unsigned)a.tiles_per_wg, (unsigned)a.ntiles - tile0);     // tiles of this workgroup
    const unsigned tpu = (unsigned)a.tiles_per_utt;
    const unsigned utt_bytes = ((unsigned)a.N & ~3u) * ESZ;             // a float4 / short4 at a multiple of 4 samples is wholly inside or wholly outside

    // sample loads of local tile `local` (wave-uniform; past the workgroup's last tile: a descriptor of zero records, no traffic)
    xreg_t x[NL];                                          // NL = 13: frames of <= 416 samples, the window table is zero behind them
    auto issue = [&](unsigned local, const int ln) {
        const unsigned lane_off = (unsigned)((ln >> 3) * a.S + 4 * (ln & 7)) * ESZ;
        const bool on = local < ntl;
        const unsigned tile = tile0 + (on ? local : 0u);
        const unsigned b = tile / tpu;
        const unsigned t0 = (tile - b * tpu) * 8u;
        const char* base = reinterpret_cast<const char*>(a.signals) + (long)b * a.sig_stride * ESZ;
        const unsigned lo = wave_uniform((unsigned)(uintptr_t)base), hi = wave_uniform((unsigned)((uintptr_t)base >> 32));
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo), 0, on ? (int)utt_bytes : 0, 0x00020000);
        const int voff = (int)(t0 * (unsigned)a.S * ESZ + lane_off);
#pragma unroll
        for (int j = 0; j < NL; ++j) {
#ifdef LBX_ABL_NOLOAD
            x[j] = (xreg_t)(unsigned)(voff + j);
#else
            if constexpr (SRC16) x[j] = __builtin_amdgcn_raw_buffer_load_b64(r, voff + 64 * j, 0, 0);
            else x[j] = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 128 * j, 0, 0);
#endif
        }
    };
    // Table loads first (one value per thread and table where the workgroup is wide enough), THEN the first tile's samples: loads
    // return in order, so the tables (L2 hits) are staged and the barrier is passed while the samples are still on their way from HBM.
    const int nthr = (int)blockDim.x;
    const float t_win = tid < 512 ? a.win512[tid] : 0.f;
    const float2 t_256 = tid < 256 ? a.tw256[tid] : make_float2(0.f, 0.f), t_512 = tid < 256 ? a.tw512[tid] : make_float2(0.f, 0.f);
    int t_meta = 0;
    float t_segw = 0.f, t_dct = 0.f;
    if (KIND != LIDBOX_FEAT_SPECTROGRAM) {
        if (tid < 192) t_meta = a.seg_meta[tid];
        if (tid < a.seg_len * 64) t_segw = a.seg_w[tid];
        if (KIND == LIDBOX_FEAT_MFCC && tid < a.M * a.ncoef) t_dct = a.dct[tid];
    }
    unsigned cur = (unsigned)wave;
    issue(cur, lane0);
    if (tid < 512) s_win[tid] = t_win;
    if (tid < 256) {
        s_tw256[tid] = t_256;
        s_tw512[tid] = t_512;
    }
    if (KIND != LIDBOX_FEAT_SPECTROGRAM) {
        if (tid < 192) s_segmeta[tid] = t_meta;
        if (tid < a.seg_len * 64) s_segw[tid] = t_segw;
        if (KIND == LIDBOX_FEAT_MFCC && tid < a.M * a.ncoef) s_dct[tid] = t_dct;
    }
    // what a narrow workgroup (or a long table) leaves over
    for (int i = tid + nthr; i < 512; i += nthr) s_win[i] = a.win512[i];
    for (int i = tid + nthr; i < 256; i += nthr) {
        s_tw256[i] = a.tw256[i];
        s_tw512[i] = a.tw512[i];
    }
    if (KIND != LIDBOX_FEAT_SPECTROGRAM) {
        for (int i = tid + nthr; i < 192; i += nthr) s_segmeta[i] = a.seg_meta[i];
        for (int i = tid + nthr; i < a.seg_len * 64; i += nthr) s_segw[i] = a.seg_w[i];
        if (KIND == LIDBOX_FEAT_MFCC)
            for (int i = tid + nthr; i < a.M * a.ncoef; i += nthr) s_dct[i] = a.dct[i];
    }
    if (tid == 0) *s_next = NW;
    __syncthreads();
    LBX_STAMP_K(13);

    char* wbuf = smem + table_bytes + wave * WAVE_SCRATCH;
    float* s_P = reinterpret_cast<float*>(wbuf);          // SPECTROGRAM: [8][P_STRIDE]; else transposed [PT_ROWS][8]; aliases the exchange
    float* s_stage = reinterpret_cast<float*>(wbuf);      // the staged [8][M] (MFCC: [M][8]) tile aliases the power buffer once the mel runs have read it
    float* s_coef = reinterpret_cast<float*>(wbuf + 4096);
    unsigned long long badmask = 0;                       // lanes that stored a value that was not finite (scalar registers)

    for (int it = 0; cur < ntl; ++it) {
        float bad = 0.f;                                  // 0, or NaN once a value this tile stored was not finite
        // Per-lane addresses (window / twiddle / exchange rows, mel run, staging slots) depend only on the lane, so hipcc hoists them out
        // of this loop as ~25 live registers -- and, with the next tile's 64 sample registers in flight across mel / store at the
        // 128-register cap, spills them: a scratch reload behind the prefetch is a vmcnt(0) that waits for every sample load.  An opaque
        // copy of the lane index per tile keeps their (cheap) computation inside the loop.
        // (The lane index itself comes from mbcnt each tile, and the not-finite state lives in scalar registers.)
        unsigned ones = ~0u;
        asm volatile("" : "+s"(ones));
        const int lane = (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
        const int q = lane & 7, f = lane >> 3;
        // the next tile of this wave (asked for now: its loads go out in the middle of this one)
        unsigned nxt = 0;
        if (lane == 0) nxt = (unsigned)__hip_atomic_fetch_add(s_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        nxt = wave_uniform(nxt);
        asm volatile("" : "+s"(nxt));
        const unsigned tile = tile0 + cur;
        const unsigned b = tile / tpu;
        const int t0 = (int)((tile - b * tpu) * 8u);

        // ---- 1. window.  lane q holds n2 = 2q (za) and 2q+1 (zb), n1 = 0..15: packed sample n = 16 n1 + n2 <-> reals 32 n1 + 4q .. + 3
        LBX_STAMP(0);
        float2 za[16], zb[16];
#pragma unroll
        for (int n1 = 0; n1 < NL; ++n1) {
            const float4 w = *reinterpret_cast<const float4*>(s_win + 32 * n1 + 4 * q);      // zero beyond the frame
            float4 v;
            if constexpr (SRC16) {
                const int p0 = (int)x[n1].x, p1 = (int)x[n1].y;
                v = make_float4((float)(short)(p0 & 0xffff), (float)(p0 >> 16), (float)(short)(p1 & 0xffff), (float)(p1 >> 16));
            } else {
                v = __builtin_bit_cast(float4, x[n1]);
            }
            za[n1] = make_float2(v.x * w.x, v.y * w.y);
            zb[n1] = make_float2(v.z * w.z, v.w * w.w);
        }
        LBX_STAMP(1);
        // ---- 2.-6. FFT, untangling, |.|^power into the wave's power buffer
        fft512_power_tile<POW2, KIND != LIDBOX_FEAT_SPECTROGRAM, NL>(za, zb, wbuf, q, f, s_tw256, s_tw512, a.power_half,
                                                                 [&](int i) { LBX_STAMP(i); });
        // ---- 1'. the next tile's samples, into registers that are dead from here to the top of the loop
        issue(nxt, lane);
        LBX_STAMP(6);

        const int nvalid = min(8, a.T - t0);
        if (KIND == LIDBOX_FEAT_SPECTROGRAM) {
            float* dst = a.out + (long)b * a.out_bs + (long)t0 * 257;
            for (int ff = 0; ff < nvalid; ++ff) {
#pragma unroll
                for (int k0 = 0; k0 < 320; k0 += 64) {
                    const int k = k0 + lane;
                    if (k < 257) {
                        const float v = s_P[ff * P_STRIDE + k];
                        dst[ff * 257 + k] = v;
                        bad = fmaf(v, 0.f, bad);
                    }
                }
            }
        } else {
#ifndef LBX_ABL_NOMEL
            segmel_tile<KIND>(lane, a.seg_len, a.seg_steps, a.M, s_segmeta, s_segw, s_P, s_stage);
#endif
            wave_lds_sync();
            LBX_STAMP(7);
            if (KIND == LIDBOX_FEAT_MFCC) {
                float wd[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) wd[u] = 0.f;
                segdct_tile<false, 2>(lane, a.dct_runs, a.dct_len, a.ncoef, a.M, wd, s_dct, s_stage, s_coef);      // two bands per batch: the next tile's samples are in flight
                wave_lds_sync();
                float* dst = a.out + (long)b * a.out_bs + (long)t0 * a.ncoef;
                const int total = nvalid * a.ncoef;
                for (int i = lane; i < total; i += 64) {
                    const float v = s_coef[i];
                    dst[i] = v;
                    bad = fmaf(v, 0.f, bad);
                }
            } else {
                float* dst = a.out + (long)b * a.out_bs + (long)t0 * a.M;
                bad = store_mel_tile<SHADOW>(lane, nvalid * a.M, dst, SHADOW ? a.out16 + (dst - a.out) : nullptr, s_stage, bad);
            }
        }
        badmask |= __builtin_amdgcn_ballot_w64(bad != bad);
        LBX_STAMP(8);
        cur = nxt;
    }
    if (a.nonfinite && badmask != 0 && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) *a.nonfinite = 1;
    LBX_STAMP_K(14);
    LBX_STAMP_K(15);
}

// ------------------------------------------------------------------------------------------------
// generic path
// ------------------------------------------------------------------------------------------------
// one workgroup per frame: windowed frame (cropped/zero-padded to nfft) in LDS, thread k computes
// bin k by a direct DFT against the twiddle table; writes |X|^power.
__global__ __launch_bounds__(256) void generic_spectrogram_kernel(
    const float* __restrict__ signals, long sig_stride, int T, int L, int S, int nfft, int F,
    const float* __restrict__ win, const float2* __restrict__ tw, float power, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* fr = reinterpret_cast<float*>(smem);
    const int t = blockIdx.x, b = blockIdx.y;
    const int Leff = L < nfft ? L : nfft;
    const float* src = signals + (long)b * sig_stride + (long)t * S;
    for (int i = threadIdx.x; i < Leff; i += blockDim.x) fr[i] = src[i] * win[i];
    __syncthreads();
    for (int k = threadIdx.x; k < F; k += blockDim.x) {
        float re = 0.f, im = 0.f;
        int idx = 0;
        for (int n = 0; n < Leff; ++n) {
            const float2 w = tw[idx];
            re = fmaf(fr[n], w.x, re);
            im = fmaf(fr[n], w.y, im);
            idx += k;
            if (idx >= nfft) idx -= nfft;
        }
        const float p2 = re * re + im * im;
        out[((long)b * T + t) * F + k] = (power == 2.0f) ? p2 : powf(p2, 0.5f * power);
    }
}

// Power-of-two fft_length other than the fused kernel's 512: one workgroup per frame, the real FFT as an H = nfft/2 point
// complex FFT of the packed samples (radix-2 Stockham autosort between two LDS buffers, log2(H) passes of H/2
// butterflies, twiddles from the plan's table e^{-2 pi i j / nfft}), then the conjugate-pair untangling and |.|^power.
// O(N log N) per frame against the direct kernel's O(N^2 / 2); LDS = 2 * H * 8 bytes (128 KB at fft_length 16384).
__global__ __launch_bounds__(256) void pow2_fft_spectrogram_kernel(
    const float* __restrict__ signals, long sig_stride, int T, int L, int S, int nfft, int F,
    const float* __restrict__ win, const float2* __restrict__ tw, float power, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = nfft >> 1;
    float2* buf0 = reinterpret_cast<float2*>(smem);
    float2* buf1 = buf0 + H;
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int Leff = L < nfft ? L : nfft;
    const float* src = signals + (long)b * sig_stride + (long)t * S;
    for (int n = tid; n < H; n += 256) {
        const int i0 = 2 * n, i1 = 2 * n + 1;
        buf0[n] = make_float2(i0 < Leff ? src[i0] * win[i0] : 0.f, i1 < Leff ? src[i1] * win[i1] : 0.f);
    }
    __syncthreads();
    float2* in = buf0;
    float2* outb = buf1;
    const int half = H >> 1;
    for (int Ns = 1; Ns < H; Ns <<= 1) {
        const int tw_step = nfft / (2 * Ns);                 // e^{-2 pi i k / (2 Ns)} = tw[k * nfft / (2 Ns)]
        for (int j = tid; j < half; j += 256) {
            const int k = j & (Ns - 1);
            const float2 w = tw[k * tw_step];
            const float2 u0 = in[j], v = in[j + half];
            const float2 u1 = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
            const int j0 = ((j - k) << 1) + k;
            outb[j0] = make_float2(u0.x + u1.x, u0.y + u1.y);
            outb[j0 + Ns] = make_float2(u0.x - u1.x, u0.y - u1.y);
        }
        __syncthreads();
        float2* tmp = in; in = outb; outb = tmp;
    }
    // untangle: X[k] = (Z[k] + conj(Z[H-k])) / 2 - i w_k (Z[k] - conj(Z[H-k])) / 2, w_k = e^{-2 pi i k / nfft}, Z[H] = Z[0]
    float* dst = out + ((long)b * T + t) * F;
    for (int k = tid; k <= H; k += 256) {
        const float2 zk = in[k == H ? 0 : k];
        const float2 zm = in[k == 0 ? 0 : H - k];
        const float er = 0.5f * (zk.x + zm.x), ei = 0.5f * (zk.y - zm.y);
        const float orr = 0.5f * (zk.y + zm.y), oi = 0.5f * (zm.x - zk.x);
        const float2 w = (k == H) ? make_float2(-1.f, 0.f) : tw[k];
        const float xr = er + (w.x * orr - w.y * oi), xi = ei + (w.x * oi + w.y * orr);
        const float p2 = xr * xr + xi * xi;
        dst[k] = (power == 2.0f) ? p2 : powf(p2, 0.5f * power);
    }
}

// fft_length that is not a power of two (tf.signal.stft takes any, reference audio.py:229): Bluestein's chirp-z transform.  One
// workgroup per frame: a_n = x_n win_n w_n (zero beyond the frame), A = FFT_m2(a), C = A . Bhat (the plan's transform of the wrapped
// conjugate chirp, 1 / m2 folded in), c = IFFT_m2(C) as conj(FFT(conj(C))), X_k = c_k w_k for k <= nfft / 2, then |.|^power.  Both
// transforms are the radix-2 Stockham passes of pow2_fft_spectrogram_kernel over m2 complex points; LDS = 2 m2 x 8 bytes (128 KB at
// m2 = 8192; m2 = 16 384 runs bluestein_inplace_spectrogram_kernel below).  O(m2 log m2) per frame against the direct DFT's
// O(N^2 / 2): 64 x 1 s at fft_length 2000 runs ~20 x faster (tests/test_features_gpu.py).
__device__ __forceinline__ float2* stockham_fft(float2* in, float2* outb, int n, const float2* __restrict__ tw, int tid) {
    const int half = n >> 1;
    for (int Ns = 1; Ns < n; Ns <<= 1) {
        const int tw_step = n / (2 * Ns);                    // e^{-2 pi i k / (2 Ns)} = tw[k * n / (2 Ns)]
        for (int j = tid; j < half; j += 256) {
            const int k = j & (Ns - 1);
            const float2 w = tw[k * tw_step];
            const float2 u0 = in[j], v = in[j + half];
            const float2 u1 = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
            const int j0 = ((j - k) << 1) + k;
            outb[j0] = make_float2(u0.x + u1.x, u0.y + u1.y);
            outb[j0 + Ns] = make_float2(u0.x - u1.x, u0.y - u1.y);
        }
        __syncthreads();
        float2* tmp = in; in = outb; outb = tmp;
    }
    return in;                                               // the buffer that holds the result
}

__global__ __launch_bounds__(256) void bluestein_spectrogram_kernel(
    const float* __restrict__ signals, long sig_stride, int T, int L, int S, int nfft, int F, int m2,
    const float* __restrict__ win, const float2* __restrict__ chirp, const float2* __restrict__ bhat, const float2* __restrict__ tw,
    float power, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* buf0 = reinterpret_cast<float2*>(smem);
    float2* buf1 = buf0 + m2;
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int Leff = L < nfft ? L : nfft;
    const float* src = signals + (long)b * sig_stride + (long)t * S;
    for (int n = tid; n < m2; n += 256) {
        float2 v = make_float2(0.f, 0.f);
        if (n < Leff) {
            const float x = src[n] * win[n];
            const float2 w = chirp[n];
            v = make_float2(x * w.x, x * w.y);
        }
        buf0[n] = v;
    }
    __syncthreads();
    float2* A = stockham_fft(buf0, buf1, m2, tw, tid);
    float2* other = A == buf0 ? buf1 : buf0;
    for (int k = tid; k < m2; k += 256) {                    // conj(A . Bhat): the inverse transform as conj(FFT(conj(.)))
        const float2 a = A[k], h = bhat[k];
        A[k] = make_float2(a.x * h.x - a.y * h.y, -(a.x * h.y + a.y * h.x));
    }
    __syncthreads();
    const float2* c = stockham_fft(A, other, m2, tw, tid);
    float* dst = out + ((long)b * T + t) * F;
    for (int k = tid; k < F; k += 256) {
        const float2 y = make_float2(c[k].x, -c[k].y), w = chirp[k];
        const float xr = y.x * w.x - y.y * w.y, xi = y.x * w.y + y.y * w.x;
        const float p2 = xr * xr + xi * xi;
        dst[k] = (power == 2.0f) ? p2 : powf(p2, 0.5f * power);
    }
}

// The same transform for m2 = 16 384 (round 6: fft_length up to 10 922 at any frame length, up to 16 384 for frames <= 8 192 samples):
// ONE LDS buffer of m2 complex values (128 KB), so both FFTs run in place -- the forward one as decimation in frequency (natural order
// in, bit-reversed out), the product reads Bhat through the bit-reversed index, the inverse one as decimation in time (bit-reversed
// in, natural out): no reordering pass, no second buffer.  1 024 threads (one workgroup per CU at this LDS size).  Any power of two
// m2 >= 2 works (LIDBOX_FEAT_BLUESTEIN_INPLACE=1 runs it at every size: the tests compare the two kernels on the same plans).
__global__ __launch_bounds__(1024) void bluestein_inplace_spectrogram_kernel(
    const float* __restrict__ signals, long sig_stride, int T, int L, int S, int nfft, int F, int m2,
    const float* __restrict__ win, const float2* __restrict__ chirp, const float2* __restrict__ bhat, const float2* __restrict__ tw,
    float power, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* x = reinterpret_cast<float2*>(smem);
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int Leff = L < nfft ? L : nfft;
    const int lg = 31 - __clz(m2), nb = m2 >> 1;
    const float* src = signals + (long)b * sig_stride + (long)t * S;
    for (int n = tid; n < m2; n += 1024) {
        float2 v = make_float2(0.f, 0.f);
        if (n < Leff) {
            const float xs = src[n] * win[n];
            const float2 w = chirp[n];
            v = make_float2(xs * w.x, xs * w.y);
        }
        x[n] = v;
    }
    __syncthreads();
    for (int len = m2; len >= 2; len >>= 1) {                // A = FFT(a), decimation in frequency
        const int half = len >> 1, tw_step = m2 / len;       // e^{-2 pi i k / len} = tw[k * m2 / len]
        for (int j = tid; j < nb; j += 1024) {
            const int k = j & (half - 1);
            const int i0 = ((j - k) << 1) + k;
            const float2 u = x[i0], v = x[i0 + half], w = tw[k * tw_step];
            const float2 d = make_float2(u.x - v.x, u.y - v.y);
            x[i0] = make_float2(u.x + v.x, u.y + v.y);
            x[i0 + half] = make_float2(d.x * w.x - d.y * w.y, d.x * w.y + d.y * w.x);
        }
        __syncthreads();
    }
    for (int i = tid; i < m2; i += 1024) {                   // conj(A . Bhat), both at bit-reversed position i
        const float2 a = x[i], h = bhat[__brev((unsigned)i) >> (32 - lg)];
        x[i] = make_float2(a.x * h.x - a.y * h.y, -(a.x * h.y + a.y * h.x));
    }
    __syncthreads();
    for (int len = 2; len <= m2; len <<= 1) {                // FFT of it, decimation in time: natural order out
        const int half = len >> 1, tw_step = m2 / len;
        for (int j = tid; j < nb; j += 1024) {
            const int k = j & (half - 1);
            const int i0 = ((j - k) << 1) + k;
            const float2 u = x[i0], v = x[i0 + half], w = tw[k * tw_step];
            const float2 vw = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
            x[i0] = make_float2(u.x + vw.x, u.y + vw.y);
            x[i0 + half] = make_float2(u.x - vw.x, u.y - vw.y);
        }
        __syncthreads();
    }
    float* dst = out + ((long)b * T + t) * F;
    for (int k = tid; k < F; k += 1024) {
        const float2 y = make_float2(x[k].x, -x[k].y), w = chirp[k];
        const float xr = y.x * w.x - y.y * w.y, xi = y.x * w.y + y.y * w.x;
        const float p2 = xr * xr + xi * xi;
        dst[k] = (power == 2.0f) ? p2 : powf(p2, 0.5f * power);
    }
}

// thread per (frame, band): banded mel (+ optional log)
__global__ void generic_mel_kernel(const float* __restrict__ spec, long nframes, int F, int M,
                                   const int* __restrict__ ms, const int* __restrict__ mc,
                                   const int* __restrict__ mo, const float* __restrict__ mw,
                                   int do_log, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nframes * M) return;
    const long fr = i / M;
    const int m = (int)(i - fr * M);
    const float* P = spec + fr * F + ms[m];
    const float* w = mw + mo[m];
    float acc = 0.f;
    for (int j = 0; j < mc[m]; ++j) acc = fmaf(P[j], w[j], acc);
    out[i] = do_log ? logf(acc + LOG_EPS) : acc;
}

// thread per (frame, coef)
__global__ void generic_dct_kernel(const float* __restrict__ logmel, long nframes, int M, int ncoef,
                                   const float* __restrict__ dct, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nframes * ncoef) return;
    const long fr = i / ncoef;
    const int c = (int)(i - fr * ncoef);
    float acc = 0.f;
    for (int n = 0; n < M; ++n) acc = fmaf(logmel[fr * M + n], dct[n * ncoef + c], acc);
    out[i] = acc;
}

// out16[b*bs + i] = bf16(out[b*bs + i]), i < per: the shadow of feature kinds whose kernel does not store it itself
__global__ void shadow_rows_kernel(const float* __restrict__ in, unsigned short* __restrict__ out16, long bs, long per) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < per) out16[blockIdx.y * bs + i] = __builtin_bit_cast(unsigned short, (__bf16)in[blockIdx.y * bs + i]);
}

// any value of out[b * bs + i], i < per, not finite -> *flag |= 1 (the kernels without a flag in their store stage)
__global__ void nonfinite_rows_kernel(const float* __restrict__ x, long bs, long per, int* __restrict__ flag) {
    float bad = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) bad = fmaf(x[blockIdx.y * bs + i], 0.f, bad);
    if (bad != bad) *flag = 1;
}

template <int KIND>
int launch_fused(const lidbox_feat_plan* p, const FusedArgs& a, bool vec4, bool segmel, size_t lds, hipStream_t st) {
    const bool pow2 = (p->power == 2.0f);
#define LBX_FUSED(V, P2, SG)                                                                          \
    hipLaunchKernelGGL((fused_feat512_kernel<KIND, V, P2, SG>), dim3(a.nwg), dim3(256), lds, st, a)
    if (segmel && KIND != LIDBOX_FEAT_SPECTROGRAM) {
        if (vec4 && pow2) LBX_FUSED(true, true, (KIND != LIDBOX_FEAT_SPECTROGRAM));
        else if (vec4) LBX_FUSED(true, false, (KIND != LIDBOX_FEAT_SPECTROGRAM));
        else if (pow2) LBX_FUSED(false, true, (KIND != LIDBOX_FEAT_SPECTROGRAM));
        else LBX_FUSED(false, false, (KIND != LIDBOX_FEAT_SPECTROGRAM));
    } else {
        if (vec4 && pow2) LBX_FUSED(true, true, false);
        else if (vec4) LBX_FUSED(true, false, false);
        else if (pow2) LBX_FUSED(false, true, false);
        else LBX_FUSED(false, false, false);
    }
#undef LBX_FUSED
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

// the streaming kernel: its LDS depends on the plan (tables + nw wave slices, up to the CU's 160 KiB), so the limit is raised once per
// (instantiation, DEVICE): a later plan with more mel bins / another sample rate, or a second GPU in the process, still launches
template <int KIND, bool POW2, bool SHADOW, bool SRC16, int NL>
int launch_stream_nl(const FusedArgs& a, int nw, size_t lds, hipStream_t st) {
    static std::atomic<unsigned long long> attr_devs{0};
    int dev = 0;
    LBX_HIP(hipGetDevice(&dev));
    if (dev >= 64 || !(attr_devs.load() >> dev & 1ull)) {
        LBX_HIP(hipFuncSetAttribute((const void*)feat512_stream_kernel<KIND, POW2, SHADOW, SRC16, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (dev < 64) attr_devs.fetch_or(1ull << dev);
    }
    hipLaunchKernelGGL((feat512_stream_kernel<KIND, POW2, SHADOW, SRC16, NL>), dim3(a.nwg), dim3(64 * nw), lds, st, a);
    LBX_LAUNCH_OK();
    return LIDBOX_OK;
}

// frames of <= 416 samples (25 ms at 16 kHz: 400): the window table is zero from sample 416 on, the three sample loads per lane behind
// it, their window products and the zero inputs of pass 1's first radix-4 stage are compiled out (NL = 13)
template <int KIND, bool POW2, bool SHADOW, bool SRC16>
int launch_stream_one(const FusedArgs& a, int nw, size_t lds, hipStream_t st) {
    static const bool no_prune = getenv("LIDBOX_FEAT_NO_PRUNE") != nullptr;      // A/B aid
    if (a.L <= 416 && !no_prune) return launch_stream_nl<KIND, POW2, SHADOW, SRC16, 13>(a, nw, lds, st);
    return launch_stream_nl<KIND, POW2, SHADOW, SRC16, 16>(a, nw, lds, st);
}

// Instantiations: power == 2 (every lidbox config) gets all of {shadow store (log-mel only), 16-bit PCM source, pruned loads};
// another power only the plain float kernel (the caller converts PCM / shadows with a pass of its own).
template <int KIND>
int launch_stream(const lidbox_feat_plan* p, const FusedArgs& a, bool src16, int nw, size_t lds, hipStream_t st) {
    if (p->power != 2.0f) return launch_stream_nl<KIND, false, false, false, 16>(a, nw, lds, st);
    if (KIND == LIDBOX_FEAT_LOGMEL && a.out16)
        return src16 ? launch_stream_one<KIND, true, KIND == LIDBOX_FEAT_LOGMEL, true>(a, nw, lds, st)
                     : launch_stream_one<KIND, true, KIND == LIDBOX_FEAT_LOGMEL, false>(a, nw, lds, st);
    return src16 ? launch_stream_one<KIND, true, false, true>(a, nw, lds, st) : launch_stream_one<KIND, true, false, false>(a, nw, lds, st);
}

}  // namespace

extern "C" int lidbox_feat_plan_is_fused(const lidbox_feat_plan* p, int kind, const float* signals,
                                         long sig_stride) {
    (void)signals; (void)sig_stride; (void)kind;
    return p && p->fused_ok ? 1 : 0;
}

extern "C" size_t lidbox_extract_features_workspace(const lidbox_feat_plan* p, int kind, int B, int N,
                                                    const float* signals, long sig_stride) {
    if (!p || lidbox_feat_plan_is_fused(p, kind, signals, sig_stride)) return 0;
    const long T = lidbox_num_frames(N, p->L, p->S);
    size_t bytes = 0;
    if (kind != LIDBOX_FEAT_SPECTROGRAM) bytes += (size_t)B * T * p->F * 4;
    if (kind == LIDBOX_FEAT_MFCC) bytes += (size_t)B * T * p->M * 4;
    return bytes;
}

extern "C" int lidbox_extract_features_fwd(const lidbox_feat_plan* p, int kind, const float* signals,
                                           int B, int N, long sig_stride, float* out,
                                           long out_batch_stride, void* workspace,
                                           size_t workspace_bytes, lidbox_stream_t stream) {
    return lidbox_extract_features_fwd_ex(p, kind, signals, LIDBOX_SRC_F32, B, N, sig_stride, out, out_batch_stride, nullptr, nullptr, workspace,
                                          workspace_bytes, stream);
}

// + out16: a bfloat16 copy of the features (round-to-nearest-even) at the same element offsets as out (same batch stride, in
// elements) -- the shadow the bf16-storage Conv1D path reads, written by the feature kernel itself.
extern "C" int lidbox_extract_features_fwd_shadow(const lidbox_feat_plan* p, int kind, const float* signals,
                                                  int B, int N, long sig_stride, float* out, long out_batch_stride, void* out16,
                                                  void* workspace, size_t workspace_bytes, lidbox_stream_t stream) {
    return lidbox_extract_features_fwd_ex(p, kind, signals, LIDBOX_SRC_F32, B, N, sig_stride, out, out_batch_stride, out16, nullptr, workspace,
                                          workspace_bytes, stream);
}

// 16-bit mono PCM read in place: bit-identical to lidbox_pcm16_to_f32(channels = 1) followed by lidbox_extract_features_fwd
extern "C" int lidbox_extract_features_fwd_pcm16(const lidbox_feat_plan* p, int kind, const int16_t* pcm, int B, int N, long sig_stride,
                                                 float* out, long out_batch_stride, int* nonfinite, lidbox_stream_t stream) {
    return lidbox_extract_features_fwd_ex(p, kind, pcm, LIDBOX_SRC_PCM16, B, N, sig_stride, out, out_batch_stride, nullptr, nonfinite, nullptr, 0, stream);
}

extern "C" int lidbox_extract_features_fwd_ex(const lidbox_feat_plan* p, int kind, const void* signals_v, int src_format,
                                              int B, int N, long sig_stride, float* out, long out_batch_stride, void* out16,
                                              int* nonfinite, void* workspace, size_t workspace_bytes, lidbox_stream_t stream) {
    LBX_ARG(p && signals_v && out, "plan, signals, out != NULL");
    LBX_ARG(kind >= LIDBOX_FEAT_SPECTROGRAM && kind <= LIDBOX_FEAT_MFCC, "kind");
    LBX_ARG(src_format == LIDBOX_SRC_F32 || src_format == LIDBOX_SRC_PCM16, "src_format");
    LBX_ARG(kind != LIDBOX_FEAT_MFCC || p->ncoef > 0, "the plan's MFCC slice [coef_begin, coef_end) is empty");
    LBX_ARG(B >= 0 && N >= 0 && sig_stride >= N, "B >= 0, N >= 0, sig_stride >= N");
    const bool src16 = src_format == LIDBOX_SRC_PCM16;
    const float* signals = reinterpret_cast<const float*>(signals_v);
    hipStream_t st = (hipStream_t)stream;
    const int T = lidbox_num_frames(N, p->L, p->S);
    if (B == 0 || T == 0) return LIDBOX_OK;
    const long chan = lidbox_feat_plan_channels(p, kind);
    if (out_batch_stride == 0) out_batch_stride = (long)T * chan;
    LBX_ARG(out_batch_stride >= (long)T * chan, "out_batch_stride >= T * channels");
    LBX_ARG(!out16 || (((uintptr_t)out16) & 7) == 0, "out16 must be 8-byte aligned");
    // bf16 shadow / finite flag of kinds and shapes whose kernel does not produce them itself: one pass over what was just written
    auto shadow_after = [&]() -> int {
        const long per = (long)T * chan;
        shadow_rows_kernel<<<dim3((unsigned)lbx_cdiv(per, 256L), (unsigned)B), 256, 0, st>>>(out, (unsigned short*)out16, out_batch_stride, per);
        LBX_LAUNCH_OK();
        return LIDBOX_OK;
    };
    auto flag_after = [&]() -> int {
        const long per = (long)T * chan;
        nonfinite_rows_kernel<<<dim3((unsigned)std::min(lbx_cdiv(per, 256L), 64L), (unsigned)B), 256, 0, st>>>(out, out_batch_stride, per, nonfinite);
        LBX_LAUNCH_OK();
        return LIDBOX_OK;
    };

    if (lidbox_feat_plan_is_fused(p, kind, signals, sig_stride)) {
        FusedArgs a;
        a.signals = signals; a.sig_stride = sig_stride; a.B = B; a.N = N; a.T = T;
        a.L = p->L; a.S = p->S; a.power_half = 0.5f * p->power;
        a.M = p->M; a.ncoef = p->ncoef; a.nnz = p->nnz;
        a.seg_len = p->seg_len; a.seg_steps = p->seg_steps; a.seg_meta = p->d_seg_meta; a.seg_w = p->d_seg_w;
        a.dct_runs = p->ncoef > 0 ? (64 / p->ncoef < 1 ? 1 : 64 / p->ncoef) : 1;
        if (a.dct_runs > p->M) a.dct_runs = p->M;
        a.dct_len = (p->M + a.dct_runs - 1) / a.dct_runs;
        a.dct_runs = (p->M + a.dct_len - 1) / a.dct_len;          // drop runs that would be empty
        static const bool no_segmel = getenv("LIDBOX_FEAT_NO_SEGMEL") != nullptr;      // A/B aid
        const bool segmel = LBX_FEAT_SEGMEL && p->seg_ok && !no_segmel && kind != LIDBOX_FEAT_SPECTROGRAM;
        a.win512 = src16 ? p->d_win512_pcm : p->d_win512; a.tw256 = p->d_tw256; a.tw512 = p->d_tw512;
        a.mel_start = p->d_mel_start; a.mel_cnt = p->d_mel_cnt; a.mel_off = p->d_mel_off;
        a.mel_w = p->d_mel_w; a.dct = p->d_dct; a.out = out; a.out_bs = out_batch_stride;
        a.out16 = nullptr;
        a.nonfinite = nonfinite;
#if defined(LBX_FEAT_TIMING) || defined(LBX_FEAT_TIMELINE)
        a.stamps = (workspace && workspace_bytes >= (size_t)4096 * 4 * 16 * 8) ? (long long*)workspace : nullptr;
#endif
        a.tiles_per_utt = (T + 7) / 8;
        a.ntiles = (long)B * a.tiles_per_utt;
        // float4 (short4) loads need 16-byte (8-byte) aligned frames
        const bool vec4 = (((uintptr_t)signals & (src16 ? 7 : 15)) == 0) && (sig_stride % 4 == 0) && (p->S % 4 == 0) && (p->L % 4 == 0);

        // ---- streaming kernel: one persistent workgroup per CU, all the waves the LDS holds, every CU the same number of consecutive
        //      tiles (+- 1) handed out dynamically.  LIDBOX_FEAT_STREAM=0 keeps the round-1 shape (A/B aid).
        {
            static const int stream_env = getenv("LIDBOX_FEAT_STREAM") ? atoi(getenv("LIDBOX_FEAT_STREAM")) : 1;
            const int table_floats = 1536 + (kind == LIDBOX_FEAT_SPECTROGRAM ? 0 : mel_table_floats(true, p->M, p->nnz, p->seg_len)) +
                                     (kind == LIDBOX_FEAT_MFCC ? p->M * p->ncoef : 0);
            const int table_bytes = (table_floats * 4 + 4 + 15) & ~15;
            int nw = (160 * 1024 - table_bytes) / WAVE_SCRATCH;
            if (nw > 16) nw = 16;
            if (const char* e = getenv("LIDBOX_FEAT_STREAM_NW")) { const int v = atoi(e); if (v >= 1 && v < nw) nw = v; }      // tuning aid
            const bool fits32 = (long)N * 4 < (1L << 31) && a.ntiles < (1L << 31) && (long)(T + 8) * p->S * 4 < (1L << 31);
            if (stream_env != 0 && vec4 && fits32 && nw >= 8 && (segmel || kind == LIDBOX_FEAT_SPECTROGRAM) && (!src16 || p->power == 2.0f)) {
                int ncu = 256;
                (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p->device);
                if (ncu < 1) ncu = 256;
                a.tiles_per_wg = (int)lbx_cdiv(a.ntiles, (long)ncu);
                a.nwg = (unsigned)lbx_cdiv(a.ntiles, (long)a.tiles_per_wg);
                if (a.tiles_per_wg < nw) nw = a.tiles_per_wg;      // tiny batches: no idle waves behind the table staging
                a.iters = 0;
                const bool shadow_in_kernel = out16 && kind == LIDBOX_FEAT_LOGMEL && p->power == 2.0f;
                if (shadow_in_kernel) a.out16 = (unsigned short*)out16;
                const size_t lds = (size_t)table_bytes + (size_t)nw * WAVE_SCRATCH;
                int rc;
                switch (kind) {
                    case LIDBOX_FEAT_SPECTROGRAM: rc = launch_stream<LIDBOX_FEAT_SPECTROGRAM>(p, a, src16, nw, lds, st); break;
                    case LIDBOX_FEAT_MEL: rc = launch_stream<LIDBOX_FEAT_MEL>(p, a, src16, nw, lds, st); break;
                    case LIDBOX_FEAT_LOGMEL: rc = launch_stream<LIDBOX_FEAT_LOGMEL>(p, a, src16, nw, lds, st); break;
                    default: rc = launch_stream<LIDBOX_FEAT_MFCC>(p, a, src16, nw, lds, st); break;
                }
                if (rc != LIDBOX_OK || !out16 || shadow_in_kernel) return rc;
                return shadow_after();
            }
        }
        LBX_ARG(!src16, "16-bit PCM sources need the streaming kernel: 8-byte aligned signals, sig_stride / frame_length / frame_step multiples of 4 "
                        "(convert with lidbox_pcm16_to_f32 otherwise)");

        // ---- round-1 shape (unaligned signals, plans outside the streaming kernel's table limits)
        // LDS: tables + 4 wave scratch blocks (must mirror the carve in the kernel)
        const bool dct_regs = kind == LIDBOX_FEAT_MFCC && segmel && a.dct_len <= 8;      // mirrors the kernel
        const int table_floats = 1536 + mel_table_floats(segmel, p->M, p->nnz, p->seg_len) +
                                 ((kind == LIDBOX_FEAT_MFCC && !dct_regs) ? p->M * p->ncoef : 0);
        const int table_bytes = (table_floats * 4 + 15) & ~15;
        const int stage_floats = (kind == LIDBOX_FEAT_SPECTROGRAM) ? 0
                                 : 8 * p->M + ((kind == LIDBOX_FEAT_MFCC && !segmel) ? 8 * p->ncoef : 0);
        const int wave_bytes = WAVE_SCRATCH + ((stage_floats * 4 + 15) & ~15);
        const size_t lds = (size_t)table_bytes + 4 * (size_t)wave_bytes;
        // grid: about four dispatch rounds of resident workgroups, equal tile counts per wave.  One tile per wave
        // while that holds (B <= ~490 at 2 s): the dispatcher then balances the tail; measured 35 vs 39 us at B = 256
        // against three tiles per wave on two thirds of the slots, and no difference at B = 2048.
        const long max_wg = 256 * LBX_FEAT_WAVES;
        const long wg_needed = lbx_cdiv(a.ntiles, 4);
        a.iters = (int)lbx_cdiv(wg_needed, 4 * max_wg);
        if (const char* e = getenv("LIDBOX_FEAT_ITERS")) { const int v = atoi(e); if (v >= 1) a.iters = v; }   // tuning aid
        a.nwg = (unsigned)lbx_cdiv(a.ntiles, 4L * a.iters);
        a.tiles_per_wg = 4 * a.iters;
        int rc;
        switch (kind) {
            case LIDBOX_FEAT_SPECTROGRAM: rc = launch_fused<LIDBOX_FEAT_SPECTROGRAM>(p, a, vec4, false, lds, st); break;
            case LIDBOX_FEAT_MEL: rc = launch_fused<LIDBOX_FEAT_MEL>(p, a, vec4, segmel, lds, st); break;
            case LIDBOX_FEAT_LOGMEL: rc = launch_fused<LIDBOX_FEAT_LOGMEL>(p, a, vec4, segmel, lds, st); break;
            default: rc = launch_fused<LIDBOX_FEAT_MFCC>(p, a, vec4, segmel, lds, st); break;
        }
        if (rc != LIDBOX_OK || !out16) return rc;
        return shadow_after();
    }
    LBX_ARG(!src16, "16-bit PCM sources need the fused path (fft_length 512); convert with lidbox_pcm16_to_f32 otherwise");

    // ---- generic path (dense output only)
    LBX_ARG(out_batch_stride == (long)T * chan, "the non-fused path needs a dense output (out_batch_stride = T*channels)");
    const size_t need = lidbox_extract_features_workspace(p, kind, B, N, signals, sig_stride);
    LBX_ARG(workspace_bytes >= need && (need == 0 || workspace), "workspace too small");
    const long nframes = (long)B * T;
    float* spec = (kind == LIDBOX_FEAT_SPECTROGRAM) ? out : (float*)workspace;
    const int Leff = p->L < p->nfft ? p->L : p->nfft;
    const bool pow2 = p->nfft >= 4 && (p->nfft & (p->nfft - 1)) == 0;
    const bool force_dft = getenv("LIDBOX_FEAT_FORCE_DFT") != nullptr;                 // A/B and test aid (read per call)
    if (pow2 && !force_dft) {
        // LDS: two buffers of nfft / 2 complex values (<= 128 KB at the largest fft_length a plan accepts)
        const size_t lds = (size_t)p->nfft * 8;
        if (lds > 65536)
            LBX_HIP(hipFuncSetAttribute((const void*)pow2_fft_spectrogram_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(pow2_fft_spectrogram_kernel, dim3(T, B), dim3(256), lds, st,
                           signals, sig_stride, T, p->L, p->S, p->nfft, p->F, p->d_win, p->d_twN, p->power, spec);
    } else if (p->bs_m2 && !force_dft) {
        // any other length: Bluestein on two power-of-two transforms of bs_m2 points -- between two LDS buffers up to 8192 points,
        // in place in one buffer at 16 384
        const char* e = getenv("LIDBOX_FEAT_BLUESTEIN_INPLACE");
        const bool inplace = p->bs_m2 > 8192 || (e && atoi(e) != 0);
        const size_t lds = (size_t)p->bs_m2 * (inplace ? 8 : 16);
        const void* fn = inplace ? (const void*)bluestein_inplace_spectrogram_kernel : (const void*)bluestein_spectrogram_kernel;
        if (lds > 65536) LBX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (inplace)
            hipLaunchKernelGGL(bluestein_inplace_spectrogram_kernel, dim3(T, B), dim3(1024), lds, st, signals, sig_stride, T, p->L, p->S,
                               p->nfft, p->F, p->bs_m2, p->d_win, p->d_bs_chirp, p->d_bs_bhat, p->d_bs_tw, p->power, spec);
        else
            hipLaunchKernelGGL(bluestein_spectrogram_kernel, dim3(T, B), dim3(256), lds, st, signals, sig_stride, T, p->L, p->S, p->nfft,
                               p->F, p->bs_m2, p->d_win, p->d_bs_chirp, p->d_bs_bhat, p->d_bs_tw, p->power, spec);
    } else {
        hipLaunchKernelGGL(generic_spectrogram_kernel, dim3(T, B), dim3(256), (size_t)Leff * 4, st,
                           signals, sig_stride, T, p->L, p->S, p->nfft, p->F, p->d_win, p->d_twN,
                           p->power, spec);
    }
    LBX_LAUNCH_OK();
    auto finish = [&]() -> int {
        if (nonfinite) { const int rc = flag_after(); if (rc != LIDBOX_OK) return rc; }
        return out16 ? shadow_after() : LIDBOX_OK;
    };
    if (kind == LIDBOX_FEAT_SPECTROGRAM) return finish();
    float* mel = (kind == LIDBOX_FEAT_MFCC) ? (float*)workspace + nframes * p->F : out;
    hipLaunchKernelGGL(generic_mel_kernel, dim3((unsigned)lbx_cdiv(nframes * p->M, 256)), dim3(256), 0, st,
                       spec, nframes, p->F, p->M, p->d_mel_start, p->d_mel_cnt, p->d_mel_off,
                       p->d_mel_w, kind != LIDBOX_FEAT_MEL ? 1 : 0, mel);
    LBX_LAUNCH_OK();
    if (kind == LIDBOX_FEAT_MFCC) {
        hipLaunchKernelGGL(generic_dct_kernel, dim3((unsigned)lbx_cdiv(nframes * p->ncoef, 256)), dim3(256), 0,
                           st, mel, nframes, p->M, p->ncoef, p->d_dct, out);
        LBX_LAUNCH_OK();
    }
    return finish();
}
