// gemm_tuned.h -- measured decompositions of the fp32 GEMM family for the x-vector layer shapes at 128 / 256 / 512
// utterances per GPU (BASELINE configs 1-3 run 256 per GPU).  GENERATED from tools/gemm_sweep.py (brute force over tile
// shape x split count per launch on an MI355X, interleaved medians): only entries that beat the cost model of gemm.hip
// (choose_rows / tn_plan) by more than 2 % are listed; every other shape goes through the model.
// kind: 0 = nn (forward), 1 = nt (dgrad), 2 = tn (wgrad; "K" is K1, the rows of the weight gradient).
#pragma once

namespace {

struct TunedGemm {
    int kind;
    long M;
    int N, K, bm, bn, splits, no_tail_split;
};

const TunedGemm TUNED_GEMM[] = {
    {0, 25344, 512, 200, 64, 64, 1, 0},   // frame1 fwd B=128: 65.7 -> 58.7 us
    {2, 12672, 512, 1536, 128, 128, 16, 0},   // frame2 wgrad B=128: 198.4 -> 187.9 us
    {1, 12672, 1024, 512, 64, 64, 1, 0},   // frame2 dgrad0 B=128: 148.7 -> 131.6 us
    {0, 4224, 512, 1536, 64, 64, 3, 0},   // frame3 fwd B=128: 79.1 -> 75.0 us
    {2, 4224, 512, 1536, 64, 64, 8, 0},   // frame3 wgrad B=128: 79.5 -> 73.4 us
    {1, 4224, 512, 512, 64, 64, 3, 0},   // frame4 dgrad0 B=128: 39.9 -> 38.9 us
    {2, 4224, 1500, 512, 64, 64, 8, 0},   // frame5 wgrad B=128: 78.8 -> 73.0 us
    {1, 4224, 512, 1500, 64, 64, 3, 0},   // frame5 dgrad0 B=128: 78.3 -> 75.6 us
    {0, 50688, 512, 200, 64, 64, 1, 1},   // frame1 fwd B=256: 109.4 -> 103.5 us
    {2, 50688, 512, 200, 64, 64, 48, 0},   // frame1 wgrad B=256: 141.8 -> 138.8 us
    {0, 25344, 512, 1536, 128, 128, 1, 0},   // frame2 fwd B=256: 331.1 -> 320.7 us
    {1, 25344, 1024, 512, 64, 64, 1, 1},   // frame2 dgrad0 B=256: 267.7 -> 245.3 us
    {1, 25344, 512, 512, 64, 64, 1, 1},   // frame2 dgrad1 B=256: 165.2 -> 138.0 us
    {2, 8448, 512, 1536, 128, 128, 16, 0},   // frame3 wgrad B=256: 138.9 -> 133.6 us
    {1, 8448, 1536, 512, 64, 64, 1, 1},   // frame3 dgrad0 B=256: 153.9 -> 134.7 us
    {2, 8448, 512, 512, 64, 64, 16, 0},   // frame4 wgrad B=256: 56.9 -> 55.2 us
    {0, 8448, 1500, 512, 64, 64, 1, 1},   // frame5 fwd B=256: 136.2 -> 122.2 us
    {2, 8448, 1500, 512, 128, 128, 16, 0},   // frame5 wgrad B=256: 139.2 -> 133.0 us
    {0, 101376, 512, 200, 64, 128, 1, 0},   // frame1 fwd B=512: 206.3 -> 198.9 us
    {2, 101376, 512, 200, 64, 64, 64, 0},   // frame1 wgrad B=512: 259.4 -> 253.3 us
    {0, 50688, 512, 1536, 128, 128, 1, 0},   // frame2 fwd B=512: 667.2 -> 625.9 us
    {1, 50688, 1024, 512, 64, 64, 1, 0},   // frame2 dgrad0 B=512: 509.6 -> 488.3 us
    {1, 50688, 512, 512, 64, 64, 1, 0},   // frame2 dgrad1 B=512: 306.5 -> 268.6 us
    {0, 16896, 512, 1536, 128, 128, 1, 0},   // frame3 fwd B=512: 240.0 -> 226.5 us
    {2, 16896, 512, 1536, 128, 128, 16, 0},   // frame3 wgrad B=512: 263.7 -> 244.5 us
    {1, 16896, 1536, 512, 64, 64, 1, 0},   // frame3 dgrad0 B=512: 266.3 -> 244.0 us
    {2, 16896, 512, 512, 64, 64, 24, 0},   // frame4 wgrad B=512: 102.4 -> 94.5 us
    {0, 16896, 1500, 512, 64, 64, 1, 0},   // frame5 fwd B=512: 248.6 -> 231.7 us
    {2, 16896, 1500, 512, 128, 128, 16, 0},   // frame5 wgrad B=512: 264.3 -> 244.2 us
    {1, 16896, 512, 1500, 64, 128, 1, 0},   // frame5 dgrad0 B=512: 246.5 -> 234.7 us
    {2, 512, 4, 512, 64, 64, 8, 0},   // outputs wgrad B=512: 8.0 -> 7.7 us
};

inline const TunedGemm* tuned_gemm(int kind, long M, int N, int K) {
    static const bool off = getenv("LIDBOX_GEMM_NO_TUNED") != nullptr;          // A/B aid
    if (off) return nullptr;
    for (const TunedGemm& t : TUNED_GEMM)
        if (t.kind == kind && t.M == M && t.N == N && t.K == K) return &t;
    return nullptr;
}

}  // namespace
