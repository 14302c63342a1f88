// skgemm.hip -- prototype of the persistent stream-K fp32 MFMA GEMM body (round 3), standalone (no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/micro/skgemm tools/micro/skgemm.hip && tools/micro/skgemm
// Structure under test (the product kernel in csrc/gemm_sk.h grew out of this file):
//   * 128 x 128 tile, 4 waves as 2 x 2, 64 x 64 per wave on v_mfma_f32_32x32x2_f32
//   * LDS ring of STAGES stages of BK = 16, filled by global_load_lds_dwordx4 (no VGPR staging, no ds_write)
//   * K-inner operands ([row][k] in HBM) land as [row][16 floats] with the four 16-byte chunks of a row XOR-swizzled by
//     (row >> 2) & 3 -- applied to the SOURCE address, LDS stays lane-linear -- and are read with conflict-free ds_read_b128:
//     lanes 0-31 take chunk 2s, lanes 32-63 chunk 2s+1, so MFMA j of sub-step s contracts k = {8s + j, 8s + 4 + j}
//   * K-outer operands ([k][col]) land as [k][128] and are read with ds_read_b32 (32 consecutive floats per half wave)
//   * one s_barrier per K step; operands of the next sub-step are read while the current one multiplies
//   * persistent grid: each workgroup runs D whole tiles, then its share of the K steps of the remaining tiles (stream-K);
//     partial tiles go to slabs in accumulator order, a fix-up kernel sums them in k order
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int A_STAGE = BM * BK, B_STAGE = BK * BN, STAGE = A_STAGE + B_STAGE;      // floats

enum { KINNER = 0, KOUTER = 1 };

struct Plan {
    int tiles_n, ntiles, nk;          // nk = K steps per tile
    int dp_rounds;                    // whole tiles per workgroup
    int sk_tiles, sk_first;           // tiles [sk_first, sk_first + sk_tiles) are streamed
};

__device__ __forceinline__ void dma16(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// saddr form: 64-bit wave-uniform base in SGPRs + one 32-bit byte offset per lane (half the address payload of the
// flat form hipcc picks for the builtin); M0 = LDS destination of the wave (lane i lands at M0 + 16 i)
__device__ __forceinline__ void dma16s(const float* sbase, unsigned voff, float* lds_wave_base) {
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_wave_base;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(dst))
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One operand of one workgroup tile: 128 "rows" (m for A, n for B) x BK.
//   KINNER: X[row][k], element (row, k) at base + row * ld + k        (A of nn / nt, B of nt)
//   KOUTER: X[k][row], element (k, row) at base + k * ld + row        (B of nn, A and B of tn)
template <int KIND>
struct Operand {
    const float* sb;                   // wave-uniform source base of the current K step (SGPR pair)
    unsigned vo[2];                    // this lane's byte offsets of its two DMA pieces (32-bit: the saddr form of the load)
    long step;                         // floats per K step
    int rd;                            // this lane's LDS read base (floats, inside the operand's stage)

    // row0: first row of the tile, nrows: rows of the matrix (clamp), wv: wave id (uniform), wsub: wave's half (wm or wn)
    __device__ __forceinline__ void init(const float* base, long ld, long row0, long nrows, int kstep0, int lane, int wv, int wsub) {
        if (KIND == KINNER) {
            sb = base + row0 * ld + (long)kstep0 * BK;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = wv * 2 + i;                        // piece: rows 16q .. 16q+15
                int r = 16 * q + (lane >> 2);
                const int slot = lane & 3;
                const int chunk = slot ^ ((r >> 2) & 3);
                if (row0 + r >= nrows) r = 0;                    // rows outside the matrix: the tile's first row (never stored)
                vo[i] = (unsigned)(((long)r * ld + chunk * 4) * 4);
            }
            step = BK;
            rd = (wsub * 64 + (lane & 31)) * 16;                 // + bi * 32 * 16, chunk slot added per read
        } else {
            sb = base + (long)kstep0 * BK * ld + row0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = wv * 2 + i;                        // piece: k rows 2q, 2q+1
                const int kr = 2 * q + (lane >> 5);
                int c = (lane & 31) * 4;
                if (row0 + c >= nrows) c = 0;
                vo[i] = (unsigned)(((long)kr * ld + c) * 4);
            }
            step = (long)BK * ld;
            rd = (4 * (lane >> 5)) * 128 + wsub * 64 + (lane & 31);
        }
    }
    __device__ __forceinline__ void issue_one(float* stage_base, int wv, int i) {
        dma16s(sb, vo[i], stage_base + (wv * 2 + i) * 256);
        if (i == 1) sb += step;
    }
    __device__ __forceinline__ void issue(float* stage_base, int wv) {
        issue_one(stage_base, wv, 0);
        issue_one(stage_base, wv, 1);
    }
};

// operand registers of one sub-step (8 of the 16 k of a stage) for the two 32-row blocks of a wave
struct Frag {
    float v[2][4];
};

template <int KIND>
__device__ __forceinline__ void read_frag(const float* st, int rd, int lane, int s2, Frag& f) {
    if (KIND == KINNER) {
        const int slot = (2 * s2 + (lane >> 5)) ^ ((lane >> 2) & 3);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(st + rd + b * 32 * 16 + slot * 4);
            f.v[b][0] = x[0]; f.v[b][1] = x[1]; f.v[b][2] = x[2]; f.v[b][3] = x[3];
        }
    } else {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) f.v[b][j] = st[rd + (8 * s2 + j) * 128 + b * 32];
    }
}

template <int J0, int J1>
__device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int j = J0; j < J1; ++j)
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 2; ++bj)
                acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[bi][j], b.v[bj][j], acc[bi][bj], 0, 0, 0);
}

// C[M][N] = A . B, both operands plain matrices.
//   AK == KINNER: A[M][lda]   KOUTER: A[K][lda] (tn)
//   BKIND == KOUTER: B[K][ldb]   KINNER: B[N][ldb] (nt)
template <int AK, int BKIND, int STAGES, int WGCU, int PRIO, int ABL = 0>
__global__ __launch_bounds__(256, WGCU) void sk_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B, long ldb,
                                                       float* __restrict__ C, long ldc, long M, int N, Plan pl, float* __restrict__ slabs, long long* __restrict__ clk = nullptr, long long* __restrict__ stamps = nullptr) {
    long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int items_ = 0;
    st_[0] = wall_clock64();
    const long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const unsigned P = gridDim.x;
    // XCD-chunk remap of the workgroup id (block b runs on XCD b % 8): neighbours in tile order share an L2
    unsigned pid;
    {
        const unsigned bid = blockIdx.x, xcd = bid & 7u, idx = bid >> 3, q = P >> 3, r = P & 7u;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int nk = pl.nk;
    const int prio_slot = blockIdx.x >> 8;

    auto run_item = [&](int tile, int kb, int ke, float* slab) {
        const int tn = tile % pl.tiles_n, tm = tile / pl.tiles_n;
        const long m0 = (long)tm * BM;
        const int n0 = tn * BN;
        Operand<AK> oa;
        Operand<BKIND> ob;
        oa.init(A, lda, m0, M, kb, lane, wv, wm);
        ob.init(B, ldb, n0, N, kb, lane, wv, wn);
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        const int n = ke - kb;
        // every wave is past the previous item's LDS reads (its MFMAs consumed them) once it arrives here
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < n) {
                oa.issue(smem + s * STAGE, wv);
                ob.issue(smem + s * STAGE + A_STAGE, wv);
            }
        // stage 0 landed: allow the pieces of the later stages to stay in flight
        if (n >= STAGES - 1) wait_vm<(STAGES - 2) * 4>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        Frag a0, b0, a1, b1;
        read_frag<AK>(smem, oa.rd, lane, 0, a0);
        read_frag<BKIND>(smem + A_STAGE, ob.rd, lane, 0, b0);
        if (ABL & 4) { a1 = a0; b1 = b0; }
        if (stamps && items_ == 0) st_[1] = wall_clock64();
        int cur = 0;                                      // stage of step t
        for (int t = 0; t < n; ++t) {
            // publish step t+1 (its pieces were issued STAGES-2 steps ago)
            if (t + 1 < n && !(ABL & 1) && !(ABL & 8)) {
                if (STAGES >= 4 && t + STAGES - 2 < n) wait_vm<(STAGES - 3) * 4>();
                else wait_vm<0>();
            }
            if (!(ABL & 2)) __builtin_amdgcn_s_barrier();
            int nxt = cur + 1;
            if (nxt == STAGES) nxt = 0;
            const bool more = t + STAGES - 1 < n && !(ABL & 1);            // a stage to refill during this step
            int tgt = cur + STAGES - 1;
            if (tgt >= STAGES) tgt -= STAGES;
            float* ta = smem + tgt * STAGE;
            const float* st = smem + cur * STAGE;
            // The four DMA pieces of this wave go out one per MFMA group (an LDS-DMA issue costs the wave 60+ cycles:
            // four in a row right behind the barrier left the matrix pipe idle), and the reads of the next sub-step
            // behind the first group, so the lgkmcnt wait ahead of a group covers reads issued a whole group earlier.
            if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
            if (PRIO == 2) {
                // age decides between equal priorities, so the oldest resident workgroup of a CU would run ahead of the
                // others for the whole launch: rotate one priority step among the residents (slot = blockIdx.x / 256)
                if ((unsigned)(t + prio_slot) % (unsigned)WGCU == 0) __builtin_amdgcn_s_setprio(2);
                else __builtin_amdgcn_s_setprio(0);
            }
            mma<0, 1>(a0, b0, acc);
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 4)) {
                read_frag<AK>(st, oa.rd, lane, 1, a1);
                read_frag<BKIND>(st + A_STAGE, ob.rd, lane, 1, b1);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma<1, 2>(a0, b0, acc);
            if (more) oa.issue_one(ta, wv, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma<2, 3>(a0, b0, acc);
            if (more) oa.issue_one(ta, wv, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma<3, 4>(a0, b0, acc);
            if (more) ob.issue_one(ta + A_STAGE, wv, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma<0, 1>(a1, b1, acc);
            if (more) ob.issue_one(ta + A_STAGE, wv, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < n && !(ABL & 4)) {
                const float* sn = smem + nxt * STAGE;
                read_frag<AK>(sn, oa.rd, lane, 0, a0);
                read_frag<BKIND>(sn + A_STAGE, ob.rd, lane, 0, b0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma<1, 4>(a1, b1, acc);
            if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
            cur = nxt;
        }
        if (stamps && items_ == 0) st_[2] = wall_clock64();
        const int h = lane >> 5, l = lane & 31;
        if (slab == nullptr) {
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long row = m0 + wm * 64 + bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row >= M) continue;
#pragma unroll
                    for (int bj = 0; bj < 2; ++bj) {
                        const int col = n0 + wn * 64 + bj * 32 + l;
                        if (col < N) C[row * ldc + col] = acc[bi][bj][r];
                    }
                }
        } else {
            // accumulator order: [(wave * 4 + block) * 4 + r4][lane][4]  -- 1 KB per wave store
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < 2; ++bj)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        f32x4 x = {acc[bi][bj][4 * r4], acc[bi][bj][4 * r4 + 1], acc[bi][bj][4 * r4 + 2], acc[bi][bj][4 * r4 + 3]};
                        *reinterpret_cast<f32x4*>(slab + (((wv * 4 + bi * 2 + bj) * 4 + r4) * 64 + lane) * 4) = x;
                    }
        }
        if (stamps) {
            if (items_ == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st_[3] = wall_clock64(); }
            ++items_;
        }
    };

    for (int d = 0; d < pl.dp_rounds; ++d) run_item(d * (int)P + (int)pid, 0, nk, nullptr);
    if (pl.sk_tiles > 0) {
        const long total = (long)pl.sk_tiles * nk;
        long it = (long)pid * total / P;
        const long end = (long)(pid + 1) * total / P;
        int which = 0;
        while (it < end) {
            const int t = (int)(it / nk);
            const int kb = (int)(it - (long)t * nk);
            int ke = kb + (int)(end - it);
            if (ke > nk) ke = nk;
            const bool whole = kb == 0 && ke == nk;
            run_item(pl.sk_first + t, kb, ke, whole ? nullptr : slabs + ((long)pid * 2 + which) * (BM * BN));
            if (!whole) ++which;
            it += ke - kb;
        }
    }
    if (stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st_[4] = wall_clock64();
        st_[5] = items_;
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        st_[6] = hwid;
        st_[7] = xcc;
        for (int i = 0; i < 8; ++i) stamps[(long)blockIdx.x * 8 + i] = st_[i];
    }
    if (clk && blockIdx.x == 0 && tid == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - c0;
        clk[1] = wall_clock64() - w0;
    }
}

// sums the slabs of every split tile in k order (= workgroup order) and stores the tile
__global__ __launch_bounds__(256) void sk_fixup_kernel(const float* __restrict__ slabs, float* __restrict__ C, long ldc, long M, int N, Plan pl,
                                                       unsigned P) {
    const int t = blockIdx.x;                       // streamed tile index
    const int nk = pl.nk;
    const long total = (long)pl.sk_tiles * nk;
    const long ib = (long)t * nk, ie = ib + nk;
    // contributors: workgroups p with [beg_p, end_p) intersecting [ib, ie), beg_p = p * total / P
    long pa = ib * P / total;
    while (pa > 0 && (pa * total / P) > ib) --pa;
    while (((pa + 1) * total / P) <= ib) ++pa;
    const int tile = pl.sk_first + t;
    const int tn = tile % pl.tiles_n, tm = tile / pl.tiles_n;
    const long m0 = (long)tm * BM;
    const int n0 = tn * BN;
    // whole tiles (one contributor covering everything) were stored by the GEMM kernel
    {
        const long b0 = pa * total / P, e0 = (pa + 1) * total / P;
        if (b0 <= ib && e0 >= ie) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1, h = lane >> 5, l = lane & 31;
    {
        const int blk = blockIdx.y >> 2, r4 = blockIdx.y & 3;
        {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (long p = pa; p < P; ++p) {
                const long b = p * total / P, e = (p + 1) * total / P;
                if (b >= ie) break;
                if (e <= ib || b == e) continue;
                // first partial item of p is the tile holding b (if it does not start a tile or ends inside it)
                const long t0 = b / nk;
                const bool first_partial = !(b == t0 * nk && e >= (t0 + 1) * nk);
                const int which = (t0 == t) ? 0 : (first_partial ? 1 : 0);
                const float* sl = slabs + (p * 2 + which) * (long)(BM * BN);
                s += *reinterpret_cast<const f32x4*>(sl + (((wv * 4 + blk) * 4 + r4) * 64 + lane) * 4);
            }
            const int bi = blk >> 1, bj = blk & 1;
            const int col = n0 + wn * 64 + bj * 32 + l;
            for (int j = 0; j < 4; ++j) {
                const long row = m0 + wm * 64 + bi * 32 + j + 8 * r4 + 4 * h;
                if (row < M && col < N) C[row * ldc + col] = s[j];
            }
        }
    }
}

// reference: one thread per element, k-ordered fma chain
__global__ void ref_kernel(const float* A, long lda, int ak, const float* B, long ldb, int bk, float* C, long ldc, long M, int N, int K) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= M * N) return;
    const long m = i / N;
    const int n = (int)(i % N);
    double s = 0.0;
    for (int k = 0; k < K; ++k) {
        const float a = ak == KINNER ? A[m * lda + k] : A[(long)k * lda + m];
        const float b = bk == KOUTER ? B[(long)k * ldb + n] : B[(long)n * ldb + k];
        s += (double)a * b;
    }
    C[m * ldc + n] = (float)s;
}

static double g_main_us = 0;
struct Shape {
    const char* name;
    int kind;   // 0 nn, 1 nt, 2 tn
    long M;
    int N, K;
};

template <int AK, int BKIND, int STAGES, int WGCU, int PRIO>
double run(const Shape& sh, const float* A, long lda, const float* B, long ldb, float* C, float* slabs, int mode, bool check,
           const float* Cref, int reps) {
    const long M = sh.M;
    const int N = sh.N, K = sh.K;
    Plan pl;
    pl.tiles_n = (N + BN - 1) / BN;
    const int tiles_m = (int)((M + BM - 1) / BM);
    pl.ntiles = tiles_m * pl.tiles_n;
    pl.nk = K / BK;
    unsigned P = 256 * WGCU;
    if (mode == 0) {                 // data parallel only: one tile per workgroup, classic grid
        P = pl.ntiles;
        pl.dp_rounds = 1; pl.sk_tiles = 0; pl.sk_first = 0;
    } else if (mode == 1) {          // hybrid: whole rounds + stream-K remainder
        pl.dp_rounds = pl.ntiles / P;
        pl.sk_first = pl.dp_rounds * P;
        pl.sk_tiles = pl.ntiles - pl.sk_first;
    } else {                         // hybrid, remainder + one round streamed ("two-tile")
        pl.dp_rounds = pl.ntiles / P;
        if (pl.dp_rounds > 0 && pl.ntiles % P != 0) --pl.dp_rounds;
        pl.sk_first = pl.dp_rounds * P;
        pl.sk_tiles = pl.ntiles - pl.sk_first;
    }
    const size_t lds = (size_t)STAGES * STAGE * sizeof(float);
    auto kern = sk_kernel<AK, BKIND, STAGES, WGCU, PRIO>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto launch = [&](bool fix = true) {
        hipLaunchKernelGGL(kern, dim3(P), dim3(256), lds, 0, A, lda, B, ldb, C, (long)N, M, N, pl, slabs, (long long*)nullptr, (long long*)nullptr);
        if (fix && pl.sk_tiles > 0) hipLaunchKernelGGL(sk_fixup_kernel, dim3(pl.sk_tiles, 16), dim3(256), 0, 0, slabs, C, (long)N, M, N, pl, P);
    };
    if (check) {
        CK(hipMemset(C, 0xff, (size_t)M * N * 4));
        launch();
        CK(hipDeviceSynchronize());
        std::vector<float> c((size_t)M * N), r((size_t)M * N);
        CK(hipMemcpy(c.data(), C, c.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(r.data(), Cref, r.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        for (size_t i = 0; i < c.size(); ++i) {
            const double e = fabs((double)c[i] - r[i]);
            if (!(e <= maxerr)) maxerr = e;
            if (fabs(r[i]) > maxref) maxref = fabs(r[i]);
        }
        if (!(maxerr <= 2e-5 * maxref * sqrt((double)K) / 8 + 1e-6)) printf("    !! MISMATCH max err %g (max |ref| %g)\n", maxerr, maxref);
    }
    for (int i = 0; i < 3; ++i) launch();
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch(false);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms2;
    CK(hipEventElapsedTime(&ms2, e0, e1));
    g_main_us = ms2 * 1e3 / reps;
    return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
    const Shape shapes[] = {
        {"frame2 fwd", 0, 25344, 512, 1536}, {"frame3 fwd", 0, 8448, 512, 1536}, {"frame4 fwd", 0, 8448, 512, 512},
        {"frame5 fwd", 0, 8448, 1500, 512},  {"frame1 fwd*", 0, 50688, 512, 192}, {"frame5 dgrad*", 1, 8448, 512, 1504},
        {"frame4 dgrad", 1, 8448, 512, 512}, {"frame3 dgrad", 1, 8448, 1536, 512}, {"frame2 dgrad0", 1, 25344, 1024, 512},
        {"frame2 dgrad1", 1, 25344, 512, 512}, {"frame2 wgrad", 2, 1536, 512, 25344}, {"frame3 wgrad", 2, 1536, 512, 8448},
        {"frame4 wgrad", 2, 512, 512, 8448}, {"frame5 wgrad", 2, 512, 1500, 8448}, {"frame1 wgrad*", 2, 256, 512, 50688},
    };
    const bool check = argc > 1 && !strcmp(argv[1], "check");
    const size_t maxA = (size_t)50688 * 1536 + 4096, maxC = (size_t)50688 * 1536;
    float *A, *B, *C, *Cref, *slabs;
    CK(hipMalloc(&A, maxA * 4)); CK(hipMalloc(&B, maxA * 4)); CK(hipMalloc(&C, maxC * 4)); CK(hipMalloc(&Cref, maxC * 4));
    CK(hipMalloc(&slabs, (size_t)768 * 2 * BM * BN * 4));
    {
        std::vector<float> h(maxA);
        unsigned s = 12345u;
        for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
        CK(hipMemcpy(A, h.data(), maxA * 4, hipMemcpyHostToDevice));
        for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
        CK(hipMemcpy(B, h.data(), maxA * 4, hipMemcpyHostToDevice));
    }
    const int reps = 20;
    if (argc > 1 && !strcmp(argv[1], "timeline")) {
        long long* stamps;
        CK(hipMalloc(&stamps, 768 * 8 * 8));
        auto tl = [&](const char* name, long M, int N, int K, int mode, int prio) {
            Plan pl;
            pl.tiles_n = (N + BN - 1) / BN;
            pl.ntiles = (int)((M + BM - 1) / BM) * pl.tiles_n;
            pl.nk = K / BK;
            const unsigned P = 768;
            pl.dp_rounds = pl.ntiles / P;
            if (mode == 2 && pl.dp_rounds > 0 && pl.ntiles % P != 0) --pl.dp_rounds;
            pl.sk_first = pl.dp_rounds * P;
            pl.sk_tiles = pl.ntiles - pl.sk_first;
            auto kern = prio ? sk_kernel<KINNER, KOUTER, 3, 3, 2, 0> : sk_kernel<KINNER, KOUTER, 3, 3, 0, 0>;
            const size_t lds = (size_t)3 * STAGE * 4;
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            for (int i = 0; i < 3; ++i)
                hipLaunchKernelGGL(kern, dim3(P), dim3(256), lds, 0, A, (long)K, B, (long)N, C, (long)N, M, N, pl, slabs, (long long*)nullptr, stamps);
            CK(hipDeviceSynchronize());
            std::vector<long long> h(768 * 8);
            CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
            long long t0 = h[0];
            for (unsigned p = 0; p < P; ++p) if (h[p * 8] < t0) t0 = h[p * 8];
            printf("%s prio-rotate %d M=%ld N=%d K=%d mode %d: dp %d sk_tiles %d; per-WG stamps in us since the first entry (min / median / max over 768 WGs)\n", name, prio, M, N, K, mode, pl.dp_rounds, pl.sk_tiles);
            const char* nm[5] = {"entry", "item0 loop start", "item0 loop end", "item0 stored", "exit"};
            for (int i = 0; i < 5; ++i) {
                std::vector<double> v;
                for (unsigned p = 0; p < P; ++p) v.push_back((h[p * 8 + i] - t0) / 100.0);
                std::sort(v.begin(), v.end());
                printf("   %-18s %8.1f %8.1f %8.1f\n", nm[i], v[0], v[P / 2], v[P - 1]);
            }
        };
        tl("frame3 fwd", 8448, 512, 1536, 2, 0);
        tl("frame3 fwd", 8448, 512, 1536, 2, 1);
        {   // group the last run by CU
            std::vector<long long> h(768 * 8);
            CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
            long long t0 = h[0];
            for (unsigned p = 0; p < 768; ++p) if (h[p * 8] < t0) t0 = h[p * 8];
            std::vector<std::pair<long long, unsigned>> keyed;
            for (unsigned p = 0; p < 768; ++p) {
                const unsigned hw = (unsigned)h[p * 8 + 6], xcc = (unsigned)h[p * 8 + 7] & 15u;
                const long long key = ((long long)xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 7) | (((hw >> 8) & 15));
                keyed.push_back({key, p});
            }
            std::sort(keyed.begin(), keyed.end());
            printf("   per CU (xcc/se/sh/cu): block ids with [entry, exit] us\n");
            int shown = 0;
            for (size_t i = 0; i < keyed.size() && shown < 40; ++i) {
                if (i == 0 || keyed[i].first != keyed[i - 1].first) { printf("\n   cu %06llx:", keyed[i].first); ++shown; }
                const unsigned p = keyed[i].second;
                printf("  b%u[%.1f, %.1f]", p, (h[p * 8] - t0) / 100.0, (h[p * 8 + 4] - t0) / 100.0);
            }
            printf("\n");
        }
        tl("frame4 fwd", 8448, 512, 512, 2, 0);
        tl("frame4 fwd", 8448, 512, 512, 2, 1);
        tl("frame2 fwd", 25344, 512, 1536, 2, 0);
        tl("frame2 fwd", 25344, 512, 1536, 2, 1);
        tl("frame2 fwd", 25344, 512, 1536, 1, 1);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "ablate")) {
        long long* clk;
        CK(hipMalloc(&clk, 16));
#define ABRUN(ST, WG, ABLV, LDA_PAD)                                                                                                      \
    {                                                                                                                            \
        const long M = 128L * 64 * WG;                                                                                           \
        const int N = 512, K = 2048;                                                                                             \
        Plan pl{4, (int)(M / 128) * 4, K / BK, 1, 0, 0};                                                                         \
        auto kern = sk_kernel<KINNER, KOUTER, ST, WG, 0, ABLV>;                                                                  \
        const size_t lds = (size_t)ST * STAGE * 4;                                                                               \
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
        hipEvent_t e0, e1;                                                                                                       \
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));                                                                        \
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256 * WG), dim3(256), lds, 0, A, (long)K + LDA_PAD, B, (long)N, C, (long)N, M, N, pl, slabs, clk, (long long*)nullptr); \
        CK(hipEventRecord(e0));                                                                                                  \
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(256 * WG), dim3(256), lds, 0, A, (long)K + LDA_PAD, B, (long)N, C, (long)N, M, N, pl, slabs, clk, (long long*)nullptr); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                                                     \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                                                          \
        long long hc[2]; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));                                                      \
        const double us = ms * 1e3 / reps, gf = 2.0 * M * N * K * 1e-9;                                                          \
        printf("ablate stages %d wg/cu %d abl %2d lda K+%-3d (1 no dma, 2 no barrier, 4 no ds_read, 8 no vmcnt wait): %7.1f us  %6.1f TF (%.3f)  wg0: %lld cycles in %.1f us = %.0f MHz, %.0f cycles per K step\n", \
               ST, WG, ABLV, LDA_PAD, us, gf / us * 1e3, gf / us * 1e3 / 157.3, hc[0], hc[1] / 100.0, hc[0] / (hc[1] / 100.0), (double)hc[0] / (K / BK));   \
    }
        ABRUN(4, 1, 0, 0) ABRUN(4, 1, 8, 0) ABRUN(4, 1, 0, 32) ABRUN(4, 1, 0, 48) ABRUN(4, 1, 0, 544) ABRUN(4, 1, 8, 32) ABRUN(4, 1, 1, 0) ABRUN(4, 1, 7, 0)
        ABRUN(4, 2, 0, 0) ABRUN(4, 2, 8, 0) ABRUN(4, 2, 0, 32) ABRUN(4, 2, 0, 48) ABRUN(4, 2, 0, 544) ABRUN(4, 2, 8, 32) ABRUN(4, 2, 1, 0) ABRUN(4, 2, 7, 0)
        ABRUN(3, 3, 0, 0) ABRUN(3, 3, 8, 0) ABRUN(3, 3, 0, 32) ABRUN(3, 3, 0, 48) ABRUN(3, 3, 1, 0)
        return 0;
    }
    for (const Shape& sh : shapes) {
        const long M = sh.M;
        const int N = sh.N, K = sh.K;
        long lda, ldb;
        int ak, bk;
        if (sh.kind == 0) { ak = KINNER; bk = KOUTER; lda = K; ldb = N; }
        else if (sh.kind == 1) { ak = KINNER; bk = KINNER; lda = K; ldb = K; }
        else { ak = KOUTER; bk = KOUTER; lda = M; ldb = N; }
        const double gflop = 2.0 * M * N * K * 1e-9;
        printf("%-14s %s M=%ld N=%d K=%d  (%.2f GFLOP, %.1f us at 157.3 TF)\n", sh.name, sh.kind == 0 ? "nn" : sh.kind == 1 ? "nt" : "tn", M, N, K,
               gflop, gflop / 157.3e3 * 1e6);
        if (check) {
            hipLaunchKernelGGL(ref_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, 0, A, lda, ak, B, ldb, bk, Cref, (long)N, M, N, K);
            CK(hipDeviceSynchronize());
        }
#define RUN(AKK, BKK, ST, WG, MODE, PRIO)                                                                                  \
    {                                                                                                                      \
        const double us = run<AKK, BKK, ST, WG, PRIO>(sh, A, lda, B, ldb, C, slabs, MODE, check, Cref, reps);              \
        printf("    stages %d  wg/cu %d  mode %d prio %d : %8.1f us  %6.1f TF  (%.3f)   main kernel alone %8.1f us (%.3f)\n", ST, WG, MODE, PRIO, us,          \
               gflop / us * 1e3, gflop / us * 1e3 / 157.3, g_main_us, gflop / g_main_us * 1e3 / 157.3);                                                                \
    }
#define RUN_ALL(AKK, BKK)                \
    RUN(AKK, BKK, 3, 3, 0, 0)            \
    RUN(AKK, BKK, 3, 3, 2, 0)            \
    RUN(AKK, BKK, 4, 2, 0, 0)            \
    RUN(AKK, BKK, 4, 2, 1, 0)            \
    RUN(AKK, BKK, 4, 2, 2, 0)            \
    RUN(AKK, BKK, 4, 2, 2, 2)            \
    RUN(AKK, BKK, 3, 3, 2, 2)            \
    RUN(AKK, BKK, 4, 1, 0, 0)            \
    RUN(AKK, BKK, 4, 1, 2, 0)
        if (sh.kind == 0) { RUN_ALL(KINNER, KOUTER) }
        else if (sh.kind == 1) { RUN_ALL(KINNER, KINNER) }
        else { RUN_ALL(KOUTER, KOUTER) }
    }
    return 0;
}
