// Microbenchmark: decompose the K-step of gemm_bf16.hip's NN loop (128x128x32 tile, 256 threads).
// FEAT bits: 1 = global loads (8 x dwordx4 per thread per K-step), 2 = cvt + LDS stores, 4 = barrier,
//            8 = LDS operand reads + MFMAs, 16 = consume loads with a cheap VALU sum when not stored
// build: hipcc -O3 --offload-arch=gfx950 -o bf16_loop bf16_loop.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int LDT = 40;

template <int FEAT, int DEPTH, int SKEW>
__global__ __launch_bounds__(256) void kloop(const float* __restrict__ A, const float* __restrict__ B, float* out, int ksteps,
                                             long lda, long ldb, int kwrap) {
    __shared__ __attribute__((aligned(16))) __bf16 As[2][128 * LDT];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][128 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 128 * LDT; i += 256) { (&As[0][0])[i] = (__bf16)(float)(i & 7); (&Bs[0][0])[i] = (__bf16)(float)(i & 3); }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int h = lane >> 5, l = lane & 31, wm = wave >> 1, wn = wave & 1;
    const int kq = tid & 7, rs = tid >> 3;
    const int r0 = (rs & 0x1a) | ((rs & 1) << 2) | ((rs >> 2) & 1);
    // A panel shared by the 4 n-tiles of an m-tile (consecutive blocks), B shared by everyone
    // SKEW 1: every workgroup starts its K walk somewhere else (wraps around), so concurrent workgroups do not
    // request the same B lines (and the 4 n-tiles of an m-tile not the same A lines) at the same time
    const int kskew = SKEW ? (int)((blockIdx.x * 37u) % (unsigned)ksteps) : 0;
    const float* ap0 = A + ((long)(blockIdx.x >> 2) * 128 + r0) * lda + 4 * kq;
    const float* bp0 = B + (long)(4 * kq) * ldb + (blockIdx.x & 3) * 128 + 4 * rs;
    const float* ap = ap0 + 32 * kskew;
    const float* bp = bp0 + (long)32 * kskew * ldb;
    int kpos = kskew;
    int kw = 0;                                        // kwrap > 0: restart the K walk every kwrap steps (small, L2-resident panels)
    f32x4 va[DEPTH][4], vb[DEPTH][4];
    for (int d = 0; d < DEPTH; ++d) for (int p = 0; p < 4; ++p) { va[d][p] = f32x4{1, 2, 3, 4}; vb[d][p] = f32x4{1, 2, 3, 4}; }
    f32x4 junk = {0, 0, 0, 0};
    // prologue for DEPTH > 1: DEPTH-1 tiles already in flight
    if (FEAT & 1) {
        for (int d = 0; d + 1 < DEPTH; ++d) {
            for (int p = 0; p < 4; ++p) va[d][p] = *reinterpret_cast<const f32x4*>(ap + (long)(32 * p) * lda);
            for (int p = 0; p < 4; ++p) vb[d][p] = *reinterpret_cast<const f32x4*>(bp + (long)p * ldb);
            ap += 32; bp += 32 * ldb;
        }
    }
#pragma unroll 1
    for (int kt0 = 0; kt0 < ksteps; kt0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int kt = kt0 + d;
            const int cur = kt & 1;
            const int slot_new = (d + DEPTH - 1) % DEPTH;      // register slot the new loads land in
            if (FEAT & 1) {
#pragma unroll
                for (int p = 0; p < 4; ++p) va[slot_new][p] = *reinterpret_cast<const f32x4*>(ap + (long)(32 * p) * lda);
#pragma unroll
                for (int p = 0; p < 4; ++p) vb[slot_new][p] = *reinterpret_cast<const f32x4*>(bp + (long)p * ldb);
                ap += 32; bp += 32 * ldb;
                if (SKEW && ++kpos == ksteps) { kpos = 0; ap = ap0; bp = bp0; }
                if (kwrap && ++kw == kwrap) { kw = 0; ap = ap0; bp = bp0; }
            }
            if (FEAT & 8) {
                const __bf16* pa = As[cur] + (wm * 64 + l) * LDT + 8 * h;
                const __bf16* pb = Bs[cur] + (wn * 64 + l) * LDT + 8 * h;
                bf16x8 a[2][2], b[2][2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        a[ks][i] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * LDT + ks * 16);
                        b[ks][i] = *reinterpret_cast<const bf16x8*>(pb + i * 32 * LDT + ks * 16);
                    }
                __builtin_amdgcn_s_setprio(3);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
            // the tile consumed now is the one loaded DEPTH-1 steps ago: slot d
            if (FEAT & 2) {
                __bf16* da = As[cur ^ 1];
                __bf16* db = Bs[cur ^ 1];
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    *reinterpret_cast<bf16x4*>(da + (r0 + 32 * p) * LDT + 4 * kq) = __builtin_convertvector(va[d][p], bf16x4);
                __bf16* dd = db + (4 * rs) * LDT + 4 * kq;
                *reinterpret_cast<bf16x4*>(dd + 0 * LDT) = __builtin_convertvector(f32x4{vb[d][0].x, vb[d][1].x, vb[d][2].x, vb[d][3].x}, bf16x4);
                *reinterpret_cast<bf16x4*>(dd + 1 * LDT) = __builtin_convertvector(f32x4{vb[d][0].y, vb[d][1].y, vb[d][2].y, vb[d][3].y}, bf16x4);
                *reinterpret_cast<bf16x4*>(dd + 2 * LDT) = __builtin_convertvector(f32x4{vb[d][0].z, vb[d][1].z, vb[d][2].z, vb[d][3].z}, bf16x4);
                *reinterpret_cast<bf16x4*>(dd + 3 * LDT) = __builtin_convertvector(f32x4{vb[d][0].w, vb[d][1].w, vb[d][2].w, vb[d][3].w}, bf16x4);
            } else if (FEAT & 16) {
#pragma unroll
                for (int p = 0; p < 4; ++p) junk += va[d][p] + vb[d][p];
            }
            if (FEAT & 4) __syncthreads();
        }
    }
    float s = junk.x + junk.y + junk.z + junk.w;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int FEAT, int DEPTH, int SKEW = 0>
static void run(const char* name, const float* A, const float* B, float* out, int grid, int ksteps, long lda, int kwrap = 0) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((kloop<FEAT, DEPTH, SKEW>), dim3(grid), dim3(256), 0, 0, A, B, out, ksteps, lda, 512L, kwrap);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((kloop<FEAT, DEPTH, SKEW>), dim3(grid), dim3(256), 0, 0, A, B, out, ksteps, lda, 512L, kwrap);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 200.0;
    const double cyc_per_step_per_cu = us * 2400.0 / ksteps / (grid / 256.0);
    const double bytes = (double)grid * ksteps * 32768.0;
    printf("  %-58s %8.1f us  %7.1f cyc/K-step/CU  %5.1f B/clk/CU  (%6.1f TF/s equiv)\n", name, us, cyc_per_step_per_cu,
           bytes / (us * 2400.0) / 256.0, (double)grid * ksteps * 2.0 * 128 * 128 * 32 / us / 1e6);
}

int main() {
    float *A, *B, *out;
    const int ksteps = 96;                                   // K = 3072
    const long lda = 96 * 32;
    (void)hipMalloc(&A, (size_t)(256 * 128 + 1) * lda * 4);       // 256 m-tiles
    (void)hipMemset(A, 0, (size_t)(256 * 128 + 1) * lda * 4);
    (void)hipMalloc(&B, (size_t)(lda + 256) * 512 * 4);      // + the prefetch overrun of the deeper variants
    (void)hipMemset(B, 0, (size_t)(lda + 256) * 512 * 4);
    (void)hipMalloc(&out, 1024 * 256 * 4);
    for (int w = 1; w <= 3; ++w) {
        const int grid = 256 * w;
        printf("%d WG/CU (grid %d), %d K-steps of 32\n", w, grid, ksteps);
        run<8, 1>("LDS reads + MFMA only", A, B, out, grid, ksteps, lda);
        run<1 | 16, 1>("global loads only (consumed by VALU adds)", A, B, out, grid, ksteps, lda);
        run<1 | 16 | 4, 1>("global loads + barrier", A, B, out, grid, ksteps, lda);
        run<1 | 2, 1>("global loads + cvt + LDS stores", A, B, out, grid, ksteps, lda);
        run<1 | 2 | 4, 1>("global loads + cvt + LDS stores + barrier", A, B, out, grid, ksteps, lda);
        run<2 | 4 | 8, 1>("everything but the global loads", A, B, out, grid, ksteps, lda);
        run<1 | 2 | 4 | 8, 1>("full loop (prefetch distance 1)", A, B, out, grid, ksteps, lda);
        run<1 | 2 | 4 | 8, 2>("full loop, prefetch distance 2", A, B, out, grid, ksteps, lda);
        run<1 | 2 | 4 | 8, 3>("full loop, prefetch distance 3", A, B, out, grid, ksteps, lda);
        run<1 | 16, 1, 1>("global loads only, K start skewed per workgroup", A, B, out, grid, ksteps, lda);
        run<1 | 2 | 4 | 8, 1, 1>("full loop, K start skewed per workgroup", A, B, out, grid, ksteps, lda);
        run<1 | 2 | 4 | 8, 2, 1>("full loop, distance 2, K start skewed", A, B, out, grid, ksteps, lda);
        run<1 | 16, 1>("global loads only, K range 8 steps re-walked (B 0.5 MB)", A, B, out, grid, ksteps, lda, 8);
        run<1 | 2 | 4 | 8, 1>("full loop, K range 8 steps re-walked", A, B, out, grid, ksteps, lda, 8);
        run<1 | 2 | 4 | 8, 2>("full loop, distance 2, K range 8 steps re-walked", A, B, out, grid, ksteps, lda, 8);
        run<1 | 16, 2>("global loads only, distance 2", A, B, out, grid, ksteps, lda);
        run<1 | 16, 3>("global loads only, distance 3", A, B, out, grid, ksteps, lda);
    }
    return 0;
}
