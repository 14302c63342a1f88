// Microbenchmark: fp32 MFMA issue rate on gfx950 with/without LDS operand reads.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_peak tools/micro/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: MFMA only (operands in registers)
// MODE 1: 2 ds_read2_b32-equivalents per 4 MFMA (as in gemm.hip), prefetched one group ahead
// MODE 2: same reads but no prefetch (read -> wait -> 4 MFMA)
// MODE 3: 16x16x4 MFMA, registers only
template <int MODE, int ITERS>
__global__ __launch_bounds__(256) void k32(float* out, int lda) {
    __shared__ float As[16 * 130], Bs[16 * 128];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16 * 130; i += 256) As[i] = (float)(i & 7);
    for (int i = tid; i < 16 * 128; i += 256) Bs[i] = (float)(i & 3);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int h = lane >> 5, l = lane & 31;
    const float* ap = As + h * lda + l;
    const float* bp = Bs + h * 128 + l;
    float a0 = ap[0], a1 = ap[32], b0 = bp[0], b1 = bp[32];
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            float na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
            if (MODE == 1) {
                const int k2 = (kk + 2) & 15;
                na0 = ap[k2 * lda]; na1 = ap[k2 * lda + 32]; nb0 = bp[k2 * 128]; nb1 = bp[k2 * 128 + 32];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 2) {
                a0 = ap[kk * lda]; a1 = ap[kk * lda + 32]; b0 = bp[kk * 128]; b1 = bp[kk * 128 + 32];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            if (MODE == 1) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int ITERS>
__global__ __launch_bounds__(256) void k16(float* out) {
    const int tid = threadIdx.x;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = (float)(tid & 3), b = (float)(tid & 7);
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <typename F>
static void run(const char* name, F launch, double flops) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.1f us  %6.1f TF/s\n", name, ms * 200.0, flops * 5 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    constexpr int IT = 400;
    for (int wgs_per_cu = 1; wgs_per_cu <= 4; ++wgs_per_cu) {
        const int grid = 256 * wgs_per_cu;
        const double fl32 = (double)grid * 4 * IT * 8 * 4 * 4096.0;       // waves * iters * kk * 4 mfma * 4096 flop
        char nm[128];
        snprintf(nm, sizeof nm, "32x32x2 regs only          %d WG/CU", wgs_per_cu);
        run(nm, [&] { hipLaunchKernelGGL((k32<0, IT>), dim3(grid), dim3(256), 0, 0, out, 130); }, fl32);
        snprintf(nm, sizeof nm, "32x32x2 + LDS reads prefetch %d WG/CU", wgs_per_cu);
        run(nm, [&] { hipLaunchKernelGGL((k32<1, IT>), dim3(grid), dim3(256), 0, 0, out, 130); }, fl32);
        snprintf(nm, sizeof nm, "32x32x2 + LDS reads serial   %d WG/CU", wgs_per_cu);
        run(nm, [&] { hipLaunchKernelGGL((k32<2, IT>), dim3(grid), dim3(256), 0, 0, out, 130); }, fl32);
        const double fl16 = (double)grid * 4 * IT * 8 * 8 * 2048.0;
        snprintf(nm, sizeof nm, "16x16x4 regs only          %d WG/CU", wgs_per_cu);
        run(nm, [&] { hipLaunchKernelGGL((k16<IT>), dim3(grid), dim3(256), 0, 0, out); }, fl16);
    }
    return 0;
}
