// Microbenchmark: issue cost of plain vs packed fp32 VALU ops on gfx950 (cycles per wave64 instruction per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a[16]; v2f p[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed * (i + 1) + threadIdx.x; p[i] = v2f{seed * i, seed + i}; }
    const float c = seed * 0.5f; const v2f pc = v2f{seed, seed * 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) a[i] = a[i] + c;                       // v_add_f32
                if (MODE == 1) a[i] = __builtin_fmaf(a[i], c, c);     // v_fma_f32
                if (MODE == 2) p[i] = p[i] + pc;                      // v_pk_add_f32
                if (MODE == 3) p[i] = p[i] * pc + pc;                 // v_pk_fma_f32
                if (MODE == 4) a[i] = a[i] * c;                       // v_mul_f32
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, float* out, int wg_per_cu) {
    const int iters = 2000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f);
    hipDeviceSynchronize(); hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)wg_per_cu * iters * 64.0;            // wave-instructions per SIMD (1 wave/SIMD/WG)
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("  %-14s %d wave/SIMD: %7.3f ms  %.2f cycles per wave-instruction at 2.4 GHz\n", name, wg_per_cu, ms, cyc / inst_per_simd);
}

int main() {
    float* out; hipMalloc(&out, 2048 * 256 * 4);
    for (int w : {1, 2, 4}) {
        run<0>("v_add_f32", out, w); run<4>("v_mul_f32", out, w); run<1>("v_fma_f32", out, w);
        run<2>("v_pk_add_f32", out, w); run<3>("v_pk_fma_f32", out, w);
    }
    return 0;
}
