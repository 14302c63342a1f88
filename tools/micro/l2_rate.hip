// Microbenchmark: per-CU load rate from L2 (L1-missing footprint) for a few access shapes.
// build: hipcc -O3 --offload-arch=gfx950 -o l2_rate l2_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: dwordx4, wave covers 1 KB contiguous      MODE 1: dwordx4, wave covers 8 rows x 128 B (row stride 256 B)
// MODE 2: dwordx2 contiguous (512 B per wave-load)   MODE 3: dword contiguous (256 B)
// MODE 4: dwordx4, 16 rows x 64 B                    MODE 5: global_load_lds dwordx4 contiguous (LDS-DMA)
template <int MODE>
__global__ __launch_bounds__(256) void rd(const float* __restrict__ g, float* out, int iters, long wg_stride, int span_floats,
                                          int row_stride) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 4 * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* base = g + (long)blockIdx.x * wg_stride;
    f32x4 acc = {0, 0, 0, 0};
    int pos = 0;                                       // walks a span_floats window cyclically (L2-resident, > L1)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {
                acc += *reinterpret_cast<const f32x4*>(base + pos + (wave * 8 + u) * 256 + lane * 4);
            } else if (MODE == 1) {
                acc += *reinterpret_cast<const f32x4*>(base + pos + (long)((wave * 8 + u) * 8 + (lane >> 3)) * 64 + (lane & 7) * 4);
            } else if (MODE == 6) {
                acc += *reinterpret_cast<const f32x4*>(base + pos + (long)((wave * 8 + u) * 8 + (lane >> 3)) * row_stride + (lane & 7) * 4);
            } else if (MODE == 2) {
                const f32x2 a = *reinterpret_cast<const f32x2*>(base + pos + (wave * 8 + u) * 128 + lane * 2);
                acc.x += a.x; acc.y += a.y;
            } else if (MODE == 3) {
                acc.x += base[pos + (wave * 8 + u) * 64 + lane];
            } else if (MODE == 4) {
                acc += *reinterpret_cast<const f32x4*>(base + pos + (long)((wave * 8 + u) * 16 + (lane >> 2)) * 32 + (lane & 3) * 4);
            } else {
                __builtin_amdgcn_global_load_lds(base + pos + (wave * 8 + u) * 256 + lane * 4,
                                                 (__attribute__((address_space(3))) void*)(lds + (wave * 8 + u) * 256), 16, 0, 0);
            }
        }
        const int step = MODE == 0 || MODE == 5 ? 8192 : MODE == 2 ? 4096 : MODE == 3 ? 2048 : 32;
        pos += step;
        if (pos >= span_floats) pos = 0;
        if (MODE == 5) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    }
    if (MODE == 5) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc.x += lds[tid]; }
    out[blockIdx.x * 256 + tid] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE>
static void run(const char* name, const float* g, float* out, int grid, int iters, long wg_stride, int span, double bytes_per_iter,
                int row_stride = 0) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(rd<MODE>, dim3(grid), dim3(256), 0, 0, g, out, iters, wg_stride, span, row_stride);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(rd<MODE>, dim3(grid), dim3(256), 0, 0, g, out, iters, wg_stride, span, row_stride);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000.0 / 3;
    const double bytes = (double)grid * iters * bytes_per_iter;
    printf("  %-52s %8.1f us  %6.1f B/clk/CU  %6.2f TB/s\n", name, us, bytes / (us * 2400.0) / 256.0, bytes / us / 1e6);
}

int main() {
    float *g, *out;
    const size_t n = ((size_t)1 << 28) + ((size_t)1 << 20);   // 1 GiB of floats + slack for the strided shapes
    (void)hipMalloc(&g, n * 4);
    (void)hipMemset(g, 0, n * 4);
    (void)hipMalloc(&out, 2048 * 256 * 4);
    const int iters = 400;
    for (int w = 1; w <= 2; ++w) {
        const int grid = 256 * w;
        printf("%d WG/CU: each workgroup cycles over its own 64 KB window (L2 hits, L1 misses; %d MB per XCD)\n", w, 2 * w);
        const long stride = 1 << 18;                   // 1 MB apart
        run<0>("dwordx4, 1 KB contiguous per wave-load", g, out, grid, iters, stride, 16384, 32768);
        run<1>("dwordx4, 8 rows x 128 B per wave-load", g, out, grid, iters, stride, 64, 32768);
        run<4>("dwordx4, 16 rows x 64 B per wave-load", g, out, grid, iters, stride, 32, 32768);
        // 256 rows x 128 B per iteration; pos walks 32 floats per iteration inside each row (span = row length used)
        run<6>("8 rows x 128 B, row stride 256 B, private window", g, out, grid, iters, stride, 64, 32768, 64);
        run<6>("8 rows x 128 B, row stride 2 KB, window shared by all", g, out, grid, iters, 0, 512, 32768, 512);
        run<6>("8 rows x 128 B, row stride 2 KB + 128 B, shared", g, out, grid, iters, 0, 512, 32768, 544);
        run<6>("8 rows x 128 B, row stride 12 KB, shared", g, out, grid, iters, 0, 512, 32768, 3072);
        run<6>("8 rows x 128 B, row stride 12 KB + 128 B, shared", g, out, grid, iters, 0, 512, 32768, 3104);
        run<6>("8 rows x 128 B, row stride 512 B, shared", g, out, grid, iters, 0, 128, 32768, 128);
        run<2>("dwordx2, 512 B contiguous per wave-load", g, out, grid, iters, stride, 16384, 16384);
        run<3>("dword, 256 B contiguous per wave-load", g, out, grid, iters, stride, 16384, 8192);
        run<5>("global_load_lds dwordx4, 1 KB contiguous", g, out, grid, iters, stride, 16384, 32768);
    }
    return 0;
}
