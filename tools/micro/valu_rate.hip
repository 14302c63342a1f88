// VALU issue-rate microbenchmark (gfx950): cycles per wave-instruction per SIMD for plain and packed f32 ops, at 1-4
// waves per SIMD.  build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
    v2f a[8], b, c;
    for (int i = 0; i < 8; ++i) a[i] = (v2f){(float)threadIdx.x * 1e-3f + i, 1.0f + i};
    unsigned long long mask = 0x5555aaaa3333ccccull ^ (unsigned long long)iters; asm volatile("" : "+s"(mask));
    b = (v2f){1.0001f, 0.9999f}; c = (v2f){1e-3f, -1e-3f};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(c));
                if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
                if (OP == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 7) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(c.y)); }
                if (OP == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(b.x));
                if (OP == 11) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "s"(mask));
                if (OP == 12) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (OP == 13) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].x) : "v"(b.x));
                if (OP == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i].x) : "v"(a[(i + 1) & 7].y));
                if (OP == 15) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 16) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i].x) : "v"(c.x));
                if (OP == 17) asm volatile("v_log_f32 %0, %0" : "+v"(a[i].x));
                if (OP == 18) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i].x) : "s"((unsigned)mask), "v"(c.x));
                if (OP == 19) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i].x) : "s"((unsigned)mask));
                if (OP == 20) asm volatile("v_add_f32 %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i].x) : "v"(c.x));
                if (OP == 21) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(a[(i+1)&7].y), "v"(c.x));
                if (OP == 22) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[i].x) : "v"(a[(i+3)&7].x), "v"(a[(i+5)&7].y));
                if (OP == 10) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "s"(c));
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
static int g_short = 0;
template <int OP> void run(const char* name, int mult) {
    float* out; long long* cyc;
    const int iters = 2000;
    // "short" mode (the PMC reconciliation run, profiles/r03_valu_issue_counters.txt): 1, 2 and 4 workgroups per CU
    for (int wg_per_cu = g_short ? 1 : 2; wg_per_cu <= 4; wg_per_cu *= 2) {
        int nwg = 256 * wg_per_cu;
        hipMalloc(&out, nwg * 256 * 4); hipMalloc(&cyc, nwg * 4 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<nwg, 256>>>(out, 10, cyc);
        hipEventRecord(e0);
        k<OP><<<nwg, 256>>>(out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
        double insts = (double)iters * 32 * mult;              // per wave
        printf("%-34s waves/SIMD=%d  wall %.1f us  memtime-ticks/inst/wave %.2f  -> SIMD clk/inst @2.4GHz (wall): %.2f\n", name, wg_per_cu, ms * 1e3,
               (double)h[0] / insts, ms * 1e-3 * 2.4e9 / (insts * wg_per_cu));
        hipFree(out); hipFree(cyc);
    }
}
int main(int argc, char** argv) {
    if (argc > 1) {          // short: the three ops the feature kernel's ceiling rests on
        g_short = 1;
        run<0>("v_fma_f32", 1); run<5>("v_add_f32", 1); run<1>("v_pk_fma_f32", 1);
        return 0;
    }
    run<0>("v_fma_f32", 1); run<5>("v_add_f32", 1); run<8>("v_mul_f32", 1); run<9>("v_cndmask_b32", 1);
    run<1>("v_pk_fma_f32", 1); run<2>("v_pk_add_f32", 1); run<3>("v_pk_add_f32 op_sel neg", 1); run<4>("v_pk_mul_f32", 1);
    run<6>("v_pk_fma_f32 op_sel neg", 1); run<7>("2 x v_add_f32 (pair)", 2); run<10>("v_pk_add_f32 sgpr src", 1);
    run<11>("v_cndmask_b32_e64 sgpr mask", 1); run<12>("v_fmac_f32_e32", 1); run<13>("v_fma_f32 a,a,b,b", 1); run<14>("v_mov_b32", 1);
    run<15>("v_lshl_add_u64", 1); run<16>("v_lshl_add_u32", 1); run<17>("v_log_f32", 1); run<18>("v_fmac_f32 sgpr", 1); run<19>("v_mul_f32 sgpr", 1);
    run<20>("v_add_f32 dpp row_shr", 1); run<21>("v_fma_f32 d=x*c+d (3 regs, mixed)", 1); run<22>("v_sub_f32 mixed regs", 1);
    return 0;
}
