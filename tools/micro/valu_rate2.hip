// VALU issue rate, second take (round 4): the round-2/3 micro (valu_rate.hip) let the compiler pick the
// registers, and its accumulators sat in even registers next to two fixed operands -- if the vector register file is banked
// (register index mod 4) its v_fma_f32 / v_add_f32 were measured WITH operand-bank conflicts.  Here every instruction is
// written with explicit physical registers inside one asm block, so the banks of dst / src0 / src1 / src2 are chosen:
//   "distinct"  src operands in three different banks (index mod 4 all different)
//   "same"      all source operands in one bank
// 64 independent instructions per loop trip (16 destinations x 4), 1 .. 8 waves per SIMD, wall clock -> SIMD cycles per
// wave-instruction at the clock s_memtime / wall gives.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate2 valu_rate2.hip ; run: ./valu_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string>
#include <vector>

#define STR2(x) #x
#define STR(x) STR2(x)
// sixteen destinations d = 16 + 4 i + DB (bank DB), sources by pattern
#define I16(OPC, DB, S0, S1, S2)                                                                                         \
    OPC(16 + DB, S0(16 + DB), S1, S2) OPC(20 + DB, S0(20 + DB), S1, S2) OPC(24 + DB, S0(24 + DB), S1, S2) OPC(28 + DB, S0(28 + DB), S1, S2) \
    OPC(32 + DB, S0(32 + DB), S1, S2) OPC(36 + DB, S0(36 + DB), S1, S2) OPC(40 + DB, S0(40 + DB), S1, S2) OPC(44 + DB, S0(44 + DB), S1, S2) \
    OPC(48 + DB, S0(48 + DB), S1, S2) OPC(52 + DB, S0(52 + DB), S1, S2) OPC(56 + DB, S0(56 + DB), S1, S2) OPC(60 + DB, S0(60 + DB), S1, S2) \
    OPC(64 + DB, S0(64 + DB), S1, S2) OPC(68 + DB, S0(68 + DB), S1, S2) OPC(72 + DB, S0(72 + DB), S1, S2) OPC(76 + DB, S0(76 + DB), S1, S2)
#define SELF(r) r
#define FMA(d, a, b, c) "v_fma_f32 v[" STR(d) "], v[" STR(a) "], v[" STR(b) "], v[" STR(c) "]\n"
#define ADD(d, a, b, c) "v_add_f32 v[" STR(d) "], v[" STR(a) "], v[" STR(b) "]\n"
#define MUL(d, a, b, c) "v_mul_f32 v[" STR(d) "], v[" STR(a) "], v[" STR(b) "]\n"
#define SUB(d, a, b, c) "v_sub_f32 v[" STR(d) "], v[" STR(a) "], v[" STR(b) "]\n"
#define CLOB "v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","s20","scc"

#define KERNEL(NAME, BODY)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, long long* cyc) {                   \
        long long t0 = __builtin_amdgcn_s_memtime();                                                       \
        asm volatile("s_mov_b32 s20, %0\n"                                                                  \
                     "v_mov_b32 v1, 1.0\n v_mov_b32 v2, 0\n v_mov_b32 v3, 1.0\n v_mov_b32 v4, 0\n v_mov_b32 v5, 1.0\n v_mov_b32 v6, 0\n v_mov_b32 v7, 1.0\n v_mov_b32 v8, 0\n" \
                     "1:\n" BODY "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" ::"s"(iters) : CLOB); \
        long long t1 = __builtin_amdgcn_s_memtime();                                                       \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                    \
        if (iters < 0) out[threadIdx.x] = 0.f;                                                             \
    }
// 64 instructions per trip: four groups of sixteen
// fma: dst/src0 bank 0; src1 v1 (bank 1), src2 v2 (bank 2): distinct.   same: src1 v4, src2 v8 (bank 0 like src0)
KERNEL(k_fma_distinct, I16(FMA, 0, SELF, 1, 2) I16(FMA, 0, SELF, 5, 6) I16(FMA, 0, SELF, 1, 6) I16(FMA, 0, SELF, 5, 2))
KERNEL(k_fma_same, I16(FMA, 0, SELF, 4, 8) I16(FMA, 0, SELF, 8, 4) I16(FMA, 0, SELF, 4, 8) I16(FMA, 0, SELF, 8, 4))
KERNEL(k_fma_two_same, I16(FMA, 0, SELF, 4, 2) I16(FMA, 0, SELF, 8, 6) I16(FMA, 0, SELF, 4, 2) I16(FMA, 0, SELF, 8, 6))
KERNEL(k_add_distinct, I16(ADD, 0, SELF, 1, 0) I16(ADD, 0, SELF, 5, 0) I16(ADD, 0, SELF, 3, 0) I16(ADD, 0, SELF, 7, 0))
KERNEL(k_add_same, I16(ADD, 0, SELF, 4, 0) I16(ADD, 0, SELF, 8, 0) I16(ADD, 0, SELF, 4, 0) I16(ADD, 0, SELF, 8, 0))
KERNEL(k_mul_distinct, I16(MUL, 0, SELF, 1, 0) I16(MUL, 0, SELF, 5, 0) I16(MUL, 0, SELF, 3, 0) I16(MUL, 0, SELF, 7, 0))
// a butterfly-like mix the FFT issues: add / sub / fma / mul in equal parts, destinations in all four banks
KERNEL(k_mix_distinct, I16(ADD, 0, SELF, 1, 0) I16(SUB, 1, SELF, 2, 0) I16(FMA, 2, SELF, 3, 1) I16(MUL, 3, SELF, 2, 0))
KERNEL(k_mix_same, I16(ADD, 0, SELF, 4, 0) I16(SUB, 1, SELF, 5, 0) I16(FMA, 2, SELF, 6, 6) I16(MUL, 3, SELF, 7, 0))
// packed: 64-bit operands (register pairs); dst pair v[16+4i : 17+4i], sources v[2:3] / v[6:7]
#define PK16(OPC, A, B)                                                                                                    \
    OPC(16, A, B) OPC(20, A, B) OPC(24, A, B) OPC(28, A, B) OPC(32, A, B) OPC(36, A, B) OPC(40, A, B) OPC(44, A, B)       \
    OPC(48, A, B) OPC(52, A, B) OPC(56, A, B) OPC(60, A, B) OPC(64, A, B) OPC(68, A, B) OPC(72, A, B) OPC(76, A, B)
#define PKFMA(d, a, b) "v_pk_fma_f32 v[" STR(d) ":" STR(d + 1) "], v[" STR(d) ":" STR(d + 1) "], v[" STR(a) ":" STR(a + 1) "], v[" STR(b) ":" STR(b + 1) "]\n"
#define PKADD(d, a, b) "v_pk_add_f32 v[" STR(d) ":" STR(d + 1) "], v[" STR(d) ":" STR(d + 1) "], v[" STR(a) ":" STR(a + 1) "]\n"
#define PKMUL(d, a, b) "v_pk_mul_f32 v[" STR(d) ":" STR(d + 1) "], v[" STR(d) ":" STR(d + 1) "], v[" STR(a) ":" STR(a + 1) "]\n"
KERNEL(k_pkfma, PK16(PKFMA, 2, 6) PK16(PKFMA, 6, 2) PK16(PKFMA, 2, 6) PK16(PKFMA, 6, 2))
KERNEL(k_pkadd, PK16(PKADD, 2, 0) PK16(PKADD, 6, 0) PK16(PKADD, 2, 0) PK16(PKADD, 6, 0))
KERNEL(k_pkmul, PK16(PKMUL, 2, 0) PK16(PKMUL, 6, 0) PK16(PKMUL, 2, 0) PK16(PKMUL, 6, 0))

typedef void (*kern_t)(float*, int, long long*);
static void run(const char* name, kern_t k, double results_per_inst) {
    const int iters = 4000;
    float* out; long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8 * 256 * 4 * 8);
    for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
        const int nwg = 256 * wg_per_cu;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<nwg, 256>>>(out, 10, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<nwg, 256>>>(out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(nwg * 4); hipMemcpy(h.data(), cyc, nwg * 4 * 8, hipMemcpyDeviceToHost);
        double ticks = 0; for (long long v : h) ticks += (double)v; ticks /= h.size();
        const double insts = (double)iters * 64;                        // per wave
        // s_memtime ticks at 100 MHz on this part: report both the tick view and the wall view at an assumed 2.4 GHz
        printf("%-16s waves/SIMD=%d  wall %8.1f us  wave ticks %9.0f  SIMD cycles/inst @2.4GHz %5.2f  (%.0f Gresults/s chip-wide)\n", name, wg_per_cu,
               ms * 1e3, ticks, ms * 1e-3 * 2.4e9 / (insts * wg_per_cu), insts * wg_per_cu * 1024 * 64 * results_per_inst / (ms * 1e-3) / 1e9);
    }
    hipFree(out); hipFree(cyc);
}
int main() {
    run("fma distinct", k_fma_distinct, 1); run("fma two-same", k_fma_two_same, 1); run("fma all-same", k_fma_same, 1);
    run("add distinct", k_add_distinct, 1); run("add same", k_add_same, 1); run("mul distinct", k_mul_distinct, 1);
    run("mix distinct", k_mix_distinct, 1); run("mix same", k_mix_same, 1);
    run("pk_fma", k_pkfma, 2); run("pk_add", k_pkadd, 2); run("pk_mul", k_pkmul, 2);
    return 0;
}
