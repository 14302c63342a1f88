// Microbenchmark: add the K-loop ingredients of gemm.hip one at a time to a pure MFMA loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// FEAT bits: 1 = LDS operand reads, 2 = LDS stores (transposed b32 x8 + b128 x2), 4 = barrier per K-step,
//            8 = global loads (4 x float4 per thread per K-step), 16 = double-buffer toggle
template <int FEAT>
__global__ __launch_bounds__(256) void kloop(const float* __restrict__ g, float* out, int ksteps, long gstride, long long* clk) {
    const long long c0 = clock64(), w0 = wall_clock64();
    __shared__ __attribute__((aligned(16))) float As[2][16 * 130];
    __shared__ __attribute__((aligned(16))) float Bs[2][16 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 16 * 130; i += 256) (&As[0][0])[i] = (float)(i & 7);
    for (int i = tid; i < 2 * 16 * 128; i += 256) (&Bs[0][0])[i] = (float)(i & 3);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int h = lane >> 5, l = lane & 31, wm = wave >> 1, wn = wave & 1;
    const int c4 = tid & 3, r0 = tid >> 2;
    // A panel shared by 8 consecutive workgroups (the n-tiles of one m-tile), B panel shared by all
    const float* gp = g + (long)(blockIdx.x >> 3) * gstride + (long)r0 * 1536 + c4 * 4;
    const float* gb = g + (long)r0 * 1536 + c4 * 4;
    float4 va[2] = {make_float4(1, 2, 3, 4), make_float4(1, 2, 3, 4)}, vb[2] = {make_float4(1, 2, 3, 4), make_float4(1, 2, 3, 4)};
    float ra0 = 1.f, ra1 = 2.f, rb0 = 3.f, rb1 = 4.f;
    for (int kt = 0; kt < ksteps; ++kt) {
        const int cur = (FEAT & 16) ? (kt & 1) : 0;
        if (FEAT & 8) {
            va[0] = *reinterpret_cast<const float4*>(gp);
            va[1] = *reinterpret_cast<const float4*>(gp + 64 * 1536);
            vb[0] = *reinterpret_cast<const float4*>(gb);
            vb[1] = *reinterpret_cast<const float4*>(gb + 64 * 1536);
            gp += 16;
            gb += 16;
        }
        const float* ap = As[cur] + h * 130 + wm * 64 + l;
        const float* bp = Bs[cur] + h * 128 + wn * 64 + l;
        if (FEAT & 32) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            if (FEAT & 1) { ra0 = ap[kk * 130]; ra1 = ap[kk * 130 + 32]; rb0 = bp[kk * 128]; rb1 = bp[kk * 128 + 32]; }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0, rb0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0, rb1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1, rb0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1, rb1, acc[1][1], 0, 0, 0);
        }
        if (FEAT & 32) __builtin_amdgcn_s_setprio(0);
        if (FEAT & 2) {
            float* da = As[cur ^ ((FEAT & 16) ? 1 : 0)];
            float* db = Bs[cur ^ ((FEAT & 16) ? 1 : 0)];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float* d = da + (c4 * 4) * 130 + r0 + p * 64;
                d[0] = va[p].x; d[130] = va[p].y; d[260] = va[p].z; d[390] = va[p].w;
            }
            *reinterpret_cast<float4*>(db + ((tid >> 5)) * 128 + (tid & 31) * 4) = vb[0];
            *reinterpret_cast<float4*>(db + ((tid >> 5) + 8) * 128 + (tid & 31) * 4) = vb[1];
        }
        if (FEAT & 4) __syncthreads();
    }
    float s = va[0].x + vb[1].y;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (clk && blockIdx.x == 0 && tid == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}


// Full loop with prefetch distance 2: loads for tile t+2 are issued at step t and written to LDS at step t+1,
// so the end-of-step wait is for loads issued a whole step earlier.
__global__ __launch_bounds__(256) void kloop_pf2(const float* __restrict__ g, float* out, int ksteps, long gstride) {
    __shared__ __attribute__((aligned(16))) float As[2][16 * 130];
    __shared__ __attribute__((aligned(16))) float Bs[2][16 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 16 * 130; i += 256) (&As[0][0])[i] = (float)(i & 7);
    for (int i = tid; i < 2 * 16 * 128; i += 256) (&Bs[0][0])[i] = (float)(i & 3);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int h = lane >> 5, l = lane & 31, wm = wave >> 1, wn = wave & 1;
    const int c4 = tid & 3, r0 = tid >> 2;
    const float* gp = g + (long)(blockIdx.x >> 3) * gstride + (long)r0 * 1536 + c4 * 4;
    const float* gb = g + (long)r0 * 1536 + c4 * 4;
    float4 va0[2], vb0[2], va1[2], vb1[2];
#define LOADSET(VA, VB) { VA[0] = *reinterpret_cast<const float4*>(gp); VA[1] = *reinterpret_cast<const float4*>(gp + 64 * 1536); \
                          VB[0] = *reinterpret_cast<const float4*>(gb); VB[1] = *reinterpret_cast<const float4*>(gb + 64 * 1536); gp += 16; gb += 16; }
#define STORESET(VA, VB, BUF) { float* da = As[BUF]; float* db = Bs[BUF]; \
        _Pragma("unroll") for (int p = 0; p < 2; ++p) { float* d = da + (c4 * 4) * 130 + r0 + p * 64; d[0] = VA[p].x; d[130] = VA[p].y; d[260] = VA[p].z; d[390] = VA[p].w; } \
        *reinterpret_cast<float4*>(db + ((tid >> 5)) * 128 + (tid & 31) * 4) = VB[0]; \
        *reinterpret_cast<float4*>(db + ((tid >> 5) + 8) * 128 + (tid & 31) * 4) = VB[1]; }
#define MMA(BUF) { const float* ap = As[BUF] + h * 130 + wm * 64 + l; const float* bp = Bs[BUF] + h * 128 + wn * 64 + l; \
        _Pragma("unroll") for (int kk = 0; kk < 16; kk += 2) { \
            const float ra0 = ap[kk * 130], ra1 = ap[kk * 130 + 32], rb0 = bp[kk * 128], rb1 = bp[kk * 128 + 32]; \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0, rb0, acc[0][0], 0, 0, 0); \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0, rb1, acc[0][1], 0, 0, 0); \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1, rb0, acc[1][0], 0, 0, 0); \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1, rb1, acc[1][1], 0, 0, 0); } }
    LOADSET(va1, vb1)                       // tile 1 (tile 0 is already "in LDS")
    for (int kt = 0; kt < ksteps; kt += 2) {
        LOADSET(va0, vb0)                   // tile kt+2
        MMA(0)
        STORESET(va1, vb1, 1)               // tile kt+1, loaded one step ago
        __syncthreads();
        LOADSET(va1, vb1)                   // tile kt+3
        MMA(1)
        STORESET(va0, vb0, 0)               // tile kt+2
        __syncthreads();
    }
    float s = va0[0].x + vb1[1].y;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

// Full loop, distance-1 prefetch, but the LDS stores of the next tile are issued in the MIDDLE of this
// tile's MFMAs (after kk = SPLIT) so they execute in the shadow of the matrix pipe.
template <int SPLIT>
__global__ __launch_bounds__(256) void kloop_mid(const float* __restrict__ g, float* out, int ksteps, long gstride) {
    __shared__ __attribute__((aligned(16))) float As[2][16 * 130];
    __shared__ __attribute__((aligned(16))) float Bs[2][16 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 16 * 130; i += 256) (&As[0][0])[i] = (float)(i & 7);
    for (int i = tid; i < 2 * 16 * 128; i += 256) (&Bs[0][0])[i] = (float)(i & 3);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int h = lane >> 5, l = lane & 31, wm = wave >> 1, wn = wave & 1;
    const int c4 = tid & 3, r0 = tid >> 2;
    const float* gp = g + (long)(blockIdx.x >> 3) * gstride + (long)r0 * 1536 + c4 * 4;
    const float* gb = g + (long)r0 * 1536 + c4 * 4;
    float4 va[2], vb[2];
    for (int kt = 0; kt < ksteps; ++kt) {
        const int cur = kt & 1;
        va[0] = *reinterpret_cast<const float4*>(gp); va[1] = *reinterpret_cast<const float4*>(gp + 64 * 1536);
        vb[0] = *reinterpret_cast<const float4*>(gb); vb[1] = *reinterpret_cast<const float4*>(gb + 64 * 1536);
        gp += 16; gb += 16;
        const float* ap = As[cur] + h * 130 + wm * 64 + l;
        const float* bp = Bs[cur] + h * 128 + wn * 64 + l;
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const float ra0 = ap[kk * 130], ra1 = ap[kk * 130 + 32], rb0 = bp[kk * 128], rb1 = bp[kk * 128 + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0, rb0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0, rb1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1, rb0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1, rb1, acc[1][1], 0, 0, 0);
            if (kk == SPLIT) {
                __builtin_amdgcn_sched_barrier(0);
                float* da = As[cur ^ 1]; float* db = Bs[cur ^ 1];
#pragma unroll
                for (int p = 0; p < 2; ++p) { float* d = da + (c4 * 4) * 130 + r0 + p * 64; d[0] = va[p].x; d[130] = va[p].y; d[260] = va[p].z; d[390] = va[p].w; }
                *reinterpret_cast<float4*>(db + ((tid >> 5)) * 128 + (tid & 31) * 4) = vb[0];
                *reinterpret_cast<float4*>(db + ((tid >> 5) + 8) * 128 + (tid & 31) * 4) = vb[1];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    float s = va[0].x + vb[1].y;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

// wgrad-like loop: BOTH operands are K-outer tiles [16][128] copied without transposition.
// DMA = false: global -> VGPR -> ds_write_b128 (what gemm_tn_kernel does)
// DMA = true : global_load_lds (LDS-DMA, 16 B per lane, no VGPR round trip, no ds_write)
template <bool DMA>
__global__ __launch_bounds__(256) void kloop_tn(const float* __restrict__ g, float* out, int ksteps, long gstride) {
    __shared__ __attribute__((aligned(16))) float As[2][16 * 128];
    __shared__ __attribute__((aligned(16))) float Bs[2][16 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 16 * 128; i += 256) { (&As[0][0])[i] = (float)(i & 7); (&Bs[0][0])[i] = (float)(i & 3); }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int h = lane >> 5, l = lane & 31, wm = wave >> 1, wn = wave & 1;
    // thread -> (row kk0 (+8), float4 column c4) of the 16 x 128 tile; rows of the source are 1536 floats apart
    const int c4 = tid & 31, kk0 = tid >> 5;
    const float* ga = g + (long)(blockIdx.x >> 3) * gstride + (long)kk0 * 1536 + c4 * 4;
    const float* gb = g + (long)kk0 * 1536 + 512 + c4 * 4;
    float4 va[2], vb[2];
    for (int kt = 0; kt < ksteps; ++kt) {
        const int cur = kt & 1, nxt = cur ^ 1;
        if (DMA) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                // wave w, pass p covers tile rows 2w + 8p .. +1  (LDS destination = wave-uniform base + lane*16)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + (long)p * 8 * 1536),
                                                 (__attribute__((address_space(3))) void*)(&As[nxt][(2 * wave + 8 * p) * 128]), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + (long)p * 8 * 1536),
                                                 (__attribute__((address_space(3))) void*)(&Bs[nxt][(2 * wave + 8 * p) * 128]), 16, 0, 0);
            }
        } else {
            va[0] = *reinterpret_cast<const float4*>(ga); va[1] = *reinterpret_cast<const float4*>(ga + 8 * 1536);
            vb[0] = *reinterpret_cast<const float4*>(gb); vb[1] = *reinterpret_cast<const float4*>(gb + 8 * 1536);
        }
        ga += 16 * 1536; gb += 16 * 1536;
        const float* ap = As[cur] + h * 128 + wm * 64 + l;
        const float* bp = Bs[cur] + h * 128 + wn * 64 + l;
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const float ra0 = ap[kk * 128], ra1 = ap[kk * 128 + 32], rb0 = bp[kk * 128], rb1 = bp[kk * 128 + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0, rb0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0, rb1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1, rb0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1, rb1, acc[1][1], 0, 0, 0);
        }
        if (!DMA) {
            *reinterpret_cast<float4*>(&As[nxt][kk0 * 128 + c4 * 4]) = va[0];
            *reinterpret_cast<float4*>(&As[nxt][(kk0 + 8) * 128 + c4 * 4]) = va[1];
            *reinterpret_cast<float4*>(&Bs[nxt][kk0 * 128 + c4 * 4]) = vb[0];
            *reinterpret_cast<float4*>(&Bs[nxt][(kk0 + 8) * 128 + c4 * 4]) = vb[1];
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

// NN-like loop where the B (weights, [K][N], L2-resident) operand never touches LDS: each lane loads its
// MFMA B registers directly (lanes 0-31 / 32-63 read two 128-byte rows: coalesced dwords), one K-step ahead.
// A (activations, K-inner) still goes global -> VGPR -> transposed LDS tile.
__global__ __launch_bounds__(256) void kloop_bdirect(const float* __restrict__ g, float* out, int ksteps, long gstride) {
    __shared__ __attribute__((aligned(16))) float As[2][16 * 130];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 16 * 130; i += 256) (&As[0][0])[i] = (float)(i & 7);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int h = lane >> 5, l = lane & 31, wm = wave >> 1, wn = wave & 1;
    const int c4 = tid & 3, r0 = tid >> 2;
    const float* gp = g + (long)(blockIdx.x >> 3) * gstride + (long)r0 * 1536 + c4 * 4;
    // B: row k has 512 floats (ldb = 512); this lane's column n = (blockIdx&7)*64?  use wn*64 + l (+32)
    const float* gb = g + 4096 + (long)h * 512 + (blockIdx.x & 7) * 64 + wn * 32 + l;   // 64-wide n tile: NJ = 1 per wave half... keep 2 regs
    float bcur[16], bnxt[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bcur[i] = gb[(long)(i >> 1) * 2 * 512 + (i & 1) * 32];
    gb += 16 * 512;
    float4 va[2];
    for (int kt = 0; kt < ksteps; ++kt) {
        const int cur = kt & 1;
        va[0] = *reinterpret_cast<const float4*>(gp); va[1] = *reinterpret_cast<const float4*>(gp + 64 * 1536);
        gp += 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) bnxt[i] = gb[(long)(i >> 1) * 2 * 512 + (i & 1) * 32];
        gb += 16 * 512;
        const float* ap = As[cur] + h * 130 + wm * 64 + l;
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const float ra0 = ap[kk * 130], ra1 = ap[kk * 130 + 32];
            const float rb0 = bcur[kk], rb1 = bcur[kk + 1];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0, rb0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0, rb1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1, rb0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1, rb1, acc[1][1], 0, 0, 0);
        }
        float* da = As[cur ^ 1];
#pragma unroll
        for (int p = 0; p < 2; ++p) { float* d = da + (c4 * 4) * 130 + r0 + p * 64; d[0] = va[p].x; d[130] = va[p].y; d[260] = va[p].z; d[390] = va[p].w; }
#pragma unroll
        for (int i = 0; i < 16; ++i) bcur[i] = bnxt[i];
        __syncthreads();
    }
    float s = va[0].x;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int FEAT>
static void run(const char* name, const float* g, float* out, int grid, int ksteps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const long gstride = 128L * 1536;   // (kloop_tn walks 1536 rows of 1536 floats: stays inside the 1.5 GB buffer)
    static long long* clk = nullptr;
    if (!clk) hipMalloc(&clk, 16);
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    hipLaunchKernelGGL((kloop<FEAT>), dim3(grid), dim3(256), 0, 0, g, out, ksteps, gstride, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((kloop<FEAT>), dim3(grid), dim3(256), 0, 0, g, out, ksteps, gstride, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double fl = (double)grid * 4 * ksteps * 8 * 4 * 4096.0;
    // clock64 = shader clock, wall_clock64 = constant-rate counter: their ratio is the sustained shader clock
    printf("  %-52s %8.1f us  %6.1f TF/s   shader clock %.0f MHz (wg0: %lld cyc)\n", name, ms * 200.0, fl * 5 / (ms * 1e-3) / 1e12,
           h[1] ? (double)h[0] / ((double)h[1] / (wall_khz * 1e3)) / 1e6 : 0.0, h[0]);
}

int main() {
    float *g, *out;
    const int maxgrid = 1024;
    hipMalloc(&g, (size_t)maxgrid * 256 * 1536 * 4);
    hipMemset(g, 0, (size_t)maxgrid * 256 * 1536 * 4);
    hipMalloc(&out, maxgrid * 256 * 4);
    const int ksteps = 960;
    for (int w = 1; w <= 4; ++w) {
        const int grid = 256 * w;
        printf("%d WG/CU (grid %d), %d K-steps\n", w, grid, ksteps);
        run<0>("MFMA only", g, out, grid, ksteps);
        run<1>("+ LDS operand reads", g, out, grid, ksteps);
        run<1 | 2>("+ LDS stores", g, out, grid, ksteps);
        run<1 | 2 | 4>("+ barrier", g, out, grid, ksteps);
        run<1 | 2 | 4 | 16>("+ double-buffer toggle", g, out, grid, ksteps);
        run<1 | 2 | 4 | 8 | 16>("+ global loads (full loop)", g, out, grid, ksteps);
        {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(kloop_pf2, dim3(grid), dim3(256), 0, 0, g, out, ksteps, 128L * 1536);
            hipDeviceSynchronize(); hipEventRecord(e0);
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kloop_pf2, dim3(grid), dim3(256), 0, 0, g, out, ksteps, 128L * 1536);
            hipEventRecord(e1); hipEventSynchronize(e1); float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            const double fl = (double)grid * 4 * ksteps * 8 * 4 * 4096.0;
            printf("  %-52s %8.1f us  %6.1f TF/s\n", "full loop, prefetch distance 2 (unroll x2)", ms * 200.0, fl * 5 / (ms * 1e-3) / 1e12);
        }
#define RUNK(KERN, NAME) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); \
            hipLaunchKernelGGL(KERN, dim3(grid), dim3(256), 0, 0, g, out, ksteps, 128L * 1536); \
            hipDeviceSynchronize(); hipEventRecord(e0); \
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(KERN, dim3(grid), dim3(256), 0, 0, g, out, ksteps, 128L * 1536); \
            hipEventRecord(e1); hipEventSynchronize(e1); float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); \
            const double fl = (double)grid * 4 * ksteps * 8 * 4 * 4096.0; \
            printf("  %-52s %8.1f us  %6.1f TF/s\n", NAME, ms * 200.0, fl * 5 / (ms * 1e-3) / 1e12); }
        RUNK(kloop_bdirect, "NN loop, B operand direct from L2 (no LDS)")
        RUNK(kloop_tn<false>, "wgrad-like loop, register staging + ds_write_b128")
        RUNK(kloop_tn<true>, "wgrad-like loop, global_load_lds (LDS-DMA)")
        RUNK(kloop_mid<6>, "full loop, LDS stores after kk=6 (mid-MMA)")
        RUNK(kloop_mid<10>, "full loop, LDS stores after kk=10")
        RUNK(kloop_mid<14>, "full loop, LDS stores after kk=14 (end, ref)")
        run<1 | 2 | 4 | 8 | 16 | 32>("full loop + s_setprio 3 in MFMA phase", g, out, grid, ksteps);
        run<32>("MFMA only + s_setprio", g, out, grid, ksteps);
        run<1 | 4 | 8 | 16>("full minus LDS stores", g, out, grid, ksteps);
        run<1 | 2 | 8 | 16>("full minus barrier", g, out, grid, ksteps);
    }
    return 0;
}
