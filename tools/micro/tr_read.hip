// What ds_read_b64_tr_b16 returns (gfx950): LDS holds u16 = its own element index; every lane passes a byte address and gets
// four u16 back.  build: hipcc --offload-arch=gfx950 -O3 -o tr_read tr_read.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned short* out, int mode, int stride_bytes) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = lane * 8;                                        // 64 consecutive 8-byte pieces
    else if (mode == 1) addr = (lane & 15) * 8 + (lane >> 4) * stride_bytes;   // 16 pieces per row-group, groups stride apart
    else addr = (lane & 3) * 8 + ((lane >> 2) & 3) * stride_bytes + (lane >> 4) * 4 * stride_bytes;  // 4 pieces x 4 rows per 16 lanes
    addr += (unsigned)(size_t)lds;                                         // LDS base (generic -> lds offset is the low bits)
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[lane * 4 + 0] = v.x & 0xffff; out[lane * 4 + 1] = v.x >> 16; out[lane * 4 + 2] = v.y & 0xffff; out[lane * 4 + 3] = v.y >> 16;
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    const int modes[][2] = {{0, 0}, {1, 128}, {1, 512}, {2, 32}, {2, 64}, {2, 80}};
    for (auto& m : modes) {
        k<<<1, 64>>>(d, m[0], m[1]);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d stride %d bytes\n", m[0], m[1]);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d: %5d %5d %5d %5d", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
            if (l % 4 == 3) printf("\n");
        }
    }
    return 0;
}
