#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    unsigned x = threadIdx.x, a = x, b = x;
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    o[threadIdx.x] = a; o[64 + threadIdx.x] = b;
    unsigned c = x, d = x;
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
    o[128 + threadIdx.x] = c; o[192 + threadIdx.x] = d;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[4] = {"p16 a", "p16 b", "p32 a", "p32 b"};
    for (int r = 0; r < 4; r++) { printf("%s:", nm[r]); for (int i = 0; i < 64; i += 8) printf(" %u", h[r * 64 + i]); printf("\n"); }
    return 0;
}
