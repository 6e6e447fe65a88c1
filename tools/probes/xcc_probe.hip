#include <hip/hip_runtime.h>
__global__ void k(unsigned* out) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 4); k<<<64, 64>>>(d); unsigned h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) printf("%u%s", h[i] & 0xf, i % 8 == 7 ? "\n" : " ");
    return 0;
}
