// Which XCD does shader engine s of the CU mask belong to?  (socioreasoner_amd/streams.py: bit i of a hipExtStreamCreateWithCUMask mask = CU i / 32 of shader
// engine i % 32.)  One stream per shader engine (all 8 CU words, one bit each), a kernel whose blocks record HW_REG_XCC_ID.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k(unsigned* hist) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    if (threadIdx.x == 0) atomicAdd(&hist[v & 15], 1u);
    for (volatile int i = 0; i < 2000; ++i) {}
}
int main() {
    unsigned* d; CK(hipMalloc(&d, 64));
    printf("{\"se_to_xcd\": [");
    for (int s = 0; s < 32; ++s) {
        unsigned mask[8];
        for (int w = 0; w < 8; ++w) mask[w] = (getenv("ONE_WORD") ? (w == atoi(getenv("ONE_WORD")) ? 1u << s : 0u) : 1u << s);
        hipStream_t st; CK(hipExtStreamCreateWithCUMask(&st, 8, mask));
        CK(hipMemsetAsync(d, 0, 64, st));
        hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, st, d);
        CK(hipStreamSynchronize(st));
        unsigned h[16]; CK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
        int best = 0, n = 0;
        for (int i = 0; i < 16; ++i) { if (h[i] > h[best]) best = i; n += h[i] > 0; }
        printf("%s[", s ? ", " : "");
        for (int i = 0; i < 8; ++i) printf("%s%u", i ? "," : "", h[i]);
        printf("]");
        (void)best; (void)n;
        CK(hipStreamDestroy(st));
    }
    printf("]}\n");
    return 0;
}
