// dev probe (GPU box): what v_permlane32_swap_b32 returns in the builtin's two results
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
    const unsigned l = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(100u + l, 200u + l, false, false);
    o[l] = r[0]; o[64 + l] = r[1];
}
int main() {
    unsigned *d, h[128];
    hipMalloc((void **)&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("r0: lane0 %u lane1 %u lane32 %u lane33 %u\n", h[0], h[1], h[32], h[33]);
    printf("r1: lane0 %u lane1 %u lane32 %u lane33 %u\n", h[64], h[65], h[96], h[97]);
    return 0;
}
