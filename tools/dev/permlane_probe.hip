// dev probe (GPU box): v_permlane32_swap_b32 -- the builtin's two results, and the inline-asm form used on 16 live registers the way the
// pooled epilogue of sepconv_ws_kernel would use it (8 swaps v[k] <-> v[8 + k], then per-column maxima), against the shuffle form
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
    const unsigned l = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(100u + l, 200u + l, false, false);
    o[l] = r[0]; o[64 + l] = r[1];
}
__global__ void k2(const float *in, float *out_swap, float *out_ref)
{
    const int l = threadIdx.x, h = l >> 5;
    float v[16], w[16];
    for (int e = 0; e < 16; e++) { v[e] = in[e * 64 + l] * 1.5f + 0.25f; w[e] = v[e]; }
    // reference: what store_tile_pool computes with __shfl_xor
    float m_ref[8], y0_ref[8];
    for (int kk = 0; kk < 8; kk++) {
        const float own = h ? w[8 + kk] : w[kk];
        const float oth = __shfl_xor(h ? w[kk] : w[8 + kk], 32);
        m_ref[kk] = fmaxf(own, oth);
        y0_ref[kk] = kk < 4 ? own : oth;
    }
    // swap form: after v_permlane32_swap_b32 A, B (A = v[k], B = v[8 + k]): A.hi <-> B.lo, so a lane of half 0 holds (own v[k], the other
    // half's v[k]) and a lane of half 1 (the other half's v[8 + k], own v[8 + k]) in (A, B)
    float m_sw[8], y0_sw[8];
    for (int kk = 0; kk < 8; kk++) {
        float A = v[kk], B = v[8 + kk];
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(A), "+v"(B));
        m_sw[kk] = fmaxf(A, B);
        y0_sw[kk] = ((h != 0) != (kk >= 4)) ? B : A;
    }
    for (int kk = 0; kk < 8; kk++) {
        out_ref[kk * 64 + l] = m_ref[kk]; out_ref[(8 + kk) * 64 + l] = y0_ref[kk];
        out_swap[kk * 64 + l] = m_sw[kk]; out_swap[(8 + kk) * 64 + l] = y0_sw[kk];
    }
}
int main() {
    unsigned *d, h[128];
    (void)hipMalloc((void **)&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("builtin r0: lane0 %u lane1 %u lane32 %u lane33 %u\n", h[0], h[1], h[32], h[33]);
    printf("builtin r1: lane0 %u lane1 %u lane32 %u lane33 %u\n", h[64], h[65], h[96], h[97]);
    float hin[1024], hs[1024], hr[1024], *din, *ds, *dr;
    for (int i = 0; i < 1024; i++) hin[i] = (float)((i * 2654435761u) % 1000) * 0.01f - 5.f;
    (void)hipMalloc((void **)&din, 4096); (void)hipMalloc((void **)&ds, 4096); (void)hipMalloc((void **)&dr, 4096);
    (void)hipMemcpy(din, hin, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, din, ds, dr);
    (void)hipMemcpy(hs, ds, 4096, hipMemcpyDeviceToHost); (void)hipMemcpy(hr, dr, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; i++) bad += hs[i] != hr[i];
    printf("inline-asm swap form vs shuffle form: %d of 1024 values differ\n", bad);
    return 0;
}
