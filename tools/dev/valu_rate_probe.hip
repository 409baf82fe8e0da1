// dev probe (GPU box): issue cost of vector instructions on gfx950, alone on a SIMD and beside f32 / bf16 MFMAs of another wave.
//   one wave:      cycles per v_pk_fma_f32 / v_fma_f32 / v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16 (independent chains)
//   two waves on one SIMD (waves 0 and 4 of a 320-thread block): wave 0 issues MFMAs, wave 4 v_pk_fma_f32: time of both
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP 64
__device__ __forceinline__ long long now() { return (long long)__builtin_readcyclecounter(); }

template <int MODE> __device__ void body(float *out, long long *cyc, int slot)
{
    const float x = (float)threadIdx.x * 1e-3f;
    if (MODE == 0 || MODE == 1) {           // 8 independent chains of packed / scalar FMAs
        f32x2 a[8];
        for (int i = 0; i < 8; i++) a[i] = f32x2{x + i, x - i};
        const f32x2 b = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
        const long long t0 = now();
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                else { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(b.y), "v"(c.y)); }
            }
        }
        const long long t1 = now();
        float s = 0; for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
        out[threadIdx.x] = s;
        if ((threadIdx.x & 63) == 0) cyc[slot] = t1 - t0;
    } else {                                // 4 independent accumulators of f32 (MODE 2) / bf16 (MODE 3) MFMAs
        f32x16 acc[4];
        for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
        bf16x8 pa, pb;
        for (int e = 0; e < 8; e++) { pa[e] = (__bf16)(x + e); pb[e] = (__bf16)(x - e); }
        const long long t0 = now();
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x + 1.f, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc[i], 0, 0, 0);
            }
        }
        const long long t1 = now();
        float s = 0; for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
        out[threadIdx.x] = s;
        if ((threadIdx.x & 63) == 0) cyc[slot] = t1 - t0;
    }
}

template <int MODE> __global__ void single(float *out, long long *cyc) { body<MODE>(out, cyc, 0); }

// waves 0 and 4 share SIMD 0 (waves are dealt round-robin to the four SIMDs): wave 0 runs MFMAs (MM = 2 f32, 3 bf16), wave 4 packed FMAs
template <int MM> __global__ void pair(float *out, long long *cyc)
{
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if (wave == 0) body<MM>(out, cyc, 0);
    else if (wave == 4) body<0>(out, cyc, 1);
}

// Roles by the SIMD a wave really runs on (HW_ID.SIMD_ID): on SIMD 0 the first NM waves run MFMAs (MMODE 2 f32, 3 bf16), the next one
// packed FMAs with priority PRIO; the other SIMDs idle.  simd_of[] reports the placement.
template <int PRIO, int MMODE, int NM> __global__ void triple(float *out, long long *cyc, int *simd_of)
{
    __shared__ int sid[16];
    const int wave = threadIdx.x >> 6;
    const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);      // HW_REG_HW_ID, bits 5:4
    if ((threadIdx.x & 63) == 0) { sid[wave] = simd; simd_of[wave] = simd; }
    __syncthreads();
    int rank = 0;
    for (int w = 0; w < wave; w++) rank += sid[w] == simd;
    if (simd != 0) return;
    if (rank == NM && PRIO) __builtin_amdgcn_s_setprio(PRIO);
    if (rank < NM) body<MMODE>(out + 64 * rank, cyc, rank);
    else if (rank == NM) body<0>(out + 128, cyc, 2);
}

int main()
{
    float *o; long long *c, h[2];
    hipMalloc((void **)&o, 4096 * 4); hipMalloc((void **)&c, 64);
    const char *names[4] = {"v_pk_fma_f32 (8 chains)", "v_fma_f32 x2 (8 chains)", "v_mfma_f32_32x32x2_f32 (4 accumulators)", "v_mfma_f32_32x32x16_bf16 (4 accumulators)"};
    for (int m = 0; m < 4; m++) {
        for (int rep = 0; rep < 2; rep++) {
            if (m == 0) hipLaunchKernelGGL(single<0>, dim3(1), dim3(64), 0, 0, o, c);
            if (m == 1) hipLaunchKernelGGL(single<1>, dim3(1), dim3(64), 0, 0, o, c);
            if (m == 2) hipLaunchKernelGGL(single<2>, dim3(1), dim3(64), 0, 0, o, c);
            if (m == 3) hipLaunchKernelGGL(single<3>, dim3(1), dim3(64), 0, 0, o, c);
            hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
        }
        const int n = m < 2 ? REP * 8 * (m == 1 ? 2 : 1) : REP * 4;
        printf("alone: %-45s %7.1f cycles per instruction (%d instructions)\n", names[m], (double)h[0] / n, n);
    }
    for (int mm = 2; mm < 4; mm++) {
        for (int rep = 0; rep < 2; rep++) {
            if (mm == 2) hipLaunchKernelGGL(pair<2>, dim3(1), dim3(320), 0, 0, o, c);
            else hipLaunchKernelGGL(pair<3>, dim3(1), dim3(320), 0, 0, o, c);
            hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
        }
        printf("same SIMD: %-40s %7.1f cycles per MFMA | v_pk_fma_f32 beside it %7.1f cycles per instruction\n", names[mm], (double)h[0] / (REP * 4), (double)h[1] / (REP * 8));
    }
    long long h3[3];
    int *sd, hs[12];
    hipMalloc((void **)&sd, 64);
    for (int v = 0; v < 8; v++) {
        for (int rep = 0; rep < 2; rep++) {
            hipMemset(c, 0, 24);
            if (v == 0) hipLaunchKernelGGL((triple<0, 2, 2>), dim3(1), dim3(768), 0, 0, o, c, sd);
            if (v == 1) hipLaunchKernelGGL((triple<3, 2, 2>), dim3(1), dim3(768), 0, 0, o, c, sd);
            if (v == 2) hipLaunchKernelGGL((triple<0, 3, 2>), dim3(1), dim3(768), 0, 0, o, c, sd);
            if (v == 3) hipLaunchKernelGGL((triple<3, 3, 2>), dim3(1), dim3(768), 0, 0, o, c, sd);
            if (v == 4) hipLaunchKernelGGL((triple<0, 2, 1>), dim3(1), dim3(768), 0, 0, o, c, sd);
            if (v == 5) hipLaunchKernelGGL((triple<3, 2, 1>), dim3(1), dim3(768), 0, 0, o, c, sd);
            if (v == 6) hipLaunchKernelGGL((triple<0, 3, 1>), dim3(1), dim3(768), 0, 0, o, c, sd);
            if (v == 7) hipLaunchKernelGGL((triple<3, 3, 1>), dim3(1), dim3(768), 0, 0, o, c, sd);
            hipMemcpy(h3, c, 24, hipMemcpyDeviceToHost);
            hipMemcpy(hs, sd, 48, hipMemcpyDeviceToHost);
        }
        const int nm = v < 4 ? 2 : 1, bf = (v >> 1) & 1;
        printf("SIMD 0: %d %s MFMA wave(s) + one v_pk_fma_f32 wave (s_setprio %d): %6.1f / %6.1f cycles per MFMA, %6.1f cycles per v_pk_fma_f32   [SIMD of waves 0-11:",
               nm, bf ? "bf16" : "f32 ", (v & 1) ? 3 : 0, (double)h3[0] / (REP * 4), (double)h3[1] / (REP * 4), (double)h3[2] / (REP * 8));
        for (int w = 0; w < 12; w++) printf(" %d", hs[w]);
        printf("]\n");
    }
    return 0;
}
