// dev probe (GPU box): do vector FMAs of one wave issue beside the MFMAs of other waves of the same SIMD?  Wall-clock (hipEvent) of one
// workgroup on one CU, every SIMD loaded the same way, so wave placement does not matter:
//   M  = 8 waves (2 per SIMD) x NM MFMAs each        V  = 4 waves (1 per SIMD) x NV v_pk_fma_f32 each        MV = both in one workgroup
// MV ~ max(M, V): side by side.  MV ~ M + V: one pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int BF> __device__ float mfma_loop(int n, float x)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    bf16x8 pa, pb;
    for (int e = 0; e < 8; e++) { pa[e] = (__bf16)(x + e); pb[e] = (__bf16)(x - e); }
    for (int r = 0; r < n; r++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x + 1.f, acc[i], 0, 0, 0);
        }
    }
    float s = 0; for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
    return s;
}
// OP: 0 v_pk_fma_f32, 1 v_fma_f32, 2 v_add_u32, 3 v_max_f32, 4 v_cndmask_b32, 5 v_pk_add_f32, 6 v_perm_b32
template <int OP> __device__ float valu_loop(int n, float x)
{
    f32x2 a[8];
    for (int i = 0; i < 8; i++) a[i] = f32x2{x + i, x - i};
    const f32x2 b = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
    constexpr int op = OP;
    for (int r = 0; r < n; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (op == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            else if (op == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
            else if (op == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
            else if (op == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
            else if (op == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(b.x));
            else if (op == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            else asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
        }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    return s;
}
// waves 0 .. nmw-1 run nm x 4 MFMAs, waves 8 .. 11 run nv x 8 packed FMAs (priority prio); launched with 768 threads
template <int BF, int OP> __global__ void k(float *out, int nmw, int nm, int nv, int prio)
{
    const int wave = threadIdx.x >> 6;
    const float x = threadIdx.x * 1e-3f;
    float s = 0.f;
    if (wave >= 8 && prio) __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    if (wave < nmw) s = mfma_loop<BF>(nm, x);
    else if (wave >= 8) s = valu_loop<OP>(nv, x);
    out[threadIdx.x] = s;
}
template <int BF, int OP> float run(float *o, int nmw, int nm, int nv, int prio)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<BF, OP>), dim3(1), dim3(768), 0, 0, o, nmw, nm, nv, prio);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}
template <int BF, int OP> void report(float *o, const char *name, float m2)
{
    const int NM = 20000, NV = BF ? 36000 : 72000;
    const float v = run<BF, OP>(o, 0, 0, NV, 0), mv0 = run<BF, OP>(o, 8, NM, NV, 0), mv3 = run<BF, OP>(o, 8, NM, NV, 1), m1v = run<BF, OP>(o, 4, NM, NV, 0);
    printf("   + one wave/SIMD of %-14s alone %.3f ms | beside 2 MFMA waves/SIMD %.3f ms (s_setprio 3: %.3f) | beside 1 MFMA wave/SIMD %.3f ms | hidden share of the vector time: %.2f\n",
           name, v, mv0, mv3, m1v, (m2 + v - mv0) / v);
}
template <int BF> void all(float *o)
{
    const int NM = 20000;
    const float m1 = run<BF, 0>(o, 4, NM, 0, 0), m2 = run<BF, 0>(o, 8, NM, 0, 0);
    printf("%s MFMA alone: 1 wave/SIMD %.3f ms, 2 waves/SIMD %.3f ms\n", BF ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32  ", m1, m2);
    report<BF, 0>(o, "v_pk_fma_f32", m2); report<BF, 1>(o, "v_fma_f32", m2); report<BF, 2>(o, "v_add_u32", m2); report<BF, 3>(o, "v_max_f32", m2);
    report<BF, 4>(o, "v_cndmask_b32", m2); report<BF, 5>(o, "v_pk_add_f32", m2); report<BF, 6>(o, "v_perm_b32", m2);
}
int main()
{
    float *o; (void)hipMalloc((void **)&o, 4096 * 4);
    all<0>(o);
    all<1>(o);
    return 0;
}
