// Microbenchmark: does a second wave on the SIMD hide a BLOCK of VALU work between MFMA phases?
// Every wave runs the c3 forward's per-row MFMA mix (54 x 16x16x4 + 108 x 4x4x1) and then N dependent-free v_fma_f32 in one
// block (a stand-in for a row's transforms), with 1 or 2 waves per SIMD; second form: the same N VALU spread between the MFMAs
// (two per 16x16x4).  Reported: ns per SIMD-step and the MFMA rate.  If two tenants hid each other's VALU blocks, the 2-wave
// rows would stay at the MFMA-only time.
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o mfma_overlap.bin mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int N, bool SPREAD>
__global__ __launch_bounds__(256) void k(const float *ab, float *out, int steps)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    float af[54], as[27], b[18];
#pragma unroll
    for (int i = 0; i < 54; ++i) { af[i] = ab[(i % 24) * 64 + lane]; asm volatile("" : "+a"(af[i])); }
#pragma unroll
    for (int i = 0; i < 27; ++i) { as[i] = ab[((i + 7) % 24) * 64 + lane]; asm volatile("" : "+a"(as[i])); }
#pragma unroll
    for (int i = 0; i < 18; ++i) { b[i] = ab[(24 + i % 8) * 64 + lane]; asm volatile("" : "+v"(b[i])); }
    v4f c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, tot = {0, 0, 0, 0};
    float e[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = b[i];
    constexpr int PER = SPREAD ? (N + 53) / 54 : 0;
    for (int s = 0; s < steps; ++s) {
        c0 = tot; c1 = tot; c2 = tot;
#pragma unroll
        for (int kk = 0; kk < 54; ++kk) {
            const float bb = b[kk % 18];
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk], bb, c0, 0, 0, 0);
            const int f1 = 2 * kk, f2 = 2 * kk + 1;
            if ((f1 & 3) == 0) { c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f1 >> 2], bb, c1, 2, 0, 0); c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f2 >> 2], bb, c2, 2, 1, 0); }
            else { c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f1 >> 2], bb, c1, 2, 2, 0); c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f2 >> 2], bb, c2, 2, 3, 0); }
            if constexpr (SPREAD) {
#pragma unroll
                for (int i = 0; i < PER; ++i)
                    if (kk * PER + i < N) { float &x = e[(kk * PER + i) & 15]; asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(x) : "v"(e[(kk * PER + i + 5) & 15])); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!SPREAD) {
#pragma unroll
            for (int i = 0; i < N; ++i) { float &x = e[i & 15]; asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(x) : "v"(e[(i + 5) & 15])); }
        }
        __builtin_amdgcn_sched_barrier(0);
        tot = c0 + c1 + c2;
        tot *= 1e-3f;
    }
    float r = tot.x + tot.y + tot.z + tot.w;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += e[i];
    out[wave * 64 + lane] = r;
}
template <int N, bool SPREAD>
void run(const float *ab, float *out, int wps, int steps)
{
    hipFuncSetAttribute((const void *)k<N, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    float ms = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<N, SPREAD>), dim3(256 * wps), dim3(256), 80 * 1024, 0, ab, out, steps);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double ns = ms * 1e6 / ((double)wps * steps);      // per SIMD: wps waves x steps
    printf("%d wave(s) per SIMD, %3d v_fma_f32 per wave-step %s: %7.1f ns per wave-step on its SIMD (MFMA rate %5.1f TFLOP/s)\n", wps, N,
           SPREAD ? "spread between the MFMAs" : "in one block after them ", ns, 1024.0 * (54 * 2048.0 + 108 * 512.0) / ns * 1e-3);
}
int main()
{
    float *ab, *out;
    hipMalloc(&ab, 32 * 64 * 4); hipMalloc(&out, 2048 * 64 * 4);
    float h[32 * 64]; srand(1);
    for (int i = 0; i < 32 * 64; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * (i < 24 * 64 ? 0.1f : 2.f);
    hipMemcpy(ab, h, sizeof h, hipMemcpyHostToDevice);
    const int steps = 2000;
    for (int wps = 1; wps <= 2; ++wps) {
        run<0, false>(ab, out, wps, steps);
        run<64, false>(ab, out, wps, steps);
        run<128, false>(ab, out, wps, steps);
        run<256, false>(ab, out, wps, steps);
        run<108, true>(ab, out, wps, steps);
        run<216, true>(ab, out, wps, steps);
    }
    return 0;
}
