// Microbenchmark: the forward kernel's MFMA stream alone, on a full chip (2 waves per SIMD, 256 CUs).
// One "step" = 54 v_mfma_f32_16x16x4_f32 on one accumulator + 108 v_mfma_f32_4x4x1_16B_f32 (cbsz 2 / abid 0..3) on two,
// A operands from 81 registers (AGPRs or VGPRs), B operands from 18 VGPRs: what c3's strip walk issues per row, with no
// loads, stores, shifts or LDS.  Ideal: 54*32 + 108*8 = 2592 cycles per step per wave-pair slot.
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o mfma_stream.bin mfma_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
// KIND bit 0: A in AGPRs; bit 1: skip the small MFMAs; bit 2: skip the big ones; bit 3: source order big,small,small fenced
template <int KIND>
__global__ __launch_bounds__(256) void k(const float *ab, float *out, int steps, unsigned long long *tm)
{
    extern __shared__ float pad[];
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (steps < 0) pad[threadIdx.x] = 0.f;
    float af[54], as[27], b[18];
#pragma unroll
    for (int i = 0; i < 54; ++i) { af[i] = ab[(i % 24) * 64 + lane]; if (KIND & 1) asm volatile("" : "+a"(af[i])); else asm volatile("" : "+v"(af[i])); }
#pragma unroll
    for (int i = 0; i < 27; ++i) { as[i] = ab[((i + 7) % 24) * 64 + lane]; if (KIND & 1) asm volatile("" : "+a"(as[i])); else asm volatile("" : "+v"(as[i])); }
#pragma unroll
    for (int i = 0; i < 18; ++i) { b[i] = ab[(24 + i % 8) * 64 + lane]; asm volatile("" : "+v"(b[i])); }
    v4f c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0}, c4 = {0, 0, 0, 0}, tot = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        c0 = tot; c1 = tot; c2 = tot; c3 = tot; c4 = tot;
#pragma unroll
        for (int kk = 0; kk < 54; ++kk) {
            const float bb = b[kk % 18];
            if (!(KIND & 4)) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk], bb, c0, 0, 0, 0);
            if ((KIND & 16) && (kk & 1)) {                          // odd k-steps accumulate into a second pair
                const int f1 = 2 * kk, f2 = 2 * kk + 1;
                c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f1 >> 2], bb, c3, 2, 2, 0); c4 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f2 >> 2], bb, c4, 2, 3, 0);
            } else if (!(KIND & 2)) {
                const int f1 = 2 * kk, f2 = 2 * kk + 1;
                switch (f1 & 3) {
                case 0: c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f1 >> 2], bb, c1, 2, 0, 0); c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f2 >> 2], bb, c2, 2, 1, 0); break;
                default: c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f1 >> 2], bb, c1, 2, 2, 0); c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f2 >> 2], bb, c2, 2, 3, 0); break;
                }
            }
            if (KIND & 8) __builtin_amdgcn_sched_barrier(0);
        }
        tot = c0 + c1 + c2 + c3 + c4;   // a few VALU per step: the accumulators must be live results
        tot *= 1e-3f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[wave * 64 + lane] = tot.x + tot.y + tot.z + tot.w;
    if (lane == 0) tm[wave] = t1 - t0;
}
template <int KIND>
void run(const char *name, const float *ab, float *out, unsigned long long *tm, int steps, double ideal, int per_simd = 2)
{
    const int waves = 256 * 4 * per_simd;
    hipFuncSetAttribute((const void *)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(waves / 4), dim3(256), 80 * 1024, 0, ab, out, steps, tm);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long t[2048]; hipMemcpy(t, tm, sizeof t, hipMemcpyDeviceToHost);
        double st = 0; for (int w = 0; w < waves; ++w) st += t[w];
        if (rep == 2) printf("%-52s %.3f ms; shader ticks per step per SIMD: %.0f  (MFMA issue alone: %.0f)  -> %.1f %%\n", name, ms,
                             st / waves / steps / per_simd, ideal, 100.0 * ideal / (st / waves / steps / per_simd));
    }
}
int main()
{
    float *ab, *out; unsigned long long *tm;
    hipMalloc(&ab, 32 * 64 * 4); hipMalloc(&out, 2048 * 64 * 4); hipMalloc(&tm, 2048 * 8);
    float h[32 * 64]; srand(1);
    for (int i = 0; i < 32 * 64; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * (i < 24 * 64 ? 0.1f : 2.f);
    hipMemcpy(ab, h, sizeof h, hipMemcpyHostToDevice);
    const int steps = 4000;
    run<0>("A in VGPRs, compiler's order", ab, out, tm, steps, 2592);
    run<1>("A in AGPRs, compiler's order", ab, out, tm, steps, 2592);
    run<8>("A in VGPRs, big/small/small fenced per k-step", ab, out, tm, steps, 2592);
    run<9>("A in AGPRs, big/small/small fenced per k-step", ab, out, tm, steps, 2592);
    run<3>("A in AGPRs, 16x16x4 only (one chain)", ab, out, tm, steps, 1728);
    run<5>("A in AGPRs, 4x4x1 only (two chains)", ab, out, tm, steps, 864);
    run<2>("A in VGPRs, 16x16x4 only (one chain)", ab, out, tm, steps, 1728);
    run<4>("A in VGPRs, 4x4x1 only (two chains)", ab, out, tm, steps, 864);
    printf("-- ONE wave per SIMD (the inverse's situation: one problem per SIMD at c3)\n");
    run<1>("1 wave: compiler's order", ab, out, tm, steps, 2592, 1);
    run<9>("1 wave: big/small/small fenced per k-step", ab, out, tm, steps, 2592, 1);
    run<17>("1 wave: four small accumulators, compiler's order", ab, out, tm, steps, 2592, 1);
    run<25>("1 wave: four small accumulators, fenced per k-step", ab, out, tm, steps, 2592, 1);
    run<3>("1 wave: 16x16x4 only", ab, out, tm, steps, 1728, 1);
    run<5>("1 wave: 4x4x1 only (two chains)", ab, out, tm, steps, 864, 1);
    run<21>("1 wave: 4x4x1 only (four chains)", ab, out, tm, steps, 864, 1);
    return 0;
}
