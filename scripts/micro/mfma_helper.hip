// Microbenchmark: can a SECOND wave on the same SIMD do VALU / LDS / VMEM work in the shadow of another wave's
// v_mfma_f32_16x16x4_f32 stream on gfx950?  512-thread workgroups: waves 0-3 (one per SIMD) run an MFMA-only loop,
// waves 4-7 (their SIMD partners) run `kind` helper work or exit at once.  Reported: time of the MFMA loop alone, with
// a partner, and the helper's own throughput.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_helper mfma_helper.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int KIND>   // 0 none, 1 valu (v_fma), 2 valu (v_cndmask/v_mov mix), 3 LDS write+read, 4 global load+store, 5 v_accvgpr moves
__global__ __launch_bounds__(512) void k(float *out, float *buf, int iters, int hiters, unsigned long long *tm)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __shared__ float lds[8 * 64 * 8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        v4f a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0}, a3 = {0, 0, 0, 0};
        float x = lane * 0.001f + 1.0f, y = 1.0f + lane * 0.01f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = a0.x + a1.y + a2.z + a3.w;
        if (lane == 0) tm[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_memtime() - t0;
    } else if (KIND != 0) {
        float w[8];
        for (int i = 0; i < 8; ++i) w[i] = lane + i;
        const int sel = lane & 1;
        float *g = buf + ((size_t)blockIdx.x * 4 + (wave - 4)) * 64 * 64 + lane * 4;
        for (int it = 0; it < hiters; ++it) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                if (KIND == 1) w[v % 8] = __builtin_fmaf(w[v % 8], 1.0001f, 0.5f);
                if (KIND == 2) { w[v % 8] = sel ? w[(v + 1) % 8] : w[v % 8]; asm volatile("v_mov_b32 %0, %0" : "+v"(w[(v + 3) % 8])); }
                if (KIND == 3) { lds[(wave * 64 + lane) * 8 + (v % 8)] = w[v % 8]; w[(v + 1) % 8] += lds[(wave * 64 + ((lane + 1) & 63)) * 8 + (v % 8)]; }
                if (KIND == 4 && v < 4) { float4 t = *reinterpret_cast<float4 *>(g + (size_t)((it * 4 + v) & 15) * 256); t.x += 1.f; *reinterpret_cast<float4 *>(g + (size_t)((it * 4 + v + 8) & 15) * 256) = t; w[0] += t.y; }
                if (KIND == 5) { float t = w[v % 8]; asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a0" : "+v"(t)::"a0"); w[v % 8] = t; }
            }
        }
        float s = 0;
        for (int i = 0; i < 8; ++i) s += w[i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if (lane == 0) tm[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_memtime() - t0;
    }
}
static unsigned long long *g_tm;
static double g_mfma_cyc, g_help_cyc;
template <int KIND>
float run(float *d, float *buf, int iters, int hiters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256, 512>>>(d, buf, 200, 200, g_tm);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipMemset(g_tm, 0, 256 * 8 * 8);
    k<KIND><<<256, 512>>>(d, buf, iters, hiters, g_tm);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[256 * 8];
    hipMemcpy(h, g_tm, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < 256; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += (double)h[i * 8 + w];
    g_mfma_cyc = a / 1024; g_help_cyc = b / 1024;
    return ms;
}
int main()
{
    float *d, *buf;
    hipMalloc(&d, 256 * 512 * 4);
    hipMalloc(&buf, (size_t)256 * 4 * 64 * 64 * 4);
    hipMemset(buf, 0, (size_t)256 * 4 * 64 * 64 * 4);
    const int iters = 20000;            // 32 MFMAs per iteration = 1024 cycles alone
    hipMalloc(&g_tm, 256 * 8 * 8);
    const float base = run<0>(d, buf, iters, 0);
    const double base_cyc = g_mfma_cyc;
    printf("MFMA loop alone: %.3f ms = %.1f cycles per MFMA @2.4GHz\n", base, base * 1e-3 * 2.4e9 / (iters * 32.0));
    const char *names[] = {"", "v_fma x16", "cndmask+mov x16", "lds write+read x16", "global ld+st 16B x4", "accvgpr wr+rd x16"};
    for (int hit : {2500, 5000, 10000, 20000}) {   // helper iterations: 16 ops each
        for (int kk = 1; kk <= 5; ++kk) {
            float t = kk == 1 ? run<1>(d, buf, iters, hit) : kk == 2 ? run<2>(d, buf, iters, hit) : kk == 3 ? run<3>(d, buf, iters, hit)
                    : kk == 4 ? run<4>(d, buf, iters, hit) : run<5>(d, buf, iters, hit);
            printf("helper %-22s %6d iters (%.2f ops per MFMA): kernel %.3f ms | MFMA waves %+.1f %% (memtime ticks) | helper waves %.0f %% of the MFMA waves' time\n",
                   names[kk], hit, hit * 16.0 / (iters * 32.0), t, (g_mfma_cyc / base_cyc - 1) * 100, g_help_cyc / g_mfma_cyc * 100);
        }
    }
    return 0;
}
