// Microbenchmark: does VALU work hide behind v_mfma_f32_16x16x4_f32 on gfx950 for ONE wave per SIMD?
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu mfma_valu.hip ; run: ./mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int V, int KIND>
__global__ __launch_bounds__(64) void k(float *out, int iters)
{
    v4f a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x, w[8];
    for (int i = 0; i < 8; ++i) w[i] = x + i;
    int sel = threadIdx.x & 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                if (KIND == 0) w[v % 8] = sel ? w[(v + 1) % 8] : w[v % 8];           // v_cndmask
                if (KIND == 1) w[v % 8] = __builtin_fmaf(w[v % 8], 1.0001f, 0.5f);   // v_fma
                if (KIND == 2) asm volatile("v_mov_b32 %0, %0" : "+v"(w[v % 8]));    // v_mov
            }
            __builtin_amdgcn_sched_barrier(0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += w[i];
    out[blockIdx.x * 64 + threadIdx.x] = a0.x + a1.y + s;
}
template <int V, int KIND>
void run(float *d, const char *name)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int blocks : {256, 1024}) {
        k<V, KIND><<<blocks, 64>>>(d, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<V, KIND><<<blocks, 64>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double ns_per_pair = ms * 1e6 / (iters * 8.0);
        printf("%s V=%d blocks=%d: %.1f ns per (2 MFMA + V valu)  = %.1f cycles @2.4GHz\n", name, V, blocks, ns_per_pair,
               ns_per_pair * 2.4);
    }
}
int main()
{
    float *d;
    hipMalloc(&d, 1024 * 64 * 4);
    run<0, 0>(d, "cndmask"); run<2, 0>(d, "cndmask"); run<4, 0>(d, "cndmask"); run<6, 0>(d, "cndmask");
    run<8, 0>(d, "cndmask"); run<12, 0>(d, "cndmask");
    run<4, 1>(d, "fma"); run<8, 1>(d, "fma");
    run<4, 2>(d, "mov"); run<8, 2>(d, "mov");
    return 0;
}
