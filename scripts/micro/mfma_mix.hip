// Microbenchmark 2: cost of ONE extra instruction of each kind between two f32 MFMAs (one wave per SIMD).
// Build+run on the GPU box: hipcc -O3 --offload-arch=gfx950 -w -o /tmp/mix mfma_mix.hip && /tmp/mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
// KIND: 0 none, 1 v_cndmask, 2 v_mov_dpp row_shr:1, 3 v_accvgpr_read, 4 ds_write_b32, 5 ds_read_b32 (result unused
// until end), 6 v_add_u32, 7 s_add (salu), 8 v_permlane32_swap, 9 ds_read2st64
template <int KIND, int V>
__global__ __launch_bounds__(64) void k(float *out, int iters, int sel)
{
    __shared__ float lds[4096];
    v4f a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x;
    float w[8];
    int c[8];
    for (int i = 0; i < 8; ++i) { w[i] = x + i; c[i] = threadIdx.x + i; }
    const bool pred = (threadIdx.x & sel) != 0;
    int sacc = iters;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int r = (v + m) % 8;
                if (KIND == 1) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(w[r]) : "v"(w[(r + 3) % 8]), "s"((unsigned long long)sel));
                if (KIND == 2) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(w[r]) : "v"(w[(r + 3) % 8]));
                if (KIND == 3) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(w[r]) : "a"(a1.x));
                if (KIND == 4) asm volatile("ds_write_b32 %0, %1" ::"v"(c[r] * 4), "v"(w[r]) : "memory");
                if (KIND == 5) asm volatile("ds_read_b32 %0, %1" : "=v"(w[r]) : "v"(c[r] * 4) : "memory");
                if (KIND == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c[r]) : "v"(c[(r + 3) % 8]));
                if (KIND == 7) asm volatile("s_add_i32 %0, %0, 3" : "+s"(sacc));
                if (KIND == 8) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(w[r]), "+v"(w[(r + 4) % 8]));
                if (KIND == 9) asm volatile("ds_read2st64_b32 %0, %1 offset1:2" : "=v"(*(double *)&w[(r & 6)]) : "v"(c[r] * 4) : "memory");
            }
            if (KIND == 5 || KIND == 9 || KIND == 4) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = lds[threadIdx.x] + sacc;
    for (int i = 0; i < 8; ++i) s += w[i] + c[i];
    out[blockIdx.x * 64 + threadIdx.x] = a0.x + a1.y + s;
}
template <int KIND, int V>
double run(float *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256;
    k<KIND, V><<<blocks, 64>>>(d, 100, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND, V><<<blocks, 64>>>(d, iters, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / (iters * 8.0);
}
template <int KIND>
void row(float *d, const char *name, double base)
{
    double t2 = run<KIND, 2>(d), t4 = run<KIND, 4>(d), t8 = run<KIND, 8>(d);
    printf("%-22s per pair: V=2 %.1f ns  V=4 %.1f ns  V=8 %.1f ns   => %.1f / %.1f / %.1f ns per instr (%.1f cycles @2.3GHz at V=8)\n",
           name, t2, t4, t8, (t2 - base) / 2, (t4 - base) / 4, (t8 - base) / 8, (t8 - base) / 8 * 2.3);
}
int main()
{
    float *d;
    hipMalloc(&d, 1024 * 64 * 4);
    double base = run<0, 0>(d);
    base = run<0, 0>(d);
    printf("2 MFMA alone: %.1f ns\n", base);
    row<1>(d, "v_cndmask", base); row<2>(d, "v_mov_dpp row_shr", base); row<3>(d, "v_accvgpr_read", base);
    row<4>(d, "ds_write_b32", base); row<5>(d, "ds_read_b32", base); row<9>(d, "ds_read2st64_b32", base);
    row<6>(d, "v_add_u32", base); row<7>(d, "s_add_i32", base); row<8>(d, "v_permlane32_swap", base);
    return 0;
}
