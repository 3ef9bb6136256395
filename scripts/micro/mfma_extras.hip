// Microbenchmark: what does ONE non-MFMA instruction cost a SIMD that is otherwise saturated with fp32 MFMAs?
// 2 waves per SIMD on all 256 CUs each run the c3 forward's MFMA stream (54 x 16x16x4 + 108 x 4x4x1 per step, see
// mfma_stream.hip) and, between steps, N extra instructions of one kind.  Reported: shader ticks per step per SIMD and
// the marginal cost per extra instruction (both waves' extras: 2N per SIMD-step).
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o mfma_extras.bin mfma_extras.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
enum { NONE, VMOV, VADD, DPP, DPPBANK, PERMSWAP, ACCREAD, DSWRITE, DSREAD, VLOAD, VLOADX4, VSTORE, VSTOREX4, SNOP, SALU };
template <int KIND, int N>
__global__ __launch_bounds__(256) void k(const float *ab, float *out, float *scratch, int steps, unsigned long long *tm)
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
    float e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = b[i];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(scratch + (size_t)wave * 4096), 0, 16384, 0x00020000);
    float *my = lds + (threadIdx.x >> 6) * 2048;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u q4 = {1, 2, 3, 4};
    int sacc = steps;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        c0 = tot; c1 = tot; c2 = tot;
#pragma unroll
        for (int kk = 0; kk < 54; ++kk) {
            const float bb = b[kk % 18];
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk], bb, c0, 0, 0, 0);
            const int f1 = 2 * kk, f2 = 2 * kk + 1;
            if ((f1 & 3) == 0) { c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f1 >> 2], bb, c1, 2, 0, 0); c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f2 >> 2], bb, c2, 2, 1, 0); }
            else { c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f1 >> 2], bb, c1, 2, 2, 0); c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f2 >> 2], bb, c2, 2, 3, 0); }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            float &x = e[i & 7], &y = e[(i + 3) & 7];
            if (KIND == VMOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));
            if (KIND == VADD) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(x));
            if (KIND == DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));
            if (KIND == DPPBANK) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0x8" : "+v"(x) : "v"(y));
            if (KIND == PERMSWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
            if (KIND == ACCREAD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(af[i % 54]));
            if (KIND == DSWRITE) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(lane * 4 + (int)(size_t)my), "v"(x), "n"((i % 16) * 256) : "memory");
            if (KIND == DSREAD) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x) : "v"(lane * 4 + (int)(size_t)my), "n"((i % 16) * 256) : "memory");
            if (KIND == VLOAD) x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, (i % 16) * 256, 0));
            if (KIND == VLOADX4) q4 = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (i % 16) * 1024, 0);
            if (KIND == VSTORE) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), rs, lane * 4, (i % 16) * 256, 0);
            if (KIND == VSTOREX4) __builtin_amdgcn_raw_buffer_store_b128(q4, rs, lane * 16, (i % 16) * 1024, 0);
            if (KIND == SNOP) asm volatile("s_nop 0");
            if (KIND == SALU) asm volatile("s_add_i32 %0, %0, 3" : "+s"(sacc));
        }
        if (KIND == DSREAD || KIND == VLOAD || KIND == VLOADX4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        tot = c0 + c1 + c2;
        tot *= 1e-3f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = tot.x + tot.y + tot.z + tot.w + (float)sacc + __builtin_bit_cast(float, q4.x);
#pragma unroll
    for (int i = 0; i < 8; ++i) r += e[i];
    out[wave * 64 + lane] = r;
    if (lane == 0) tm[wave] = t1 - t0;
}
static double base_ticks = 0;
template <int KIND, int N>
void run(const char *name, const float *ab, float *out, float *scratch, unsigned long long *tm, int steps)
{
    const int waves = 256 * 4 * 2;
    hipFuncSetAttribute((const void *)k<KIND, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    double ticks = 0;
    float ms = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, N>), dim3(waves / 4), dim3(256), 80 * 1024, 0, ab, out, scratch, steps, tm);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        hipDeviceSynchronize();
        static unsigned long long t[2048]; hipMemcpy(t, tm, sizeof t, hipMemcpyDeviceToHost);
        double st = 0; for (int w = 0; w < waves; ++w) st += t[w];
        ticks = st / waves / steps / 2;
    }
    if (KIND == NONE) base_ticks = ticks;
    // wall clock: 2 waves x steps per SIMD in ms -> ns per SIMD-step; MFMA flop of a step = 54*2048 + 108*512
    const double ns = ms * 1e6 / (2.0 * steps);
    printf("%-44s x%3d per wave-step: %6.0f ticks, %6.1f ns per SIMD-step (%5.1f TFLOP/s of MFMA)", name, N, ticks, ns,
           1024.0 * (54 * 2048.0 + 108 * 512.0) / ns * 1e-3);
    if (N) printf("  = +%5.1f per instruction", (ticks - base_ticks) / (2.0 * N));
    printf("\n");
}
int main()
{
    float *ab, *out, *scratch; unsigned long long *tm;
    hipMalloc(&ab, 32 * 64 * 4); hipMalloc(&out, 2048 * 64 * 4); hipMalloc(&tm, 2048 * 8); hipMalloc(&scratch, (size_t)2048 * 16384);
    float h[32 * 64]; srand(1);
    for (int i = 0; i < 32 * 64; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * (i < 24 * 64 ? 0.1f : 2.f);
    hipMemcpy(ab, h, sizeof h, hipMemcpyHostToDevice);
    hipMemset(scratch, 0, (size_t)2048 * 16384);
    const int steps = 2000;
    run<NONE, 0>("MFMA stream only", ab, out, scratch, tm, steps);
    run<VMOV, 32>("v_mov_b32", ab, out, scratch, tm, steps);
    run<VADD, 32>("v_add_f32", ab, out, scratch, tm, steps);
    run<DPP, 32>("v_mov_b32_dpp row_shr", ab, out, scratch, tm, steps);
    run<DPPBANK, 32>("v_mov_b32_dpp row_shr bank_mask:0x8", ab, out, scratch, tm, steps);
    run<PERMSWAP, 32>("v_permlane32_swap_b32", ab, out, scratch, tm, steps);
    run<ACCREAD, 32>("v_accvgpr_read_b32", ab, out, scratch, tm, steps);
    run<DSWRITE, 16>("ds_write_b32", ab, out, scratch, tm, steps);
    run<DSREAD, 16>("ds_read_b32 (+wait)", ab, out, scratch, tm, steps);
    run<VLOAD, 12>("buffer_load_dword (L2 hits, +wait)", ab, out, scratch, tm, steps);
    run<VLOADX4, 6>("buffer_load_dwordx4 (L2 hits, +wait)", ab, out, scratch, tm, steps);
    run<VSTORE, 12>("buffer_store_dword", ab, out, scratch, tm, steps);
    run<VSTOREX4, 6>("buffer_store_dwordx4", ab, out, scratch, tm, steps);
    run<SNOP, 32>("s_nop 0", ab, out, scratch, tm, steps);
    run<SALU, 32>("s_add_i32", ab, out, scratch, tm, steps);
    return 0;
}
