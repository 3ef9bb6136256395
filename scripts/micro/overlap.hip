// Micro-benchmark: does the latency of an LDS read overlap a chain of fp32 MFMAs that does not depend on it (one wave per SIMD)?
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form scripts/micro/overlap.hip -o ablate_build/overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
#define SB() __builtin_amdgcn_sched_barrier(0)

// MODE bits: 1 = LDS read at the top (used at the end), 2 = NM MFMAs (two chains), 4 = LDS write + lgkmcnt(0) at the end, 8 = s_barrier,
// 16 = the MFMAs' operands come from a DPP of the previous result (the recurrence)
template <int MODE, int NM>
__global__ void k(unsigned long long *out, float *sink, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char *b = (char *)lds;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0.001f * i;
    __syncthreads();
    const float a = 0.001f * lane;
    float q0 = 1.f * lane, q1 = 2.f;
    v4f keep = {0.f, 0.f, 0.f, 0.f};
    const int cell = wave * 4096 + lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < iters; ++t) {
        v4f r = {0.f, 0.f, 0.f, 0.f};
        if (MODE & 1) r = *(const v4f *)(b + cell + (t & 3) * 1024);
        SB();
        v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (MODE & 2) {
#pragma unroll
            for (int m = 0; m < NM / 2; ++m) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, q0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, q1, acc1, 0, 0, 0);
            }
        }
        const v4f x = (acc0 + r) + acc1;
        if (MODE & 4) {
            *(v4f *)(b + cell + ((t + 1) & 3) * 1024) = x;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (MODE & 16) {
            const float x0 = x.x, x1 = x.y;
            q0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x0), 0x111, 0xf, 0xf, false));
            q1 = x1 * 0.5f;
        } else keep += x;
        if (MODE & 8) asm volatile("s_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = q0 + q1 + keep.x + keep.y;
}

template <int MODE, int NM>
void run(const char *name)
{
    unsigned long long *d; float *s;
    (void)hipMalloc(&d, 256 * 8 * 8); (void)hipMalloc(&s, 256 * 512 * 4);
    const int iters = 4000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<MODE, NM>), dim3(256), dim3(256), 65536, 0, d, s, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[8];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-72s %7.1f cycles / iteration\n", name, (double)h[0] / iters);
    (void)hipFree(d); (void)hipFree(s);
}

int main()
{
    run<2, 6>("6 MFMAs (2 chains), independent iterations");
    run<2 | 16, 6>("6 MFMAs, operands from the previous result (DPP): the recurrence");
    run<1, 0>("LDS read b128, used by an add");
    run<1 | 2, 6>("LDS read + 6 MFMAs (independent of it)");
    run<1 | 2 | 16, 6>("LDS read + 6 MFMAs, recurrence");
    run<1 | 4, 0>("LDS read, write, lgkmcnt(0)");
    run<1 | 2 | 4 | 16, 6>("LDS read + 6 MFMAs + write + lgkmcnt(0), recurrence");
    run<1 | 2 | 4 | 8 | 16, 6>("... + s_barrier");
    run<1 | 4 | 8, 0>("LDS read, write, lgkmcnt(0), s_barrier");
    run<2 | 16, 2>("2 MFMAs, recurrence");
    run<2 | 16, 4>("4 MFMAs, recurrence");
    return 0;
}
