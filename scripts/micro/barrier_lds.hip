// Micro-benchmark: what one lockstep step of a 4-wave workgroup costs on gfx950 before any arithmetic --
// s_barrier alone, + an LDS write before it and a dependent LDS read behind it (the hand-over of finc_chain.hip),
// + a chain of dependent MFMAs.  One workgroup per CU, s_memtime around the loop, cycles per iteration.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/barrier_lds.hip -o /tmp/barrier_lds && /tmp/barrier_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE, int NMFMA>
__global__ void k(unsigned long long *out, float *sink, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char *b = (char *)lds;
    v4f x = {1.f * lane, 2.f, 3.f, 4.f};
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    float a = 0.001f * lane;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1) {   // read what the "other" wave wrote last iteration
            const v4f r = *(const v4f *)(b + ((wave ^ 1) * 1024 + lane * 16) + (it & 1) * 4096);
            x += r;
        }
        if (MODE >= 2) {
#pragma unroll
            for (int m = 0; m < NMFMA; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x.x, acc, 0, 0, 0);
            x += acc;
        }
        if (MODE >= 1) *(v4f *)(b + (wave * 1024 + lane * 16) + ((it + 1) & 1) * 4096) = x;
        if (MODE == 3) {   // counted wait: nothing but the write is outstanding anyway
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x.x + acc.y;
}

template <int MODE, int NMFMA>
void run(const char *name, int waves)
{
    unsigned long long *d; float *s;
    hipMalloc(&d, 256 * 8 * 8); hipMalloc(&s, 256 * 512 * 4);
    const int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<MODE, NMFMA>), dim3(256), dim3(64 * waves), 32768, 0, d, s, iters);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-58s waves %d: %7.1f cycles / iteration\n", name, waves, (double)h[0] / iters);
    hipFree(d); hipFree(s);
}

int main()
{
    for (int w : {1, 2, 4, 5, 8}) {
        run<0, 0>("s_barrier only", w);
        run<1, 0>("LDS read b128 + add + write b128 + barrier", w);
        run<2, 6>("... + 6 dependent 16x16x4 MFMAs", w);
    }
    return 0;
}
