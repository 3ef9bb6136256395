// Microbenchmark 4: what ONE buffer_load/store_dwordx4 costs a wave that is otherwise issuing MFMAs, as a function of
// how the 64 lanes' addresses are laid out (one wave per SIMD, 1024 waves like the c3 inverse).
// Build+run on the GPU box: hipcc -O3 --offload-arch=gfx950 -w -o /tmp/vm vmem_issue.hip && /tmp/vm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// MODE 0: no memory op.  1: load, lanes contiguous (16 B apart: 1 KiB per instruction).  2: load, lanes 256 B apart
// (every lane its own 128-B line, 16 lanes = 16 rows of a 64-px image, lane rows q = channel planes 16 KiB apart).
// 3: store contiguous.  4: store 256 B apart.  5: load, 4-lane groups cover one 64-B sector, groups 256 B apart.
// 6: load, lane PAIRS cover one 32-B piece (8 rows x 4 channel planes).  7: load, 256 B apart, only 32 lanes active
// (the current SEC parity classes).  8 / 9: the same two as stores.
template <int MODE, int NM>
__global__ __launch_bounds__(64) void k(float *buf, float *out, int iters, unsigned span)
{
    const int l = threadIdx.x;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)(buf + (size_t)blockIdx.x * (span / 4)), 0, (int)span, 0x00020000);
    unsigned off;
    if (MODE == 1 || MODE == 3) off = l * 16;
    else if (MODE == 5) off = (l >> 2) * 256 + (l & 3) * 16;
    else if (MODE == 6 || MODE == 8) off = (l & 1) * 16 + ((l >> 1) & 7) * 256 + (l >> 4) * 16384;
    else off = (l & 15) * 256 + (l >> 4) * 16384;
    v4f a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    v4u acc = {0, 0, 0, 0};
    v4u pend[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};   // loads are consumed 4 iterations later
    float x = l * 0.001f, y = 1.0f + l;
    unsigned step = 0;
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        }
        if (MODE == 7) {
            asm volatile("" ::"v"(pend[0]));
            pend[0] = pend[1]; pend[1] = pend[2]; pend[2] = pend[3];
            if ((l >> 2) & 1) pend[3] = __builtin_amdgcn_raw_buffer_load_b128(r, off + step, 0, 0);
        }
        if (MODE == 9) {
            v4u v = {step, step, step, step};
            __builtin_amdgcn_raw_buffer_store_b128(v, r, ((l >> 2) & 1) ? off + step : 0x80000000u, 0, 0);
        }
        if (MODE == 1 || MODE == 2 || MODE == 5 || MODE == 6) {
            asm volatile("" ::"v"(pend[0]));
            pend[0] = pend[1]; pend[1] = pend[2]; pend[2] = pend[3];
            pend[3] = __builtin_amdgcn_raw_buffer_load_b128(r, off + step, 0, 0);
        }
        if (MODE == 3 || MODE == 4 || MODE == 8) {
            v4u v = {step, step, step, step};
            __builtin_amdgcn_raw_buffer_store_b128(v, r, off + step, 0, 0);
        }
        step = (step + 32) & 0xff;      // walk along the row (stay inside the slab)
        if (MODE == 1 || MODE == 3) step = (step * 32) & 0x7fff;
      }
    }
    out[blockIdx.x * 64 + l] = a0.x + a1.y + acc.x;
}
template <int MODE, int NM>
double run(float *buf, float *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 1024;
    const unsigned span = 393216;       // one (image, group) slab at c3
    k<MODE, NM><<<blocks, 64>>>(buf, d, 100, span);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, NM><<<blocks, 64>>>(buf, d, iters, span);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / iters;
}
template <int NM>
void table(float *buf, float *d)
{
    const double base = run<0, NM>(buf, d);
    printf("%d MFMA pairs per memory op: base %.1f ns | +load contiguous %.1f | +load 256B-strided %.1f | +load 64B-sector groups %.1f | "
           "+store contiguous %.1f | +store 256B-strided %.1f  (ns per op on top of the MFMAs)\n",
           NM, base, run<1, NM>(buf, d) - base, run<2, NM>(buf, d) - base, run<5, NM>(buf, d) - base, run<3, NM>(buf, d) - base,
           run<4, NM>(buf, d) - base);
    printf("      load lane-pairs/32B %.1f | load 32 active lanes strided %.1f | store lane-pairs/32B %.1f | store 32 active strided %.1f\n",
           run<6, NM>(buf, d) - base, run<7, NM>(buf, d) - base, run<8, NM>(buf, d) - base, run<9, NM>(buf, d) - base);
}
int main()
{
    float *buf, *d;
    hipMalloc(&buf, (size_t)1024 * 393216);
    hipMalloc(&d, 1024 * 64 * 4);
    hipMemset(buf, 0, (size_t)1024 * 393216);
    table<4>(buf, d);
    table<8>(buf, d);
    table<16>(buf, d);
    return 0;
}
