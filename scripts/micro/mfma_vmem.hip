// Microbenchmark: the c3 forward's MFMA stream (mfma_stream.hip) plus the forward's MEMORY pattern and nothing else, on a
// full chip (2 waves per SIMD).  Each wave streams through its own region of a 0.4-GB input and output buffer, one "row"
// per step, with real HBM traffic (the regions do not repeat), so the cost of the memory instructions beside the MFMAs can
// be read off variant by variant: MODE bit 0 = row loads, bit 1 = halo loads, bit 2 = stores; LOADS/STORES pick the width.
// The loaded values are only kept alive (the MFMA operands do not depend on them: no data-dependent clock effects).
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o mfma_vmem.bin mfma_vmem.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
// LW: 1 = 6 buffer_load_dword per row (lane (q,p): channel 4j+q, column p: 64-byte sectors, as the kernel), 4 = the same
//     bytes as 2 dwordx4 loads (lane: 16 bytes);  SW: 1 = 24 dword stores, 4 = 2 dwordx4 stores (as shipped)
template <int MODE, int LW, int SW, int PRIO = 0>
__global__ __launch_bounds__(256) void k(const float *ab, const float *in, float *outb, float *out, int steps, unsigned long long *tm)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int q = lane >> 4, p = lane & 15;
    float af[54], as[27], b[18];
#pragma unroll
    for (int i = 0; i < 54; ++i) { af[i] = ab[(i % 24) * 64 + lane]; asm volatile("" : "+a"(af[i])); }
#pragma unroll
    for (int i = 0; i < 27; ++i) { as[i] = ab[((i + 7) % 24) * 64 + lane]; asm volatile("" : "+a"(as[i])); }
#pragma unroll
    for (int i = 0; i < 18; ++i) { b[i] = ab[(24 + i % 8) * 64 + lane]; asm volatile("" : "+v"(b[i])); }
    v4f c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, tot = {0, 0, 0, 0};
    // a wave's region: `steps` rows of 24 channels x 16 columns x 4 bytes = 1536 bytes per row, laid out like a strip of a
    // 64-wide image: channel stride HW*4 with H = steps, W = 64 (this wave = strip blockIdx-ish); keep it simple: the wave
    // owns a [24][steps][16] block, so every row access is 24 sectors of 64 bytes, 6 KB apart per channel.
    const size_t wave_floats = (size_t)24 * steps * 16;
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)(in + wave * wave_floats), 0, (int)(wave_floats * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(outb + wave * wave_floats), 0, (int)(wave_floats * 4), 0x00020000);
    const unsigned chs = (unsigned)steps * 64u;                     // channel stride in bytes
    const unsigned l1 = (unsigned)q * chs + (unsigned)p * 4u;       // dword form: channel q (+4j), column p
    const unsigned lh = p < 2 ? (unsigned)q * chs + (unsigned)p * 4u : 0x40000000u;   // halo: 2 lanes per lane row
    const unsigned l4 = (unsigned)(lane >> 2) * chs + (unsigned)(lane & 3) * 16u;     // x4 form: channel lane/4 (+16i), 16-byte piece
    float nx[6], nh[6]; v4u n4[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int j = 0; j < 6; ++j) nx[j] = nh[j] = 0.f;
    // PRIO 1: the waves of odd workgroups run at priority 3 for the whole kernel; 2: the same by hardware wave slot (HW_ID
    // bits 3:0) parity; 3: every wave raises its priority for its memory phase only
    if (PRIO == 1 && (blockIdx.x & 1)) __builtin_amdgcn_s_setprio(3);
    if (PRIO == 2) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id)); if (id & 1) __builtin_amdgcn_s_setprio(3); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        const unsigned row = (unsigned)s * 64u;
        if (PRIO == 3) __builtin_amdgcn_s_setprio(3);
        // consume the row that arrived (keeps the loads honest), then ask for the next one
#pragma unroll
        for (int j = 0; j < 6; ++j) { asm volatile("" ::"v"(nx[j])); asm volatile("" ::"v"(nh[j])); }
        asm volatile("" ::"v"(n4[0]), "v"(n4[1]));
        if (MODE & 4) {
            if (SW == 4) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    v4u v = {__builtin_bit_cast(unsigned, tot.x), __builtin_bit_cast(unsigned, tot.y), (unsigned)s, (unsigned)i};
                    __builtin_amdgcn_raw_buffer_store_b128(v, ro, i == 1 && (lane >> 2) >= 8 ? 0x40000000u : l4, 16u * i * chs + row, 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 6; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tot.x) + r, ro, l1, (4u * j + 0u) * chs + row + 0u * r, 0);
                }
            }
        }
        if (MODE & 1) {
            if (LW == 4) {
#pragma unroll
                for (int i = 0; i < 2; ++i) n4[i] = __builtin_amdgcn_raw_buffer_load_b128(ri, i == 1 && (lane >> 2) >= 8 ? 0x40000000u : l4, 16u * i * chs + row, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 6; ++j) nx[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, l1, 4u * j * chs + row, 0));
            }
        }
        if (MODE & 2) {
#pragma unroll
            for (int j = 0; j < 6; ++j) nh[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, lh, 4u * j * chs + row, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (PRIO == 3) __builtin_amdgcn_s_setprio(0);
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
        tot = c0 + c1 + c2;
        tot *= 1e-3f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = tot.x + tot.y + tot.z + tot.w + __builtin_bit_cast(float, n4[0].x) + __builtin_bit_cast(float, n4[1].y);
#pragma unroll
    for (int j = 0; j < 6; ++j) r += nx[j] + nh[j];
    out[wave * 64 + lane] = r;
    if (lane == 0) tm[wave] = t1 - t0;
}
static double base_ticks = 0;
template <int MODE, int LW, int SW, int PRIO = 0>
void run(const char *name, const float *ab, const float *in, float *outb, float *out, unsigned long long *tm, int steps)
{
    const int waves = 256 * 4 * 2;
    hipFuncSetAttribute((const void *)k<MODE, LW, SW, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    double ticks = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k<MODE, LW, SW, PRIO>), dim3(waves / 4), dim3(256), 80 * 1024, 0, ab, in, outb, out, steps, tm);
        hipDeviceSynchronize();
        static unsigned long long t[2048]; hipMemcpy(t, tm, sizeof t, hipMemcpyDeviceToHost);
        double st = 0; for (int w = 0; w < waves; ++w) st += t[w];
        ticks = st / waves / steps / 2;
    }
    if (MODE == 0) base_ticks = ticks;
    printf("%-66s %6.0f ticks per SIMD-step (+%4.0f)\n", name, ticks, ticks - base_ticks);
}
int main()
{
    const int steps = 128;                                          // 2048 waves x 128 rows x 1536 B = 0.40 GB each way: c3's image
    float *ab, *out, *in, *outb; unsigned long long *tm;
    const size_t bytes = (size_t)2048 * 24 * steps * 16 * 4;
    hipMalloc(&ab, 32 * 64 * 4); hipMalloc(&out, 2048 * 64 * 4); hipMalloc(&tm, 2048 * 8); hipMalloc(&in, bytes); hipMalloc(&outb, bytes);
    float h[32 * 64]; srand(1);
    for (int i = 0; i < 32 * 64; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * (i < 24 * 64 ? 0.1f : 2.f);
    hipMemcpy(ab, h, sizeof h, hipMemcpyHostToDevice);
    hipMemset(in, 0x3c, bytes); hipMemset(outb, 0, bytes);
    printf("bytes each way: %.3f GB\n", bytes * 1e-9);
    run<0, 1, 4>("MFMA stream only", ab, in, outb, out, tm, steps);
    run<1, 1, 4>("+ 6 row loads (dword)", ab, in, outb, out, tm, steps);
    run<3, 1, 4>("+ 6 row loads + 6 halo loads (dword)", ab, in, outb, out, tm, steps);
    run<4, 1, 4>("+ 2 stores (dwordx4)", ab, in, outb, out, tm, steps);
    run<4, 1, 1>("+ 24 stores (dword)", ab, in, outb, out, tm, steps);
    run<7, 1, 4>("+ 12 loads (dword) + 2 stores (dwordx4)   [the shipped pattern]", ab, in, outb, out, tm, steps);
    run<7, 1, 1>("+ 12 loads (dword) + 24 stores (dword)     [before ab22]", ab, in, outb, out, tm, steps);
    run<1, 4, 4>("+ 2 row loads (dwordx4)", ab, in, outb, out, tm, steps);
    run<5, 4, 4>("+ 2 row loads (dwordx4) + 2 stores (dwordx4)", ab, in, outb, out, tm, steps);
    run<7, 4, 4>("+ 2 row loads (dwordx4) + 6 halo loads + 2 stores (dwordx4)", ab, in, outb, out, tm, steps);
    run<7, 1, 4, 1>("shipped pattern, odd workgroups at priority 3", ab, in, outb, out, tm, steps);
    run<7, 1, 4, 2>("shipped pattern, odd hardware wave slots at priority 3", ab, in, outb, out, tm, steps);
    run<7, 1, 4, 3>("shipped pattern, priority 3 during the memory phase", ab, in, outb, out, tm, steps);
    run<7, 1, 1, 1>("12 loads + 24 dword stores, odd workgroups at priority 3", ab, in, outb, out, tm, steps);
    return 0;
}
