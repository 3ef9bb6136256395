// Microbenchmark: a candidate data path for the c3 forward, piece by piece, beside the real MFMA stream (2 waves per SIMD,
// all CUs, real HBM traffic of c3's size).  A row of the strip (24 channels x 16 columns) arrives as TWO dwordx4 loads
// (lane = channel, 16-byte piece), is transposed through LDS into MFMA B operands by ds_read_b32 (the three column shifts are
// three read addresses: no DPP, no halo loads -- the halo columns come from a third, 24-lane load of the neighbour's last
// piece), and the result row leaves through the staging tile as two dwordx4 stores.  Everything is software-pipelined one
// step ahead so that no wait sits in front of the MFMAs.  PIECES bits: 1 loads, 2 LDS transpose + operand reads (else the
// operands stay constant), 4 result staging (reduce + ds_write + ds_read_b128), 8 stores, 16 halo piece.
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o fwd_path.bin fwd_path.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ inline float block_reduce(const v4f &acc)
{
    const float r0 = acc.x, r1 = acc.y, r2 = acc.z, r3 = acc.w;
    const v2u a = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, r0), __builtin_bit_cast(unsigned, r2), false, false);
    const v2u b = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, r1), __builtin_bit_cast(unsigned, r3), false, false);
    const unsigned a0 = a.x, a1 = a.y, b0 = b.x, b1 = b.y;
    const float s02 = __builtin_bit_cast(float, a0) + __builtin_bit_cast(float, a1);
    const float s13 = __builtin_bit_cast(float, b0) + __builtin_bit_cast(float, b1);
    const v2u c = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s02), __builtin_bit_cast(unsigned, s13), false, false);
    const unsigned c0 = c.x, c1 = c.y;
    return __builtin_bit_cast(float, c0) + __builtin_bit_cast(float, c1);
}
constexpr int IP = 24;   // input tile pitch in floats: [channel][4 halo + 16 + pad]
constexpr int OP = 20;   // output tile pitch
template <int PIECES>
__global__ __launch_bounds__(256) void k(const float *ab, const float *in, float *outb, float *out, int steps, unsigned long long *tm)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wave = blockIdx.x * 4 + wv;
    const int q = lane >> 4, p = lane & 15;
    float af[54], as[27];
#pragma unroll
    for (int i = 0; i < 54; ++i) { af[i] = ab[(i % 24) * 64 + lane]; asm volatile("" : "+a"(af[i])); }
#pragma unroll
    for (int i = 0; i < 27; ++i) { as[i] = ab[((i + 7) % 24) * 64 + lane]; asm volatile("" : "+a"(as[i])); }
    float *itile = lds + wv * (2 * 24 * IP + 32 * OP);               // two input tiles (rows alternate) + one output tile
    float *otile = itile + 2 * 24 * IP;
    for (int i = lane; i < 2 * 24 * IP + 32 * OP; i += 64) itile[i] = 0.f;
    // layout: own block [24][steps][16] per wave (rows of a channel are 64 bytes apart), or -- PIECES bit 128 -- the
    // image's: the 4 waves of a workgroup are the 4 strips of a [24][steps][64] slab (a channel row is 256 bytes, a wave's
    // sector every 256 bytes, each 128-byte line shared by two waves)
    constexpr bool IMG = (PIECES & 128) != 0;
    const size_t wave_floats = (size_t)24 * steps * 16;
    const size_t base = IMG ? (size_t)(wave >> 2) * 4 * wave_floats + (size_t)(wave & 3) * 16 : wave * wave_floats;
    const int recs = (int)((IMG ? 4 * wave_floats : wave_floats) * 4) - (IMG ? (wave & 3) * 64 : 0);
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)(in + base), 0, recs, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(outb + base), 0, recs, 0x00020000);
    const unsigned RS = IMG ? 256u : 64u;                           // row stride in bytes
    const unsigned chs = (unsigned)steps * RS;                      // channel stride in bytes
    const int c16 = lane >> 2, pc = lane & 3;
    const unsigned l4 = (unsigned)c16 * chs + (unsigned)pc * 16u;   // x4 form: channel c16 (+16i), 16-byte piece pc
    const unsigned l4b = c16 < 8 ? l4 : 0x40000000u;                // second instruction: channels 16..23 only
    // halo piece: lanes 0..23 = channel, the LAST 16-byte piece of the row before (stands for the neighbour strip's sector)
    const unsigned l1 = (unsigned)q * chs + (unsigned)p * 4u;       // classic: channel q (+4j), column p
    const unsigned lh = p < 2 ? (unsigned)q * chs + (unsigned)p * 4u : 0x40000000u;
    float nxt[6] = {0, 0, 0, 0, 0, 0}, nxh[6] = {0, 0, 0, 0, 0, 0};
    const unsigned lhp = lane < 24 ? (unsigned)lane * chs + 48u : 0x40000000u;
    float X[4][3][6];                                               // [row slot][shift][k-step]
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int j = 0; j < 6; ++j) { X[s][b][j] = ab[(24 + (s + b + j) % 8) * 64 + lane]; asm volatile("" : "+v"(X[s][b][j])); }
    v4u LL[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}}, LHH[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, ostv[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    v4f acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float tot = 0.f;
    auto step = [&](auto sc, int s) {
        constexpr int S = decltype(sc)::value;                      // slot of row s; row s+1 goes to slot (S+1)%4
        constexpr int SN = (S + 1) % 4;
        const unsigned row = (unsigned)s * RS;
        float *it = itile + (S & 1) * 24 * IP;
        // PIECES bit 32: loads run TWO steps ahead (two register sets, by step parity); else one step ahead, one set
        constexpr int DEEP = (PIECES & 32) ? 1 : 0;
        v4u (&L)[2] = LL[DEEP ? (S & 1) : 0];              // the set that arrived / is refilled now
        v4u &LH = LHH[DEEP ? (S & 1) : 0];
        if (PIECES & 8) {                                           // the row staged last step leaves
#pragma unroll
            for (int i = 0; i < 2; ++i) __builtin_amdgcn_raw_buffer_store_b128(ostv[i], ro, i ? l4b : l4, 16u * i * chs + row, 0);
        }
        if (PIECES & 64) {                                          // THE SHIPPED SCHEME: 6 + 6 dword loads a step ahead, DPP shifts
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                X[SN][0][j] = nxt[j];
                X[SN][2][j] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, nxh[j]), __builtin_bit_cast(int, nxt[j]), 0x112, 0xf, 0xf, false));
                const int sh = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, nxh[j]), 0x101, 0xf, 0xf, true);
                X[SN][1][j] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(sh, __builtin_bit_cast(int, nxt[j]), 0x111, 0xf, 0xf, false));
            }
            const unsigned r2 = (unsigned)((s + 2) % steps) * RS;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                nxt[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, l1, 4u * j * chs + r2, 0));
                nxh[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, lh, 4u * j * chs + r2, 0));
            }
        }
        if (PIECES & 256) {                                         // DIRECT: row s+2 goes HBM -> LDS (tile s&1), no registers, no ds_write
            const unsigned r2 = (unsigned)((s + 2) % steps) * RS;
            float *dt = itile + (S & 1) * 24 * IP;                  // 64 lanes x 16 bytes land contiguously from dt on (M0 base)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ri, (__attribute__((address_space(3))) void *)dt, 16, l4, r2, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ri, (__attribute__((address_space(3))) void *)(dt + 256), 16, l4b, 16u * chs + r2, 0, 0);
            float *rt = itile + ((S + 1) & 1) * 24 * IP;            // the tile filled a step ago
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int j = 0; j < 6; ++j) X[SN][b][j] = rt[(4 * j + q) * 16 + (p >= b ? p - b : p)];
        }
        if (PIECES & 2) {                                           // row s+1 (loaded during step s-1) -> LDS, [channel][4 + 16]
            *reinterpret_cast<v4u *>(&it[c16 * IP + 4 + 4 * pc]) = L[0];
            if (c16 < 8) *reinterpret_cast<v4u *>(&it[(16 + c16) * IP + 4 + 4 * pc]) = L[1];
            if ((PIECES & 16) && lane < 24) *reinterpret_cast<v4u *>(&it[lane * IP]) = LH;
        }
        if (PIECES & 1) {                                           // row s+2 is asked for
            const unsigned r2 = (unsigned)((s + 2 + DEEP) % steps) * RS;
            L[0] = __builtin_amdgcn_raw_buffer_load_b128(ri, l4, r2, 0);
            L[1] = __builtin_amdgcn_raw_buffer_load_b128(ri, l4b, 16u * chs + r2, 0);
            if (PIECES & 16) LH = __builtin_amdgcn_raw_buffer_load_b128(ri, lhp, r2 >= RS ? r2 - RS : 0u, 0);
        }
        if (PIECES & 2) {                                           // operands of row s+1: lane (q,p) = channel 4j+q, column p-b
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int j = 0; j < 6; ++j) X[SN][b][j] = it[(4 * j + q) * IP + 4 + p - b];
        }
        if (PIECES & 4) {                                           // result of row s-1 -> staging tile -> 16-byte pieces
            const float v[4] = {acc[0].x, acc[0].y, acc[0].z, acc[0].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) otile[(4 * q + r) * OP + p] = v[r];
            otile[(16 + q) * OP + p] = block_reduce(acc[1]);
            otile[(20 + q) * OP + p] = block_reduce(acc[2]);
#pragma unroll
            for (int i = 0; i < 2; ++i) ostv[i] = *reinterpret_cast<const v4u *>(&otile[(16 * i + c16) * OP + 4 * pc]);
        } else {
            tot += acc[0].x + acc[1].y + acc[2].z;
        }
        __builtin_amdgcn_sched_barrier(0);
        v4f c0 = {1e-3f, 0, 0, 0}, c1 = c0, c2 = c0;
#pragma unroll
        for (int a = 2; a >= 0; --a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int kk = (a * 3 + b) * 6 + j;
                    const float bb = X[(S + 4 - a) % 4][b][j];
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk], bb, c0, 0, 0, 0);
                    const int f1 = 2 * kk, f2 = 2 * kk + 1;
                    if ((f1 & 3) == 0) { c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f1 >> 2], bb, c1, 2, 0, 0); c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f2 >> 2], bb, c2, 2, 1, 0); }
                    else { c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f1 >> 2], bb, c1, 2, 2, 0); c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(as[f2 >> 2], bb, c2, 2, 3, 0); }
                }
        acc[0] = c0; acc[1] = c1; acc[2] = c2;
        __builtin_amdgcn_sched_barrier(0);
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s += 4) {
        step(std::integral_constant<int, 0>{}, s); step(std::integral_constant<int, 1>{}, s + 1);
        step(std::integral_constant<int, 2>{}, s + 2); step(std::integral_constant<int, 3>{}, s + 3);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < 6; ++j) tot += nxt[j] + nxh[j];
    float r = tot + acc[0].x + acc[1].x + acc[2].x + __builtin_bit_cast(float, LL[0][0].x) + __builtin_bit_cast(float, LL[0][1].y) + __builtin_bit_cast(float, LHH[0].x) +
              __builtin_bit_cast(float, LL[1][0].x) + __builtin_bit_cast(float, LL[1][1].y) + __builtin_bit_cast(float, LHH[1].x) +
              __builtin_bit_cast(float, ostv[0].x) + __builtin_bit_cast(float, ostv[1].x);
    out[wave * 64 + lane] = r;
    if (lane == 0) tm[wave] = t1 - t0;
}
static double base_ticks = 0;
template <int PIECES>
void run(const char *name, const float *ab, const float *in, float *outb, float *out, unsigned long long *tm, int steps)
{
    const int waves = 256 * 4 * 2;
    hipFuncSetAttribute((const void *)k<PIECES>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    double ticks = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k<PIECES>), dim3(waves / 4), dim3(256), 80 * 1024, 0, ab, in, outb, out, steps, tm);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
        static unsigned long long t[2048]; hipMemcpy(t, tm, sizeof t, hipMemcpyDeviceToHost);
        double st = 0; for (int w = 0; w < waves; ++w) st += t[w];
        ticks = st / waves / steps / 2;
    }
    if (PIECES == 0) base_ticks = ticks;
    printf("%-72s %6.0f ticks per SIMD-step (+%4.0f)\n", name, ticks, ticks - base_ticks);
}
int main()
{
    const int steps = 128;
    float *ab, *out, *in, *outb; unsigned long long *tm;
    const size_t bytes = (size_t)2048 * 24 * steps * 16 * 4;
    hipMalloc(&ab, 32 * 64 * 4); hipMalloc(&out, 2048 * 64 * 4); hipMalloc(&tm, 2048 * 8); hipMalloc(&in, bytes); hipMalloc(&outb, bytes);
    float h[32 * 64]; srand(1);
    for (int i = 0; i < 32 * 64; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * (i < 24 * 64 ? 0.1f : 2.f);
    hipMemcpy(ab, h, sizeof h, hipMemcpyHostToDevice);
    hipMemset(in, 0x3c, bytes); hipMemset(outb, 0, bytes);
    run<0>("MFMA stream only (operands from 72 rotating registers)", ab, in, outb, out, tm, steps);
    run<1>("+ 2 dwordx4 row loads", ab, in, outb, out, tm, steps);
    run<3>("+ LDS transpose: 2 ds_write_b128 + 18 ds_read_b32 (operands now loaded data)", ab, in, outb, out, tm, steps);
    run<19>("+ halo piece (1 more dwordx4 load, 24 lanes, + ds_write_b128)", ab, in, outb, out, tm, steps);
    run<23>("+ result staging (12 VALU reduce, 6 ds_write_b32, 2 ds_read_b128)", ab, in, outb, out, tm, steps);
    run<31>("+ 2 dwordx4 stores   [the whole path]", ab, in, outb, out, tm, steps);
    run<33>("2 dwordx4 row loads, two steps ahead", ab, in, outb, out, tm, steps);
    run<63>("the whole path, loads two steps ahead", ab, in, outb, out, tm, steps);
    run<64>("SHIPPED scheme: 12 dword loads + 18 DPP/18 mov", ab, in, outb, out, tm, steps);
    run<76>("SHIPPED scheme + staging + 2 dwordx4 stores   [the shipped kernel's path]", ab, in, outb, out, tm, steps);
    run<128 + 31>("the whole path, IMAGE layout (4 strips share a 256-byte row)", ab, in, outb, out, tm, steps);
    run<128 + 76>("the shipped kernel's path, IMAGE layout", ab, in, outb, out, tm, steps);
    run<128 + 1>("2 dwordx4 row loads, IMAGE layout", ab, in, outb, out, tm, steps);
    run<128 + 12>("staging + stores, IMAGE layout", ab, in, outb, out, tm, steps);
    run<256>("DIRECT-to-LDS row loads (2 buffer_load_dwordx4 ... lds) + 18 operand reads", ab, in, outb, out, tm, steps);
    run<256 + 12>("DIRECT-to-LDS loads + operand reads + staging + stores", ab, in, outb, out, tm, steps);
    run<12>("only staging + stores", ab, in, outb, out, tm, steps);
    run<4>("only staging", ab, in, outb, out, tm, steps);
    return 0;
}
