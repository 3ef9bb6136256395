// Microbenchmark 3: v_mfma_f32_4x4x1_16B_f32 -- operand/result lane layout and issue cost next to 16x16x4.
// Build+run on the GPU box: hipcc -O3 --offload-arch=gfx950 -w -o /tmp/m4 mfma4x4.hip && /tmp/m4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float v4f __attribute__((ext_vector_type(4)));

// ---- layout: D = A x B with A[lane], B[lane] arbitrary; dump the 4 result registers per lane
__global__ void layout(const float *a, const float *b, float *d)
{
    const int l = threadIdx.x;
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    d[l * 4 + 0] = c.x; d[l * 4 + 1] = c.y; d[l * 4 + 2] = c.z; d[l * 4 + 3] = c.w;
}

// ---- the 4x4 transpose-reduce of pack_d: out lane row q = sum over lane rows of register q
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void treduce(const float *r, float *o)
{
    const int l = threadIdx.x;
    const float r0 = r[l], r1 = r[64 + l], r2 = r[128 + l], r3 = r[192 + l];
    const v2u a = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, r0), __builtin_bit_cast(unsigned, r2), false, false);
    const v2u b = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, r1), __builtin_bit_cast(unsigned, r3), false, false);
    const unsigned a0 = a.x, a1 = a.y, b0 = b.x, b1 = b.y;
    const float s02 = __builtin_bit_cast(float, a0) + __builtin_bit_cast(float, a1);
    const float s13 = __builtin_bit_cast(float, b0) + __builtin_bit_cast(float, b1);
    const v2u c = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s02), __builtin_bit_cast(unsigned, s13), false, false);
    const unsigned c0 = c.x, c1 = c.y;
    o[l] = __builtin_bit_cast(float, c0) + __builtin_bit_cast(float, c1);
}

// ---- throughput: per iteration NB 16x16x4 and NS 4x4x1, independent accumulators, interleaved
template <int NB, int NS, int NACC>
__global__ __launch_bounds__(64) void thr(float *out, int iters)
{
    v4f big[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    v4f sm[NACC];
    for (int i = 0; i < NACC; ++i) sm[i] = (v4f){0, 0, 0, 0};
    float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
            for (int i = 0; i < NB; ++i) big[i & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, big[i & 1], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NS; ++i) sm[i % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, sm[i % NACC], 0, 0, 0);
        }
    }
    float s = big[0].x + big[1].y;
    for (int i = 0; i < NACC; ++i) s += sm[i].x;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NB, int NS, int NACC>
double run(float *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256;
    thr<NB, NS, NACC><<<blocks, 64>>>(d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    thr<NB, NS, NACC><<<blocks, 64>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / (iters * 8.0);
}

int main()
{
    float ha[64], hb[64], hd[256], *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024 * 64 * 4);
    // A[lane] = 1 + lane, B[lane] = 100 + lane: the product D names both source lanes
    for (int l = 0; l < 64; ++l) { ha[l] = 1 + l; hb[l] = 100 + l; }
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    layout<<<1, 64>>>(a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    // hypothesis: block = lane / 4; D register i of lane 4*blk + j = A[4*blk + i] * B[4*blk + j]
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const int blk = l / 4, j = l % 4;
            const float want = ha[4 * blk + i] * hb[4 * blk + j];
            if (hd[l * 4 + i] != want) ++bad;
        }
    printf("layout hypothesis (blk=lane/4, D.reg i @ lane 4blk+j = A[4blk+i]*B[4blk+j]): %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    if (bad)
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int i = 0; i < 4; ++i) {
                // decode which (la, lb) produced it
                int fa = -1, fb = -1;
                for (int la = 0; la < 64 && fa < 0; ++la)
                    for (int lb = 0; lb < 64; ++lb)
                        if (ha[la] * hb[lb] == hd[l * 4 + i]) { fa = la; fb = lb; break; }
                printf("  r%d=A%d*B%d", i, fa, fb);
            }
            printf("\n");
        }
    {
        float hr[256], ho[64], *r;
        hipMalloc(&r, 1024);
        for (int i = 0; i < 256; ++i) hr[i] = (float)(1 << (i % 7)) + i * 1024.f;   // exact in fp32 sums
        hipMemcpy(r, hr, 1024, hipMemcpyHostToDevice);
        treduce<<<1, 64>>>(r, d);
        hipMemcpy(ho, d, 256, hipMemcpyDeviceToHost);
        int badt = 0;
        for (int l = 0; l < 64; ++l) {
            const int q = l >> 4, pp = l & 15;
            float want = 0;
            for (int qq = 0; qq < 4; ++qq) want += hr[q * 64 + qq * 16 + pp];
            if (ho[l] != want) ++badt;
        }
        printf("transpose-reduce (permlane32_swap x2, permlane16_swap): %s (%d mismatches)\n", badt ? "WRONG" : "ok", badt);
    }
    double b2 = run<2, 0, 1>(d);
    b2 = run<2, 0, 1>(d);
    printf("2 x 16x16x4                  : %.1f ns\n", b2);
    printf("4 x 4x4x1 (4 accumulators)   : %.1f ns\n", run<0, 4, 4>(d));
    printf("8 x 4x4x1 (4 accumulators)   : %.1f ns\n", run<0, 8, 4>(d));
    printf("8 x 4x4x1 (2 accumulators)   : %.1f ns\n", run<0, 8, 2>(d));
    printf("8 x 4x4x1 (1 accumulator)    : %.1f ns\n", run<0, 8, 1>(d));
    printf("2 big + 4 small (2 acc)      : %.1f ns\n", run<2, 4, 2>(d));
    printf("2 big + 4 small (4 acc)      : %.1f ns\n", run<2, 4, 4>(d));
    printf("1 big + 2 small (2 acc)      : %.1f ns\n", run<1, 2, 2>(d));
    printf("1 big + 3 small (3 acc)      : %.1f ns\n", run<1, 3, 3>(d));
    printf("0 big + 1 small (1 acc)      : %.1f ns\n", run<0, 1, 1>(d));
    printf("1 big alone (dependent chain): %.1f ns\n", run<1, 0, 1>(d));
    printf("1 big + 4 small (2 acc: sA sB sA sB): %.1f ns\n", run<1, 4, 2>(d));
    printf("1 big + 4 small (4 acc)      : %.1f ns\n", run<1, 4, 4>(d));
    return 0;
}
