// Micro-benchmark: the lockstep skeleton of finc_chain.hip -- one wave ("A") whose step is 6 dependent-pair MFMAs on its own
// registers + the sum of three cells the others prepared, three waves ("B") whose step is 4 MFMAs on old operands + 3 on a cell A
// wrote in the step before -- with the sync variants under discussion.  One workgroup per CU, cycles per step by s_memtime.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form scripts/micro/chain_model.hip -o ablate_build/chain_model
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
#define SB() __builtin_amdgcn_sched_barrier(0)

// SYNC: 0 = lgkmcnt(0) + s_barrier (acknowledged writes), 1 = s_barrier only, 2 = no barrier at all (timing only), 3 = tags + polling, no barrier
// AM / BF / BN: MFMAs of A, B's old-operand MFMAs, B's new-operand MFMAs
template <int SYNC, int AM, int BF, int BN, int NB>
__global__ void k(unsigned long long *out, float *sink, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char *b = (char *)lds;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    const float a = 0.001f * lane;
    v4f q = {1.f * lane, 2.f, 3.f, 0.f};
    v4f keep = {0.f, 0.f, 0.f, 0.f};
    const int xcell = lane * 16, pcell = 8192 + lane * 16;     // x ring: 8 slots of 1 KB; prep: [parity][wave][lane]
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave == 0) {
        for (int t = 0; t < iters; ++t) {
            v4f prep[3];
            if (SYNC != 3)
                for (int i = 0; i < NB; ++i) prep[i] = *(const v4f *)(b + pcell + ((t & 1) * 3 + i) * 1024);
            SB();
            v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < AM / 2; ++m) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, q.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, q.y, acc1, 0, 0, 0);
            }
            SB();
            if (SYNC == 3) {
                bool bad;
                do {
                    bad = false;
                    for (int i = 0; i < NB; ++i) {
                        prep[i] = *(const volatile v4f *)(b + pcell + ((t & 1) * 3 + i) * 1024);
                        const float w = prep[i].w;
                        bad = bad || __builtin_bit_cast(int, w) != t;
                    }
                } while (__builtin_amdgcn_ballot_w64(bad) != 0);
            }
            v4f ps = prep[0];
            for (int i = 1; i < NB; ++i) ps += prep[i];
            v4f x = acc0 + acc1 + ps;
            x.w = __builtin_bit_cast(float, t);
            *(v4f *)(b + xcell + (t & 7) * 1024) = x;
            q.x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (float)x.x), 0x111, 0xf, 0xf, false));
            q.y = x.y * 0.5f;
            if (SYNC == 0) __syncthreads();
            if (SYNC == 1) asm volatile("s_barrier" ::: "memory");
        }
    } else if (wave <= NB) {
        v4f old = {0.5f, 0.25f, 0.125f, 0.f};
        for (int t = 0; t < iters; ++t) {
            // prepares step t + 1 from x(t - 1)
            v4f nw;
            if (SYNC != 3) nw = *(const v4f *)(b + xcell + ((t - 1) & 7) * 1024);
            SB();
            v4f accf0 = {0.f, 0.f, 0.f, 0.f}, accf1 = {0.f, 0.f, 0.f, 0.f}, accn = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < BF / 2; ++m) {
                accf0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, old.x, accf0, 0, 0, 0);
                accf1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, old.y, accf1, 0, 0, 0);
            }
            SB();
            if (SYNC == 3 && t > 0) {
                bool bad;
                do {
                    nw = *(const volatile v4f *)(b + xcell + ((t - 1) & 7) * 1024);
                    const float w = nw.w;
                    bad = __builtin_bit_cast(int, w) != t - 1;
                } while (__builtin_amdgcn_ballot_w64(bad) != 0);
            } else if (SYNC == 3) nw = (v4f){0.f, 0.f, 0.f, 0.f};
            const float n0 = nw.x, n1 = nw.y, n2 = nw.z;
            const float ns[3] = {n0, n1, n2};
#pragma unroll
            for (int m = 0; m < BN; ++m) accn = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ns[m % 3], accn, 0, 0, 0);
            v4f c = accf0 + accf1 + accn;
            c.w = __builtin_bit_cast(float, t + 1);
            *(v4f *)(b + pcell + (((t + 1) & 1) * 3 + (wave - 1)) * 1024) = c;
            old = nw;
            keep += c;
            if (SYNC == 0) __syncthreads();
            if (SYNC == 1) asm volatile("s_barrier" ::: "memory");
        }
    } else {
        for (int t = 0; t < iters; ++t) {   // a bystander (the I/O wave): one LDS copy per step
            const v4f v = *(const v4f *)(b + xcell + ((t - 3) & 7) * 1024);
            *(v4f *)(b + 32768 + lane * 16) = v;
            if (SYNC == 0) __syncthreads();
            if (SYNC == 1) asm volatile("s_barrier" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = q.x + keep.x + keep.y;
}

template <int SYNC, int AM, int BF, int BN, int NB = 3>
void run(const char *name, int waves)
{
    unsigned long long *d; float *s;
    (void)hipMalloc(&d, 256 * 8 * 8); (void)hipMalloc(&s, 256 * 512 * 4);
    const int iters = 4000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<SYNC, AM, BF, BN, NB>), dim3(256), dim3(64 * waves), 65536, 0, d, s, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[8];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-64s waves %d: A %6.1f  B0 %6.1f cycles / step\n", name, waves, (double)h[0] / iters, (double)h[1] / iters);
    (void)hipFree(d); (void)hipFree(s);
}

int main()
{
    for (int w : {4, 5}) {
        run<0, 6, 4, 3>("ack + barrier, A 6 MFMA, B 4 + 3", w);
        run<1, 6, 4, 3>("barrier only (no ack wait)", w);
        run<2, 6, 4, 3>("no sync at all (timing only)", w);
        run<3, 6, 4, 3>("tags + polling, no barrier", w);
        run<0, 0, 0, 0>("ack + barrier, no MFMA", w);
        run<1, 0, 0, 0>("barrier only, no MFMA", w);
        run<3, 0, 0, 0>("tags + polling, no MFMA", w);
        run<0, 6, 0, 3>("ack + barrier, A 6, B 0 + 3", w);
        run<0, 6, 4, 0>("ack + barrier, A 6, B 4 + 0", w);
        run<0, 0, 4, 3>("ack + barrier, A 0, B 4 + 3", w);
        run<3, 6, 0, 3>("tags, A 6, B 0 + 3", w);
        run<3, 6, 0, 3, 1>("tags, A 6, ONE B wave 0 + 3", w);
    }
    return 0;
}
