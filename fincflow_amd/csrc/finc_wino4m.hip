// 3x3 forward (and, with transposed fragments, grad-input) with HALF THE MULTIPLIES for the banks one wave cannot hold:
// Winograd F(4,3) along W, M-split over the waves of a workgroup (gfx950 only; round 4).
//
// finc_wino.hip's F(4,3) keeps the whole bank -- 18 x NK x MT fragments -- in ONE wave's registers: that ends at 24 channels per
// group, and the banks above it (FastFlowUnit at C = 100 .. 256, CINCFlowUnit at C = 25 .. 64) ran the direct K-split strip
// kernel.  Here a workgroup of NW = Cq/16 waves owns a strip of 16 column QUADS (64 columns) of one (image, group) slab:
//   * wave w holds the fragments of OUTPUT tile w only (18 x NK registers: 144 / 216 / 288 at Cq = 32 / 48 / 64) and computes
//     those 16 channels completely -- no partial sums, no exchange of results;
//   * the input transform is shared: wave w loads and transforms the k-steps [4w, 4w+4) of a row (a quad + the pair left of it per
//     lane and k-step, as in finc_wino.hip) and parks V in LDS, [slot][k-step][lane] as a 16-byte + an 8-byte cell; after ONE
//     barrier per row every wave reads all of V for its MFMAs.  Four row slots (row & 3): a row is written while the three
//     rows above it are still being read, so one barrier per row is all the synchronisation (a plain s_barrier behind an
//     lgkmcnt wait -- __syncthreads() would fence, i.e. drain the loads and stores in flight);
//   * B^T, A^T, the points (0, +-1, +-3/2, infinity) and the packed bank are finc_wino.hip's F(4,3) (wino_pack_kernel, NF = 6);
//     frequency 1 enters all four outputs with weight +1, so a folded shift rides in as its start value.
// LDS: 4 x NK x 64 x 24 bytes (98 KB at Cq = 64) + the shift: one workgroup per compute unit, one wave per SIMD.
#include "finc_common.h"
#include "finc_tile.h"

#include <stdlib.h>

#include <type_traits>
#include <utility>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr unsigned OFF_INVALID = 0x80000000u;
constexpr unsigned OFF_BAD_CHANNEL = 0x40000000u;

template <int I>
using IC = std::integral_constant<int, I>;
#define FINC_SB() __builtin_amdgcn_sched_barrier(0)

template <int CQP>
struct W4mCfg {
    static constexpr int NW = CQP / 16, NK = CQP / 4, NKL = NK / NW, NA = 3, NF = 6;
    static_assert(CQP % 16 == 0 && NKL == 4, "");
    static constexpr int NGRP = NA * NF * NK;                     // fragments one wave holds: (a, f, j) of its tile
    static constexpr int NFRAG = NGRP * NW;                       // fragment (a, f, j, mt) at ((a*NF + f)*NK + j)*NW + mt
    static constexpr int NPACK = NFRAG + 4 * NW;                  // + the output shift in accumulator layout
    static constexpr int VA = 4 * NK * 64 * 4, VB = 4 * NK * 64 * 2;   // floats: 16-byte cells (f0..f3), 8-byte cells (f4, f5)
    static constexpr size_t LDS_BYTES = sizeof(float) * (size_t)(VA + VB + NW * 256);
};

template <int CQP, bool FW>
__device__ __forceinline__ void wino4m_walk(const __amdgpu_buffer_rsrc_t rin, const __amdgpu_buffer_rsrc_t rout,
                                            const float *__restrict__ packed, float *__restrict__ lds, int g, int CQ, int H, int W,
                                            int strip, int RC, bool fh)
{
    using C = W4mCfg<CQP>;
    constexpr int NW = C::NW, NK = C::NK, NKL = C::NKL, NF = C::NF, NGRP = C::NGRP;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, p = lane & 15;
    const int HW = H * W;
    v4f *const vA = reinterpret_cast<v4f *>(lds);
    v2f *const vB = reinterpret_cast<v2f *>(lds + C::VA);
    float *const blds = lds + C::VA + C::VB + wv * 256;

    // ---- the bank: this wave's output tile, every (row tap, frequency, k-step); pinned to the accumulation registers as far as they go
    float af[NGRP];
    {
        // (a walking pointer, opaque to the compiler: 288 base + constant addresses would each take a scalar register pair)
        const float *pk = packed + (size_t)g * C::NPACK * 64 + wv * 64 + lane;
#pragma unroll
        for (int t = 0; t < NGRP; ++t) {
            af[t] = *pk;
            pk += NW * 64;
            asm volatile("" : "+v"(pk));
        }
#pragma unroll
        for (int t = 0; t < NGRP; ++t)
            if (t < 248) asm volatile("" : "+a"(af[t]));
        const float *pb = packed + ((size_t)g * C::NPACK + C::NFRAG) * 64 + lane;
        *reinterpret_cast<v4f *>(blds + lane * 4) =
            (v4f){pb[(size_t)(4 * wv + 0) * 64], pb[(size_t)(4 * wv + 1) * 64], pb[(size_t)(4 * wv + 2) * 64], pb[(size_t)(4 * wv + 3) * 64]};
    }

    // ---- addressing (finc_wino.hip's scheme).  W % 4 == 0: a quad never straddles the right edge
    const int wt = strip * 64 + 4 * p;                    // canonical columns wt .. wt+3 of this lane's quad
    const bool colok = wt < W, leftok = colok && wt > 0;  // (left of the first quad: the zero padding of layers/conv.py:41-55)
    const unsigned coff = (unsigned)(FW ? W - 4 - wt : wt) * 4u;
    const unsigned loff = (unsigned)(FW ? W - wt : wt - 2) * 4u;          // the pair wt-2, wt-1 (mirrored: wt-1 first)
    unsigned lin[NKL], ll[NKL];                           // this wave's k-steps jg = wv*NKL + jl: channel 4 jg + q
#pragma unroll
    for (int jl = 0; jl < NKL; ++jl) {
        const int ch = 4 * (wv * NKL + jl) + q;
        lin[jl] = (colok && ch < CQ) ? coff + (unsigned)ch * HW * 4u : OFF_BAD_CHANNEL;
        ll[jl] = (leftok && ch < CQ) ? loff + (unsigned)ch * HW * 4u : OFF_BAD_CHANNEL;
    }
    unsigned lo[4];                                       // output register r: channel 16 wv + 4 q + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ch = 16 * wv + 4 * q + r;
        lo[r] = (colok && ch < CQ) ? coff + (unsigned)ch * HW * 4u : OFF_BAD_CHANNEL;
    }
    auto rowoff = [&](int h) { return (h >= 0 && h < H) ? (unsigned)((fh ? H - 1 - h : h) * W) * 4u : OFF_INVALID; };

    for (int i = threadIdx.x; i < C::VA + C::VB; i += 64 * NW) lds[i] = 0.f;
    v4u nx[NKL];                                          // the next row: this lane's quad ...
    v2u nl[NKL];                                          // ... and the pair left of it (memory order)
    auto issue = [&](int h) {
        const unsigned ro = rowoff(h);
#pragma unroll
        for (int jl = 0; jl < NKL; ++jl) {
            nx[jl] = __builtin_amdgcn_raw_buffer_load_b128(rin, ro + lin[jl], 0, 0);
            nl[jl] = __builtin_amdgcn_raw_buffer_load_b64(rin, ro + ll[jl], 0, 0);
        }
    };
    auto pair_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // transform this wave's k-steps of the row in nx / nl and park them in slot S
    auto transform = [&](auto slot_c) {
        constexpr int S = decltype(slot_c)::value;
#pragma unroll
        for (int jl = 0; jl < NKL; ++jl) {
            const v4u m = nx[jl];
            const v2u l = nl[jl];
            const float d0 = __builtin_bit_cast(float, FW ? l.y : l.x), d1 = __builtin_bit_cast(float, FW ? l.x : l.y);
            const float d2 = __builtin_bit_cast(float, FW ? m.w : m.x), d3 = __builtin_bit_cast(float, FW ? m.z : m.y);
            const float d4 = __builtin_bit_cast(float, FW ? m.y : m.z), d5 = __builtin_bit_cast(float, FW ? m.x : m.w);
            const float t1 = __builtin_fmaf(-2.25f, d2, d4), t2 = __builtin_fmaf(-2.25f, d1, d3);
            const float t3 = d4 - d2, t4 = d3 - d1;
            const int cell = (S * NK + wv * NKL + jl) * 64 + lane;
            vA[cell] = (v4f){__builtin_fmaf(2.25f, d0, __builtin_fmaf(-3.25f, d2, d4)), t1 + t2, t1 - t2, __builtin_fmaf(1.5f, t4, t3)};
            vB[cell] = (v2f){__builtin_fmaf(-1.5f, t4, t3), __builtin_fmaf(2.25f, d1, __builtin_fmaf(-3.25f, d3, d5))};
        }
    };
    auto store16 = [&](const v4u &v, unsigned voff) { __builtin_amdgcn_raw_buffer_store_b128(v, rout, voff, 0, 0); };
    const int r0 = blockIdx.y * RC, r1 = r0 + RC < H ? r0 + RC : H;       // output rows of this chunk
    auto step = [&](auto slot_c, int h) {                                 // row h sits in nx / nl; S = (h - hs) & 3
        constexpr int S = decltype(slot_c)::value;
        transform(slot_c);
        FINC_SB();
        issue(h + 1);                                                     // (lands during this step's MFMAs)
        FINC_SB();
        pair_barrier();                                                   // row h is in its slot, all k-steps
        if (h < r0) return;                                               // (filling the slots of a chunk: no output row)
        v4f acc[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[f] = f == 1 ? *reinterpret_cast<const v4f *>(blds + lane * 4) : (v4f){0.f, 0.f, 0.f, 0.f};
        float vb[2][NF];
        auto fetch = [&](int t, float (&dst)[NF]) {                        // t = a*NK + j: row h-a sits in slot (S - a) & 3
            const int a = t / NK, j = t % NK, slot = (S + 4 - a) & 3;
            const v4f lo4 = vA[(slot * NK + j) * 64 + lane];
            const v2f hi2 = vB[(slot * NK + j) * 64 + lane];
            dst[0] = lo4.x; dst[1] = lo4.y; dst[2] = lo4.z; dst[3] = lo4.w; dst[4] = hi2.x; dst[5] = hi2.y;
        };
        fetch(0, vb[0]);
#pragma unroll
        for (int t = 0; t < 3 * NK; ++t) {
            if (t + 1 < 3 * NK) fetch(t + 1, vb[(t + 1) & 1]);
            const int a = t / NK, j = t % NK;
#pragma unroll
            for (int f = 0; f < NF; ++f)
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[(a * NF + f) * NK + j], vb[t & 1][f], acc[f], 0, 0, 0);
            FINC_SB();
        }
        // output transform and stores: one quad per output register and lane -- 16 lanes write 256 contiguous bytes
        const v4f s1 = acc[1] + acc[2], e1 = acc[1] - acc[2], s2 = acc[3] + acc[4], e2 = acc[3] - acc[4];
        const v4f y0 = acc[0] + s1 + s2, y1 = e1 + 1.5f * e2, y2 = s1 + 2.25f * s2, y3 = e1 + 3.375f * e2 + acc[5];
        const float a0[4] = {y0.x, y0.y, y0.z, y0.w}, a1[4] = {y1.x, y1.y, y1.z, y1.w};
        const float a2[4] = {y2.x, y2.y, y2.z, y2.w}, a3[4] = {y3.x, y3.y, y3.z, y3.w};
        const unsigned ro = rowoff(h);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v4u v;
            v.x = __builtin_bit_cast(unsigned, FW ? a3[r] : a0[r]);
            v.y = __builtin_bit_cast(unsigned, FW ? a2[r] : a1[r]);
            v.z = __builtin_bit_cast(unsigned, FW ? a1[r] : a2[r]);
            v.w = __builtin_bit_cast(unsigned, FW ? a0[r] : a3[r]);
            store16(v, ro + lo[r]);
        }
    };
    // rows r0-2, r0-1 fill the slots (rows above the image load zeros); then one output row per step, slots rotating
    const int hs = r0 - 2;
    pair_barrier();                                                        // (the zeroed slots)
    issue(hs);
    for (int h = hs; h < r1; h += 4) {
        step(IC<0>{}, h);
        if (h + 1 < r1) step(IC<1>{}, h + 1);
        if (h + 2 < r1) step(IC<2>{}, h + 2);
        if (h + 3 < r1) step(IC<3>{}, h + 3);
    }
}

// grid = (B*G*NS strips of 64 columns, row chunks); NW wavefronts each, one per SIMD
template <int CQP>
__global__ __launch_bounds__(64 * (CQP / 16)) __attribute__((amdgpu_waves_per_eu(1, 1))) void finc_wino4m_kernel(
    const float *__restrict__ in, const float *__restrict__ packed, float *__restrict__ out, int G, int CQ, int H, int W, int NS, int RC,
    unsigned orient)
{
    extern __shared__ __attribute__((aligned(16))) float lds4m[];
    const int strip = blockIdx.x % NS, bg = blockIdx.x / NS;
    const int g = bg % G;
    const unsigned o = finc_group_orient(orient, g);
    const bool fh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 4u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    if (fw) wino4m_walk<CQP, true>(rin, rout, packed, lds4m, g, CQ, H, W, strip, RC, fh);
    else wino4m_walk<CQP, false>(rin, rout, packed, lds4m, g, CQ, H, W, strip, RC, fh);
}

typedef void (*wino4m_fn)(const float *, const float *, float *, int, int, int, int, int, int, unsigned);
struct W4mInst {
    int cqp, nw, nk, npack;
    size_t lds;
    wino4m_fn fn;
};
template <int CQP>
constexpr W4mInst make_w4m()
{
    using C = W4mCfg<CQP>;
    return W4mInst{CQP, C::NW, C::NK, C::NPACK, C::LDS_BYTES, finc_wino4m_kernel<CQP>};
}
const W4mInst g_w4m[] = {make_w4m<32>(), make_w4m<48>(), make_w4m<64>()};

const W4mInst *find_w4m(int Cq)
{
    if (Cq <= 24) return nullptr;                                          // (finc_wino.hip's one-wave kernels)
    const W4mInst *best = nullptr;
    for (const W4mInst &i : g_w4m)
        if (i.cqp >= Cq && (!best || i.cqp < best->cqp)) best = &i;
    return best;
}

} // namespace

size_t finc_wino4m_packed_bytes(int G, int Cq, int KH, int KW)
{
    if (KH != 3 || KW != 3) return 0;
    const W4mInst *i = find_w4m(Cq);
    return i ? (size_t)i->npack * 64 * sizeof(float) * (size_t)G : 0;
}

bool finc_wino4m_takes(const float *in, const float *out, const FincShape &s)
{
    if (s.KH != 3 || s.KW != 3 || !find_w4m(s.Cq) || finc_wino_disabled()) return false;
    if (s.W % 4 != 0 || s.W < 4) return false;
    if ((((uintptr_t)in) | ((uintptr_t)out)) & 15u) return false;          // a row arrives as 16-byte windows
    if ((size_t)s.Cq * s.H * s.W * 4 >= ((size_t)1 << 30)) return false;
    // strips of 64 columns: at least three quarters of what they cover is image, and enough workgroups for the chip
    const int NS = (s.W + 63) / 64;
    return 4 * s.W >= 3 * NS * 64 && (long long)s.B * s.G * NS * (s.H >= 16 ? s.H / 8 : 1) >= 32;
}

int finc_wino4m_pack(const float *wc, void *packed, int G, int Cq, bool transpose, hipStream_t st, const float *scale, const float *shift)
{
    const W4mInst *i = find_w4m(Cq);
    if (!i) return FINC_ERR_UNSUPPORTED;
    return finc_wino_pack_bank(wc, (float *)packed, G, Cq, i->nw, i->nw, i->nk, 6, transpose, st, scale, shift);
}

int finc_wino4m_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st)
{
    const W4mInst *i = find_w4m(s.Cq);
    if (!i) return FINC_ERR_UNSUPPORTED;
    const int NS = (s.W + 63) / 64;
    const long long wgs = (long long)s.B * s.G * NS;
    const long long fill = 256 * (4 / i->nw);                              // workgroups of one wave per SIMD that a chip holds at once (every
                                                                           // chunk recomputes two rows of operands)
    int nrc = finc_row_chunks(wgs, fill, s.H, 8, 2);                       // (rounds x rows per chunk: finc_common.h)
    const int RC = (s.H + nrc - 1) / nrc;
    nrc = (s.H + RC - 1) / RC;
    if (int e = finc_ensure_dynamic_lds((const void *)i->fn, i->lds)) return e;
    hipLaunchKernelGGL(i->fn, dim3(s.B * s.G * NS, nrc), dim3(64 * i->nw), i->lds, st, in, (const float *)packed, out, s.G, s.Cq, s.H, s.W, NS,
                       RC, s.orient);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

unsigned finc_build_flags_wino4m() { return FINC_BUILD_FLAGS; }
