// 5x5 forward (and, with transposed fragments, grad-input) with FEWER MULTIPLIES: Winograd F(2,5) along W (gfx950 only).
//
// configs[4] of BASELINE.json (5x5, C = 192: Cq = 48) is the most compute-bound shape of the path (300 flop per byte): its
// forward ran the direct K-split strip kernel of finc_conv.hip at 77 % of the fp32 MFMA peak, i.e. at the roof of the direct
// sum.  As for the 3x3 banks (finc_wino.hip) the only way further is fewer multiplies, and the forward has no recurrence, so
// any exact reformulation of layers/conv.py:102-107 is allowed.  Along W two neighbouring outputs of one row share their inputs:
//
//     d[m] = x[i, h-a, wt-4+m], m = 0..5         the six columns the outputs wt, wt+1 of row tap a read
//     V = B^T d = (d0-5d2+4d4, (4d4-d2)+(4d3-d1), (4d4-d2)-(4d3-d1), 2(d2-d4)+(d1-d3), 2(d2-d4)-(d1-d3), d1-5d3+4d5)
//     U = G g    (g_k = w[o, i, KH-1-a, k], k = 0..4; once per weight version, fp64):
//         (g0, (g0+g1+g2+g3+g4)/6, (g0-g1+g2-g3+g4)/6, (16g0+8g1+4g2+2g3+g4)/12, (16g0-8g1+4g2-2g3+g4)/12, g4/4)
//     M_f = sum_a sum_i U_{a,f}[o,i] * V_{a,f}[i]                  6 frequencies x 5 row taps instead of 25 taps x 2 outputs
//     y(wt) = M0+M1+M2+M3+M4,   y(wt+1) = (M1-M2) + (M3-M4)/2 + M5
//
// (Cook-Toom with the points 0, +-1, +-1/2, infinity; every constant of B^T and A^T is exact in fp32; the matrices are restated
// with rational arithmetic in tests/test_winograd_algebra.py.)  30 multiplies per pair of outputs and (o, i) instead of 50:
// 0.6 x the MFMAs of the direct sum.  Error of this arithmetic in fp32 at the c5 bank: 2.2e-6 of the largest output (direct
// fp32 sum: 1.4e-6; numpy model in the same test); the GPU tests hold the kernel to BASELINE.json's 1e-5.
//
// Mapping: a workgroup of NW waves owns a strip of 16 column PAIRS (32 columns) of one (image, group) slab and walks it top to
// bottom, one row per step; lane (q,p) = pair p, k-slot q.  Wave w owns the input-channel k-steps [w*NKL, (w+1)*NKL) of every
// (row tap, frequency) -- 1/NW of the bank, register-resident (270 fragments at Cq = 48, NW = 4) -- and the partial OUTPUT
// pairs (the output transform is linear: it is applied to the partial sums) are summed through LDS once per row, each wave
// finalising and storing its share of the channels (finc_conv.hip's K-split).  A row arrives as three dwordx2 per lane and
// k-step -- the pairs wt-4, wt-2, wt: a pair is either wholly on the image or wholly in the zero padding of
// layers/conv.py:41-55, so no edge is special; neighbouring lanes overlap and meet in L1 -- and leaves as one dwordx2 per
// output register (16 lanes = 128 contiguous bytes).  The transformed rows h-1 .. h-4 wait in LDS (a lane reads back only
// what it wrote: no barrier for them).
#include "finc_common.h"
#include "finc_tile.h"

#include <stdlib.h>

#include <type_traits>
#include <utility>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

constexpr unsigned OFF_INVALID = 0x80000000u;
constexpr unsigned OFF_BAD_CHANNEL = 0x40000000u;

template <int I>
using IC = std::integral_constant<int, I>;
#define FINC_SB() __builtin_amdgcn_sched_barrier(0)

template <int CQP, int NW>
struct W5Cfg {
    static constexpr int MTB = CQP / 16, NSM = (CQP % 16) / 4, MT = MTB + NSM, NK = CQP / 4, NKL = NK / NW;
    static constexpr int NA = 5, NF = 6;
    static constexpr int NFRAG = NA * NF * NKL * MT;              // fragments one wave holds
    static constexpr int NFRAGT = NA * NF * NK * MT;              // fragment (a, f, j, mt) at ((a*NF + f)*NK + j)*MT + mt
    static constexpr int NPACK = NFRAGT + 4 * MT;                 // + the output shift in accumulator layout (finc_conv.hip)
    static constexpr int NOUT = 4 * MTB + NSM;                    // output registers of a pixel
    static constexpr int DREG = NOUT / NW;                        // ... this wave finalises
    static_assert(NK % NW == 0 && NOUT % NW == 0, "K-split must divide the k-steps and the output registers");
    static constexpr int VSLOT = NF * NKL * 64;                   // floats of one transformed row of one wave
    static constexpr int VWAVE = 4 * VSLOT + MT * 256;            // rows h-1 .. h-4 + the shift (4 floats per tile and lane)
    static constexpr int XCH = NW > 1 ? 2 * NW * NW * DREG * 2 * 64 : 0;   // [parity][dst][src][reg][output][lane]
    static constexpr size_t LDS_BYTES = sizeof(float) * (size_t)(NW * VWAVE + XCH);
};

template <int CQP, int NW, bool FW>
__device__ __forceinline__ void wino5_walk(const __amdgpu_buffer_rsrc_t rin, const __amdgpu_buffer_rsrc_t rout,
                                           const float *__restrict__ packed, float *__restrict__ lds, int g, int CQ, int H, int W,
                                           int strip, int RC, bool fh)
{
    using C = W5Cfg<CQP, NW>;
    constexpr int MT = C::MT, MTB = C::MTB, NSM = C::NSM, NK = C::NK, NKL = C::NKL, NA = C::NA, NF = C::NF, NOUT = C::NOUT, DREG = C::DREG;
    const int wv = NW > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, p = lane & 15;
    const int HW = H * W;
    float *const vlds = lds + wv * C::VWAVE;
    float *const blds = vlds + 4 * C::VSLOT;
    float *const xch = lds + NW * C::VWAVE;

    // ---- the bank: this wave's k-steps of every (row tap, frequency).  16-row-tile fragments one register each, 4-row-block
    // fragments four to a register (finc_tile.h); pinned to the accumulation registers as far as those go
    constexpr int NGRP = NA * NF * NKL;                   // (a, f, jl) triples
    constexpr int NSMALL = NGRP * NSM, NSR = (NSMALL + 3) / 4;
    constexpr int NBIG = NGRP * MTB;
    float af[NBIG > 0 ? NBIG : 1];
    float afs[NSR > 0 ? NSR : 1];
    {
        const float *pk = packed + (size_t)g * C::NPACK * 64 + lane;
        auto gidx = [&](int t, int mt) {                  // t = (a*NF + f)*NKL + jl  ->  packed fragment index
            const int af_ = t / NKL, jl = t % NKL;
            return (af_ * NK + wv * NKL + jl) * MT + mt;
        };
#pragma unroll
        for (int t = 0; t < NGRP; ++t)
#pragma unroll
            for (int mt = 0; mt < MTB; ++mt) af[t * MTB + mt] = pk[gidx(t, mt) * 64];
        const int quad = (lane & 15) >> 2;
#pragma unroll
        for (int r = 0; r < NSR; ++r) {
            int gi = 0;
#pragma unroll
            for (int a = 3; a >= 0; --a) {
                constexpr int NSMD = NSM > 0 ? NSM : 1;
                const int sfr = 4 * r + a < NSMALL ? 4 * r + a : NSMALL - 1;
                const int ga = gidx(sfr / NSMD, MTB + sfr % NSMD);
                gi = (a == 3 || quad == a) ? ga : gi;
            }
            afs[r] = pk[gi * 64];
        }
        constexpr int NPIN = NBIG + NSR <= 252 ? NBIG : 252 - NSR;
#pragma unroll
        for (int t = 0; t < NBIG; ++t)
            if (t < NPIN) asm volatile("" : "+a"(af[t]));
#pragma unroll
        for (int r = 0; r < NSR; ++r) asm volatile("" : "+a"(afs[r]));
    }
    auto mma = [&](v4f &acc_, int t, int mt, float b) {   // t = (a*NF + f)*NKL + jl
        if (mt < MTB) acc_ = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t * MTB + mt], b, acc_, 0, 0, 0);
        else {
            const int sfr = t * NSM + (mt - MTB);
            finc_mma_small(acc_, afs[sfr >> 2], b, sfr & 3);
        }
    };
    // the output shift (accumulator layout) waits in LDS; it enters once: wave 0, frequency 1 (weight +1 in both outputs)
    {
        const float *pb = packed + ((size_t)g * C::NPACK + C::NFRAGT) * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            v4f b = (v4f){pb[(4 * mt + 0) * 64], pb[(4 * mt + 1) * 64], pb[(4 * mt + 2) * 64], pb[(4 * mt + 3) * 64]};
            if (wv != 0) b = (v4f){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<v4f *>(blds + (mt * 64 + lane) * 4) = b;
        }
    }

    // ---- addressing (finc_conv.hip's scheme): offset = row part (scalar; OFF_INVALID for a row off the image) + lane part
    // (pair + the lane row's share of the channel; OFF_BAD_CHANNEL for a pair off the image or a padded channel) + the uniform
    // share of the channel in the instruction's scalar offset.  W is even: a pair never straddles an edge.
    const int wt = strip * 32 + 2 * p;                    // canonical columns wt, wt+1 of this lane's pair
    const unsigned qoff = (unsigned)q * HW * 4u;
    unsigned lin[3][NKL];                                 // input pairs wt-4, wt-2, wt of k-step jl
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int c = wt - 4 + 2 * k;
        const bool ok = c >= 0 && c < W;
        const unsigned coff = (unsigned)(FW ? W - 2 - c : c) * 4u;
#pragma unroll
        for (int jl = 0; jl < NKL; ++jl)
            lin[k][jl] = (ok && 4 * (wv * NKL + jl) + q < CQ) ? coff + qoff : OFF_BAD_CHANNEL;
    }
    unsigned lout[DREG];                                  // output registers this wave finalises: d = wv*DREG + dl
    {
        const bool colok = wt < W;
        const unsigned coff = (unsigned)(FW ? W - 2 - wt : wt) * 4u;
#pragma unroll
        for (int dl = 0; dl < DREG; ++dl) {
            const int d = wv * DREG + dl;
            const int c = d < 4 * MTB ? 16 * (d >> 2) + 4 * q + (d & 3) : 16 * MTB + 4 * (d - 4 * MTB) + q;
            lout[dl] = (colok && c < CQ) ? coff + (unsigned)c * HW * 4u : OFF_BAD_CHANNEL;
        }
    }
    auto rowoff = [&](int h) { return (h >= 0 && h < H) ? (unsigned)((fh ? H - 1 - h : h) * W) * 4u : OFF_INVALID; };

    // ---- state: the transformed row h in registers; rows h-1 .. h-4 in LDS, slot = row & 3, layout [slot][f][jl][lane].  A row
    // enters its slot at the END of its step, when the row it replaces (h-4) has been read for the last time.
    float Vc[NF][NKL];
    for (int i = lane; i < 4 * C::VSLOT; i += 64) vlds[i] = 0.f;
    v2u nx[NKL][3];                                       // raw pairs of the next row (memory order)
    auto issue = [&](int h) {
        const unsigned ro = rowoff(h);
#pragma unroll
        for (int jl = 0; jl < NKL; ++jl)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                nx[jl][k] = __builtin_amdgcn_raw_buffer_load_b64(rin, ro + lin[k][jl], 4 * (wv * NKL + jl) * HW * 4, 0);
    };
    auto transform = [&]() {
#pragma unroll
        for (int jl = 0; jl < NKL; ++jl) {
            // canonical order of a pair = memory (x, y), or (y, x) when the group is W-flipped
            const float d0 = __builtin_bit_cast(float, FW ? nx[jl][0].y : nx[jl][0].x), d1 = __builtin_bit_cast(float, FW ? nx[jl][0].x : nx[jl][0].y);
            const float d2 = __builtin_bit_cast(float, FW ? nx[jl][1].y : nx[jl][1].x), d3 = __builtin_bit_cast(float, FW ? nx[jl][1].x : nx[jl][1].y);
            const float d4 = __builtin_bit_cast(float, FW ? nx[jl][2].y : nx[jl][2].x), d5 = __builtin_bit_cast(float, FW ? nx[jl][2].x : nx[jl][2].y);
            const float t1 = 4.f * d4 - d2, t2 = 4.f * d3 - d1, t3 = 2.f * (d2 - d4), t4 = d1 - d3;
            Vc[0][jl] = 4.f * d4 + (d0 - 5.f * d2);
            Vc[1][jl] = t1 + t2;
            Vc[2][jl] = t1 - t2;
            Vc[3][jl] = t3 + t4;
            Vc[4][jl] = t3 - t4;
            Vc[5][jl] = 4.f * d5 + (d1 - 5.f * d3);
        }
    };
    auto keep = [&](auto slot_c) {                        // Vc -> the slot of this row
        constexpr int S = decltype(slot_c)::value;
#pragma unroll
        for (int jl = 0; jl < NKL; ++jl)
#pragma unroll
            for (int f = 0; f < NF; ++f) vlds[((S * NF + f) * NKL + jl) * 64 + lane] = Vc[f][jl];
    };
    const int r0 = blockIdx.y * RC, r1 = r0 + RC < H ? r0 + RC : H;       // output rows of this chunk
    int parity = 0;
    auto step = [&](auto slot_c, int h) {                                 // row h sits in nx; S = (h - hs) & 3
        constexpr int S = decltype(slot_c)::value;
        transform();
        FINC_SB();
        issue(h + 1);                                                     // (lands during this step's MFMAs)
        FINC_SB();
        if (h < r0) { keep(slot_c); return; }                             // (filling the slots of a chunk: no output row)
        v4f acc[NF][MT];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[f][mt] = f == 1 ? *reinterpret_cast<const v4f *>(blds + (mt * 64 + lane) * 4) : (v4f){0.f, 0.f, 0.f, 0.f};
        // row tap a = 0 from registers; a = 1 .. 4 from LDS, one k-step (6 frequencies) ahead of its MFMAs
        float vb[2][NF];
        auto fetch = [&](int t, float (&dst)[NF]) {                        // t = (a-1)*NKL + jl, a = 1 .. 4
            const int a = 1 + t / NKL, jl = t % NKL, slot = (S + 4 - a) & 3;   // (row h-a sits in slot (S - a) mod 4)
#pragma unroll
            for (int f = 0; f < NF; ++f) dst[f] = vlds[((slot * NF + f) * NKL + jl) * 64 + lane];
        };
        fetch(0, vb[0]);
#pragma unroll
        for (int jl = 0; jl < NKL; ++jl) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) mma(acc[f][mt], (0 * NF + f) * NKL + jl, mt, Vc[f][jl]);
        }
        FINC_SB();
#pragma unroll
        for (int t = 0; t < 4 * NKL; ++t) {
            if (t + 1 < 4 * NKL) fetch(t + 1, vb[(t + 1) & 1]);
            const int a = 1 + t / NKL, jl = t % NKL;
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) mma(acc[f][mt], (a * NF + f) * NKL + jl, mt, vb[t & 1][f]);
            FINC_SB();
        }
        keep(slot_c);                                                     // (row h-4 has been read for the last time)
        // output transform on the PARTIAL sums (linear), 4-row blocks reduced behind it
        float yy[2][NOUT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const v4f s12 = acc[1][mt] + acc[2][mt], d12 = acc[1][mt] - acc[2][mt];
            const v4f s34 = acc[3][mt] + acc[4][mt], d34 = acc[3][mt] - acc[4][mt];
            const v4f y0 = (acc[0][mt] + s12) + s34;
            const v4f y1 = (d12 + acc[5][mt]) + 0.5f * d34;
            if (mt < MTB) {
                const float a0[4] = {y0.x, y0.y, y0.z, y0.w}, a1[4] = {y1.x, y1.y, y1.z, y1.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) { yy[0][4 * mt + r] = a0[r]; yy[1][4 * mt + r] = a1[r]; }
            } else {
                yy[0][4 * MTB + (mt - MTB)] = finc_block_reduce(y0);
                yy[1][4 * MTB + (mt - MTB)] = finc_block_reduce(y1);
            }
        }
        const unsigned ro = rowoff(h);
        if constexpr (NW == 1) {
#pragma unroll
            for (int d = 0; d < NOUT; ++d) {
                v2u v;
                v.x = __builtin_bit_cast(unsigned, FW ? yy[1][d] : yy[0][d]);
                v.y = __builtin_bit_cast(unsigned, FW ? yy[0][d] : yy[1][d]);
                __builtin_amdgcn_raw_buffer_store_b64(v, rout, ro + lout[d], 0, 0);
            }
        } else {
            // exchange: output register d belongs to wave d / DREG; everybody ships the registers it does not own, one barrier,
            // the owner adds the NW-1 partials it received and stores (double-buffered by the parity of the row)
            float *xb = xch + parity * (NW * NW * DREG * 2 * 64);
#pragma unroll
            for (int d = 0; d < NOUT; ++d) {
                const int dst = d / DREG;
                if (dst != wv) {
                    xb[(((dst * NW + wv) * DREG + d % DREG) * 2 + 0) * 64 + lane] = yy[0][d];
                    xb[(((dst * NW + wv) * DREG + d % DREG) * 2 + 1) * 64 + lane] = yy[1][d];
                }
            }
            __syncthreads();
#pragma unroll
            for (int dl = 0; dl < DREG; ++dl) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int d = 0; d < NOUT; ++d)
                    if (d / DREG == wv && d % DREG == dl) { s0 = yy[0][d]; s1 = yy[1][d]; }   // own partial (wv is wave-uniform)
#pragma unroll
                for (int src = 0; src < NW; ++src)
                    if (src != wv) {
                        s0 += xb[(((wv * NW + src) * DREG + dl) * 2 + 0) * 64 + lane];
                        s1 += xb[(((wv * NW + src) * DREG + dl) * 2 + 1) * 64 + lane];
                    }
                v2u v;
                v.x = __builtin_bit_cast(unsigned, FW ? s1 : s0);
                v.y = __builtin_bit_cast(unsigned, FW ? s0 : s1);
                __builtin_amdgcn_raw_buffer_store_b64(v, rout, ro + lout[dl], 0, 0);
            }
            parity ^= 1;
        }
    };
    // rows r0-4 .. r0-1 fill the slots (rows above the image load zeros); then one output row per step, slots rotating
    const int hs = r0 - 4;
    issue(hs);
    for (int h = hs; h < r1; h += 4) {
        step(IC<0>{}, h);
        if (h + 1 < r1) step(IC<1>{}, h + 1);
        if (h + 2 < r1) step(IC<2>{}, h + 2);
        if (h + 3 < r1) step(IC<3>{}, h + 3);
    }
}

// grid = (B*G*NS strips of 32 columns, row chunks); NW wavefronts each, one per SIMD
template <int CQP, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(1, 1))) void finc_wino5_kernel(
    const float *__restrict__ in, const float *__restrict__ packed, float *__restrict__ out, int G, int CQ, int H, int W, int NS, int RC,
    unsigned orient)
{
    extern __shared__ __attribute__((aligned(16))) float lds5[];
    const int strip = blockIdx.x % NS, bg = blockIdx.x / NS;
    const int g = bg % G;
    const unsigned o = finc_group_orient(orient, g);
    const bool fh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 4u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    if (fw) wino5_walk<CQP, NW, true>(rin, rout, packed, lds5, g, CQ, H, W, strip, RC, fh);
    else wino5_walk<CQP, NW, false>(rin, rout, packed, lds5, g, CQ, H, W, strip, RC, fh);
}

// -----------------------------------------------------------------------------------------------
// Bank: U_{a,f} = filter transform of row a of the (canonical) 5x5 kernel, in the fragment layout of the strip kernels (lane
// (q,i) of fragment (a, f, j, mt) = U[row(mt,i)][4j+q]); `transpose` swaps in/out channels (grad-input); `scale` / `shift`
// fold an output-side affine map (finc_conv.hip conv_pack_kernel).  fp64 arithmetic.
// -----------------------------------------------------------------------------------------------
__global__ void wino5_pack_kernel(const float *__restrict__ wc, const float *__restrict__ scale, const float *__restrict__ shift,
                                  float *__restrict__ packed, int Cq, int MT, int MTB, int NK, int transpose)
{
    const int g = blockIdx.y;
    const float *wg = wc + (size_t)g * Cq * Cq * 25;
    const int nfrag = 5 * 6 * NK * MT, npack = nfrag + 4 * MT;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 4 * MT * 64; e += gridDim.x * blockDim.x) {
        const int lane = e & 63, f = e >> 6;
        const int q = lane >> 4, mt = f >> 2, r = f & 3;
        const int row = mt < MTB ? 16 * mt + 4 * q + r : (q == 0 ? 16 * MTB + 4 * (mt - MTB) + r : Cq);
        packed[((size_t)g * npack + nfrag + f) * 64 + lane] = (shift && row < Cq) ? shift[g * Cq + row] : 0.f;
    }
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nfrag * 64; e += gridDim.x * blockDim.x) {
        const int lane = e & 63, fr = e >> 6;
        const int q = lane >> 4, i = lane & 15;
        const int mt = fr % MT, j = (fr / MT) % NK, f = (fr / (MT * NK)) % 6, a = fr / (MT * NK * 6);
        const int row = finc_tile_row(MTB, mt, i), col = 4 * j + q;
        double v = 0.0;
        if (row < Cq && col < Cq) {
            const int oc = transpose ? col : row, ic = transpose ? row : col;
            const float *w5 = wg + ((size_t)oc * Cq + ic) * 25 + (4 - a) * 5;     // g_k = w[o, i, KH-1-a, k]
            const double g0 = w5[0], g1 = w5[1], g2 = w5[2], g3 = w5[3], g4 = w5[4];
            v = f == 0 ? g0
              : f == 1 ? (g0 + g1 + g2 + g3 + g4) / 6.0
              : f == 2 ? (g0 - g1 + g2 - g3 + g4) / 6.0
              : f == 3 ? (16.0 * g0 + 8.0 * g1 + 4.0 * g2 + 2.0 * g3 + g4) / 12.0
              : f == 4 ? (16.0 * g0 - 8.0 * g1 + 4.0 * g2 - 2.0 * g3 + g4) / 12.0
              : g4 / 4.0;
            if (scale) v *= (double)scale[g * Cq + row];
        }
        packed[((size_t)g * npack + fr) * 64 + lane] = (float)v;
    }
}

typedef void (*wino5_fn)(const float *, const float *, float *, int, int, int, int, int, int, unsigned);
struct W5Inst {
    int cqp, nw, mt, mtb, nk, npack;
    size_t lds;
    wino5_fn fn;
};
template <int CQP, int NW>
constexpr W5Inst make_w5()
{
    using C = W5Cfg<CQP, NW>;
    return W5Inst{CQP, NW, C::MT, C::MTB, C::NK, C::NPACK, C::LDS_BYTES, finc_wino5_kernel<CQP, NW>};
}
// the 5x5 banks of the strip kernel's table (finc_conv.hip g_conv: same padded channel counts, so one rule says which bank a
// channel count runs on)
const W5Inst g_w5[] = {make_w5<4, 1>(), make_w5<8, 1>(), make_w5<12, 1>(), make_w5<16, 1>(), make_w5<24, 2>(), make_w5<32, 4>(), make_w5<48, 4>()};

const W5Inst *find_w5(int Cq)
{
    const W5Inst *best = nullptr;
    for (const W5Inst &i : g_w5)
        if (i.cqp >= Cq && (!best || i.cqp < best->cqp)) best = &i;
    if (best && best->nw == 1 && best->cqp - Cq > 3) return nullptr;      // (finc_conv.hip find_conv's rule)
    return best;
}

// FINC_NO_WINO5=1 keeps the 5x5 forward on the direct strip kernel (A/B timing, tests of that path)
bool no_wino5()
{
    static const bool off = [] { const char *e = finc_env("FINC_NO_WINO5"); return e && e[0] == '1'; }();
    return off || finc_wino_form_override() == 1;                         // (finc_debug_set_forward_form(1): the direct kernels everywhere)
}

} // namespace

size_t finc_wino5_packed_bytes(int G, int Cq, int KH, int KW)
{
    if (KH != 5 || KW != 5) return 0;
    const W5Inst *i = find_w5(Cq);
    return i ? (size_t)i->npack * 64 * sizeof(float) * (size_t)G : 0;
}

bool finc_wino5_takes(const float *in, const float *out, const FincShape &s)
{
    if (s.KH != 5 || s.KW != 5 || no_wino5() || !find_w5(s.Cq)) return false;
    if (s.W % 2 != 0 || s.W < 2) return false;                             // pairs
    if ((((uintptr_t)in) | ((uintptr_t)out)) & 7u) return false;           // 8-byte pieces
    return (size_t)s.Cq * s.H * s.W * 4 < ((size_t)1 << 30);
}

int finc_wino5_pack(const float *wc, void *packed, int G, int Cq, bool transpose, hipStream_t st, const float *scale, const float *shift)
{
    const W5Inst *i = find_w5(Cq);
    if (!i) return FINC_ERR_UNSUPPORTED;
    const int total = (i->npack - 4 * i->mt) * 64;
    int blocks = (total + 255) / 256;
    if (blocks > 128) blocks = 128;
    hipLaunchKernelGGL(wino5_pack_kernel, dim3(blocks, G), dim3(256), 0, st, wc, scale, shift, (float *)packed, Cq, i->mt, i->mtb, i->nk,
                       transpose ? 1 : 0);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

int finc_wino5_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st)
{
    const W5Inst *i = find_w5(s.Cq);
    if (!i) return FINC_ERR_UNSUPPORTED;
    const int NS = (s.W + 31) / 32;
    // one workgroup per CU at a time (its waves hold 1/NW of the bank each: one wave per SIMD); row chunks when the strips
    // alone do not fill the chip (every chunk recomputes 4 rows of operands)
    const long long wgs = (long long)s.B * s.G * NS;
    int nrc = finc_row_chunks(wgs, 256, s.H, 8, 4);                        // (rounds x rows per chunk: finc_common.h)
    const int RC = (s.H + nrc - 1) / nrc;
    nrc = (s.H + RC - 1) / RC;
    if (int e = finc_ensure_dynamic_lds((const void *)i->fn, i->lds)) return e;
    hipLaunchKernelGGL(i->fn, dim3(s.B * s.G * NS, nrc), dim3(64 * i->nw), i->lds, st, in, (const float *)packed, out, s.G, s.Cq, s.H, s.W, NS,
                       RC, s.orient);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

unsigned finc_build_flags_wino5() { return FINC_BUILD_FLAGS; }
