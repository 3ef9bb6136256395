// 3x3 forward (and, with transposed fragments, grad-input) with FEWER MULTIPLIES: Winograd along W (gfx950 only) -- F(2,3),
// described first, and F(4,3) (finc_wino4_kernel, further down), which a call runs whenever its strips of 64 columns fill the chip.
//
// The strip kernel of finc_conv.hip runs at ~78 % of the fp32 MFMA peak and the shape is compute-bound: the only
// way to a faster forward is fewer multiplies.  The forward has no recurrence, so any exact reformulation of the masked
// convolution (layers/conv.py:102-107: F.pad on one corner + cross-correlation) is allowed.  Along W, two neighbouring outputs
// of one row share their inputs:
//
//     d[m] = x[i, h-a, wt-2+m], m = 0..3        the four columns two outputs (wt, wt+1) of row tap a read
//     V = (d0-d2, d1+d2, d2-d1, d1-d3)          input transform   (4 add/sub per channel)
//     U = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)  filter transform  (g_k = w[o, i, KH-1-a, k]; once per weight version, fp64)
//     M_f = sum_a sum_i U_{a,f}[o,i] * V_{a,f}[i]                 4 "frequencies" x 3 row taps instead of 9 taps x 2 outputs
//     y(wt) = M0+M1+M2,  y(wt+1) = M1-M2-M3     output transform  (4 add/sub per channel)
//
// 12 multiplies per pair of outputs and (o,i) instead of 18: 1.5x fewer MFMAs for the same result (exact in exact arithmetic;
// in fp32 the error stays at the 1e-6 level of the direct sum -- the transforms add and halve, nothing is amplified).  A 2-D
// F(2x2,3x3) would save 2.25x but needs 16 frequencies x 12 accumulator registers per lane: it does not fit a wave beside
// its operands (profiles/r03/notes/winograd.md).
//
// Mapping: one wavefront owns a strip of 16 column PAIRS (32 columns) of one (image, group) slab and walks it top to bottom, one
// row per step; lane (q,p) = pair p, k-slot q.  A row arrives as ONE dwordx4 per lane and k-step -- the lane's own four
// columns wt-2 .. wt+1 (neighbouring lanes overlap by a pair: the second fetch is an L1 hit; no halo loads, no DPP) -- and
// leaves as one dwordx2 per output register (16 lanes = 128 contiguous bytes).  The transformed rows h, h-1, h-2 stay in
// registers (3 rotating slots); the bank -- 3 x 4 x NK x MT fragments -- is register-resident; no LDS.
#include "finc_common.h"
#include "finc_tile.h"

#include <stdlib.h>

#include <atomic>

#include <type_traits>
#include <utility>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr unsigned OFF_INVALID = 0x80000000u;
constexpr unsigned OFF_BAD_CHANNEL = 0x40000000u;

template <int I>
using IC = std::integral_constant<int, I>;
#define FINC_SB() __builtin_amdgcn_sched_barrier(0)
#define FINC_SB4() do { if constexpr (!(FINC_WINO_ABLATE & 256)) __builtin_amdgcn_sched_barrier(0); } while (0)   // (256: F(4,3) without its scheduling barriers)
#ifndef FINC_WINO_ABLATE   // timing-only bits (results wrong): 1 no loads, 2 no stores, 4 no MFMAs of the row taps 1 and 2, 8 no LDS slots
                           // (F(4,3): 8 no LDS reads, 16 no output transform, 32 no input transform, 64 no LDS writes)
#define FINC_WINO_ABLATE 0
#endif

template <int CQP, int MO = 2>                                    // MO outputs per tile: F(2,3) or F(4,3)
struct WCfg {
    static constexpr int MTB = CQP / 16, NSM = (CQP % 16) / 4, MT = MTB + NSM, NK = CQP / 4;
    static constexpr int NA = 3, NF = MO + 2;                     // row taps, frequencies
    static constexpr int NFRAG = NA * NF * NK * MT;               // fragment (a, f, j, mt) at ((a*NF + f)*NK + j)*MT + mt
    static constexpr int NPACK = NFRAG + 4 * MT;                  // + the output shift in accumulator layout (finc_conv.hip)
};

// The walk of one strip, specialised on the two wave-uniform facts that would otherwise cost selects in every step: FW (the
// group is W-flipped: a window arrives mirrored) and EDGE (strip 0: its first lane sits at the image's left edge, where the
// left half of the window is the zero padding of layers/conv.py:41-55).
template <int CQP, bool FW, bool EDGE>
__device__ __forceinline__ void wino_walk(const __amdgpu_buffer_rsrc_t rin, const __amdgpu_buffer_rsrc_t rout,
                                          const float *__restrict__ packed, float *__restrict__ vlds, int g, int CQ, int H, int W,
                                          int strip, int RC, bool fh)
{
    using C = WCfg<CQP>;
    constexpr int MT = C::MT, MTB = C::MTB, NSM = C::NSM, NK = C::NK, NF = C::NF, NFRAG = C::NFRAG;
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, p = lane & 15;
    const int HW = H * W;

    // ---- the bank.  16-row-tile fragments: one accumulation register each; 4-row-block fragments: four to a register,
    // selected by the MFMA's ABID (finc_tile.h) -- 72 + 36 registers at Cq = 24, which leaves room for two waves per SIMD.
    constexpr int NGRP = NFRAG / MT;                      // (a, f, j) triples
    constexpr int NSMALL = NGRP * NSM, NSR = (NSMALL + 3) / 4;
    float af[NGRP * (MTB > 0 ? MTB : 1)];
    float afs[NSR > 0 ? NSR : 1];
    {
        const float *pk = packed + (size_t)g * C::NPACK * 64 + lane;
#pragma unroll
        for (int t = 0; t < NGRP; ++t)
#pragma unroll
            for (int mt = 0; mt < MTB; ++mt) af[t * MTB + mt] = pk[(t * MT + mt) * 64];
        const int quad = (lane & 15) >> 2;
#pragma unroll
        for (int r = 0; r < NSR; ++r) {
            int gi = 0;
#pragma unroll
            for (int a = 3; a >= 0; --a) {
                constexpr int NSMD = NSM > 0 ? NSM : 1;
                const int sfr = 4 * r + a < NSMALL ? 4 * r + a : NSMALL - 1;
                const int ga = (sfr / NSMD) * MT + MTB + sfr % NSMD;
                gi = (a == 3 || quad == a) ? ga : gi;
            }
            afs[r] = pk[gi * 64];
        }
#pragma unroll
        for (int t = 0; t < NGRP * MTB; ++t) asm volatile("" : "+a"(af[t]));
#pragma unroll
        for (int r = 0; r < NSR; ++r) asm volatile("" : "+a"(afs[r]));
    }
    auto mma = [&](v4f &acc_, int t, int mt, float b) {   // t = (a*NF + f)*NK + j
        if (mt < MTB) acc_ = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t * MTB + mt], b, acc_, 0, 0, 0);
        else {
            const int sfr = t * NSM + (mt - MTB);
            finc_mma_small(acc_, afs[sfr >> 2], b, sfr & 3);
        }
    };
    // the output shift (accumulator layout, finc_conv.hip) waits in LDS: 4*MT registers less, three ds_read_b128 per step
    float *const blds = vlds + 2 * NF * NK * 64;
    {
        const float *pb = packed + ((size_t)g * C::NPACK + NFRAG) * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            *reinterpret_cast<v4f *>(blds + (mt * 64 + lane) * 4) =
                (v4f){pb[(4 * mt + 0) * 64], pb[(4 * mt + 1) * 64], pb[(4 * mt + 2) * 64], pb[(4 * mt + 3) * 64]};
    }

    // ---- addressing (finc_conv.hip's scheme): offset = row part (scalar; OFF_INVALID for a row off the image) + lane part
    // (columns + the lane row's share of the channel; OFF_BAD_CHANNEL for a pair off the image or a padded channel) + the
    // uniform share of the channel in the instruction's scalar offset.  W is even: a pair never straddles the edge.
    const int wt = strip * 32 + 2 * p;                    // canonical columns wt, wt+1 of this lane's pair
    const bool colok = wt < W;
    const unsigned coff = (unsigned)(FW ? W - 2 - wt : wt) * 4u;
    const unsigned qoff = (unsigned)q * HW * 4u;
    const bool lastok = 4 * (NK - 1) + q < CQ;            // (one-wave banks: padded channels sit in the last group of four)
    const unsigned lin0 = colok ? coff + qoff : OFF_BAD_CHANNEL, lin1 = (colok && lastok) ? coff + qoff : OFF_BAD_CHANNEL;
    unsigned lo_tile[4];                                  // 16-row tile: channel 16mt + 4q + r (masked for the last tile)
#pragma unroll
    for (int r = 0; r < 4; ++r) lo_tile[r] = (colok && 16 * (MTB - 1) + 4 * q + r < CQ) ? coff + 4u * qoff : OFF_BAD_CHANNEL;
    const unsigned lo_base = colok ? coff + 4u * qoff : OFF_BAD_CHANNEL;
    // the input window: canonical columns wt-2 .. wt+1 = 16 contiguous bytes, mirrored in memory for a W-flipped group.  The
    // lane at the image's left edge (EDGE, wt == 0) loads the columns 0 .. 3 instead and takes its pair from the other half
    const bool edge = EDGE && wt == 0;
    const int w0 = edge ? 0 : wt - 2;                     // first canonical column of the window this lane loads
    const unsigned woff = (unsigned)(FW ? W - 4 - w0 : w0) * 4u;
    const bool winok = colok && w0 + 3 < W;               // (W >= 4)
    const unsigned lw0 = winok ? woff + qoff : OFF_BAD_CHANNEL, lw1 = (winok && lastok) ? woff + qoff : OFF_BAD_CHANNEL;
    auto rowoff = [&](int h) { return (h >= 0 && h < H) ? (unsigned)((fh ? H - 1 - h : h) * W) * 4u : OFF_INVALID; };

    // ---- state: the transformed row h in registers; rows h-1, h-2 in LDS (two slots, slot = row & 1; layout
    // [slot][f][j][lane]: a lane reads back only what it wrote, no barrier) -- that is what makes two waves per SIMD fit.  A
    // row enters its slot at the END of its step, when the row it replaces (h-2) has been read for the last time.
    float Vc[NF][NK];
    for (int i = lane; i < 2 * NF * NK * 64; i += 64) vlds[i] = 0.f;
    v4u nx[NK];                                           // raw window of the next row (memory order)
    auto issue = [&](int h) {
        const unsigned ro = rowoff(h);
        if constexpr (FINC_WINO_ABLATE & 1) {
#pragma unroll
            for (int j = 0; j < NK; ++j) nx[j] = (v4u){ro, ro + 1u, ro + 2u, (unsigned)j};
            return;
        }
#pragma unroll
        for (int j = 0; j < NK; ++j)
            nx[j] = __builtin_amdgcn_raw_buffer_load_b128(rin, ro + (j == NK - 1 ? lw1 : lw0), 4 * j * HW * 4, 0);
    };
    auto transform = [&](auto slot_c) {
        constexpr int S = decltype(slot_c)::value;
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            // canonical order of the window = memory (x, y, z, w), or (w, z, y, x) when the group is W-flipped
            const float m0 = __builtin_bit_cast(float, FW ? nx[j].w : nx[j].x), m1 = __builtin_bit_cast(float, FW ? nx[j].z : nx[j].y);
            const float m2 = __builtin_bit_cast(float, FW ? nx[j].y : nx[j].z), m3 = __builtin_bit_cast(float, FW ? nx[j].x : nx[j].w);
            float d0 = m0, d1 = m1, d2 = m2, d3 = m3;
            if constexpr (EDGE) {                          // the edge lane loaded columns 0 .. 3: zero padding | its pair
                d0 = edge ? 0.f : m0; d1 = edge ? 0.f : m1; d2 = edge ? m0 : m2; d3 = edge ? m1 : m3;
            }
            Vc[0][j] = d0 - d2;
            Vc[1][j] = d1 + d2;
            Vc[2][j] = d2 - d1;
            Vc[3][j] = d1 - d3;
        }
        (void)S;
    };
    auto keep = [&](auto par_c) {                                         // Vc -> the slot of this row's parity
        constexpr int PAR = decltype(par_c)::value;
#pragma unroll
        for (int j = 0; j < NK; ++j)
#pragma unroll
            for (int f = 0; f < NF; ++f) vlds[((PAR * NF + f) * NK + j) * 64 + lane] = Vc[f][j];
    };
    const int r0 = blockIdx.y * RC, r1 = r0 + RC < H ? r0 + RC : H;       // output rows of this chunk
    auto step = [&](auto slot_c, int h) {                                 // row h sits in nx; S = h's parity
        constexpr int S = decltype(slot_c)::value;
        transform(slot_c);
        FINC_SB();
        issue(h + 1);                                                     // (lands during this step's MFMAs)
        FINC_SB();
        if (h < r0) { keep(slot_c); return; }                             // (filling the slots of a chunk: no output row)
        // (frequency 1 enters both outputs with weight +1: its accumulators start from the shift, which costs nothing)
        v4f acc[NF][MT];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[f][mt] = f == 1 ? *reinterpret_cast<const v4f *>(blds + (mt * 64 + lane) * 4) : (v4f){0.f, 0.f, 0.f, 0.f};
        // row tap a = 0 from registers; a = 1, 2 from LDS, one k-step (4 frequencies) ahead of its MFMAs
        float vb[2][NF];
        auto fetch = [&](int t, float (&dst)[NF]) {                        // t = (a-1)*NK + j, a = 1, 2
            const int a = 1 + t / NK, j = t % NK, slot = (S + a) & 1;      // (row h-a has the parity of h+a)
#pragma unroll
            for (int f = 0; f < NF; ++f) dst[f] = vlds[((slot * NF + f) * NK + j) * 64 + lane];
        };
        fetch(0, vb[0]);
#pragma unroll
        for (int j = 0; j < NK; ++j) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) mma(acc[f][mt], (0 * NF + f) * NK + j, mt, Vc[f][j]);
        }
        FINC_SB();
#pragma unroll
        for (int t = 0; t < ((FINC_WINO_ABLATE & 4) ? 0 : 2 * NK); ++t) {
            if (t + 1 < 2 * NK) fetch(t + 1, vb[(t + 1) & 1]);
            const int a = 1 + t / NK, j = t % NK;
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) mma(acc[f][mt], (a * NF + f) * NK + j, mt, vb[t & 1][f]);
            FINC_SB();
        }
        keep(slot_c);                                                     // (row h-2 has been read for the last time)
        // output transform (the shift rode in with frequency 1) and stores: one pair per output register and lane -- 16 lanes
        // write 128 contiguous bytes.  (Collecting the row in an LDS tile and storing whole 16-byte pieces -- CQP/8 dwordx4
        // stores instead of 4*MTB + NSM dwordx2 ones -- was measured 5 % SLOWER: profiles/r03/notes/winograd.md.)
        const unsigned ro = (FINC_WINO_ABLATE & 2) ? (h == 12345 ? 0u : OFF_INVALID) : rowoff(h);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const v4f y0 = acc[0][mt] + acc[1][mt] + acc[2][mt];
            const v4f y1 = acc[1][mt] - acc[2][mt] - acc[3][mt];
            if (mt < MTB) {
                const float a0[4] = {y0.x, y0.y, y0.z, y0.w}, a1[4] = {y1.x, y1.y, y1.z, y1.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v2u v;
                    v.x = __builtin_bit_cast(unsigned, FW ? a1[r] : a0[r]);
                    v.y = __builtin_bit_cast(unsigned, FW ? a0[r] : a1[r]);
                    const unsigned lo = (NSM == 0 && mt == MTB - 1) ? lo_tile[r] : lo_base;
                    __builtin_amdgcn_raw_buffer_store_b64(v, rout, ro + lo, (16 * mt + r) * HW * 4, 0);
                }
            } else {
                const float s0 = finc_block_reduce(y0), s1 = finc_block_reduce(y1);
                v2u v;
                v.x = __builtin_bit_cast(unsigned, FW ? s1 : s0);
                v.y = __builtin_bit_cast(unsigned, FW ? s0 : s1);
                const int sb = mt - MTB;
                __builtin_amdgcn_raw_buffer_store_b64(v, rout, ro + (sb == NSM - 1 ? lin1 : lin0), (16 * MTB + 4 * sb) * HW * 4, 0);
            }
        }
    };
    // rows r0-2, r0-1 fill the slots (rows above the image load zeros); then one output row per step, slots rotating
    const int hs = r0 - 2;
    issue(hs);
    // (rows are walked in pairs: the slot of a row is its parity relative to hs)
    for (int h = hs; h < ((FINC_WINO_ABLATE & 16) ? hs + 2 : r1); h += 2) {
        step(IC<0>{}, h);
        if (h + 1 < r1) step(IC<1>{}, h + 1);
    }
}

// grid = (B*G*NS strips of 32 columns, row chunks); one wavefront each; 2 * 4 * NK * 256 + MT * 1024 bytes of LDS
template <int CQP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void finc_wino_kernel(
    const float *__restrict__ in, const float *__restrict__ packed, float *__restrict__ out, int G, int CQ, int H, int W, int NS, int RC,
    unsigned orient, int skew_mask, int skew_sleep)
{
    extern __shared__ __attribute__((aligned(16))) float vlds[];
    // Two waves share a SIMD and run the same fixed-length steps: started together they stay in phase -- both in their MFMA
    // block, then both in their transforms and stores, the matrix core idle.  Half of the waves therefore start half a step
    // late (the workgroups whose index has a bit of skew_mask set; the dispatcher deals workgroups round-robin, so the two
    // tenants of a SIMD differ in a high bit), and one wave's VALU / VMEM work falls into the other's MFMA block.
    if ((int)blockIdx.x & skew_mask)
        for (int i = 0; i < skew_sleep; ++i) __builtin_amdgcn_s_sleep(16);
    const int strip = blockIdx.x % NS, bg = blockIdx.x / NS;
    const int g = bg % G;
    const unsigned o = finc_group_orient(orient, g);
    const bool fh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 4u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    if (strip == 0) {
        if (fw) wino_walk<CQP, true, true>(rin, rout, packed, vlds, g, CQ, H, W, strip, RC, fh);
        else wino_walk<CQP, false, true>(rin, rout, packed, vlds, g, CQ, H, W, strip, RC, fh);
    } else {
        if (fw) wino_walk<CQP, true, false>(rin, rout, packed, vlds, g, CQ, H, W, strip, RC, fh);
        else wino_walk<CQP, false, false>(rin, rout, packed, vlds, g, CQ, H, W, strip, RC, fh);
    }
}

// -----------------------------------------------------------------------------------------------
// F(4,3) along W: FOUR neighbouring outputs of a row share six inputs -- 6 multiplies per (o, i, row tap) instead of 12 (2x
// fewer MFMAs than the direct sum, 1.33x fewer than F(2,3)).  Interpolation points 0, +-1, +-3/2, infinity -- the textbook
// set (Lavin & Gray) has +-2; +-3/2 costs the same operations, every constant is still exact in fp32, and the error is
// 1.27x smaller (numpy model of this arithmetic: profiles/r03/notes/winograd.md):
//
//     d[m] = x[i, h-a, wt-2+m], m = 0..5        the six columns the outputs wt .. wt+3 of row tap a read
//     V = B^T d   = (2.25d0-3.25d2+d4, (d4-2.25d2)+-(d3-2.25d1), (d4-d2)+-1.5(d3-d1), 2.25d1-3.25d3+d5)
//     U = G g     = (g0/2.25, -(g0+-g1+g2)/2.5, (g0+-1.5g1+2.25g2)/5.625, g2)                           (pack kernel, fp64)
//     y = A^T M   : y0 = M0+(M1+M2)+(M3+M4), y1 = (M1-M2)+1.5(M3-M4), y2 = (M1+M2)+2.25(M3+M4), y3 = (M1-M2)+3.375(M3-M4)+M5
//
// The constants cost accuracy: 1.1e-6 of the largest output against 3e-7 for F(2,3) and 5e-7 for the direct fp32 sum at the c3
// bank (numpy model of this exact arithmetic; the GPU tests hold the kernel to 1e-5).  Mapping: one wavefront owns 16 column
// QUADS = 64 columns of one (image, group) slab -- the whole width at 64x64 --, lane (q,p) = quad p, k-slot q; one wave per SIMD
// (162 bank registers + 72 accumulators at Cq = 24).  A row arrives as one dwordx4 (the lane's own quad) + one dwordx2 (the pair
// left of it: the neighbour's bytes, an L1 hit; the zero padding for the first quad of a row) per lane and k-step and leaves
// as one dwordx4 per output register (16 lanes = 256 contiguous bytes).  Rows h-1, h-2 wait transformed in LDS as before.
// -----------------------------------------------------------------------------------------------
template <int CQP, bool FW>
__device__ __forceinline__ void wino4_walk(const __amdgpu_buffer_rsrc_t rin, const __amdgpu_buffer_rsrc_t rout,
                                           const float *__restrict__ packed, float *__restrict__ vlds, int g, int CQ, int H, int W,
                                           int strip, int RC, bool fh)
{
    using C = WCfg<CQP, 4>;
    constexpr int MT = C::MT, MTB = C::MTB, NSM = C::NSM, NK = C::NK, NF = C::NF, NFRAG = C::NFRAG;
    constexpr int NB = CQP >= 20 ? 1 : 2;                 // rows of loads in flight (see below)
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, p = lane & 15;
    const int HW = H * W;

    // ---- the bank (as in wino_walk: 16-row-tile fragments one register each, 4-row-block fragments four to a register)
    constexpr int NGRP = NFRAG / MT;                      // (a, f, j) triples
    constexpr int NSMALL = NGRP * NSM, NSR = (NSMALL + 3) / 4;
    float af[NGRP * (MTB > 0 ? MTB : 1)];
    float afs[NSR > 0 ? NSR : 1];
    {
        const float *pk = packed + (size_t)g * C::NPACK * 64 + lane;
#pragma unroll
        for (int t = 0; t < NGRP; ++t)
#pragma unroll
            for (int mt = 0; mt < MTB; ++mt) af[t * MTB + mt] = pk[(t * MT + mt) * 64];
        const int quad = (lane & 15) >> 2;
#pragma unroll
        for (int r = 0; r < NSR; ++r) {
            int gi = 0;
#pragma unroll
            for (int a = 3; a >= 0; --a) {
                constexpr int NSMD = NSM > 0 ? NSM : 1;
                const int sfr = 4 * r + a < NSMALL ? 4 * r + a : NSMALL - 1;
                const int ga = (sfr / NSMD) * MT + MTB + sfr % NSMD;
                gi = (a == 3 || quad == a) ? ga : gi;
            }
            afs[r] = pk[gi * 64];
        }
#pragma unroll
        for (int t = 0; t < NGRP * MTB; ++t) asm volatile("" : "+a"(af[t]));
#pragma unroll
        for (int r = 0; r < NSR; ++r) asm volatile("" : "+a"(afs[r]));
    }
    auto mma = [&](v4f &acc_, int t, int mt, float b) {   // t = (a*NF + f)*NK + j
        if (mt < MTB) acc_ = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t * MTB + mt], b, acc_, 0, 0, 0);
        else {
            const int sfr = t * NSM + (mt - MTB);
            finc_mma_small(acc_, afs[sfr >> 2], b, sfr & 3);
        }
    };
    // LDS of this wave: frequencies 0..3 of the rows h-1, h-2 as 16-byte cells [slot][j][lane], frequencies 4, 5 as 8-byte
    // cells behind them (a lane reads back only what it wrote: no barrier), then the output shift in accumulator layout
    v4f *const vA = reinterpret_cast<v4f *>(vlds);
    v2f *const vB = reinterpret_cast<v2f *>(vlds + 2 * NK * 64 * 4);
    float *const blds = vlds + 2 * NK * 64 * 6;
    {
        const float *pb = packed + ((size_t)g * C::NPACK + NFRAG) * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            *reinterpret_cast<v4f *>(blds + (mt * 64 + lane) * 4) =
                (v4f){pb[(4 * mt + 0) * 64], pb[(4 * mt + 1) * 64], pb[(4 * mt + 2) * 64], pb[(4 * mt + 3) * 64]};
    }

    // ---- addressing (wino_walk's scheme).  W % 4 == 0: a quad never straddles the right edge
    const int wt = strip * 64 + 4 * p;                    // canonical columns wt .. wt+3 of this lane's quad
    const bool colok = wt < W, leftok = colok && wt > 0;  // (left of the first quad: the zero padding of layers/conv.py:41-55)
    const unsigned coff = (unsigned)(FW ? W - 4 - wt : wt) * 4u;
    const unsigned loff = (unsigned)(FW ? W - wt : wt - 2) * 4u;          // the pair wt-2, wt-1 (mirrored: wt-1 first)
    const unsigned qoff = (unsigned)q * HW * 4u;
    const bool lastok = 4 * (NK - 1) + q < CQ;
    const unsigned lin0 = colok ? coff + qoff : OFF_BAD_CHANNEL, lin1 = (colok && lastok) ? coff + qoff : OFF_BAD_CHANNEL;
    const unsigned ll0 = leftok ? loff + qoff : OFF_BAD_CHANNEL, ll1 = (leftok && lastok) ? loff + qoff : OFF_BAD_CHANNEL;
    unsigned lo_tile[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) lo_tile[r] = (colok && 16 * (MTB - 1) + 4 * q + r < CQ) ? coff + 4u * qoff : OFF_BAD_CHANNEL;
    const unsigned lo_base = colok ? coff + 4u * qoff : OFF_BAD_CHANNEL;
    auto rowoff = [&](int h) { return (h >= 0 && h < H) ? (unsigned)((fh ? H - 1 - h : h) * W) * 4u : OFF_INVALID; };

    float Vc[NF][NK];
    for (int i = lane; i < 2 * NK * 64 * 6; i += 64) vlds[i] = 0.f;
    // The small banks keep TWO rows of loads in flight (buffer = row parity; a row's loads are issued two steps before its
    // transform: their steps are shorter than the load latency of a chip whose 1,024 waves all fetch at once -- Cq = 12: -5 %,
    // Cq = 16: -8 %).  The banks of 20 and 24 channels keep one: their step (3.5 us at c3) covers the latency, and the second
    // buffer's 36 registers pushed the allocation to its limit (42 register-to-register copies per step).
    v4u nx[NB][NK];                                       // rows h+1 (, h+2): this lane's quad ...
    v2u nl[NB][NK];                                        // ... and the pair left of it (memory order)
    auto issue = [&](auto buf_c, int h) {
        constexpr int BUF = decltype(buf_c)::value % NB;
        const unsigned ro = rowoff(h);
        if constexpr (FINC_WINO_ABLATE & 1) {
#pragma unroll
            for (int j = 0; j < NK; ++j) { nx[BUF][j] = (v4u){ro, ro + 1u, ro + 2u, (unsigned)j}; nl[BUF][j] = (v2u){ro + 3u, ro + 5u}; }
            return;
        }
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            nx[BUF][j] = __builtin_amdgcn_raw_buffer_load_b128(rin, ro + (j == NK - 1 ? lin1 : lin0), 4 * j * HW * 4, 0);
            nl[BUF][j] = __builtin_amdgcn_raw_buffer_load_b64(rin, ro + (j == NK - 1 ? ll1 : ll0), 4 * j * HW * 4, 0);
        }
    };
    auto transform = [&](auto buf_c) {
        constexpr int BUF = decltype(buf_c)::value % NB;
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            // canonical order = memory order, or mirrored when the group is W-flipped
            const v4u m = nx[BUF][j];
            const v2u l = nl[BUF][j];
            const float d0 = __builtin_bit_cast(float, FW ? l.y : l.x), d1 = __builtin_bit_cast(float, FW ? l.x : l.y);
            const float d2 = __builtin_bit_cast(float, FW ? m.w : m.x), d3 = __builtin_bit_cast(float, FW ? m.z : m.y);
            const float d4 = __builtin_bit_cast(float, FW ? m.y : m.z), d5 = __builtin_bit_cast(float, FW ? m.x : m.w);
            if constexpr (FINC_WINO_ABLATE & 32) {
                Vc[0][j] = d0; Vc[1][j] = d1; Vc[2][j] = d2; Vc[3][j] = d3; Vc[4][j] = d4; Vc[5][j] = d5;
                continue;
            }
            const float t1 = __builtin_fmaf(-2.25f, d2, d4), t2 = __builtin_fmaf(-2.25f, d1, d3);
            const float t3 = d4 - d2, t4 = d3 - d1;
            Vc[0][j] = __builtin_fmaf(2.25f, d0, __builtin_fmaf(-3.25f, d2, d4));
            Vc[1][j] = t1 + t2;
            Vc[2][j] = t1 - t2;
            Vc[3][j] = __builtin_fmaf(1.5f, t4, t3);
            Vc[4][j] = __builtin_fmaf(-1.5f, t4, t3);
            Vc[5][j] = __builtin_fmaf(2.25f, d1, __builtin_fmaf(-3.25f, d3, d5));
        }
    };
    auto keep = [&](auto par_c) {                                         // Vc -> the slot of this row's parity
        constexpr int PAR = decltype(par_c)::value;
        if constexpr (FINC_WINO_ABLATE & 64) return;
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            vA[(PAR * NK + j) * 64 + lane] = (v4f){Vc[0][j], Vc[1][j], Vc[2][j], Vc[3][j]};
            vB[(PAR * NK + j) * 64 + lane] = (v2f){Vc[4][j], Vc[5][j]};
        }
    };
    // A 16-byte store whose channel offset rides in the instruction's SCALAR offset lost its second dword in the lanes 12..15
    // of every row when the next instruction overwrote the data registers (seen on gfx950 with a v_pk_add_f32 right behind
    // the store: the compiler's hazard table only covers the immediate-offset form).  The channel offset therefore goes into
    // the vector offset (one v_add per store), the form whose wait state the compiler inserts.
    auto store16 = [&](const v4u &v, unsigned voff, int choff) {
        __builtin_amdgcn_raw_buffer_store_b128(v, rout, voff + (unsigned)choff, 0, 0);
    };
    const int r0 = blockIdx.y * RC, r1 = r0 + RC < H ? r0 + RC : H;       // output rows of this chunk
    auto step = [&](auto slot_c, int h) {                                 // row h sits in nx / nl; S = h's parity
        constexpr int S = decltype(slot_c)::value;
        transform(slot_c);
        FINC_SB4();
        issue(slot_c, h + NB);                                            // (into the buffer this row just left)
        FINC_SB4();
        // (frequency 1 enters all four outputs with weight +1: its accumulators start from the shift)
        v4f acc[NF][MT];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[f][mt] = f == 1 ? *reinterpret_cast<const v4f *>(blds + (mt * 64 + lane) * 4) : (v4f){0.f, 0.f, 0.f, 0.f};
        float vb[2][NF];
        auto fetch = [&](int t, float (&dst)[NF]) {                        // t = (a-1)*NK + j, a = 1, 2
            const int a = 1 + t / NK, j = t % NK, slot = (S + a) & 1;      // (row h-a has the parity of h+a)
            if constexpr (FINC_WINO_ABLATE & 8) {
#pragma unroll
                for (int f = 0; f < NF; ++f) dst[f] = Vc[f][j];
                return;
            }
            const v4f lo = vA[(slot * NK + j) * 64 + lane];
            const v2f hi = vB[(slot * NK + j) * 64 + lane];
            dst[0] = lo.x; dst[1] = lo.y; dst[2] = lo.z; dst[3] = lo.w; dst[4] = hi.x; dst[5] = hi.y;
        };
        fetch(0, vb[0]);
#pragma unroll
        for (int j = 0; j < NK; ++j) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) mma(acc[f][mt], (0 * NF + f) * NK + j, mt, Vc[f][j]);
        }
        FINC_SB4();
#pragma unroll
        for (int t = 0; t < ((FINC_WINO_ABLATE & 4) ? 0 : 2 * NK); ++t) {
            if (t + 1 < 2 * NK) fetch(t + 1, vb[(t + 1) & 1]);
            const int a = 1 + t / NK, j = t % NK;
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) mma(acc[f][mt], (a * NF + f) * NK + j, mt, vb[t & 1][f]);
            FINC_SB4();
        }
        keep(slot_c);                                                     // (row h-2 has been read for the last time)
        // output transform and stores: one quad per output register and lane -- 16 lanes write 256 contiguous bytes
        const unsigned ro = (FINC_WINO_ABLATE & 2) ? (h == 12345 ? 0u : OFF_INVALID) : rowoff(h);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const v4f s1 = acc[1][mt] + acc[2][mt], e1 = acc[1][mt] - acc[2][mt];
            const v4f s2 = acc[3][mt] + acc[4][mt], e2 = acc[3][mt] - acc[4][mt];
            const v4f y0 = (FINC_WINO_ABLATE & 16) ? acc[0][mt] + acc[4][mt] : acc[0][mt] + s1 + s2;
            const v4f y1 = (FINC_WINO_ABLATE & 16) ? acc[1][mt] : e1 + 1.5f * e2;
            const v4f y2 = (FINC_WINO_ABLATE & 16) ? acc[2][mt] : s1 + 2.25f * s2;
            const v4f y3 = (FINC_WINO_ABLATE & 16) ? acc[3][mt] + acc[5][mt] : e1 + 3.375f * e2 + acc[5][mt];
            if (mt < MTB) {
                const float a0[4] = {y0.x, y0.y, y0.z, y0.w}, a1[4] = {y1.x, y1.y, y1.z, y1.w};
                const float a2[4] = {y2.x, y2.y, y2.z, y2.w}, a3[4] = {y3.x, y3.y, y3.z, y3.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v4u v;
                    v.x = __builtin_bit_cast(unsigned, FW ? a3[r] : a0[r]);
                    v.y = __builtin_bit_cast(unsigned, FW ? a2[r] : a1[r]);
                    v.z = __builtin_bit_cast(unsigned, FW ? a1[r] : a2[r]);
                    v.w = __builtin_bit_cast(unsigned, FW ? a0[r] : a3[r]);
                    const unsigned lo = (NSM == 0 && mt == MTB - 1) ? lo_tile[r] : lo_base;
                    store16(v, ro + lo, (16 * mt + r) * HW * 4);
                }
            } else {
                const float s0 = finc_block_reduce(y0), sa = finc_block_reduce(y1), sb2 = finc_block_reduce(y2), sc = finc_block_reduce(y3);
                v4u v;
                v.x = __builtin_bit_cast(unsigned, FW ? sc : s0);
                v.y = __builtin_bit_cast(unsigned, FW ? sb2 : sa);
                v.z = __builtin_bit_cast(unsigned, FW ? sa : sb2);
                v.w = __builtin_bit_cast(unsigned, FW ? s0 : sc);
                const int sb = mt - MTB;
                store16(v, ro + (sb == NSM - 1 ? lin1 : lin0), (16 * MTB + 4 * sb) * HW * 4);
            }
        }
    };
    // Rows r0-2, r0-1 fill the slots (no output row).  The walk below is shaped for the compiler's wait-count pass: a step
    // waits for ITS row's loads with the stores of the row before still in flight (vmcnt = the number of those stores) only
    // if every path into the step has issued the same loads and stores in the same order -- with the fill steps inside the
    // loop it fell back to vmcnt(0) and every step waited for the previous row's stores to reach memory (47 of 245 us at c3).
    // So: fill steps apart, the first output row peeled, then pairs (slot parities 1, 0), then the odd row left over.
    auto fill = [&](auto slot_c, int h) {
        transform(slot_c);
        FINC_SB4();
        issue(slot_c, h + NB);
        FINC_SB4();
        keep(slot_c);
    };
    const int hs = r0 - 2;
    issue(IC<0>{}, hs);
    if constexpr (NB == 2) issue(IC<1>{}, hs + 1);
    fill(IC<0>{}, hs);
    fill(IC<1>{}, hs + 1);
    int h = r0;
    step(IC<0>{}, h);
    ++h;
    for (; h + 1 < r1; h += 2) {
        step(IC<1>{}, h);
        step(IC<0>{}, h + 1);
    }
    if (h < r1) step(IC<1>{}, h);
}

// grid = (B*G*NS strips of 64 columns, row chunks); one wavefront each; 2 * NK * 64 * 24 + MT * 1024 bytes of LDS
template <int CQP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void finc_wino4_kernel(
    const float *__restrict__ in, const float *__restrict__ packed, float *__restrict__ out, int G, int CQ, int H, int W, int NS, int RC,
    unsigned orient, int, int)
{
    extern __shared__ __attribute__((aligned(16))) float vlds[];
    const int strip = blockIdx.x % NS, bg = blockIdx.x / NS;
    const int g = bg % G;
    const unsigned o = finc_group_orient(orient, g);
    const bool fh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 4u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    if (fw) wino4_walk<CQP, true>(rin, rout, packed, vlds, g, CQ, H, W, strip, RC, fh);
    else wino4_walk<CQP, false>(rin, rout, packed, vlds, g, CQ, H, W, strip, RC, fh);
}

// the same walk with TWO waves per SIMD for the banks of 4 and 8 channels (93 / 192 registers): their steps are short and
// memory-bound, and a second tenant fills the gaps -- C = 16 at 64x64, B = 256: 30.7 -> 24.9 us (0.55 -> 0.67 of the HBM peak),
// C = 32: 65.7 -> 58.2; at 16 channels per group (244 registers, MFMA-bound) it gains nothing (139 -> 143 us), at 12 it spills.
// grid as above; 2 * NK * 64 * 24 + MT * 1024 bytes of LDS
template <int CQP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void finc_wino4t_kernel(
    const float *__restrict__ in, const float *__restrict__ packed, float *__restrict__ out, int G, int CQ, int H, int W, int NS, int RC,
    unsigned orient, int, int)
{
    extern __shared__ __attribute__((aligned(16))) float vlds[];
    const int strip = blockIdx.x % NS, bg = blockIdx.x / NS;
    const int g = bg % G;
    const unsigned o = finc_group_orient(orient, g);
    const bool fh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 4u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    if (fw) wino4_walk<CQP, true>(rin, rout, packed, vlds, g, CQ, H, W, strip, RC, fh);
    else wino4_walk<CQP, false>(rin, rout, packed, vlds, g, CQ, H, W, strip, RC, fh);
}

// -----------------------------------------------------------------------------------------------
// Bank: U_{a,f} = filter transform of row a of the (canonical) 3x3 kernel, in the fragment layout of the strip kernels
// (lane (q,i) of fragment (a, f, j, mt) = U[row(mt,i)][4j+q]); `transpose` swaps in/out channels (grad-input); `scale` /
// `shift` fold an output-side affine map (finc_conv.hip conv_pack_kernel).  fp64 arithmetic.
// -----------------------------------------------------------------------------------------------
__global__ void wino_pack_kernel(const float *__restrict__ wc, const float *__restrict__ scale, const float *__restrict__ shift,
                                 float *__restrict__ packed, int Cq, int MT, int MTB, int NK, int NF, int transpose)
{
    const int g = blockIdx.y;
    const float *wg = wc + (size_t)g * Cq * Cq * 9;
    const int nfrag = 3 * NF * NK * MT, npack = nfrag + 4 * MT;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 4 * MT * 64; e += gridDim.x * blockDim.x) {
        const int lane = e & 63, f = e >> 6;
        const int q = lane >> 4, mt = f >> 2, r = f & 3;
        const int row = mt < MTB ? 16 * mt + 4 * q + r : (q == 0 ? 16 * MTB + 4 * (mt - MTB) + r : Cq);
        packed[((size_t)g * npack + nfrag + f) * 64 + lane] = (shift && row < Cq) ? shift[g * Cq + row] : 0.f;
    }
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nfrag * 64; e += gridDim.x * blockDim.x) {
        const int lane = e & 63, fr = e >> 6;
        const int q = lane >> 4, i = lane & 15;
        const int mt = fr % MT, j = (fr / MT) % NK, f = (fr / (MT * NK)) % NF, a = fr / (MT * NK * NF);
        const int row = finc_tile_row(MTB, mt, i), col = 4 * j + q;
        double v = 0.0;
        if (row < Cq && col < Cq) {
            const int oc = transpose ? col : row, ic = transpose ? row : col;
            const float *w3 = wg + ((size_t)oc * Cq + ic) * 9 + (2 - a) * 3;      // g_k = w[o, i, KH-1-a, k]
            const double g0 = w3[0], g1 = w3[1], g2 = w3[2];
            if (NF == 4) v = f == 0 ? g0 : f == 1 ? 0.5 * (g0 + g1 + g2) : f == 2 ? 0.5 * (g0 - g1 + g2) : g2;
            else
                v = f == 0 ? g0 / 2.25 : f == 1 ? -(g0 + g1 + g2) / 2.5 : f == 2 ? -(g0 - g1 + g2) / 2.5
                  : f == 3 ? (g0 + 1.5 * g1 + 2.25 * g2) / 5.625 : f == 4 ? (g0 - 1.5 * g1 + 2.25 * g2) / 5.625 : g2;
            if (scale) v *= (double)scale[g * Cq + row];
        }
        packed[((size_t)g * npack + fr) * 64 + lane] = (float)v;
    }
}

typedef void (*wino_fn)(const float *, const float *, float *, int, int, int, int, int, int, unsigned, int, int);
template <int CQP>
constexpr wino_fn finc_wino4_pick()
{
    if constexpr (CQP <= 8) return finc_wino4t_kernel<CQP>;
    else return finc_wino4_kernel<CQP>;
}
struct WInst {
    int cqp, mt, mtb, nk, npack, npack4;          // npack: the F(2,3) bank, npack4: the F(4,3) bank behind it (two waves per SIMD for cqp <= 8)
    wino_fn fn, fn4;
};
template <int CQP>
constexpr WInst make_winst()
{
    return WInst{CQP, WCfg<CQP>::MT, WCfg<CQP>::MTB, WCfg<CQP>::NK, WCfg<CQP>::NPACK, WCfg<CQP, 4>::NPACK, finc_wino_kernel<CQP>, finc_wino4_pick<CQP>()};
}

// banks whose fragments (12 * NK * MT for F(2,3), two waves per SIMD; 18 * NK * MT for F(4,3), one wave per SIMD) fit the
// accumulation registers
const WInst g_winsts[] = {make_winst<4>(), make_winst<8>(), make_winst<12>(), make_winst<16>(), make_winst<20>(), make_winst<24>()};

const WInst *find_winst(int Cq)
{
    const int cqp = (Cq + 3) / 4 * 4;
    for (const WInst &i : g_winsts)
        if (i.cqp == cqp) return &i;
    return nullptr;
}

// FINC_NO_WINO=1 keeps the forward on the direct strip kernel (A/B timing, tests of that path); FINC_WINO_FORM=2|4 pins the
// Winograd form; finc_debug_set_forward_form() does either at run time (tests of each form on one shape, in one process)
std::atomic<int> g_form_override{0};               // 0: none, 1: direct strip kernel, 2: F(2,3), 4: F(4,3)
bool finc_no_wino()
{
    static const bool off = [] { const char *e = finc_env("FINC_NO_WINO"); return e && e[0] == '1'; }();
    const int o = g_form_override.load(std::memory_order_relaxed);
    return o ? o == 1 : off;
}
int finc_wino_forced_form()
{
    static const int f = [] { const char *e = finc_env("FINC_WINO_FORM"); return e ? atoi(e) : 0; }();
    const int o = g_form_override.load(std::memory_order_relaxed);
    if (o == 2 || o == 4) return o;
    return f == 2 || f == 4 ? f : 0;
}

} // namespace

size_t finc_wino_packed_bytes(int G, int Cq, int KH, int KW)
{
    if (KH != 3 || KW != 3) return 0;
    const WInst *i = find_winst(Cq);
    return i ? (size_t)G * (i->npack + i->npack4) * 64 * sizeof(float) : 0;
}

bool finc_wino_takes(const float *in, const float *out, const FincShape &s)
{
    if (s.KH != 3 || s.KW != 3 || s.W % 4 != 0 || s.W < 4 || finc_no_wino() || !find_winst(s.Cq)) return false;
    if ((size_t)s.Cq * s.H * s.W * 4 >= ((size_t)1 << 30)) return false;
    return ((((uintptr_t)in) | ((uintptr_t)out)) & 15u) == 0;             // a row arrives as 16-byte windows
}

// Which form a call runs: F(4,3) has 1.33x fewer MFMAs but runs one wave per SIMD over strips of 64 columns, so it wants
// (a) its lanes filled -- at least three quarters of the columns its strips cover are image -- and (b) a wave for every SIMD
// out of row chunks of 8 rows or more (a chunk recomputes two rows of operands).  Measured (profiles/r03/notes/winograd.md):
// Cq = 24, 64x64: B = 32 (128 strips x 8 chunks): 40 vs 46 us, B = 16: 35 vs 27 us the other way; 128x128, B = 8: 41 vs 45 us.
int finc_wino_form(const FincShape &s)
{
    if (const int f = finc_wino_forced_form()) return f;
    const int NS4 = (s.W + 63) / 64;
    const long long waves4 = (long long)s.B * s.G * NS4 * (s.H >= 16 ? s.H / 8 : 1);
    return (4 * s.W >= 3 * NS4 * 64 && waves4 >= 1024) ? 4 : 2;
}

int finc_wino_pack(const float *wc, void *packed, int G, int Cq, bool transpose, hipStream_t st, const float *scale, const float *shift)
{
    const WInst *i = find_winst(Cq);
    if (!i) return FINC_ERR_UNSUPPORTED;
    for (int NF = 4; NF <= 6; NF += 2) {
        const int total = 3 * NF * i->nk * i->mt * 64;
        int blocks = (total + 255) / 256;
        if (blocks > 64) blocks = 64;
        float *dst = (float *)packed + (NF == 4 ? 0 : (size_t)G * i->npack * 64);
        hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks, G), dim3(256), 0, st, wc, scale, shift, dst, Cq, i->mt, i->mtb, i->nk, NF,
                           transpose ? 1 : 0);
        FINC_CHECK_LAUNCH();
    }
    return FINC_OK;
}

int finc_wino_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st)
{
    const WInst *i = find_winst(s.Cq);
    if (!i) return FINC_ERR_UNSUPPORTED;
    const bool f4 = finc_wino_form(s) == 4;
    const int NS = f4 ? (s.W + 63) / 64 : (s.W + 31) / 32;
    const long long waves = (long long)s.B * s.G * NS;
    // F(2,3): two waves per SIMD, row chunks up to about one; F(4,3): one wave per SIMD, row chunks (of 8 rows or more: a
    // chunk recomputes two rows of operands) up to that
    static const int force_chunks = finc_env("FINC_WINO_CHUNKS") ? atoi(finc_env("FINC_WINO_CHUNKS")) : 0;   // experiment switch
    int nrc;
    if (f4) {
        const long long fill4 = i->cqp <= 8 ? 2048 : 1024;                 // (the small banks: two waves per SIMD)
        nrc = finc_row_chunks(waves, fill4, s.H, 8, 2);                    // (rounds x rows per chunk: finc_common.h)
    } else {
        nrc = finc_row_chunks(waves, 1024, s.H, 4, 2, 14);                 // (two waves per SIMD: finc_common.h)
    }
    if (force_chunks > 0) nrc = force_chunks;
    if (nrc > s.H / 4) nrc = s.H / 4 > 0 ? s.H / 4 : 1;
    const int RC = (s.H + nrc - 1) / nrc;
    nrc = (s.H + RC - 1) / RC;
    static const int skew_mask = finc_env("FINC_WINO_SKEW") ? atoi(finc_env("FINC_WINO_SKEW")) : 0;          // (experiment switches)
    static const int skew_sleep = finc_env("FINC_WINO_SLEEP") ? atoi(finc_env("FINC_WINO_SLEEP")) : 2;
    const float *bank = (const float *)packed + (f4 ? (size_t)s.G * i->npack * 64 : 0);
    const size_t lds = f4 ? (size_t)2 * i->nk * 64 * 24 + (size_t)i->mt * 1024 : (size_t)2 * 4 * i->nk * 256 + (size_t)i->mt * 1024;
    hipLaunchKernelGGL(f4 ? i->fn4 : i->fn, dim3(s.B * s.G * NS, nrc), dim3(64), lds, st, in, bank, out, s.G, s.Cq, s.H, s.W, NS, RC,
                       s.orient, skew_mask, skew_sleep);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

int finc_wino_form_override() { return g_form_override.load(std::memory_order_relaxed); }

// (for finc_wino4m.hip: the same switches and the same bank, M-split)
bool finc_wino_disabled() { return finc_no_wino(); }
int finc_wino_pack_bank(const float *wc, float *packed, int G, int Cq, int MT, int MTB, int NK, int NF, bool transpose, hipStream_t st,
                        const float *scale, const float *shift)
{
    const int total = 3 * NF * NK * MT * 64;
    int blocks = (total + 255) / 256;
    if (blocks > 128) blocks = 128;
    hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks, G), dim3(256), 0, st, wc, scale, shift, packed, Cq, MT, MTB, NK, NF, transpose ? 1 : 0);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

int finc_wino_set_form(int form)
{
    if (form != 0 && form != 1 && form != 2 && form != 4) return FINC_ERR_BAD_DIMS;
    g_form_override.store(form, std::memory_order_relaxed);
    return FINC_OK;
}

unsigned finc_build_flags_wino() { return FINC_BUILD_FLAGS; }
