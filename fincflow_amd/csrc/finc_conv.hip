// Forward masked convolution (and, with transposed fragments, its input gradient) for gfx950.
//
// The forward has no recurrence, so it does not use the skewed wavefront of finc_mfma.hip.  One wavefront
// owns a STRIP of 16 canonical columns of one (image, group) slab and walks it top to bottom, one row per step:
//
//   * lane (q,p): column w0+p, k-slot q.  The 16 lanes of a row q read 16 consecutive floats of one channel row
//     = one 64-byte sector per load and per store: HBM traffic is coalesced by construction, no LDS at all.
//   * per step: z[Cq x 16px] = sum_{a,b} W_ab * x[(h-a, w-b)]: per tap and k-step one v_mfma_f32_16x16x4_f32 per full
//     16-channel tile plus one v_mfma_f32_4x4x1_16B_f32 per remaining 4-channel block (finc_tile.h), all independent
//     across rows (one accumulator set per unrolled sub-step).
//   * column shifts b come from DPP row_shr:b, the b columns left of the strip from a tiny masked "halo" load;
//     row shifts a are the operands of the previous rows, kept in registers (the row slot rotates with the
//     loop, which is unrolled by KH, so ageing a row costs no instruction).
//   * filter fragments live in AGPRs for the whole kernel (same packing as the wavefront kernel: fragment
//     (tap, j, mt), lane (q,i) = W[16mt+i][4j+q]).
//
// grad_input of this conv is the same operator on the H- and W-flipped image with in/out channels transposed
// (DESIGN.md 3.2), so finc_backward_f32 calls this kernel with `transpose` fragments and orient ^ 3 per group.
// Replaces F.pad + cuDNN conv (layers/conv.py:102-107) x4 + chunk/cat (fastflow.py:31-50).
#include "finc_common.h"
#include "finc_tile.h"

#include <type_traits>
#include <utility>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr unsigned OFF_INVALID = 0x80000000u;
constexpr unsigned OFF_BAD_CHANNEL = 0x40000000u;

template <int I>
using IC = std::integral_constant<int, I>;

template <int N>
__device__ inline float row_shr(float old, float src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                 __builtin_bit_cast(int, src), 0x110 + N, 0xf, 0xf,
                                                                 false));
}
template <int N>
__device__ inline float row_shl(float src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), 0x100 + N, 0xf, 0xf,
                                                                 true));
}

// NW > 1 ("K-split"): the workgroup has NW waves; wave w owns the input-channel k-steps [w*JL, (w+1)*JL) of every
// tap -- 1/NW of the fragments and of the input rows -- and the partial output tiles are summed through LDS once per
// row.  That is how a filter bank that does not fit one wave's registers (Cq=48, 5x5: 900 fragments) still runs with
// every fragment register-resident.
// fragments one wave of the strip kernel keeps in registers
template <int CQP, int KH, int KW, int NW>
constexpr int conv_nfrag() { return KH * KW * (CQP / 4 / NW) * (CQP / 16 + (CQP % 16) / 4); }
// registers the bank occupies: 16-row-tile fragments one each, 4-row-block fragments four to a register (finc_tile.h)
template <int CQP, int KH, int KW, int NW>
constexpr int conv_nreg()
{
    return KH * KW * (CQP / 4 / NW) * (CQP / 16) + (KH * KW * (CQP / 4 / NW) * ((CQP % 16) / 4) + 3) / 4;
}

// Two waves per SIMD (256 registers each) is what hides one wave's loads, stores and VALU behind the other's MFMAs:
// single-wave workgroups ask for that register budget when the bank leaves room for the working set (a 252-fragment
// bank squeezed into 256 registers spills: 10x slower; the 111 registers of <28,3,3> beside the dword form's working set
// spill 27: 517 vs 269 us at B=128, profiles/r02/notes/ab34); the K-split variants keep theirs (their waves are many).
#ifndef FINC_CONV_2W_MAX   // largest bank (registers) that still asks for two waves per SIMD
#define FINC_CONV_2W_MAX 104
#endif
template <int CQP, int KH, int KW, int NW, bool WIDE>
__global__ __launch_bounds__(64 * NW)
    __attribute__((amdgpu_waves_per_eu(NW == 1 && conv_nreg<CQP, KH, KW, NW>() <= FINC_CONV_2W_MAX ? 2 : 1))) void finc_conv_kernel(const float *__restrict__ in,
                                                            const float *__restrict__ packed, float *__restrict__ out,
                                                            int G, int CQ, int H, int W, int NS, int RC,
                                                            unsigned orient)
{
    constexpr int MTB = CQP / 16, NSM = (CQP % 16) / 4, MT = MTB + NSM, NKZT = CQP / 4, NTAP = KH * KW;
    constexpr int NOUT = 4 * MTB + NSM;                   // output registers of a row: 4 per 16-row tile, 1 per reduced block
    static_assert(NW == 1 || (NKZT % NW == 0 && NOUT % NW == 0), "K-split must divide the k-steps and the output registers");
    constexpr int NKZ = NKZT / NW;                        // k-steps this wave owns
    constexpr int NFRAG = NTAP * NKZ * MT;                // fragments this wave holds
    constexpr int DREG = (4 * (CQP / 16) + (CQP % 16) / 4) / NW; // output registers this wave finalises and stores
    __shared__ float xch[NW > 1 ? 2 * NW * NW * DREG * 64 : 1]; // [parity][dst wave][src wave][reg][lane]
    // Output staging (WIDE).  Straight from the accumulators a store instruction covers 4 channel rows x 64 bytes, 24 of them
    // per row at c3.  Through LDS the row leaves as [channel][16 pixels]: a lane writes its values with ds_write_b32 (cheap
    // beside MFMAs), reads back 16 bytes of one channel, and ceil(Cq/16) buffer_store_dwordx4 -- 16 whole 64-byte sectors
    // each -- go out at the start of the next step, when the read-back has long arrived.
    constexpr int OPITCH = 20;                            // floats per channel row in the staging buffer (16 + pad: bank spread)
    constexpr int NOI = (CQP + 15) / 16;                  // 16-channel store instructions per row
    __shared__ __attribute__((aligned(16))) float ostg[WIDE ? NOI * 16 * OPITCH : 4];
    // Input staging (WIDE): the row arrives as whole 16-byte pieces -- lane = (channel, piece), ceil(5*Cq/64) dwordx4 loads
    // instead of 2*Cq/4 dword loads -- and is turned into MFMA B operands by ds_read_b32: [channel][4 halo + 16 columns]
    // in MEMORY order, so a column shift is a read address (no DPP) and the columns left of the strip are the last piece
    // of the neighbouring sector, one more lane group of the same loads (no halo loads).
    constexpr int IPITCH = 24;                            // floats per channel row (48 and 80 -- conflict-free for all 64 lanes -- time the same: ab27)
    constexpr int NII = (5 * CQP + 63) / 64;              // dwordx4 load instructions per row
    __shared__ __attribute__((aligned(16))) float istg[WIDE ? CQP * IPITCH + 4 : 4];
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u ostv[NOI];                                        // WIDE: the result row read back, waiting for the next step
    unsigned ost_row = 0;                                 // ... its (scalar) row offset
    bool ost_ok = false;                                  // ... and whether that row exists in this chunk (else its stores are dropped)
    const int wv = NW > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, p = lane & 15;
    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs, each with its own L2, and the strips of a slab
    // share the sectors at their seams (the two halo columns left of a strip are the last columns of its neighbour's
    // sector).  So the NS strips of a slab are placed 8 workgroups apart -- on ONE XCD -- and the seam sector is an L2 hit
    // for whichever strip asks second (plain order: FETCH_SIZE 1.25x the image).  Speed only: nothing depends on it.
    const int BGN = (int)gridDim.x / NS;                  // slabs in this launch
    const int full = (BGN / 8) * 8 * NS;                  // workgroups in whole tiles of 8 slabs
    int strip, bg;
    if ((int)blockIdx.x < full) {
        const int tile = blockIdx.x / (8 * NS), within = blockIdx.x % (8 * NS);
        strip = within / 8;
        bg = tile * 8 + within % 8;
    } else {
        const int idx = blockIdx.x - full;
        strip = idx % NS;
        bg = (BGN / 8) * 8 + idx / NS;
    }
    const int g = bg % G;
    const unsigned o = finc_group_orient(orient, g);
    const bool fh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 4u;
    const __amdgpu_buffer_rsrc_t rin =
        __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout =
        __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);

    // 16-row-tile fragments: one register each; 4-row-block fragments: four to a register, selected by the MFMA's ABID
    // (finc_tile.h).  The packed bank in memory keeps one 64-lane fragment per (tap, k-step, tile): a lane of a packed
    // register reads the fragment its pixel quad stands for.
    constexpr int NSMALL = (NFRAG / MT) * NSM, NSR = (NSMALL + 3) / 4;
    float af[NFRAG];
    float afs[NSR > 0 ? NSR : 1];
    {
        // global fragment index ((tap*NKZT + j)*MT + mt); this wave's j = wv*NKZ + jl
        const float *pk = packed + (size_t)g * (NTAP * NKZT * MT + 4 * MT) * 64 + lane;
        auto gindex = [&](int f) {
            const int mt = f % MT, jl = (f / MT) % NKZ, tap = f / (MT * NKZ);
            return (tap * NKZT + wv * NKZ + jl) * MT + mt;
        };
#pragma unroll
        for (int f = 0; f < NFRAG; ++f) {
            if (f % MT >= MTB) continue;
            af[f] = pk[gindex(f) * 64];
        }
        const int quad = (lane & 15) >> 2;
#pragma unroll
        for (int r = 0; r < NSR; ++r) {
            int gi = 0;
#pragma unroll
            for (int a = 3; a >= 0; --a) {
                constexpr int NSMD = NSM > 0 ? NSM : 1;
                const int sfr = 4 * r + a < NSMALL ? 4 * r + a : NSMALL - 1;
                const int ga = gindex((sfr / NSMD) * MT + MTB + sfr % NSMD);
                gi = (a == 3 || quad == a) ? ga : gi;
            }
            afs[r] = pk[gi * 64];
        }
        // MFMA A operands: keep them out of the VGPRs.  With the 256-register budget of two waves per SIMD the files are
        // split 128 : 128, so a bank larger than that pins what fits and leaves the rest to the allocator.
        constexpr int NBIG = (NFRAG / MT) * MTB;
        constexpr int NPIN = ((NW == 1 || NW >= 8) && NBIG + NSR > 124 && NBIG + NSR <= 176) ? 124 - NSR : NBIG;   // (NW >= 8: two waves per SIMD by the size of the workgroup)
        int pinned = 0;
#pragma unroll
        for (int f = 0; f < NFRAG; ++f) {
            if (f % MT >= MTB) continue;
            if (pinned++ < NPIN) asm volatile("" : "+a"(af[f]));
        }
#pragma unroll
        for (int r = 0; r < NSR; ++r) asm volatile("" : "+a"(afs[r]));
    }
    // Output-side affine map folded into the bank (z' = scale * conv(x) + shift: the ActNorm that follows the unit in the
    // model, layers/actnorm.py:39-46): the rows of the filters carry the scale, and a row's accumulators start from its
    // shift -- 4*MT registers in accumulator layout behind the fragments (zeros for a plain bank; a K-split adds them once)
    v4f bias[MT];
    {
        const float *pb = packed + ((size_t)g * (NTAP * NKZT * MT + 4 * MT) + NTAP * NKZT * MT) * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            bias[mt] = (v4f){pb[(4 * mt + 0) * 64], pb[(4 * mt + 1) * 64], pb[(4 * mt + 2) * 64], pb[(4 * mt + 3) * 64]};
            if (NW > 1 && wv != 0) bias[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
        }
    }
    auto mma = [&](v4f &acc_, int f, float b) {
        const int mt = f % MT;
        if (mt < MTB) acc_ = __builtin_amdgcn_mfma_f32_16x16x4f32(af[f], b, acc_, 0, 0, 0);
        else {
            const int sfr = (f / MT) * NSM + (mt - MTB);
            finc_mma_small(acc_, afs[sfr >> 2], b, sfr & 3);
        }
    };

    if constexpr (WIDE) {
        // ------------------------------------------------------------------------------------------------------------
        // W % 16 == 0, one wave per strip.  What the MFMA pipe of a SIMD loses to this kernel is, above all, its VMEM
        // instructions -- ~25 cycles each whatever their width (scripts/micro/fwd_path.hip: the dword scheme below costs
        // 3,380 cycles per row step on a saturated SIMD, this one 3,000, the MFMAs alone 2,290) -- so the row moves in as
        // few as possible: ceil(5 Cq / 64) dwordx4 loads and ceil(Cq / 16) dwordx4 stores, both transposed through LDS.
        // Pipeline of step h (everything ahead of the MFMAs, nothing waits in front of them):
        //   stores of row h-2 | row h+1 (registers) -> input tile | loads of row h+2 | operand reads of row h+1 (land
        //   during this step's MFMAs) | result of row h-1 -> output tile -> registers | MFMAs of row h.
        // ------------------------------------------------------------------------------------------------------------
        static_assert(NW == 1 && KW <= 5, "one wave per strip; the halo is one 16-byte piece");
        constexpr int U = KH + 1;                         // row slots: slot (S+1)%U fills while slots S..S-KH+1 are read
        const float *const in_slab = in + (size_t)bg * CQ * HW;
        float *const out_slab = out + (size_t)bg * CQ * HW;
        // a row's buffer resource: the slab, or EMPTY when the row does not exist (loads give 0, stores are dropped);
        // row validity is wave-uniform: one scalar select, and the row's byte offset rides in the scalar offset
        auto rsrc_in = [&](bool ok) { return __builtin_amdgcn_make_buffer_rsrc((void *)in_slab, 0, ok ? (int)slab_bytes : 0, 0x00020000); };
        auto rsrc_out = [&](bool ok) { return __builtin_amdgcn_make_buffer_rsrc((void *)out_slab, 0, ok ? (int)slab_bytes : 0, 0x00020000); };
        auto rowbytes = [&](int h) { return (unsigned)((fh ? H - 1 - h : h) * W) * 4u; };
        const int r0 = blockIdx.y * RC, r1 = r0 + RC < H ? r0 + RC : H;
        const int ms = fw ? W - 16 - strip * 16 : strip * 16;       // memory column where the strip's sector starts
        const int hm = fw ? ms + 16 : ms - 4;                       // ... and the piece holding the KW-1 columns left of it
        unsigned lvo[NII];                                          // load slot 64i+lane: byte offset in the slab
        int lwr[NII];                                               // ... and where its 16 bytes go in the tile (floats)
#pragma unroll
        for (int i = 0; i < NII; ++i) {
            const int t = 64 * i + lane;
            lvo[i] = OFF_BAD_CHANNEL;
            lwr[i] = CQP * IPITCH;                                  // scratch piece behind the tile
            if (t < 4 * CQP) {
                const int c = t >> 2, k = t & 3;
                if (c < CQ) lvo[i] = (unsigned)c * HW * 4u + (unsigned)(ms + 4 * k) * 4u;
                lwr[i] = c * IPITCH + (fw ? 0 : 4) + 4 * k;
            } else if (t < 5 * CQP) {
                const int c = t - 4 * CQP;
                if (c < CQ && hm >= 0 && hm < W) lvo[i] = (unsigned)c * HW * 4u + (unsigned)hm * 4u;
                lwr[i] = c * IPITCH + (fw ? 16 : 0);
            }
        }
        int ird[KW];                                                // lane (q,p), shift b: canonical column p-b of channel q
#pragma unroll
        for (int b = 0; b < KW; ++b) ird[b] = q * IPITCH + (fw ? 15 - (p - b) : 4 + (p - b));
        const int pp = fw ? 15 - p : p;                             // memory order inside the strip's sector
        const unsigned ost_col = (unsigned)(ms + 4 * (lane & 3)) * 4u;
        const unsigned ost_lane = ost_col + (unsigned)(lane >> 2) * HW * 4u;
        const unsigned ost_lane_last = (16 * (NOI - 1) + (lane >> 2) < CQ) ? ost_lane : OFF_BAD_CHANNEL;
        v4u L[NII];
#pragma unroll
        for (int i = 0; i < NII; ++i) L[i] = (v4u){0u, 0u, 0u, 0u};
        float X[U][KW][NKZ];
#pragma unroll
        for (int sl = 0; sl < U; ++sl)
#pragma unroll
            for (int b = 0; b < KW; ++b)
#pragma unroll
                for (int j = 0; j < NKZ; ++j) X[sl][b][j] = 0.f;
        v4f prev[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) prev[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
        auto issue = [&](int h) {
            const bool ok = h >= 0 && h < H;
            const __amdgpu_buffer_rsrc_t r = rsrc_in(ok);
            const unsigned ro = ok ? rowbytes(h) : 0u;
#pragma unroll
            for (int i = 0; i < NII; ++i) L[i] = __builtin_amdgcn_raw_buffer_load_b128(r, lvo[i], ro, 0);
        };
        auto flush_staged = [&]() {
            const __amdgpu_buffer_rsrc_t r = rsrc_out(ost_ok);
#pragma unroll
            for (int i = 0; i < NOI; ++i)
                __builtin_amdgcn_raw_buffer_store_b128(ostv[i], r, i == NOI - 1 ? ost_lane_last : ost_lane,
                                                       (unsigned)(16 * i) * HW * 4u + ost_row, 0);
            ost_ok = false;
        };
        auto step = [&](auto s_c, int h) {
            constexpr int S = decltype(s_c)::value, SN = (S + 1) % U;
            flush_staged();
#pragma unroll
            for (int i = 0; i < NII; ++i) reinterpret_cast<v4u *>(istg)[lwr[i] >> 2] = L[i];   // (index in pieces: ds_write_b128)
            issue(h + 2);
#pragma unroll
            for (int b = 0; b < KW; ++b)
#pragma unroll
                for (int j = 0; j < NKZ; ++j) X[SN][b][j] = istg[ird[b] + 4 * j * IPITCH];
            {                                                       // the result of row h-1: accumulators -> [channel][16] -> pieces
#pragma unroll
                for (int mt = 0; mt < MTB; ++mt) {
                    const float v[4] = {prev[mt].x, prev[mt].y, prev[mt].z, prev[mt].w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) ostg[(16 * mt + 4 * q + r) * OPITCH + pp] = v[r];
                }
#pragma unroll
                for (int sb = 0; sb < NSM; ++sb) ostg[(16 * MTB + 4 * sb + q) * OPITCH + pp] = finc_block_reduce(prev[MTB + sb]);
#pragma unroll
                for (int i = 0; i < NOI; ++i)
                    ostv[i] = *reinterpret_cast<const v4u *>(&ostg[(16 * i + (lane >> 2)) * OPITCH + 4 * (lane & 3)]);
                const int hp = h - 1;
                ost_ok = hp >= r0 && hp < r1;
                ost_row = ost_ok ? rowbytes(hp) : 0u;
            }
            __builtin_amdgcn_sched_barrier(0);                      // all of the above stays ahead of the MFMAs ...
            // rows outside [r0, r1) are walked only to fill the operand slots (and to drain the pipeline): no MFMAs for them
            if (h >= r0 && h < r1) {
                v4f ac[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) ac[mt] = bias[mt];
#pragma unroll
                for (int a = KH - 1; a >= 0; --a)
#pragma unroll
                    for (int b = 0; b < KW; ++b)
#pragma unroll
                        for (int j = 0; j < NKZ; ++j)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                mma(ac[mt], ((a * KW + b) * NKZ + j) * MT + mt, X[(S + U - a) % U][b][j]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) prev[mt] = ac[mt];
            }
            __builtin_amdgcn_sched_barrier(0);                      // ... and the next step's memory work behind them
        };
        // Row hs itself is never loaded (its operands are the zeros above): it must not reach a stored row, so the walk
        // starts KH rows early; the rows before r0 only fill the slots (a slot is a position in the unrolled group, so any
        // start row will do).
        const int hs = r0 - KH;
        issue(hs + 1);
        for (int h0 = hs; h0 < r1 + 1; h0 += U) {
            [&]<int... I>(std::integer_sequence<int, I...>) { (step(IC<I>{}, h0 + I), ...); }
            (std::make_integer_sequence<int, U>{});
        }
        flush_staged();                                             // the last row, if no step followed it
        return;
    }

    // Addressing is branch-free and select-free: a buffer offset = (row part, scalar) + (lane part, constant).
    // An off-image row contributes OFF_INVALID (2^31), an off-image column or padded channel OFF_BAD_CHANNEL
    // (2^30); any such sum lands beyond the slab (< 2^30 bytes), where loads return 0 and stores are dropped.
    const int col = strip * 16 + p;
    const int hcol = strip * 16 - (KW - 1) + p;           // lanes p < KW-1 also hold the columns left of the strip
    const bool colok = col < W;
    const bool hok = p < KW - 1 && hcol >= 0;
    const unsigned coloff = (unsigned)(fw ? W - 1 - col : col) * 4u;
    const unsigned hcoloff = (unsigned)(fw ? W - 1 - hcol : hcol) * 4u;
    // Lane part of an offset: column + k-slot/lane-row channel q; the uniform channel part (4j, 16mt+r, ...) rides in the
    // instruction's scalar offset.  Channels >= CQ exist only in the LAST group of four, so only the last k-step / the
    // last output group needs its own (masked) lane part.  (Validity must sit in the VGPR offset: the scalar offset is
    // not part of the buffer range check.)
    const unsigned qoff = (unsigned)q * HW * 4u;
    const bool lastok = 4 * (wv * NKZ + NKZ - 1) + q < CQ;
    const unsigned lin0 = colok ? coloff + qoff : OFF_BAD_CHANNEL;
    const unsigned lin1 = (colok && lastok) ? coloff + qoff : OFF_BAD_CHANNEL;
    const unsigned lhal0 = hok ? hcoloff + qoff : OFF_BAD_CHANNEL;
    const unsigned lhal1 = (hok && lastok) ? hcoloff + qoff : OFF_BAD_CHANNEL;
    unsigned linj[NW > 1 ? NKZ : 1], lhalj[NW > 1 ? NKZ : 1];   // (K-split) lane offsets per k-step: OFF_BAD_CHANNEL where channel 4j + q >= CQ
    if constexpr (NW > 1) {
#pragma unroll
        for (int j = 0; j < NKZ; ++j) {
            const bool chok = 4 * (wv * NKZ + j) + q < CQ;
            linj[j] = (colok && chok) ? coloff + qoff : OFF_BAD_CHANNEL;
            lhalj[j] = (hok && chok) ? hcoloff + qoff : OFF_BAD_CHANNEL;
        }
    }
    const unsigned lo_base = colok ? coloff + 4u * qoff : OFF_BAD_CHANNEL;   // 16-row tile: channel 16mt + 4q + r
    unsigned lo_tail[4];                                  // last 16-row tile when it ends the group (Cq % 16 == 0 shapes)
#pragma unroll
    for (int r = 0; r < 4; ++r)
        lo_tail[r] = (colok && 16 * (MTB - 1) + 4 * q + r < CQ) ? coloff + 4u * qoff : OFF_BAD_CHANNEL;
    unsigned lout[NW > 1 ? 4 * MTB + NSM : 1];            // K-split store path: per-register lane offsets
    if constexpr (NW > 1) {
#pragma unroll
        for (int d = 0; d < 4 * MTB + NSM; ++d) {
            const int c = d < 4 * MTB ? 16 * (d >> 2) + 4 * q + (d & 3) : 16 * MTB + 4 * (d - 4 * MTB) + q;
            lout[d] = (colok && c < CQ) ? coloff + (unsigned)c * HW * 4u : OFF_BAD_CHANNEL;
        }
    }
    // staged stores: lane (c16, k) = channel 16i + c16, 16-byte piece k of the strip's sector (memory order)
    auto rowoff = [&](int h) {                            // scalar
        return (h >= 0 && h < H) ? (unsigned)((fh ? H - 1 - h : h) * W) * 4u : OFF_INVALID;
    };
    // Row chunks (blockIdx.y): a small problem set has too few strips to fill the chip and each strip walk is a chain of
    // row-load latencies, so the host cuts the walk into chunks of RC rows.  A chunk starts KH-1 rows early (those
    // steps only fill the operand slots; their results are not stored) at a multiple of KH (the slot rotation).
    const int r0 = blockIdx.y * RC, r1 = r0 + RC < H ? r0 + RC : H;
    auto rowoff_st = [&](int h) {                         // scalar: rows this chunk owns
        return (h >= r0 && h < r1) ? (unsigned)((fh ? H - 1 - h : h) * W) * 4u : OFF_INVALID;
    };

    float X[KH][KW][NKZ];                                 // X[s][b]: row slot s, shifted b columns
#pragma unroll
    for (int s = 0; s < KH; ++s)
#pragma unroll
        for (int b = 0; b < KW; ++b)
#pragma unroll
            for (int j = 0; j < NKZ; ++j) X[s][b][j] = 0.f;
    float nxt[NKZ], nxh[NKZ];                             // the row loaded one step ahead (+ its halo)
#ifndef FINC_CONV_ABLATE   // timing-only experiment bits (results wrong): 1 = no loads, 2 = no stores
#define FINC_CONV_ABLATE 0
#endif
    auto issue = [&](int h) {
        if constexpr (FINC_CONV_ABLATE & 1) {
#pragma unroll
            for (int j = 0; j < NKZ; ++j) { nxt[j] = (float)(h + j) * 1e-3f; nxh[j] = nxt[j]; }
            return;
        }
        const unsigned ro = rowoff(h);
        const unsigned v0 = ro + lin0, v1 = ro + lin1, h0 = ro + lhal0, h1 = ro + lhal1;
#pragma unroll
        for (int j = 0; j < NKZ; ++j) {
            const int so = 4 * (wv * NKZ + j) * HW * 4;   // uniform: channel 4j of this wave's k-step j
            // K-split banks also serve channel counts well below CQP (Cq = 50 on the 64-channel bank): there ANY k-step of the
            // last waves may hold padded channels, so every k-step has its own lane offsets (one mark at most: no wrap past 2^32)
            const unsigned vj = NW > 1 ? ro + linj[j] : (j == NKZ - 1 ? v1 : v0);
            const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(rin, vj, so, 0);
            nxt[j] = __builtin_bit_cast(float, u);
            if constexpr (KW > 1) {
                const unsigned hj = NW > 1 ? ro + lhalj[j] : (j == NKZ - 1 ? h1 : h0);
                const unsigned uh = __builtin_amdgcn_raw_buffer_load_b32(rin, hj, so, 0);
                nxh[j] = __builtin_bit_cast(float, uh);
            }
        }
    };
    v4f acc[KH][MT];                                      // one accumulator set per unrolled sub-step
#pragma unroll
    for (int s = 0; s < KH; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[s][mt] = (v4f){0.f, 0.f, 0.f, 0.f};

    int parity = 0;
    auto store_row = [&](const v4f (&ac)[MT], int h) {   // h = the row those accumulators belong to
        if constexpr (FINC_CONV_ABLATE & 2) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(ac[mt]));
            return;
        }
        const unsigned ro = rowoff_st(h);
        if constexpr (NW == 1) {
            const unsigned vb = ro + lo_base;
#pragma unroll
            for (int mt = 0; mt < MTB; ++mt) {
                const float v[4] = {ac[mt].x, ac[mt].y, ac[mt].z, ac[mt].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned vo = (NSM == 0 && mt == MTB - 1) ? ro + lo_tail[r] : vb;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), rout, vo, (16 * mt + r) * HW * 4, 0);
                }
            }
            // a reduced 4-row block leaves channel 16*MTB + 4sb + q in lane row q: the lane part of an input k-step
#pragma unroll
            for (int sb = 0; sb < NSM; ++sb) {
                const float v = finc_block_reduce(ac[MTB + sb]);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, ro + (sb == NSM - 1 ? lin1 : lin0),
                                                      (16 * MTB + 4 * sb) * HW * 4, 0);
            }
        } else {
            // exchange: register d of the output belongs to wave d / DREG; everybody ships the registers it does not
            // own, one barrier, the owner adds the NW-1 partials it received and stores (double-buffered by parity)
            float vv[NOUT];                               // partial sums; a 4-row block is reduced first (linear)
#pragma unroll
            for (int mt = 0; mt < MTB; ++mt) {
                const float v0 = ac[mt].x, v1 = ac[mt].y, v2 = ac[mt].z, v3 = ac[mt].w;
                vv[4 * mt + 0] = v0; vv[4 * mt + 1] = v1; vv[4 * mt + 2] = v2; vv[4 * mt + 3] = v3;
            }
#pragma unroll
            for (int sb = 0; sb < NSM; ++sb) vv[4 * MTB + sb] = finc_block_reduce(ac[MTB + sb]);
            float *xb = xch + parity * (NW * NW * DREG * 64);
#pragma unroll
            for (int d = 0; d < NOUT; ++d) {
                const int dst = d / DREG;
                if (dst != wv) xb[((dst * NW + wv) * DREG + d % DREG) * 64 + lane] = vv[d];
            }
            __syncthreads();
#pragma unroll
            for (int dl = 0; dl < DREG; ++dl) {
                float sum = 0.f;
#pragma unroll
                for (int d = 0; d < NOUT; ++d)
                    if (d / DREG == wv && d % DREG == dl) sum = vv[d];       // own partial (wv is wave-uniform)
#pragma unroll
                for (int src = 0; src < NW; ++src)
                    if (src != wv) sum += xb[((wv * NW + src) * DREG + dl) * 64 + lane];
                unsigned off = OFF_BAD_CHANNEL;
#pragma unroll
                for (int d = 0; d < NOUT; ++d)
                    if (d / DREG == wv && d % DREG == dl) off = lout[d];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum), rout, ro + off, 0, 0);
            }
            parity ^= 1;
        }
    };

    auto step = [&](auto s_c, int h) {
        constexpr int S = decltype(s_c)::value;           // row slot of row h  (h % KH == S)
        // the row that arrived becomes column-shift 0 of slot S; its shifted copies follow
#pragma unroll
        for (int j = 0; j < NKZ; ++j) X[S][0][j] = nxt[j];
        if constexpr (KW > 1) {
            float hl[NKZ];
#pragma unroll
            for (int j = 0; j < NKZ; ++j) hl[j] = nxh[j];
            // lane p < b needs column w0+p-b = halo lane p-b+KW-1: shift the halo left by KW-1-b first
#pragma unroll
            for (int j = 0; j < NKZ; ++j) {
                if constexpr (KW > 1) X[S][KW - 1][j] = row_shr<KW - 1>(hl[j], nxt[j]);
                if constexpr (KW > 2) X[S][KW - 2][j] = row_shr<KW - 2>(row_shl<1>(hl[j]), nxt[j]);
                if constexpr (KW > 3) X[S][KW - 3][j] = row_shr<KW - 3>(row_shl<2>(hl[j]), nxt[j]);
                if constexpr (KW > 4) X[S][KW - 4][j] = row_shr<KW - 4>(row_shl<3>(hl[j]), nxt[j]);
                if constexpr (KW > 5) X[S][KW - 5][j] = row_shr<KW - 5>(row_shl<4>(hl[j]), nxt[j]);
                if constexpr (KW > 6) X[S][KW - 6][j] = row_shr<KW - 6>(row_shl<5>(hl[j]), nxt[j]);
            }
        }
        issue(h + 1);                                     // next row: a whole step of MFMAs to arrive
        // the previous row's result leaves while this row's MFMAs run
        store_row(acc[(S + KH - 1) % KH], h - 1);
        // rows outside [r0, r1) are walked only to fill the operand slots: no MFMAs for them (their results are never stored)
        if (h >= r0 && h < r1) {
            v4f ac[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) ac[mt] = bias[mt];
            // rows h-a live in slot (S - a) mod KH; older rows first (their operands are long ready)
#pragma unroll
            for (int a = KH - 1; a >= 0; --a)
#pragma unroll
                for (int b = 0; b < KW; ++b)
#pragma unroll
                    for (int j = 0; j < NKZ; ++j)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            mma(ac[mt], ((a * KW + b) * NKZ + j) * MT + mt, X[(S + KH - a) % KH][b][j]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[S][mt] = ac[mt];
        }
    };

    int hs = r0 - (KH - 1);
    hs = hs >= 0 ? hs / KH * KH : -((-hs + KH - 1) / KH * KH);   // floor to a multiple of KH
    issue(hs);
    for (int h0 = hs; h0 < r1 + 1; h0 += KH) {
        [&]<int... I>(std::integer_sequence<int, I...>) { (step(IC<I>{}, h0 + I), ...); }
        (std::make_integer_sequence<int, KH>{});
    }
}

// fragment (tap (a,b), j, mt), lane (q,i): W[row finc_tile_row(mt,i)][col 4j+q][KH-1-a][KW-1-b]; `transpose` swaps row/col
// `scale` / `shift` ([G*Cq] or nullptr): the per-output-channel affine map folded behind the conv (rows scaled; shift in
// the 4*MT bias registers behind the fragments: 16-row tile: lane (q,p), register r = row 16mt+4q+r; 4-row block: register
// i = row base+i in lane row 0 only, because the block's 4 lane rows are summed)
__global__ void conv_pack_kernel(const float *__restrict__ wc, const float *__restrict__ scale, const float *__restrict__ shift,
                                 float *__restrict__ packed, int Cq, int KH, int KW, int MT, int MTB, int NKZ, int transpose,
                                 int nfrag)
{
    const int g = blockIdx.y;
    const float *wg = wc + (size_t)g * Cq * Cq * KH * KW;
    const int KK = KH * KW;
    const int npack = nfrag + 4 * MT;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 4 * MT * 64; e += gridDim.x * blockDim.x) {
        const int lane = e & 63, f = e >> 6;
        const int q = lane >> 4, mt = f >> 2, r = f & 3;
        const int row = mt < MTB ? 16 * mt + 4 * q + r : (q == 0 ? 16 * MTB + 4 * (mt - MTB) + r : Cq);
        packed[((size_t)g * npack + nfrag + f) * 64 + lane] = (shift && row < Cq) ? shift[g * Cq + row] : 0.f;
    }
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nfrag * 64; e += gridDim.x * blockDim.x) {
        const int lane = e & 63, f = e >> 6;
        const int q = lane >> 4, i = lane & 15;
        const int mt = f % MT, j = (f / MT) % NKZ, tap = f / (MT * NKZ);
        const int row = finc_tile_row(MTB, mt, i), col = 4 * j + q;
        const int a = tap / KW, b = tap % KW;
        float v = 0.f;
        if (row < Cq && col < Cq) {
            const int oc = transpose ? col : row, ic = transpose ? row : col;
            v = wg[((size_t)oc * Cq + ic) * KK + (KH - 1 - a) * KW + (KW - 1 - b)];
            if (scale) v *= scale[g * Cq + row];
        }
        packed[((size_t)g * npack + f) * 64 + lane] = v;
    }
}

typedef void (*conv_fn)(const float *, const float *, float *, int, int, int, int, int, int, unsigned);
struct ConvInst {
    int cqp, kh, kw;
    conv_fn fn;
    conv_fn fn_wide;             // W % 16 == 0, one wave per strip: rows move as 16-byte pieces through LDS (nullptr: none)
    int mt, mtb, nkz, nfrag, nw; // mt = mtb 16-row tiles + 4-row blocks: fragments per (tap, k-step)
};
template <int CQP, int KH, int KW, int NW>
constexpr conv_fn wide_fn()
{
    // the staged form holds KH+1 row slots of operands and its pieces in flight: only where that fits beside the bank
    // (<16,5,5> would spill inside the 256 registers of two waves per SIMD; <28,3,3> and <32,3,3> run one wave per SIMD)
    constexpr int NREG = conv_nreg<CQP, KH, KW, NW>();
    if constexpr (NW == 1 && KW <= 5 && NREG + (KH + 1) * KW * (CQP / 4) + 64 <= (NREG <= FINC_CONV_2W_MAX ? 256 : 512))
        return finc_conv_kernel<CQP, KH, KW, NW, true>;
    else return nullptr;
}
template <int CQP, int KH, int KW, int NW = 1>
constexpr ConvInst make_conv()
{
    constexpr int MTB = CQP / 16, MT = MTB + (CQP % 16) / 4;
    return ConvInst{CQP, KH, KW, finc_conv_kernel<CQP, KH, KW, NW, false>, wide_fn<CQP, KH, KW, NW>(), MT, MTB, CQP / 4,
                    KH * KW * (CQP / 4) * MT, NW};
}
const ConvInst g_conv[] = {
    make_conv<4, 3, 3>(),  make_conv<8, 3, 3>(),  make_conv<12, 3, 3>(), make_conv<16, 3, 3>(), make_conv<20, 3, 3>(),
    make_conv<24, 3, 3>(), make_conv<28, 3, 3>(), make_conv<32, 3, 3>(), make_conv<40, 3, 3, 2>(), make_conv<48, 3, 3, 2>(), make_conv<64, 3, 3, 4>(), make_conv<96, 3, 3, 8>(),
    make_conv<4, 2, 2>(),  make_conv<8, 2, 2>(),  make_conv<12, 2, 2>(), make_conv<16, 2, 2>(), make_conv<24, 2, 2>(),
    make_conv<32, 2, 2>(),
    make_conv<4, 5, 5>(),  make_conv<8, 5, 5>(),  make_conv<12, 5, 5>(), make_conv<16, 5, 5>(), make_conv<24, 5, 5, 2>(), make_conv<32, 5, 5, 4>(),
    make_conv<48, 5, 5, 4>(),
    make_conv<4, 3, 5>(),  make_conv<8, 3, 5>(),  make_conv<16, 3, 5>(),
    make_conv<4, 2, 3>(),  make_conv<8, 2, 3>(),  make_conv<16, 2, 3>(),
    make_conv<4, 1, 3>(),  make_conv<4, 3, 1>(),
};
// the smallest compiled bank that holds Cq channels (padded channels are masked in-kernel: one-wave banks hold up to 3 of
// them -- the table has every multiple of 4 there --, the K-split banks any number)
const ConvInst *find_conv(int Cq, int KH, int KW)
{
    const ConvInst *best = nullptr;
    for (const ConvInst &i : g_conv)
        if (i.cqp >= Cq && i.kh == KH && i.kw == KW && (!best || i.cqp < best->cqp)) best = &i;
    if (best && best->nw == 1 && best->cqp - Cq > 3) return nullptr;   // (a one-wave kernel masks the last group of four only)
    return best;
}

} // namespace

// banks beyond this table: the streaming-bank kernel in its forward form (finc_stream.hip)
static bool stream_bank(int Cq, int KH, int KW) { return !find_conv(Cq, KH, KW) && finc_stream_bank_ok(Cq, KH, KW); }

bool finc_conv_supported(int Cq, int H, int W, int KH, int KW)
{
    if (stream_bank(Cq, KH, KW)) return finc_stream_supported(Cq, H, W, KH, KW, false);
    if (!find_conv(Cq, KH, KW)) return false;
    if ((size_t)Cq * H * W * 4 >= ((size_t)1 << 30)) return false;
    return true;
}

// the strip kernels' bank; behind it (3x3 banks one wave holds) the Winograd bank of finc_wino.hip: which of the two a launch
// reads is decided per call (width parity, alignment), so a packed buffer carries both
static size_t conv_bank_bytes(const ConvInst *i, int G) { return (size_t)(i->nfrag + 4 * i->mt) * 64 * sizeof(float) * (size_t)G; }
size_t finc_conv_packed_bytes(int G, int Cq, int KH, int KW)
{
    if (stream_bank(Cq, KH, KW)) return finc_stream_packed_bytes(G, Cq, KH, KW, false);
    const ConvInst *i = find_conv(Cq, KH, KW);
    return i ? conv_bank_bytes(i, G) + finc_wino_packed_bytes(G, Cq, KH, KW) + finc_bigfwd_packed_bytes(G, Cq, KH, KW) +
                   finc_wino5_packed_bytes(G, Cq, KH, KW) + finc_wino4m_packed_bytes(G, Cq, KH, KW) : 0;   // (at most one of the four exists for a bank)
}

int finc_conv_pack(const float *wc, void *packed, int G, int Cq, int KH, int KW, bool transpose, hipStream_t st,
                   const float *scale, const float *shift)
{
    if (stream_bank(Cq, KH, KW)) return finc_stream_pack(wc, scale, shift, packed, G, Cq, KH, KW, false, transpose, st);
    const ConvInst *i = find_conv(Cq, KH, KW);
    if (!i) return FINC_ERR_UNSUPPORTED;
    const int total = i->nfrag * 64;
    int blocks = (total + 255) / 256;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(conv_pack_kernel, dim3(blocks, G), dim3(256), 0, st, wc, scale, shift, (float *)packed, Cq, KH, KW,
                       i->mt, i->mtb, i->nkz, transpose ? 1 : 0, i->nfrag);
    FINC_CHECK_LAUNCH();
    if (finc_wino_packed_bytes(G, Cq, KH, KW))
        return finc_wino_pack(wc, (char *)packed + conv_bank_bytes(i, G), G, Cq, transpose, st, scale, shift);
    if (finc_bigfwd_packed_bytes(G, Cq, KH, KW))   // (the banks beyond the one-wave kernels: the M-split of finc_big.hip)
        return finc_bigfwd_pack(wc, (char *)packed + conv_bank_bytes(i, G), G, Cq, KH, KW, transpose, st, scale, shift);
    if (finc_wino5_packed_bytes(G, Cq, KH, KW))    // (5x5: Winograd F(2,5) along W, finc_wino5.hip)
        return finc_wino5_pack(wc, (char *)packed + conv_bank_bytes(i, G), G, Cq, transpose, st, scale, shift);
    if (finc_wino4m_packed_bytes(G, Cq, KH, KW))   // (3x3 banks of 25 .. 64 channels: F(4,3), M-split, finc_wino4m.hip)
        return finc_wino4m_pack(wc, (char *)packed + conv_bank_bytes(i, G), G, Cq, transpose, st, scale, shift);
    return FINC_OK;
}

int finc_conv_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st)
{
    if (stream_bank(s.Cq, s.KH, s.KW)) return finc_stream_launch(in, packed, out, s, false, st);
    const ConvInst *i = find_conv(s.Cq, s.KH, s.KW);
    if (!i || !finc_conv_supported(s.Cq, s.H, s.W, s.KH, s.KW)) return FINC_ERR_UNSUPPORTED;
    // 3x3 with fewer multiplies (Winograd F(2,3) along W: finc_wino.hip) where the call allows it
    if (finc_wino_packed_bytes(s.G, s.Cq, s.KH, s.KW) && finc_wino_takes(in, out, s))
        return finc_wino_launch(in, (const char *)packed + conv_bank_bytes(i, s.G), out, s, st);
    if (finc_bigfwd_packed_bytes(s.G, s.Cq, s.KH, s.KW) && finc_bigfwd_takes(in, out, s))
        return finc_bigfwd_launch(in, (const char *)packed + conv_bank_bytes(i, s.G), out, s, st);
    // 5x5 with 0.6 x the multiplies (Winograd F(2,5) along W: finc_wino5.hip) where the call allows it
    if (finc_wino5_packed_bytes(s.G, s.Cq, s.KH, s.KW) && finc_wino5_takes(in, out, s))
        return finc_wino5_launch(in, (const char *)packed + conv_bank_bytes(i, s.G), out, s, st);
    // 3x3 banks of 25 .. 64 channels with half the multiplies (F(4,3), M-split over a workgroup's waves: finc_wino4m.hip)
    if (finc_wino4m_packed_bytes(s.G, s.Cq, s.KH, s.KW) && finc_wino4m_takes(in, out, s))
        return finc_wino4m_launch(in, (const char *)packed + conv_bank_bytes(i, s.G), out, s, st);
    const int NS = (s.W + 15) / 16;
    // row chunks of at least 4 rows: the count that minimises rounds x (rows per chunk + KH: every chunk recomputes KH-1 rows of operands
    // and loads the bank once more), two waves per SIMD (finc_common.h finc_row_chunks; profiles/r05/notes/row_chunks.txt)
    const long long waves = (long long)s.B * s.G * NS * i->nw;
    int nrc = finc_row_chunks(waves, 1024, s.H, 4, s.KH, 14);
    static const int force_chunks = finc_env("FINC_CONV_CHUNKS") ? atoi(finc_env("FINC_CONV_CHUNKS")) : 0;   // experiment switch
    if (force_chunks > 0) nrc = force_chunks;
    if (nrc > s.H / 4) nrc = s.H / 4 > 0 ? s.H / 4 : 1;
    const int RC = (s.H + nrc - 1) / nrc;
    nrc = (s.H + RC - 1) / RC;
    static const bool no_wide = finc_env("FINC_CONV_NO_WIDE") != nullptr;   // experiment switch: the dword form everywhere
    // the staged form moves 16-byte pieces: activations that are only float-aligned (a view into a larger tensor) take the
    // dword form, as the inverse sends them to its strict kernel
    const bool aligned16 = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    const conv_fn fn = (i->fn_wide && s.W % 16 == 0 && aligned16 && !no_wide) ? i->fn_wide : i->fn;
    hipLaunchKernelGGL(fn, dim3(s.B * s.G * NS, nrc), dim3(64 * i->nw), 0, st, in, (const float *)packed, out, s.G, s.Cq,
                       s.H, s.W, NS, RC, s.orient);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

int finc_conv_variant(int B, int G, int Cq, int H, int W, int KH, int KW, int *info)
{
    if (stream_bank(Cq, KH, KW)) {              // (7: the streaming-bank kernel, one or four waves per problem)
        if (!finc_conv_supported(Cq, H, W, KH, KW)) return FINC_ERR_UNSUPPORTED;
        int waves = 0;
        (void)finc_stream_info(FincShape{B, G, Cq, H, W, KH, KW, 0}, false, nullptr, nullptr, nullptr, &waves);
        info[0] = waves; info[1] = 7; info[2] = 1;
        return FINC_OK;
    }
    const ConvInst *i = find_conv(Cq, KH, KW);
    if (!i || !finc_conv_supported(Cq, H, W, KH, KW)) return FINC_ERR_UNSUPPORTED;
    static const bool no_wide = finc_env("FINC_CONV_NO_WIDE") != nullptr;
    info[0] = i->nw;
    info[1] = (i->fn_wide && W % 16 == 0 && !no_wide) ? 1 : 0;
    info[2] = (W + 15) / 16;
    const FincShape s{B, G, Cq, H, W, KH, KW, 0};
    if (finc_wino_packed_bytes(G, Cq, KH, KW) && finc_wino_takes(nullptr, nullptr, s)) info[1] = finc_wino_form(s) == 4 ? 4 : 2;   // (2: Winograd F(2,3), 4: F(4,3))
    if (finc_bigfwd_packed_bytes(G, Cq, KH, KW) && finc_bigfwd_takes(nullptr, nullptr, s)) info[1] = 3;   // (3: the big banks' M-split)
    if (finc_wino5_packed_bytes(G, Cq, KH, KW) && finc_wino5_takes(nullptr, nullptr, s)) info[1] = 5;     // (5: Winograd F(2,5), 5x5)
    if (finc_wino4m_packed_bytes(G, Cq, KH, KW) && finc_wino4m_takes(nullptr, nullptr, s)) info[1] = 6;   // (6: Winograd F(4,3), M-split)
    return FINC_OK;
}

unsigned finc_build_flags_conv() { return FINC_BUILD_FLAGS; }
