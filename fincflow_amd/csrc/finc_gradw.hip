// Grad-weight kernels of the forward conv for gfx950 -- their own translation unit because they want the
// OPPOSITE accumulator placement from the kernels of finc_conv.hip / finc_mfma.hip: the NTAP*MT*MT accumulator tiles are
// written by every MFMA of the kernel and read once, at the very end, so they belong in AGPRs (this file is compiled
// WITHOUT -amdgpu-mfma-vgpr-form), which leaves the 256 VGPRs of a one-wave-per-SIMD kernel to the operand slots.
#include "finc_common.h"
#include "finc_tile.h"

#include <stdlib.h>

#include <type_traits>
#include <utility>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr unsigned OFF_INVALID = 0x80000000u;
constexpr unsigned OFF_BAD_CHANNEL = 0x40000000u;

template <int I>
using IC = std::integral_constant<int, I>;

// -----------------------------------------------------------------------------------------------
// grad_w: gw[o][i][KH-1-a][KW-1-b] = sum_{image, h, w} gz[o,h,w] * x[i,h-a,w-b]  per group.
// Same strip walk, but the PIXELS are the MFMA K dimension: A = gz (lane (q,m): channel 16mo+m, pixel 4kk+q),
// B = x shifted by the tap (lane (q,n): channel 16mi+n, pixel 4kk+q-b), D = a 16x16 (o,i) tile per tap, kept in
// NTAP*MT*MT accumulators for the whole kernel.  A wave walks several (image, strip) units of one group and then
// writes its partial tiles; gradw_reduce_kernel sums the partials, un-tiles them and applies the corner-tap mask
// (PaddedConv2d.reset_gradients, layers/conv.py:98-99).  Column shifts are just shifted load addresses here.
// -----------------------------------------------------------------------------------------------
template <int CQP, int KH, int KW>
__global__ __launch_bounds__(64) void finc_gradw_kernel(const float *__restrict__ gz, const float *__restrict__ x,
                                                        float *__restrict__ part, int G, int CQ, int H, int W, int NS,
                                                        int B, int WPG, unsigned orient)
{
    constexpr int MT = (CQP + 15) / 16, NTAP = KH * KW, RS = KH + 1; // RS row slots: rows h+1 (arriving), h, .., h-KH+1
    const int lane = threadIdx.x;
    const int q = lane >> 4, m = lane & 15;
    const int g = blockIdx.x / WPG, wslot = blockIdx.x % WPG;
    const unsigned o = finc_group_orient(orient, g);
    const bool fh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 4u;

    v4f acc[NTAP][MT][MT];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int mo = 0; mo < MT; ++mo)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) acc[t][mo][mi] = (v4f){0.f, 0.f, 0.f, 0.f};

    unsigned choff[MT];                                   // channel 16mt+m of this lane (same for gz and x)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) choff[mt] = (16 * mt + m) < CQ ? (unsigned)(16 * mt + m) * HW * 4u : OFF_BAD_CHANNEL;

    for (int u = wslot; u < B * NS; u += WPG) {
        const int b = u / NS, strip = u % NS;
        const size_t slab = ((size_t)b * G + g) * CQ * HW;
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)(gz + slab), 0, (int)slab_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)(x + slab), 0, (int)slab_bytes, 0x00020000);
        // lane offsets: pixel 4kk+q of the strip, shifted left by b columns for x (invalid columns -> beyond the slab)
        unsigned og[MT][4], ox[KW][MT][4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int col = strip * 16 + 4 * kk + q;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                og[mt][kk] = col < W ? (unsigned)(fw ? W - 1 - col : col) * 4u + choff[mt] : OFF_BAD_CHANNEL;
#pragma unroll
                for (int bb = 0; bb < KW; ++bb) {
                    const int c = col - bb;
                    ox[bb][mt][kk] = (c >= 0 && col < W) ? (unsigned)(fw ? W - 1 - c : c) * 4u + choff[mt] : OFF_BAD_CHANNEL;
                }
            }
        }
        float GA[2][MT][4];                               // gz rows: [arriving / current]
        float XB[RS][KW][MT][4];                          // x rows by slot
#pragma unroll
        for (int s = 0; s < RS; ++s)
#pragma unroll
            for (int bb = 0; bb < KW; ++bb)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) XB[s][bb][mt][kk] = 0.f;
        auto rowbytes = [&](int h) { return (unsigned)((fh ? H - 1 - h : h) * W) * 4u; };
        auto load_row = [&](int h, float (&ga)[MT][4], float (&xb)[KW][MT][4]) {
            const bool rok = h < H;
            const unsigned ro = rok ? rowbytes(h) : OFF_INVALID;      // scalar
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    ga[mt][kk] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, ro + og[mt][kk], 0, 0));
#pragma unroll
                    for (int bb = 0; bb < KW; ++bb)
                        xb[bb][mt][kk] =
                            __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ro + ox[bb][mt][kk], 0, 0));
                }
        };
        auto step = [&](auto i_c, int h) {                // row h lives in slot S, row h+1 arrives into slot S+1
            constexpr int I = decltype(i_c)::value;       // h % UN
            constexpr int S = I % RS, PAR = I & 1;        // x row slot, gz ping-pong buffer
            load_row(h + 1, GA[PAR ^ 1], XB[(S + 1) % RS]);
#pragma unroll
            for (int a = 0; a < KH; ++a)
#pragma unroll
                for (int bb = 0; bb < KW; ++bb)
#pragma unroll
                    for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk)
                                acc[a * KW + bb][mo][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    GA[PAR][mo][kk], XB[(S + RS - a) % RS][bb][mi][kk], acc[a * KW + bb][mo][mi], 0, 0, 0);
        };
        load_row(0, GA[0], XB[0]);
        // RS and the 2-deep gz ring must rotate together: unroll by lcm(RS, 2)
        constexpr int UN = (RS % 2 == 0) ? RS : 2 * RS;
        for (int h0 = 0; h0 < H; h0 += UN) {
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ((h0 + I < H ? step(IC<I>{}, h0 + I) : (void)0), ...);
            }(std::make_integer_sequence<int, UN>{});
        }
    }
    // partial tiles: part[((g*WPG + wslot)*NTAP*MT*MT + tile)*256 + r*64 + lane]
    float *dst = part + (size_t)blockIdx.x * (NTAP * MT * MT) * 256 + lane;
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int mo = 0; mo < MT; ++mo)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const v4f v = acc[t][mo][mi];
                const float v0 = v.x, v1 = v.y, v2 = v.z, v3 = v.w;
                float *d = dst + ((t * MT + mo) * MT + mi) * 256;
                d[0] = v0; d[64] = v1; d[128] = v2; d[192] = v3;
            }
}

// -----------------------------------------------------------------------------------------------
// grad_w, staged form (W % 4 == 0 -- the last strip of a row may be partial --, 16-byte aligned activations).  Same strip walk, same partial tiles for
// gradw_reduce_kernel, two differences:
//   * MEMORY: a row of gz and of x arrives as 16-byte pieces (lane = (channel, piece); the columns left of the strip are
//     one more piece per x channel, as in the staged forward), is parked in an LDS tile and read back as MFMA operands
//     by ds_read_b32 -- the column shift b is a read address.  ceil(4Cq/64) + ceil(5Cq/64) dwordx4 loads per row instead
//     of (1 + KW) * 4 * ceil(Cq/16) dword loads (c3: 4 instead of 32), issued two rows ahead of their use.
//   * OUTPUT CHANNELS beyond the last full 16: v_mfma_f32_4x4x1_16B_f32 per 4-channel block (finc_tile.h) instead of a
//     padded 16-row tile: A = gz with lane (q,n) -> channel base + (n & 3), the B operand is the very register the 16-row
//     tile reads, and the accumulator keeps, per lane row q, the partial sum over the pixels == q (mod 4); the 4x4
//     transpose-reduce runs ONCE, before the partials are written (c3: 48 instead of 64 MFMA cycles per tap, i tile and
//     4 pixels).  SMALL = false keeps the padded tile where 4-row blocks would need too many accumulators.
//   * INPUT CHANNELS (FLAT, round 3): the B operand's 16 columns need not be 16 channels of ONE tap.  The KW taps of a filter
//     row times the Cq channels are laid side by side -- column j = b * Cq + i -- and cut into tiles of 16: ceil(KW*Cq/16)
//     tiles per filter row instead of KW * ceil(Cq/16) (c3: 5 instead of 6, Cq = 20: 4 instead of 6, Cq = 8: 2 instead of 3).
//     A lane simply reads its own (channel, shifted column) from the LDS tile; the partial tiles are written back in the old
//     (tap, o tile, i tile) layout, so gradw_reduce_kernel does not notice.
// -----------------------------------------------------------------------------------------------
template <int CQP, int KH, int KW, bool SMALL, bool FLAT = false>
__global__ __launch_bounds__(64) void finc_gradw_staged_kernel(const float *__restrict__ gz, const float *__restrict__ x,
                                                               float *__restrict__ part, int G, int CQ, int H, int W, int NS,
                                                               int B, int WPG, unsigned orient)
{
    constexpr int MT = (CQP + 15) / 16, NTAP = KH * KW, RS = KH + 1;   // MT: tiles of the partial layout (both dimensions)
    constexpr int MTB = SMALL ? CQP / 16 : MT;                          // full 16-row tiles of output channels
    constexpr int NSM = SMALL ? (CQP % 16) / 4 : 0;                     // 4-row blocks behind them
    constexpr int MTBD = MTB > 0 ? MTB : 1, NSMD = NSM > 0 ? NSM : 1;
    constexpr int ROWJ = KW * CQP;                                      // FLAT: columns (b, i) of one filter row ...
    constexpr int NIT = FLAT ? (ROWJ + 15) / 16 : KW * MT;              // ... and the B tiles of a filter row (else: tap b, i tile mi)
    static_assert(!FLAT || CQP % 16 != 0, "FLAT reads its zeros from the channel rows behind the bank");
    constexpr int XP = 24, GP = 20;                                     // tile pitches (floats): [channel][4 halo + 16], [channel][16]
    constexpr int NXI = (5 * CQP + 63) / 64, NGI = (4 * CQP + 63) / 64; // dwordx4 loads per row
    static_assert(KW <= 5, "the halo is one 16-byte piece");
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float xt[16 * MT * XP + 4];
    __shared__ __attribute__((aligned(16))) float gt[16 * MT * GP + 4];
    const int lane = threadIdx.x;
    const int q = lane >> 4, m = lane & 15;
    const int g = blockIdx.x / WPG, wslot = blockIdx.x % WPG;
    const unsigned o = finc_group_orient(orient, g);
    const bool fh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 4u;
    for (int i = lane; i < 16 * MT * XP + 4; i += 64) xt[i] = 0.f;      // channel rows >= CQP are never written: they stay 0
    for (int i = lane; i < 16 * MT * GP + 4; i += 64) gt[i] = 0.f;

    // accumulators: [filter row a][B tile of the row][o tile / o block]  (not FLAT: B tile = b * MT + mi)
    v4f acc[KH][NIT][MTBD], accs[KH][NIT][NSMD];
#pragma unroll
    for (int a = 0; a < KH; ++a)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
#pragma unroll
            for (int mo = 0; mo < MTBD; ++mo) acc[a][it][mo] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sb = 0; sb < NSMD; ++sb) accs[a][it][sb] = (v4f){0.f, 0.f, 0.f, 0.f};
        }
    // read positions of lane (q,m): canonical column c of the strip sits at tile column (fw ? 15 - c : c) (+4: the halo piece
    // comes first in memory order when the strip is not mirrored; mirrored, it comes last and the index runs down)
    int xrd[FLAT ? NIT : KW][4], grd[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int c = 4 * kk + q;
        grd[kk] = m * GP + (fw ? 15 - c : c);
        if constexpr (FLAT) {       // column j = 16 it + m of the filter row = tap b = j / CQP, channel i = j % CQP; past the end: a zero row
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int j = 16 * it + m, bb = j / CQP, ic = j % CQP;
                xrd[it][kk] = j < ROWJ ? ic * XP + (fw ? 15 - (c - bb) : 4 + (c - bb)) : CQP * XP + 4;
            }
        } else {
#pragma unroll
            for (int bb = 0; bb < KW; ++bb) xrd[bb][kk] = m * XP + (fw ? 15 - (c - bb) : 4 + (c - bb));
        }
    }
    const int gsm = (16 * MTB + (m & 3)) * GP - m * GP;                 // 4-row block operand: channel base + (m & 3) instead of m

    for (int u = wslot; u < B * NS; u += WPG) {
        const int b = u / NS, strip = u % NS;
        const size_t slab = ((size_t)b * G + g) * CQ * HW;
        auto rsrc = [&](const float *base, bool ok) {
            return __builtin_amdgcn_make_buffer_rsrc((void *)(base + slab), 0, ok ? (int)slab_bytes : 0, 0x00020000);
        };
        const int ms = fw ? W - 16 - strip * 16 : strip * 16;           // memory column where the strip's sector starts
        const int hm = fw ? ms + 16 : ms - 4;                           // ... and the piece holding the columns left of it
        unsigned lvx[NXI], lvg[NGI];
        int lwx[NXI], lwg[NGI];
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int t = 64 * i + lane;
            lvx[i] = OFF_BAD_CHANNEL;
            lwx[i] = 16 * MT * XP;                                      // scratch piece behind the tile
            if (t < 4 * CQP) {
                const int c = t >> 2, k = t & 3;
                // (W % 4 == 0: a piece is inside the row or outside -- the last strip of a row that is not a multiple of 16 wide)
                if (c < CQ && ms + 4 * k >= 0 && ms + 4 * k < W) lvx[i] = (unsigned)c * HW * 4u + (unsigned)(ms + 4 * k) * 4u;
                lwx[i] = c * XP + (fw ? 0 : 4) + 4 * k;
            } else if (t < 5 * CQP) {
                const int c = t - 4 * CQP;
                if (c < CQ && hm >= 0 && hm < W) lvx[i] = (unsigned)c * HW * 4u + (unsigned)hm * 4u;
                lwx[i] = c * XP + (fw ? 16 : 0);
            }
        }
#pragma unroll
        for (int i = 0; i < NGI; ++i) {
            const int t = 64 * i + lane;
            lvg[i] = OFF_BAD_CHANNEL;
            lwg[i] = 16 * MT * GP;
            if (t < 4 * CQP) {
                const int c = t >> 2, k = t & 3;
                if (c < CQ && ms + 4 * k >= 0 && ms + 4 * k < W) lvg[i] = (unsigned)c * HW * 4u + (unsigned)(ms + 4 * k) * 4u;
                lwg[i] = c * GP + 4 * k;
            }
        }
        auto rowbytes = [&](int h) { return (unsigned)((fh ? H - 1 - h : h) * W) * 4u; };
        // pieces in flight: TWO rows ahead (one wave per SIMD: nobody else hides a row's HBM latency), two register sets by
        // row parity
        v4u LX[2][NXI], LG[2][NGI];
        auto issue = [&](auto par_c, int h) {
            constexpr int PAR = decltype(par_c)::value;
            const bool ok = h >= 0 && h < H;
            const __amdgpu_buffer_rsrc_t rx = rsrc(x, ok), rg = rsrc(gz, ok);
            const unsigned ro = ok ? rowbytes(h) : 0u;
#pragma unroll
            for (int i = 0; i < NXI; ++i) LX[PAR][i] = __builtin_amdgcn_raw_buffer_load_b128(rx, lvx[i], ro, 0);
#pragma unroll
            for (int i = 0; i < NGI; ++i) LG[PAR][i] = __builtin_amdgcn_raw_buffer_load_b128(rg, lvg[i], ro, 0);
        };
        float GA[2][MTBD][4], GS[2][NSMD][4];                           // gz operands: [arriving / current]
        float XB[RS][NIT][4];                                           // x operands by row slot and B tile of the row
#pragma unroll
        for (int sl = 0; sl < RS; ++sl)
#pragma unroll
            for (int it = 0; it < NIT; ++it)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) XB[sl][it][kk] = 0.f;
        // stage(row): its pieces (asked for two steps ago) go registers -> tiles -> operand registers of the row's slot, and
        // row + 2 is asked for
        auto stage = [&](auto sn_c, auto pn_c, int row) {
            constexpr int SN = decltype(sn_c)::value, PN = decltype(pn_c)::value;
#pragma unroll
            for (int i = 0; i < NXI; ++i) reinterpret_cast<v4u *>(xt)[lwx[i] >> 2] = LX[PN][i];   // (index in pieces: ds_write_b128)
#pragma unroll
            for (int i = 0; i < NGI; ++i) reinterpret_cast<v4u *>(gt)[lwg[i] >> 2] = LG[PN][i];
            issue(IC<PN>{}, row + 2);                                   // into the set just emptied
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int mo = 0; mo < MTB; ++mo) GA[PN][mo][kk] = gt[grd[kk] + 16 * mo * GP];
#pragma unroll
                for (int sb = 0; sb < NSM; ++sb) GS[PN][sb][kk] = gt[grd[kk] + gsm + 4 * sb * GP];
                if constexpr (FLAT) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) XB[SN][it][kk] = xt[xrd[it][kk]];
                } else {
#pragma unroll
                    for (int bb = 0; bb < KW; ++bb)
#pragma unroll
                        for (int mi = 0; mi < MT; ++mi) XB[SN][bb * MT + mi][kk] = xt[xrd[bb][kk] + 16 * mi * XP];
                }
            }
        };
        // step of row h (slot h % RS, gz buffer h & 1): row h+1 is staged (it lands during the MFMAs), then the MFMAs of row h
        auto step = [&](auto i_c, int h) {
            constexpr int I = decltype(i_c)::value;
            constexpr int SC = I % RS, PC = I & 1;
            stage(IC<(I + 1) % RS>{}, IC<PC ^ 1>{}, h + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < KH; ++a)
#pragma unroll
                for (int it = 0; it < NIT; ++it)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const float xb = XB[(SC + RS - a) % RS][it][kk];
#pragma unroll
                        for (int mo = 0; mo < MTB; ++mo)
                            acc[a][it][mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(GA[PC][mo][kk], xb, acc[a][it][mo], 0, 0, 0);
#pragma unroll
                        for (int sb = 0; sb < NSM; ++sb)
                            accs[a][it][sb] = __builtin_amdgcn_mfma_f32_4x4x1f32(GS[PC][sb][kk], xb, accs[a][it][sb], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        };
        issue(IC<0>{}, 0);
        issue(IC<1>{}, 1);
        stage(IC<0>{}, IC<0>{}, 0);                                     // row 0 into slot 0 (asks for row 2)
        constexpr int UN = (RS % 2 == 0) ? RS : 2 * RS;                 // row slots and the 2-deep gz ring rotate together
        for (int h0 = 0; h0 < H; h0 += UN) {
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ((h0 + I < H ? step(IC<I>{}, h0 + I) : (void)0), ...);
            }(std::make_integer_sequence<int, UN>{});
        }
    }
    // partial tiles, D layout for gradw_reduce_kernel: part[((g*WPG + wslot)*NTAP*MT*MT + tile)*256 + r*64 + lane]
    float *dst = part + (size_t)blockIdx.x * (NTAP * MT * MT) * 256;
#pragma unroll
    for (int a = 0; a < KH; ++a)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            // column m of this B tile in the partial layout: tap t, i tile mi, column n (FLAT: a lane's own; columns past the
            // end of the filter row belong to nothing -- the positions they would fill hold channels >= Cq, which the reduce skips)
            int t, mi, n;
            bool live = true;
            if constexpr (FLAT) {
                const int j = 16 * it + m;
                live = j < ROWJ;
                t = a * KW + j / CQP; mi = (j % CQP) >> 4; n = (j % CQP) & 15;
            } else {
                t = a * KW + it / MT; mi = it % MT; n = m;
            }
#pragma unroll
            for (int mo = 0; mo < MTB; ++mo) {
                const v4f v = acc[a][it][mo];
                const float v0 = v.x, v1 = v.y, v2 = v.z, v3 = v.w;
                float *d = dst + ((t * MT + mo) * MT + mi) * 256 + q * 16 + n;
                if (live) { d[0] = v0; d[64] = v1; d[128] = v2; d[192] = v3; }
            }
            // a reduced block leaves channel 16*MTB + 4sb + q in lane row q: in the D layout of tile mo = MTB that is
            // register q of lane (sb, n)
#pragma unroll
            for (int sb = 0; sb < NSM; ++sb) {
                const float v = finc_block_reduce(accs[a][it][sb]);
                if (live) dst[((t * MT + MTB) * MT + mi) * 256 + q * 64 + sb * 16 + n] = v;
            }
        }
}

// -----------------------------------------------------------------------------------------------
// grad_w, tiled form: ONE 16x16 (o, i) tile pair per workgroup, for the banks whose NTAP * MT * MT accumulator tiles do not fit
// one wave (Cq > 32 at 3x3, Cq > 16 at 5x5: until round 2 those ran on the direct kernel, 24 .. 78 ms per training step at
// the c5 bank).  blockIdx = (((g * MTT + mo) * MTT + mi) * WPG + wslot); the wave stages only the 16 gz channels of tile mo
// and the 16 x channels of tile mi (1 + 2 dwordx4 loads per row) and keeps NTAP accumulators.  A gz tile is read once per
// mi and an x tile once per mo -- MTT times the minimum, out of L2.  Same partial layout, same reduce kernel.
// -----------------------------------------------------------------------------------------------
template <int KH, int KW>
__global__ __launch_bounds__(64) void finc_gradw_tiled_kernel(const float *__restrict__ gz, const float *__restrict__ x,
                                                              float *__restrict__ part, int G, int CQ, int H, int W, int NS,
                                                              int B, int WPG, unsigned orient, int MTT)
{
    constexpr int NTAP = KH * KW, RS = KH + 1;
    constexpr int XP = 24, GP = 20;
    constexpr int NXI = 2;                                              // 16 channels x (4 pieces + the halo piece) = 80 load slots
    static_assert(KW <= 5, "the halo is one 16-byte piece");
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float xt[16 * XP + 4];
    __shared__ __attribute__((aligned(16))) float gt[16 * GP + 4];
    const int lane = threadIdx.x;
    const int q = lane >> 4, m = lane & 15;
    int bi = blockIdx.x;
    const int wslot = bi % WPG; bi /= WPG;
    const int mi = bi % MTT; bi /= MTT;
    const int mo = bi % MTT;
    const int g = bi / MTT;
    const unsigned o = finc_group_orient(orient, g);
    const bool fh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 4u;
    const int cg = 16 * mo, cx = 16 * mi;                               // first channel of the gz tile / of the x tile

    v4f acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
    int xrd[KW][4], grd[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int c = 4 * kk + q;
        grd[kk] = m * GP + (fw ? 15 - c : c);
#pragma unroll
        for (int bb = 0; bb < KW; ++bb) xrd[bb][kk] = m * XP + (fw ? 15 - (c - bb) : 4 + (c - bb));
    }
    for (int u = wslot; u < B * NS; u += WPG) {
        const int b = u / NS, strip = u % NS;
        const size_t slab = ((size_t)b * G + g) * CQ * HW;
        auto rsrc = [&](const float *base, bool ok) {
            return __builtin_amdgcn_make_buffer_rsrc((void *)(base + slab), 0, ok ? (int)slab_bytes : 0, 0x00020000);
        };
        const int ms = fw ? W - 16 - strip * 16 : strip * 16;
        const int hm = fw ? ms + 16 : ms - 4;
        unsigned lvx[NXI], lvg;
        int lwx[NXI], lwg;
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int t = 64 * i + lane;
            lvx[i] = OFF_BAD_CHANNEL;
            lwx[i] = 16 * XP;                                           // scratch piece behind the tile
            if (t < 64) {
                const int c = t >> 2, k = t & 3;
                if (cx + c < CQ && ms + 4 * k >= 0 && ms + 4 * k < W) lvx[i] = (unsigned)(cx + c) * HW * 4u + (unsigned)(ms + 4 * k) * 4u;
                lwx[i] = c * XP + (fw ? 0 : 4) + 4 * k;
            } else if (t < 80) {
                const int c = t - 64;
                if (cx + c < CQ && hm >= 0 && hm < W) lvx[i] = (unsigned)(cx + c) * HW * 4u + (unsigned)hm * 4u;
                lwx[i] = c * XP + (fw ? 16 : 0);
            }
        }
        {
            const int c = lane >> 2, k = lane & 3;
            lvg = (cg + c < CQ && ms + 4 * k >= 0 && ms + 4 * k < W) ? (unsigned)(cg + c) * HW * 4u + (unsigned)(ms + 4 * k) * 4u : OFF_BAD_CHANNEL;
            lwg = c * GP + 4 * k;
        }
        auto rowbytes = [&](int h) { return (unsigned)((fh ? H - 1 - h : h) * W) * 4u; };
        v4u LX[2][NXI], LG[2];                                          // pieces in flight, two rows ahead, by row parity
        auto issue = [&](auto par_c, int h) {
            constexpr int PAR = decltype(par_c)::value;
            const bool ok = h >= 0 && h < H;
            const __amdgpu_buffer_rsrc_t rx = rsrc(x, ok), rg = rsrc(gz, ok);
            const unsigned ro = ok ? rowbytes(h) : 0u;
#pragma unroll
            for (int i = 0; i < NXI; ++i) LX[PAR][i] = __builtin_amdgcn_raw_buffer_load_b128(rx, lvx[i], ro, 0);
            LG[PAR] = __builtin_amdgcn_raw_buffer_load_b128(rg, lvg, ro, 0);
        };
        float GA[2][4];
        float XB[RS][KW][4];
#pragma unroll
        for (int sl = 0; sl < RS; ++sl)
#pragma unroll
            for (int bb = 0; bb < KW; ++bb)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) XB[sl][bb][kk] = 0.f;
        auto stage = [&](auto sn_c, auto pn_c, int row) {
            constexpr int SN = decltype(sn_c)::value, PN = decltype(pn_c)::value;
#pragma unroll
            for (int i = 0; i < NXI; ++i) reinterpret_cast<v4u *>(xt)[lwx[i] >> 2] = LX[PN][i];
            reinterpret_cast<v4u *>(gt)[lwg >> 2] = LG[PN];
            issue(IC<PN>{}, row + 2);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                GA[PN][kk] = gt[grd[kk]];
#pragma unroll
                for (int bb = 0; bb < KW; ++bb) XB[SN][bb][kk] = xt[xrd[bb][kk]];
            }
        };
        auto step = [&](auto i_c, int h) {
            constexpr int I = decltype(i_c)::value;
            constexpr int SC = I % RS, PC = I & 1;
            stage(IC<(I + 1) % RS>{}, IC<PC ^ 1>{}, h + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < KH; ++a)
#pragma unroll
                for (int bb = 0; bb < KW; ++bb)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        acc[a * KW + bb] = __builtin_amdgcn_mfma_f32_16x16x4f32(GA[PC][kk], XB[(SC + RS - a) % RS][bb][kk], acc[a * KW + bb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        issue(IC<0>{}, 0);
        issue(IC<1>{}, 1);
        stage(IC<0>{}, IC<0>{}, 0);
        constexpr int UN = (RS % 2 == 0) ? RS : 2 * RS;
        for (int h0 = 0; h0 < H; h0 += UN) {
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ((h0 + I < H ? step(IC<I>{}, h0 + I) : (void)0), ...);
            }(std::make_integer_sequence<int, UN>{});
        }
    }
    float *dst = part + ((size_t)(g * WPG + wslot) * (NTAP * MTT * MTT)) * 256 + lane;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        const v4f v = acc[t];
        const float v0 = v.x, v1 = v.y, v2 = v.z, v3 = v.w;
        float *d = dst + (size_t)((t * MTT + mo) * MTT + mi) * 256;
        d[0] = v0; d[64] = v1; d[128] = v2; d[192] = v3;
    }
}

// gw[g][o][i][kh][kw] = sum over the WPG partials; D layout: lane (q,n), reg r -> o = 16mo+4q+r, i = 16mi+n.
// A block of 256 threads owns 32 consecutive entries; its 8 thread groups sum 8 interleaved slices of the partials
// (w = j, j+8, ...) and the slices meet in LDS in a FIXED order: 8x the loads in flight of one thread per entry, and
// the same bits on every run.
__global__ __launch_bounds__(256) void gradw_reduce_kernel(const float *__restrict__ part, float *__restrict__ gw, int Cq, int KH,
                                                           int KW, int MT, int WPG)
{
    __shared__ float slice[8][32];
    const int g = blockIdx.y;
    const int ntap = KH * KW;
    const int per = ntap * MT * MT * 256;
    const int el = threadIdx.x & 31, j = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;                    // per is a multiple of 256: every block is full
    float s = 0.f;
    const float *p = part + (size_t)g * WPG * per + e;
    for (int w = j; w < WPG; w += 8) s += p[(size_t)w * per];
    slice[j][el] = s;
    __syncthreads();
    if (j != 0) return;
    s = ((slice[0][el] + slice[1][el]) + (slice[2][el] + slice[3][el])) + ((slice[4][el] + slice[5][el]) + (slice[6][el] + slice[7][el]));
    const int lane = e & 63, r = (e >> 6) & 3, tile = e >> 8;
    const int mi = tile % MT, mo = (tile / MT) % MT, t = tile / (MT * MT);
    const int oc = 16 * mo + 4 * (lane >> 4) + r, ic = 16 * mi + (lane & 15);
    if (oc < Cq && ic < Cq) {
        const int a = t / KW, b = t % KW;
        const bool masked = (a == 0 && b == 0) && ic >= oc;
        gw[(((size_t)(g * Cq + oc) * Cq + ic) * KH + (KH - 1 - a)) * KW + (KW - 1 - b)] = masked ? 0.f : s;
    }
}

typedef void (*gradw_fn)(const float *, const float *, float *, int, int, int, int, int, int, int, unsigned);
typedef void (*gradw_tiled_fn)(const float *, const float *, float *, int, int, int, int, int, int, int, unsigned, int);
struct GradwInst {
    int cqp, kh, kw;
    int mtg;            // ceil(Cq/16) tiles in both dimensions of the partial layout
    gradw_fn gw;        // dword loads, any W (nullptr: NTAP*MT*MT accumulators would not fit)
    gradw_fn gw_staged; // W % 4 == 0, 16-byte aligned activations (nullptr: none)
    gradw_tiled_fn gw_tiled;   // one (o, i) tile pair per workgroup, same conditions: for the banks gw cannot hold (nullptr: KW > 5)
};
template <int CQP, int KH, int KW>
constexpr gradw_fn gradw_staged_fn()
{
    constexpr int MTG = (CQP + 15) / 16, NTAP = KH * KW;
    // 4-row blocks for the channels behind the last full 16 where their accumulators fit beside the operand slots
    constexpr int ACC_SMALL = NTAP * (CQP / 16 + (CQP % 16) / 4) * MTG * 4;
    if constexpr (KW > 5 || NTAP * MTG * MTG * 4 > 200) return nullptr;
    else if constexpr (CQP % 16 != 0 && ACC_SMALL <= 224) {
        // the taps of a filter row side by side in the B tiles, where that saves tiles (FLAT)
        if constexpr ((KW * CQP + 15) / 16 < KW * MTG) return finc_gradw_staged_kernel<CQP, KH, KW, true, true>;
        else return finc_gradw_staged_kernel<CQP, KH, KW, true>;
    } else return finc_gradw_staged_kernel<CQP, KH, KW, false>;
}
template <int CQP, int KH, int KW>
constexpr GradwInst make_gradw()
{
    constexpr int MTG = (CQP + 15) / 16;
    if constexpr (KH * KW * MTG * MTG * 4 <= 200)
        return GradwInst{CQP, KH, KW, MTG, finc_gradw_kernel<CQP, KH, KW>, gradw_staged_fn<CQP, KH, KW>(), nullptr};
    else if constexpr (KW <= 5) return GradwInst{CQP, KH, KW, MTG, nullptr, nullptr, finc_gradw_tiled_kernel<KH, KW>};
    else return GradwInst{CQP, KH, KW, MTG, nullptr, nullptr, nullptr};
}
// the (Cq, K) pairs of finc_conv.hip's table
const GradwInst g_gradw[] = {
    make_gradw<4, 3, 3>(),  make_gradw<8, 3, 3>(),  make_gradw<12, 3, 3>(), make_gradw<16, 3, 3>(), make_gradw<20, 3, 3>(),
    make_gradw<24, 3, 3>(), make_gradw<28, 3, 3>(), make_gradw<32, 3, 3>(), make_gradw<40, 3, 3>(), make_gradw<48, 3, 3>(), make_gradw<64, 3, 3>(), make_gradw<96, 3, 3>(),
    make_gradw<4, 2, 2>(),  make_gradw<8, 2, 2>(),  make_gradw<12, 2, 2>(), make_gradw<16, 2, 2>(), make_gradw<24, 2, 2>(),
    make_gradw<32, 2, 2>(),
    make_gradw<4, 5, 5>(),  make_gradw<8, 5, 5>(),  make_gradw<12, 5, 5>(), make_gradw<16, 5, 5>(), make_gradw<24, 5, 5>(), make_gradw<32, 5, 5>(),
    make_gradw<48, 5, 5>(),
    make_gradw<4, 3, 5>(),  make_gradw<4, 1, 3>(),  make_gradw<4, 3, 1>(),
};
// the smallest compiled bank that holds Cq channels (every kernel here tests `channel < CQ` per lane: any padding is fine)
const GradwInst *find_gradw(int Cq, int KH, int KW)
{
    const GradwInst *best = nullptr;
    for (const GradwInst &i : g_gradw)
        if (i.cqp >= Cq && i.kh == KH && i.kw == KW && (!best || i.cqp < best->cqp)) best = &i;
    return best;
}

} // namespace

static int gradw_wpg(const FincShape &s)
{
    const int units = s.B * ((s.W + 15) / 16);
    return units < 256 ? units : 256;
}
// tiled form: G * MTT^2 * WPG workgroups of one wave; about two per SIMD
static int gradw_wpg_tiled(const FincShape &s, int mtt)
{
    const int units = s.B * ((s.W + 15) / 16);
    int w = 2048 / (s.G * mtt * mtt);
    if (w < 1) w = 1;
    if (w > 256) w = 256;
    return units < w ? units : w;
}
static bool gradw_use_tiled(const GradwInst *i, const FincShape &s) { return i && !i->gw && i->gw_tiled && s.W % 4 == 0; }

size_t finc_gradw_workspace_bytes(const FincShape &s)
{
    const GradwInst *i = find_gradw(s.Cq, s.KH, s.KW);
    if (!i || !finc_conv_supported(s.Cq, s.H, s.W, s.KH, s.KW)) return 0;
    if (gradw_use_tiled(i, s)) return (size_t)s.G * gradw_wpg_tiled(s, i->mtg) * s.KH * s.KW * i->mtg * i->mtg * 256 * sizeof(float);
    if (!i->gw) return 0;
    return (size_t)s.G * gradw_wpg(s) * s.KH * s.KW * i->mtg * i->mtg * 256 * sizeof(float);
}

// FINC_ERR_UNSUPPORTED: no MFMA grad-weight kernel for this call (the caller falls back to the direct kernel)
int finc_gradw_launch(const float *gz, const float *x, float *gw, void *workspace, const FincShape &s, hipStream_t st)
{
    const GradwInst *i = find_gradw(s.Cq, s.KH, s.KW);
    if (!i) return FINC_ERR_UNSUPPORTED;
    const int NS = (s.W + 15) / 16;
    const bool aligned16 = (((uintptr_t)gz | (uintptr_t)x) & 15) == 0;
    int WPG;
    if (gradw_use_tiled(i, s)) {
        if (!aligned16) return FINC_ERR_UNSUPPORTED;
        WPG = gradw_wpg_tiled(s, i->mtg);
        hipLaunchKernelGGL(i->gw_tiled, dim3(s.G * i->mtg * i->mtg * WPG), dim3(64), 0, st, gz, x, (float *)workspace, s.G, s.Cq,
                           s.H, s.W, NS, s.B, WPG, s.orient, i->mtg);
    } else {
        if (!i->gw) return FINC_ERR_UNSUPPORTED;
        WPG = gradw_wpg(s);
        static const bool no_staged = finc_env("FINC_GRADW_NO_STAGED") != nullptr;   // experiment switch
        const gradw_fn fn = (i->gw_staged && s.W % 4 == 0 && aligned16 && !no_staged) ? i->gw_staged : i->gw;
        hipLaunchKernelGGL(fn, dim3(s.G * WPG), dim3(64), 0, st, gz, x, (float *)workspace, s.G, s.Cq, s.H, s.W, NS, s.B,
                           WPG, s.orient);
    }
    FINC_CHECK_LAUNCH();
    const int per = s.KH * s.KW * i->mtg * i->mtg * 256;
    const int blocks = per / 32;
    hipLaunchKernelGGL(gradw_reduce_kernel, dim3(blocks, s.G), dim3(256), 0, st, (const float *)workspace, gw, s.Cq, s.KH,
                       s.KW, i->mtg, WPG);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

// which grad-weight kernel finc_backward_f32 runs for this shape, given 16-byte aligned activations and a full workspace:
// 0 direct (no MFMA instantiation), 1 dword MFMA kernel, 2 staged (16-byte pieces through LDS), 3 tiled (one tile pair per workgroup)
int finc_gradw_variant(const FincShape &s)
{
    const GradwInst *i = find_gradw(s.Cq, s.KH, s.KW);
    if (!i || finc_gradw_workspace_bytes(s) == 0) return 0;
    if (gradw_use_tiled(i, s)) return 3;
    if (!i->gw) return 0;
    static const bool no_staged = finc_env("FINC_GRADW_NO_STAGED") != nullptr;
    return (i->gw_staged && s.W % 4 == 0 && !no_staged) ? 2 : 1;
}

unsigned finc_build_flags_gradw() { return FINC_BUILD_FLAGS; }
