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
    {   // four partial chains: the loads of a trip are independent (the order of the additions is still fixed)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int w = j;
        for (; w + 24 < WPG; w += 32) {
            s0 += p[(size_t)w * per]; s1 += p[(size_t)(w + 8) * per]; s2 += p[(size_t)(w + 16) * per]; s3 += p[(size_t)(w + 24) * per];
        }
        for (; w < WPG; w += 8) s0 += p[(size_t)w * per];
        s = (s0 + s1) + (s2 + s3);
    }
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

// -----------------------------------------------------------------------------------------------
// grad_w with fewer multiplies (round 4): the Winograd F(4,3) of finc_wino.hip TRANSPOSED, for 3-wide filters.
//
// The forward's tile  y_i = sum_k g_k d_{i+k}  (4 outputs, 3 taps, 6 inputs) is the trilinear form
//     T(y', g, d) = sum_f (A y')_f (G g)_f (B^T d)_f ,   A = (A^T)^T  (6 x 4),
// so the filter's gradient is  dg_k = sum_f G[f][k] * (A gz)_f * (B^T x)_f : per tile of 4 columns SIX products per (o, i, filter
// row) instead of 12.  The products are summed over all tiles, rows and images IN the frequency domain -- the sums are the
// accumulators of this kernel, M[f][a][o][i] -- and G^T is applied once, by gradw_wino_reduce_kernel.  Points 0, +-1, +-3/2,
// infinity as in the forward (tests/test_winograd_algebra.py checks the matrices exactly, this form included):
//     U = A gz :  g0 | e + o | e - o | e' + o' | e' - o' | g3          e = g0 + g2, o = g1 + g3, e' = g0 + 2.25 g2, o' = 1.5 g1 + 3.375 g3
//     V = B^T x:  2.25 d0 - 3.25 d2 + d4 | p + r | p - r | s + 1.5 u | s - 1.5 u | 2.25 d1 - 3.25 d3 + d5
//                 p = d4 - 2.25 d2, r = d3 - 2.25 d1, s = d4 - d2, u = d3 - d1          (d_j = x[4t - 2 + j], gz tile = columns 4t .. 4t+3)
//     dW[b = 2 - k] = dg_k:  k = 0: M0/2.25 - 0.4 (M1 + M2) + (8/45)(M3 + M4);  k = 1: -0.4 (M1 - M2) + (4/15)(M3 - M4);
//                            k = 2: -0.4 (M1 + M2) + 0.4 (M3 + M4) + M5
// The MFMA K dimension is the TILE: 16 columns are 4 tiles = the 4 k-slots of one MFMA per (frequency, filter row, o tile,
// i tile); a strip is 16 or 32 columns (KS = 1 / 2 k-steps).  A wave walks (image, strip) units row by row like the staged
// kernel.  The loads are wide -- lane = (channel, tile), so the 8 lanes of a channel ask for 128 contiguous bytes of a 32-column
// strip and the pieces leave L2 as 128-byte requests: with 64-byte ones (16-column strips) the kernel's time followed its
// fetched bytes at 45 G requests/s = 2.8 TB/s whatever else was changed (profiles/r04/notes/gradw_winograd.md) -- and land in
// two raw LDS tiles as they came.  From there every lane reads, for each of its OPERAND registers (lane (q, n) = tile 4 ks + q
// of channel n), the gz piece and the six x columns of that (channel, tile) and transforms them in place: U and V never exist
// in memory, the stage of a row is ONE LDS round trip, and it sits between the MFMA groups of the row before (phase A | filter
// row 0 | phase B | filter row 1 | phase C | filter row 2).  The loads are inline asm with a counted vmcnt.
// Channels behind the last full 16 (Cq = 24: 8) run on 4x4x1 MFMAs without any
// operand of their own: the block operand of gz holds channel 16 MTB + 4 ((n >> 3) & 1) + (n & 3) in lane (q, n), the one of x
// channel 16 MTB + 4 ((n >> 2) & 1) + (n & 3), and then
//     o block sb x i tile:   A = gz block operand, broadcast of block 2 sb (CBSZ = 2);  B = the x tile operand
//     i block ig x o tile:   A = x block operand, broadcast of block ig;  B = the gz tile operand        (the transposed product)
//     o blocks x i blocks:   ONE plain 4x4x1: block p of a lane row pairs o block p >> 1 with i block p & 1
// -- 72 MFMA cycles per (frequency, filter row) at Cq = 24, no padding, 24 accumulator registers.  Those are what limits the
// form: 6 frequencies x 3 rows x 24 = 432 registers, so the frequencies are split over FS = 2 waves -- the two waves of one
// workgroup, which share the raw tiles (each loads half of the pieces; one barrier per row) and transform only their own three
// frequencies -- and the bank must be 16 (FS = 1), 24 or 32 channels.
// Partial sums: part[((g * FS + fh) * WPG + w) * PER + ((fl * KH + a) * CQP + o) * CQP + i], PER = (6 / FS) KH CQP^2, complete values
// (the 4-row blocks are transpose-reduced first).
// -----------------------------------------------------------------------------------------------
template <int CBSZ, int ABID>
__device__ inline void gw_mma4(v4f &acc, float a, float b)
{
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, CBSZ, ABID, 0);
}

// FH: the frequencies of this wave (0: f0..f2, 1: f3..f5, 2: all six); FW: the group's strips are mirrored -- both are
// uniform per workgroup and compile-time here (the kernel picks the body), so the loop is straight-line code: no selects, no
// exec masks (lanes without a (channel, tile) park their transforms in a scratch word of their own), and the stage of row h+1
// is cut into four phases that sit BETWEEN the MFMA groups of row h -- every LDS round trip of the chain
// raw piece -> tile -> transform -> [frequency][channel][tile] -> operand has a filter row's MFMAs to complete behind.
template <int CQP, int KH, int KS, int FH, bool FW, bool SH>
__device__ __forceinline__ void gradw_wino_body(const float *__restrict__ gz, const float *__restrict__ x, float *__restrict__ part,
                                                int G, int CQ, int H, int W, int NS, int B, int WPG, int g, int wslot, bool fhh,
                                                float *part_dst, float *xt, float *gt)
{
    constexpr int MTB = CQP / 16, NSM = (CQP % 16) / 4, NF = FH == 2 ? 6 : 3, RS = KH + 1;
    static_assert(MTB >= 1 && (NSM == 0 || NSM == 2), "see the comment above");
    constexpr int NA = MTB + (NSM ? 1 : 0);                             // operand registers per side, frequency, k-step and row
    constexpr int NT = 4 * KS, SWC = 16 * KS;                           // tiles / columns of a strip (KS MFMA k-steps of 4 tiles)
    constexpr int XP = SWC + 4, GP = SWC + 4;                           // raw tiles: x [channel][4 halo + SWC], gz [channel][SWC + 4 pad]
                                                                        // (pitch = 4 mod 32 floats: 8 lanes' 16-byte reads cover the banks)
    constexpr int NPC = (NT * CQP + 63) / 64;                           // lane sets of (channel, tile): the loads
    constexpr int NXI = ((NT + 1) * CQP + 63) / 64;                     // dwordx4 loads of x per row (tiles + halo pieces)
    constexpr int PD = 2;                                               // rows of pieces in flight (4, 6, 8 measured: no faster)
    static_assert(PD % 2 == 0, "the piece sets rotate with the gz buffers");
    // SH: the two frequency halves of a strip are the two waves of ONE workgroup and share the raw tiles -- wave 0 loads the gz
    // pieces, wave 1 the x pieces, every byte crosses the fabric and L2 once; the tiles are double-buffered by row parity and one
    // barrier per row (behind filter row 0's MFMAs) says "row h+1 is in its tiles"
    constexpr bool LOADX = !SH || FH == 1, LOADG = !SH || FH == 0;
    constexpr int NLD = (LOADX ? NXI : 0) + (LOADG ? NPC : 0);          // loads of this wave per row
    constexpr int TST = CQP * XP + 4;                                   // floats per tile buffer (XP == GP)
    static_assert(XP == GP && (TST * 4) % 16 == 0, "");
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, n = lane & 15;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 4u;
    for (int i = lane; i < (SH ? 2 : 1) * TST; i += 64) { xt[i] = 0.f; gt[i] = 0.f; }     // (SH: both waves, the same zeros)
    auto pair_barrier = [&]() {
        if constexpr (SH) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // my tile writes are done (not __syncthreads: its fence would
            __builtin_amdgcn_s_barrier();                               //  also drain the pieces in flight)
            asm volatile("" ::: "memory");
        }
    };

    // accumulators per (filter row, frequency of this wave): tiles, o blocks x i tiles, i blocks x o tiles, blocks x blocks
    constexpr int MS = NSM ? MTB : 1;
    v4f accT[KH][NF][MTB][MTB], accO[KH][NF][2][MS], accI[KH][NF][2][MS], accB[KH][NF];
#pragma unroll
    for (int a = 0; a < KH; ++a)
#pragma unroll
        for (int f = 0; f < NF; ++f) {
#pragma unroll
            for (int mo = 0; mo < MTB; ++mo)
#pragma unroll
                for (int mi = 0; mi < MTB; ++mi) accT[a][f][mo][mi] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int mt = 0; mt < MS; ++mt) {
                    accO[a][f][sb][mt] = (v4f){0.f, 0.f, 0.f, 0.f};
                    accI[a][f][sb][mt] = (v4f){0.f, 0.f, 0.f, 0.f};
                }
            accB[a][f] = (v4f){0.f, 0.f, 0.f, 0.f};
        }
    // raw reads of lane (q, n) for operand register (k-step ks, set mt): channel ch(mt, n), CANONICAL tile 4 ks + q
    int grd[KS][NA], xrd[KS][NA];
#pragma unroll
    for (int mt = 0; mt < NA; ++mt) {
        const int chu = mt < MTB ? 16 * mt + n : 16 * MTB + 4 * ((n >> 3) & 1) + (n & 3);      // gz side
        const int chv = mt < MTB ? 16 * mt + n : 16 * MTB + 4 * ((n >> 2) & 1) + (n & 3);      // x side
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kc = 4 * ks + q;
            grd[ks][mt] = chu * GP + 4 * (FW ? NT - 1 - kc : kc);       // the piece as it came from memory
            // columns 4kc-2 .. 4kc+3: at 2 + 4kc .. (halo first); mirrored: at SWC + 1 - 4kc downwards (halo last) -- the address
            // is the 16-byte part (columns 4kc .. 4kc+3), the 8-byte part sits 2 floats below it / 4 floats above it
            xrd[ks][mt] = chv * XP + (FW ? SWC - 4 - 4 * kc : 4 + 4 * kc);
        }
    }
    // load lanes: set i, lane -> (channel c, memory tile k); lanes behind the bank park their piece in the scratch piece
    int lwg[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int t = 64 * i + lane;
        lwg[i] = t < NT * CQP ? (t / NT) * GP + 4 * (t % NT) : CQP * GP;
    }

    for (int u = wslot; u < B * NS; u += WPG) {
        const int b = u / NS, strip = u % NS;
        const size_t slab = ((size_t)b * G + g) * CQ * HW;
        auto rsrc = [&](const float *base, bool ok) {
            return __builtin_amdgcn_make_buffer_rsrc((void *)(base + slab), 0, ok ? (int)slab_bytes : 0, 0x00020000);
        };
        const int ms = FW ? W - SWC - strip * SWC : strip * SWC;         // memory column where the strip starts
        const int hm = FW ? ms + SWC : ms - 4;                          // ... and the piece holding the columns left of it
        unsigned lvx[NXI], lvg[NPC];
        int lwx[NXI];
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int t = 64 * i + lane;
            lvx[i] = OFF_BAD_CHANNEL;
            lwx[i] = CQP * XP;                                          // scratch piece behind the tile
            if (t < NT * CQP) {
                const int c = t / NT, k = t % NT;
                if (c < CQ && ms + 4 * k >= 0 && ms + 4 * k < W) lvx[i] = (unsigned)c * HW * 4u + (unsigned)(ms + 4 * k) * 4u;
                lwx[i] = c * XP + (FW ? 0 : 4) + 4 * k;
            } else if (t < (NT + 1) * CQP) {
                const int c = t - NT * CQP;
                if (c < CQ && hm >= 0 && hm < W) lvx[i] = (unsigned)c * HW * 4u + (unsigned)hm * 4u;
                lwx[i] = c * XP + (FW ? SWC : 0);
            }
        }
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int t = 64 * i + lane;
            lvg[i] = OFF_BAD_CHANNEL;
            if (t < NT * CQP) {
                const int c = t / NT, k = t % NT;
                if (c < CQ && ms + 4 * k >= 0 && ms + 4 * k < W) lvg[i] = (unsigned)c * HW * 4u + (unsigned)(ms + 4 * k) * 4u;
            }
        }
        auto rowbytes = [&](int h) { return (unsigned)((fhh ? H - 1 - h : h) * W) * 4u; };
        v4u LX[PD][NXI], LG[PD][NPC];                                   // pieces in flight, PD rows ahead, by row % PD
        auto issue = [&](auto par_c, int h) {
            constexpr int PAR = decltype(par_c)::value;
            const bool ok = h >= 0 && h < H;
            const __amdgpu_buffer_rsrc_t rx = rsrc(x, ok), rg = rsrc(gz, ok);
            const unsigned ro = ok ? rowbytes(h) : 0u;
            // asm: hipcc's own vmcnt bookkeeping does not survive the unrolled loop (it waits for all but the newest row, which
            // makes PD pointless); phase A waits for exactly the oldest row by count
            auto &lx = LX[PAR]; auto &lg = LG[PAR]; auto &ovx = lvx; auto &ovg = lvg;   // (clang: asm operands do not capture by themselves)
            if constexpr (LOADX) {
#pragma unroll
                for (int i = 0; i < NXI; ++i)
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(lx[i]) : "v"(ovx[i]), "s"(rx), "s"(ro) : "memory");
            }
            if constexpr (LOADG) {
#pragma unroll
                for (int i = 0; i < NPC; ++i)
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(lg[i]) : "v"(ovg[i]), "s"(rg), "s"(ro) : "memory");
            }
        };
        float GA[2][NF][KS][NA];                                        // gz operands [arriving / current]
        float XB[RS][NF][KS][NA];                                       // x operands by row slot
        v4f RG[KS][NA], RX4[KS][NA];                                    // raw operands of a row, between two phases
        v2f RX2[KS][NA];
#pragma unroll
        for (int sl = 0; sl < RS; ++sl)
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int mt = 0; mt < NA; ++mt) XB[sl][f][ks][mt] = 0.f;
        // The stage of a row, in three phases.  A: its pieces (asked for PD steps ago) go to the raw tiles as they came; the
        // pieces of row + PD are asked for.
        auto phase_a = [&](auto ln_c, auto tb_c, int row) {              // TB: the tile buffer of the row (its parity when SH)
            constexpr int LN = decltype(ln_c)::value, TB = decltype(tb_c)::value;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * NLD) : "memory");   // rows row+1 .. row+PD-1 stay in flight
            auto &lx = LX[LN]; auto &lg = LG[LN];
            if constexpr (LOADX) {
#pragma unroll
                for (int i = 0; i < NXI; ++i) asm volatile("" : "+v"(lx[i]));
#pragma unroll
                for (int i = 0; i < NXI; ++i) reinterpret_cast<v4u *>(xt + TB * TST)[lwx[i] >> 2] = LX[LN][i];
            }
            if constexpr (LOADG) {
#pragma unroll
                for (int i = 0; i < NPC; ++i) asm volatile("" : "+v"(lg[i]));
#pragma unroll
                for (int i = 0; i < NPC; ++i) reinterpret_cast<v4u *>(gt + TB * TST)[lwg[i] >> 2] = LG[LN][i];
            }
            issue(IC<LN>{}, row + PD);                                  // into the set just emptied
        };
        // B: every lane reads, for each of its operand registers, the gz piece and the six x columns of that (channel, tile)
        auto phase_b = [&](auto tb_c) {
            constexpr int TB = decltype(tb_c)::value;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int mt = 0; mt < NA; ++mt) {
                    RG[ks][mt] = *reinterpret_cast<const v4f *>(&gt[TB * TST + grd[ks][mt]]);
                    RX4[ks][mt] = *reinterpret_cast<const v4f *>(&xt[TB * TST + xrd[ks][mt]]);
                    RX2[ks][mt] = *reinterpret_cast<const v2f *>(&xt[TB * TST + xrd[ks][mt] + (FW ? 4 : -2)]);
                }
        };
        // C: U = A gz and V = B^T x in the operand layout: the registers the MFMAs of the next steps read
        auto phase_c = [&](auto sn_c, auto pn_c) {
            constexpr int SN = decltype(sn_c)::value, PN = decltype(pn_c)::value;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int mt = 0; mt < NA; ++mt) {
                    const v4f rg = RG[ks][mt], r4 = RX4[ks][mt];
                    const v2f r2 = RX2[ks][mt];
                    const float m0 = rg.x, m1 = rg.y, m2 = rg.z, m3 = rg.w, b0 = r4.x, b1 = r4.y, b2 = r4.z, b3 = r4.w, c0 = r2.x,
                                c1 = r2.y;
                    const float g0 = FW ? m3 : m0, g1 = FW ? m2 : m1, g2 = FW ? m1 : m2, g3 = FW ? m0 : m3;
                    // not mirrored: [d0 d1] = the 8-byte part, [d2 .. d5] the 16-byte part; mirrored: [d5 d4 d3 d2], [d1 d0]
                    const float d0 = FW ? c1 : c0, d1 = FW ? c0 : c1, d2 = FW ? b3 : b0, d3 = FW ? b2 : b1, d4 = FW ? b1 : b2,
                                d5 = FW ? b0 : b3;
                    if constexpr (FH != 1) {
                        const float e = g0 + g2, od = g1 + g3;
                        GA[PN][0][ks][mt] = g0; GA[PN][1][ks][mt] = e + od; GA[PN][2][ks][mt] = e - od;
                        const float pp = __builtin_fmaf(-2.25f, d2, d4), rr = __builtin_fmaf(-2.25f, d1, d3);
                        XB[SN][0][ks][mt] = __builtin_fmaf(2.25f, d0, __builtin_fmaf(-3.25f, d2, d4));
                        XB[SN][1][ks][mt] = pp + rr; XB[SN][2][ks][mt] = pp - rr;
                    }
                    if constexpr (FH != 0) {
                        const float e = __builtin_fmaf(2.25f, g2, g0), od = __builtin_fmaf(3.375f, g3, 1.5f * g1);
                        GA[PN][NF - 3][ks][mt] = e + od; GA[PN][NF - 2][ks][mt] = e - od; GA[PN][NF - 1][ks][mt] = g3;
                        const float ss = d4 - d2, uq = d3 - d1;
                        XB[SN][NF - 3][ks][mt] = __builtin_fmaf(1.5f, uq, ss); XB[SN][NF - 2][ks][mt] = __builtin_fmaf(-1.5f, uq, ss);
                        XB[SN][NF - 1][ks][mt] = __builtin_fmaf(2.25f, d1, __builtin_fmaf(-3.25f, d3, d5));
                    }
                }
        };
        // the MFMAs of gz row h (buffer PC) against the x row a rows above it (slot SC - a)
        auto mfmas = [&](auto sc_c, auto pc_c, auto a_c) {
            constexpr int SC = decltype(sc_c)::value, PC = decltype(pc_c)::value, a = decltype(a_c)::value;
            constexpr int sx = (SC + RS - a) % RS;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int f = 0; f < NF; ++f) {
#pragma unroll
                    for (int mo = 0; mo < MTB; ++mo)
#pragma unroll
                        for (int mi = 0; mi < MTB; ++mi)
                            accT[a][f][mo][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(GA[PC][f][ks][mo], XB[sx][f][ks][mi],
                                                                                      accT[a][f][mo][mi], 0, 0, 0);
                    if constexpr (NSM != 0) {
#pragma unroll
                        for (int mt = 0; mt < MTB; ++mt) {
                            gw_mma4<2, 0>(accO[a][f][0][mt], GA[PC][f][ks][MTB], XB[sx][f][ks][mt]);
                            gw_mma4<2, 2>(accO[a][f][1][mt], GA[PC][f][ks][MTB], XB[sx][f][ks][mt]);
                            gw_mma4<2, 0>(accI[a][f][0][mt], XB[sx][f][ks][MTB], GA[PC][f][ks][mt]);
                            gw_mma4<2, 1>(accI[a][f][1][mt], XB[sx][f][ks][MTB], GA[PC][f][ks][mt]);
                        }
                        gw_mma4<0, 0>(accB[a][f], GA[PC][f][ks][MTB], XB[sx][f][ks][MTB]);
                    }
                }
        };
        // step of row h (slot h % RS, gz buffer h & 1): the phases of row h+1 between the MFMA groups of row h
        // step of row h (slot h % RS, gz buffer h & 1): the phases of row h+1 between the MFMA groups of row h, so that the one
        // LDS round trip of the stage has a filter row's MFMAs to complete behind
        auto step = [&](auto i_c, int h) {
            constexpr int I = decltype(i_c)::value;
            constexpr int SC = I % RS, PC = I & 1;
            constexpr int TB = SH ? (I + 1) & 1 : 0;
            phase_a(IC<(I + 1) % PD>{}, IC<TB>{}, h + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(IC<SC>{}, IC<PC>{}, IC<0>{});
            __builtin_amdgcn_sched_barrier(0);
            pair_barrier();
            phase_b(IC<TB>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (KH > 1) mfmas(IC<SC>{}, IC<PC>{}, IC<1>{});
            __builtin_amdgcn_sched_barrier(0);
            phase_c(IC<(I + 1) % RS>{}, IC<PC ^ 1>{});
            __builtin_amdgcn_sched_barrier(0);
            [&]<int... A>(std::integer_sequence<int, A...>) {
                ((mfmas(IC<SC>{}, IC<PC>{}, IC<A + 2>{})), ...);
            }(std::make_integer_sequence<int, (KH > 2 ? KH - 2 : 0)>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        static_assert(PD <= 8, "");
        issue(IC<0>{}, 0);
        issue(IC<1 % PD>{}, 1);
        if constexpr (PD > 2) { issue(IC<2 % PD>{}, 2); issue(IC<3 % PD>{}, 3); }
        if constexpr (PD > 4) { issue(IC<4 % PD>{}, 4); issue(IC<5 % PD>{}, 5); }
        if constexpr (PD > 6) { issue(IC<6 % PD>{}, 6); issue(IC<7 % PD>{}, 7); }
        pair_barrier();                                                 // (SH: the partner is done with the tiles of the unit before)
        phase_a(IC<0>{}, IC<0>{}, 0);                                   // row 0 into slot 0 (asks for row PD)
        pair_barrier();
        phase_b(IC<0>{});
        phase_c(IC<0>{}, IC<0>{});
        // hipcc may park accumulators in scratch around this set-up; its wait for their reloads must not end up in the loop
        // (a vmcnt(0) there drains the pieces in flight every step): every accumulator is "used" here, before the loop
#pragma unroll
        for (int a = 0; a < KH; ++a)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
#pragma unroll
                for (int mo = 0; mo < MTB; ++mo)
#pragma unroll
                    for (int mi = 0; mi < MTB; ++mi) asm volatile("" : "+a"(accT[a][f][mo][mi]));
                if constexpr (NSM != 0) {
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                        for (int mt = 0; mt < MTB; ++mt) {
                            asm volatile("" : "+a"(accO[a][f][sb][mt]));
                            asm volatile("" : "+a"(accI[a][f][sb][mt]));
                        }
                    asm volatile("" : "+a"(accB[a][f]));
                }
            }
        constexpr int UN0 = (RS % 2 == 0) ? RS : 2 * RS, UN = UN0 % PD == 0 ? UN0 : UN0 * PD / 2;
        static_assert(UN % PD == 0 && UN % RS == 0 && UN % 2 == 0, "");
        for (int h0 = 0; h0 < H; h0 += UN) {
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ((h0 + I < H ? step(IC<I>{}, h0 + I) : (void)0), ...);
            }(std::make_integer_sequence<int, UN>{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the rows asked for beyond the image: their registers are free only now
    }
    float *dst = part_dst;
#pragma unroll
    for (int a = 0; a < KH; ++a)
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            float *d = dst + (size_t)(f * KH + a) * CQP * CQP;
#pragma unroll
            for (int mo = 0; mo < MTB; ++mo)
#pragma unroll
                for (int mi = 0; mi < MTB; ++mi) {
                    const v4f v = accT[a][f][mo][mi];
                    const float v0 = v.x, v1 = v.y, v2 = v.z, v3 = v.w;
                    float *e = d + (16 * mo + 4 * q) * CQP + 16 * mi + n;
                    e[0] = v0; e[CQP] = v1; e[2 * CQP] = v2; e[3 * CQP] = v3;
                }
            if constexpr (NSM != 0) {
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int mt = 0; mt < MTB; ++mt) {
                        const float vo = finc_block_reduce(accO[a][f][sb][mt]);     // lane row q: o = 16 MTB + 4 sb + q, i = 16 mt + n
                        d[(16 * MTB + 4 * sb + q) * CQP + 16 * mt + n] = vo;
                        const float vi = finc_block_reduce(accI[a][f][sb][mt]);     // lane row q: i = 16 MTB + 4 sb + q, o = 16 mt + n
                        d[(16 * mt + n) * CQP + 16 * MTB + 4 * sb + q] = vi;
                    }
                const float vb = finc_block_reduce(accB[a][f]);                     // block p = n >> 2: o block p >> 1, i block p & 1
                d[(16 * MTB + 4 * (n >> 3) + q) * CQP + 16 * MTB + 4 * ((n >> 2) & 1) + (n & 3)] = vb;
            }
        }
}

template <int CQP, int KH, int FS, int KS>
__global__ __launch_bounds__(64 * FS) __attribute__((amdgpu_waves_per_eu(1, 1))) void finc_gradw_wino_kernel(
    const float *__restrict__ gz, const float *__restrict__ x, float *__restrict__ part, int G, int CQ, int H, int W, int NS, int B,
    int WPG, unsigned orient)
{
    constexpr int NF = 6 / FS, TST = CQP * (16 * KS + 4) + 4;
    __shared__ __attribute__((aligned(16))) float xt[FS * TST];         // (FS = 2: the pair's two tile buffers)
    __shared__ __attribute__((aligned(16))) float gt[FS * TST];
    const int wslot = blockIdx.x % WPG, g = blockIdx.x / WPG;
    const int fh = threadIdx.x >> 6;                                     // FS = 2: wave 0 has f0..f2, wave 1 f3..f5
    const unsigned o = finc_group_orient(orient, g);
    const bool fhh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    float *dst = part + ((size_t)(g * FS + fh) * WPG + wslot) * (NF * KH * CQP * CQP);
#define FINC_GW_BODY(FH, FW) \
    gradw_wino_body<CQP, KH, KS, FH, FW, FS == 2>(gz, x, part, G, CQ, H, W, NS, B, WPG, g, wslot, fhh, dst, xt, gt)
    if constexpr (FS == 1) {
        if (fw) FINC_GW_BODY(2, true); else FINC_GW_BODY(2, false);
    } else if (fh == 0) {
        if (fw) FINC_GW_BODY(0, true); else FINC_GW_BODY(0, false);
    } else {
        if (fw) FINC_GW_BODY(1, true); else FINC_GW_BODY(1, false);
    }
#undef FINC_GW_BODY
}

// gw[g][o][i][KH-1-a][k] = G^T applied to the sums of the partials (see above; k = 2 - b).  Same fixed-order slice scheme as
// gradw_reduce_kernel: a block owns 32 consecutive (a, o, i) entries, its 8 thread groups sum 8 interleaved slices.
__global__ __launch_bounds__(256) void gradw_wino_reduce_kernel(const float *__restrict__ part, float *__restrict__ gw, int Cq,
                                                                int CQP, int KH, int FS, int WPG)
{
    __shared__ float slice[8][6][32];
    const int g = blockIdx.y;
    const int NF = 6 / FS;
    const int per = NF * KH * CQP * CQP, plane = KH * CQP * CQP;
    const int el = threadIdx.x & 31, j = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;                    // (a, o, i) flat; plane is a multiple of 32
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float *p[6];
#pragma unroll
    for (int f = 0; f < 6; ++f) p[f] = part + ((size_t)(g * FS + f / NF) * WPG) * per + (size_t)(f % NF) * plane + e;
    // (w outside, f inside, unrolled: 24 independent loads in flight per trip -- one accumulator chain per loop was a chain of
    // memory round trips: 30 us for 21 MB)
#pragma unroll 4
    for (int w = j; w < WPG; w += 8) {
#pragma unroll
        for (int f = 0; f < 6; ++f) s[f] += p[f][(size_t)w * per];
    }
#pragma unroll
    for (int f = 0; f < 6; ++f) slice[j][f][el] = s[f];
    __syncthreads();
    if (j != 0) return;
#pragma unroll
    for (int f = 0; f < 6; ++f)
        s[f] = ((slice[0][f][el] + slice[1][f][el]) + (slice[2][f][el] + slice[3][f][el])) +
               ((slice[4][f][el] + slice[5][f][el]) + (slice[6][f][el] + slice[7][f][el]));
    const int ic = e % CQP, oc = (e / CQP) % CQP, a = e / (CQP * CQP);
    if (oc >= Cq || ic >= Cq) return;
    const float s12 = s[1] + s[2], d12 = s[1] - s[2], s34 = s[3] + s[4], d34 = s[3] - s[4];
    const float k0 = s[0] * (1.f / 2.25f) - 0.4f * s12 + (8.f / 45.f) * s34;
    const float k1 = -0.4f * d12 + (4.f / 15.f) * d34;
    const float k2 = 0.4f * (s34 - s12) + s[5];
    float *o = gw + (((size_t)(g * Cq + oc) * Cq + ic) * KH + (KH - 1 - a)) * 3;
    o[0] = k0;                                             // k = 0: tap b = 2
    o[1] = k1;
    o[2] = (a == 0 && ic >= oc) ? 0.f : k2;                // the corner tap's mask (b = 0; PaddedConv2d.reset_gradients)
}

// -----------------------------------------------------------------------------------------------
// The Winograd grad-weight for the banks one wave cannot hold (Cq > 32 at 3x3, Cq > 16 at 5x5 -- the c5 bank): ONE 16x16 (o, i)
// tile pair per wave as in finc_gradw_tiled_kernel, six frequency accumulators per filter row.  3-wide filters: F(4,3)
// transposed as above (tiles of 4 columns, 32-column strips); 5-wide filters: F(2,5) of finc_wino5.hip transposed -- tiles of
// TWO columns, d_j = x[2t - 4 + j], six products per tile, filter row and (o, i) instead of ten (16-column strips):
//     U = A gz :  g0 | g0 + g1 | g0 - g1 | g0 + g1/2 | g0 - g1/2 | g1
//     V = B^T x:  d0 - 5 d2 + 4 d4 | p + r | p - r | 2 s + u | 2 s - u | d1 - 5 d3 + 4 d5      p = 4 d4 - d2, r = 4 d3 - d1, s = d2 - d4, u = d1 - d3
//     dW[b = 4 - k]:  k0 = M0 + S/6 + 4 T/3;  k1 = D/6 + 2 E/3;  k2 = S/6 + T/3;  k3 = D/6 + E/6;  k4 = S/6 + T/12 + M5/4
//                     S = M1 + M2, D = M1 - M2, T = M3 + M4, E = M3 - M4
// Same stage as the pair kernel (raw tiles, operands transformed in place, phases between the MFMA groups, asm loads), one wave
// per workgroup.  blockIdx = (((g * MTT + mo) * MTT + mi) * WPG + wslot); partials part[blockIdx][f][a][o][i] (16 x 16, complete).
// -----------------------------------------------------------------------------------------------
template <int KH, int KW, bool FW>
__device__ __forceinline__ void gradw_winot_body(const float *__restrict__ gz, const float *__restrict__ x, float *__restrict__ dst,
                                                 int G, int CQ, int H, int W, int NS, int B, int WPG, int g, int wslot, int cg, int cx,
                                                 bool fhh, float *xt, float *gt)
{
    static_assert(KW == 3 || KW == 5, "");
    constexpr int KS = 2, TW = KW == 3 ? 4 : 2, HL = 6 - TW, NTW = 4 * KS, SWC = TW * NTW, NP = SWC / 4, RS = KH + 1;
    constexpr int XP = SWC + 4, GP = SWC + 4;
    constexpr int NGI = (16 * NP + 63) / 64, NXI = (16 * (NP + 1) + 63) / 64, NLD = NGI + NXI;
    constexpr int PD = 2;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x;
    const int q = lane >> 4, n = lane & 15;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 4u;
    for (int i = lane; i < 16 * XP + 4; i += 64) { xt[i] = 0.f; gt[i] = 0.f; }
    v4f acc[KH][6];
#pragma unroll
    for (int a = 0; a < KH; ++a)
#pragma unroll
        for (int f = 0; f < 6; ++f) acc[a][f] = (v4f){0.f, 0.f, 0.f, 0.f};
    // raw reads of lane (q, n): channel n, canonical tile 4 ks + q: the gz tile as it lies in memory, the six x columns ascending
    int grd[KS], xrd[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int kc = 4 * ks + q;
        grd[ks] = n * GP + (FW ? SWC - TW - TW * kc : TW * kc);
        xrd[ks] = n * XP + (FW ? SWC - 6 - TW * kc + HL : 4 + TW * kc - HL);
    }
    int lwg[NGI];
#pragma unroll
    for (int i = 0; i < NGI; ++i) {
        const int t = 64 * i + lane;
        lwg[i] = t < 16 * NP ? (t / NP) * GP + 4 * (t % NP) : 16 * GP;
    }
    for (int u = wslot; u < B * NS; u += WPG) {
        const int b = u / NS, strip = u % NS;
        const size_t slab = ((size_t)b * G + g) * CQ * HW;
        auto rsrc = [&](const float *base, bool ok) {
            return __builtin_amdgcn_make_buffer_rsrc((void *)(base + slab), 0, ok ? (int)slab_bytes : 0, 0x00020000);
        };
        const int ms = FW ? W - SWC - strip * SWC : strip * SWC;
        const int hm = FW ? ms + SWC : ms - 4;
        unsigned lvx[NXI], lvg[NGI];
        int lwx[NXI];
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int t = 64 * i + lane;
            lvx[i] = OFF_BAD_CHANNEL;
            lwx[i] = 16 * XP;
            if (t < 16 * NP) {
                const int c = t / NP, k = t % NP;
                if (cx + c < CQ && ms + 4 * k >= 0 && ms + 4 * k < W) lvx[i] = (unsigned)(cx + c) * HW * 4u + (unsigned)(ms + 4 * k) * 4u;
                lwx[i] = c * XP + (FW ? 0 : 4) + 4 * k;
            } else if (t < 16 * (NP + 1)) {
                const int c = t - 16 * NP;
                if (cx + c < CQ && hm >= 0 && hm < W) lvx[i] = (unsigned)(cx + c) * HW * 4u + (unsigned)hm * 4u;
                lwx[i] = c * XP + (FW ? SWC : 0);
            }
        }
#pragma unroll
        for (int i = 0; i < NGI; ++i) {
            const int t = 64 * i + lane;
            lvg[i] = OFF_BAD_CHANNEL;
            if (t < 16 * NP) {
                const int c = t / NP, k = t % NP;
                if (cg + c < CQ && ms + 4 * k >= 0 && ms + 4 * k < W) lvg[i] = (unsigned)(cg + c) * HW * 4u + (unsigned)(ms + 4 * k) * 4u;
            }
        }
        auto rowbytes = [&](int h) { return (unsigned)((fhh ? H - 1 - h : h) * W) * 4u; };
        v4u LX[PD][NXI], LG[PD][NGI];
        auto issue = [&](auto par_c, int h) {
            constexpr int PAR = decltype(par_c)::value;
            const bool ok = h >= 0 && h < H;
            const __amdgpu_buffer_rsrc_t rx = rsrc(x, ok), rg = rsrc(gz, ok);
            const unsigned ro = ok ? rowbytes(h) : 0u;
            auto &lx = LX[PAR]; auto &lg = LG[PAR]; auto &ovx = lvx; auto &ovg = lvg;   // (clang: asm operands do not capture by themselves)
#pragma unroll
            for (int i = 0; i < NXI; ++i)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(lx[i]) : "v"(ovx[i]), "s"(rx), "s"(ro) : "memory");
#pragma unroll
            for (int i = 0; i < NGI; ++i)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(lg[i]) : "v"(ovg[i]), "s"(rg), "s"(ro) : "memory");
        };
        float GA[2][6][KS], XB[RS][6][KS];
        float RGZ[KS][TW];
        v2f RXR[KS][3];
#pragma unroll
        for (int sl = 0; sl < RS; ++sl)
#pragma unroll
            for (int f = 0; f < 6; ++f)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) XB[sl][f][ks] = 0.f;
        auto phase_a = [&](auto ln_c, int row) {
            constexpr int LN = decltype(ln_c)::value;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * NLD) : "memory");
            auto &lx = LX[LN]; auto &lg = LG[LN];
#pragma unroll
            for (int i = 0; i < NXI; ++i) asm volatile("" : "+v"(lx[i]));
#pragma unroll
            for (int i = 0; i < NGI; ++i) asm volatile("" : "+v"(lg[i]));
#pragma unroll
            for (int i = 0; i < NXI; ++i) reinterpret_cast<v4u *>(xt)[lwx[i] >> 2] = LX[LN][i];
#pragma unroll
            for (int i = 0; i < NGI; ++i) reinterpret_cast<v4u *>(gt)[lwg[i] >> 2] = LG[LN][i];
            issue(IC<LN>{}, row + PD);
        };
        auto phase_b = [&]() {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if constexpr (TW == 4) {
                    const v4f r = *reinterpret_cast<const v4f *>(&gt[grd[ks]]);
                    RGZ[ks][0] = r.x; RGZ[ks][1] = r.y; RGZ[ks][2] = r.z; RGZ[ks][3] = r.w;
                } else {
                    const v2f r = *reinterpret_cast<const v2f *>(&gt[grd[ks]]);
                    RGZ[ks][0] = r.x; RGZ[ks][1] = r.y;
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) RXR[ks][j] = *reinterpret_cast<const v2f *>(&xt[xrd[ks] + 2 * j]);
            }
        };
        auto phase_c = [&](auto sn_c, auto pn_c) {
            constexpr int SN = decltype(sn_c)::value, PN = decltype(pn_c)::value;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float r[6];
#pragma unroll
                for (int j = 0; j < 3; ++j) { const v2f t = RXR[ks][j]; r[2 * j] = t.x; r[2 * j + 1] = t.y; }
                const float d0 = r[FW ? 5 : 0], d1 = r[FW ? 4 : 1], d2 = r[FW ? 3 : 2], d3 = r[FW ? 2 : 3], d4 = r[FW ? 1 : 4],
                            d5 = r[FW ? 0 : 5];
                if constexpr (KW == 3) {
                    const float g0 = RGZ[ks][FW ? 3 : 0], g1 = RGZ[ks][FW ? 2 : 1], g2 = RGZ[ks][FW ? 1 : 2], g3 = RGZ[ks][FW ? 0 : 3];
                    const float e = g0 + g2, od = g1 + g3;
                    const float e2 = __builtin_fmaf(2.25f, g2, g0), o2 = __builtin_fmaf(3.375f, g3, 1.5f * g1);
                    GA[PN][0][ks] = g0; GA[PN][1][ks] = e + od; GA[PN][2][ks] = e - od;
                    GA[PN][3][ks] = e2 + o2; GA[PN][4][ks] = e2 - o2; GA[PN][5][ks] = g3;
                    const float pp = __builtin_fmaf(-2.25f, d2, d4), rr = __builtin_fmaf(-2.25f, d1, d3);
                    const float ss = d4 - d2, uq = d3 - d1;
                    XB[SN][0][ks] = __builtin_fmaf(2.25f, d0, __builtin_fmaf(-3.25f, d2, d4));
                    XB[SN][1][ks] = pp + rr; XB[SN][2][ks] = pp - rr;
                    XB[SN][3][ks] = __builtin_fmaf(1.5f, uq, ss); XB[SN][4][ks] = __builtin_fmaf(-1.5f, uq, ss);
                    XB[SN][5][ks] = __builtin_fmaf(2.25f, d1, __builtin_fmaf(-3.25f, d3, d5));
                } else {
                    const float g0 = RGZ[ks][FW ? 1 : 0], g1 = RGZ[ks][FW ? 0 : 1];
                    const float hg = 0.5f * g1;
                    GA[PN][0][ks] = g0; GA[PN][1][ks] = g0 + g1; GA[PN][2][ks] = g0 - g1;
                    GA[PN][3][ks] = g0 + hg; GA[PN][4][ks] = g0 - hg; GA[PN][5][ks] = g1;
                    const float pp = __builtin_fmaf(4.f, d4, -d2), rr = __builtin_fmaf(4.f, d3, -d1);
                    const float ss = d2 - d4, uq = d1 - d3;
                    XB[SN][0][ks] = __builtin_fmaf(4.f, d4, __builtin_fmaf(-5.f, d2, d0));
                    XB[SN][1][ks] = pp + rr; XB[SN][2][ks] = pp - rr;
                    XB[SN][3][ks] = __builtin_fmaf(2.f, ss, uq); XB[SN][4][ks] = __builtin_fmaf(2.f, ss, -uq);
                    XB[SN][5][ks] = __builtin_fmaf(4.f, d5, __builtin_fmaf(-5.f, d3, d1));
                }
            }
        };
        auto mfmas = [&](auto sc_c, auto pc_c, auto a_c) {
            constexpr int SC = decltype(sc_c)::value, PC = decltype(pc_c)::value, a = decltype(a_c)::value;
            constexpr int sx = (SC + RS - a) % RS;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int f = 0; f < 6; ++f)
                    acc[a][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(GA[PC][f][ks], XB[sx][f][ks], acc[a][f], 0, 0, 0);
        };
        auto step = [&](auto i_c, int h) {
            constexpr int I = decltype(i_c)::value;
            constexpr int SC = I % RS, PC = I & 1;
            phase_a(IC<(I + 1) % PD>{}, h + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(IC<SC>{}, IC<PC>{}, IC<0>{});
            __builtin_amdgcn_sched_barrier(0);
            phase_b();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (KH > 1) mfmas(IC<SC>{}, IC<PC>{}, IC<1>{});
            __builtin_amdgcn_sched_barrier(0);
            phase_c(IC<(I + 1) % RS>{}, IC<PC ^ 1>{});
            __builtin_amdgcn_sched_barrier(0);
            [&]<int... A>(std::integer_sequence<int, A...>) {
                ((mfmas(IC<SC>{}, IC<PC>{}, IC<A + 2>{})), ...);
            }(std::make_integer_sequence<int, (KH > 2 ? KH - 2 : 0)>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        issue(IC<0>{}, 0);
        issue(IC<1>{}, 1);
        phase_a(IC<0>{}, 0);
        phase_b();
        phase_c(IC<0>{}, IC<0>{});
#pragma unroll
        for (int a = 0; a < KH; ++a)
#pragma unroll
            for (int f = 0; f < 6; ++f) asm volatile("" : "+a"(acc[a][f]));      // (see the pair kernel: no compiler wait inside the loop)
        constexpr int UN = (RS % 2 == 0) ? RS : 2 * RS;
        for (int h0 = 0; h0 < H; h0 += UN) {
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ((h0 + I < H ? step(IC<I>{}, h0 + I) : (void)0), ...);
            }(std::make_integer_sequence<int, UN>{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int a = 0; a < KH; ++a)
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            const v4f v = acc[a][f];
            const float v0 = v.x, v1 = v.y, v2 = v.z, v3 = v.w;
            float *e = dst + (f * KH + a) * 256 + (4 * q) * 16 + n;
            e[0] = v0; e[16] = v1; e[32] = v2; e[48] = v3;
        }
}

template <int KH, int KW>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void finc_gradw_winot_kernel(
    const float *__restrict__ gz, const float *__restrict__ x, float *__restrict__ part, int G, int CQ, int H, int W, int NS, int B,
    int WPG, unsigned orient, int MTT)
{
    constexpr int SWC = (KW == 3 ? 4 : 2) * 8;
    __shared__ __attribute__((aligned(16))) float xt[16 * (SWC + 4) + 4];
    __shared__ __attribute__((aligned(16))) float gt[16 * (SWC + 4) + 4];
    int bi = blockIdx.x;
    const int wslot = bi % WPG; bi /= WPG;
    const int mi = bi % MTT; bi /= MTT;
    const int mo = bi % MTT;
    const int g = bi / MTT;
    const unsigned o = finc_group_orient(orient, g);
    const bool fhh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    float *dst = part + (size_t)blockIdx.x * (6 * KH * 256);
    if (fw) gradw_winot_body<KH, KW, true>(gz, x, dst, G, CQ, H, W, NS, B, WPG, g, wslot, 16 * mo, 16 * mi, fhh, xt, gt);
    else gradw_winot_body<KH, KW, false>(gz, x, dst, G, CQ, H, W, NS, B, WPG, g, wslot, 16 * mo, 16 * mi, fhh, xt, gt);
}

// G^T over the summed partials of finc_gradw_winot_kernel; entries e = ((mo * MTT + mi) * KH + a) * 256 + o * 16 + i per group.
template <int KW>
__global__ __launch_bounds__(256) void gradw_winot_reduce_kernel(const float *__restrict__ part, float *__restrict__ gw, int Cq, int KH,
                                                                 int MTT, int WPG)
{
    __shared__ float slice[8][6][32];
    const int g = blockIdx.y;
    const int per = 6 * KH * 256;
    const int el = threadIdx.x & 31, j = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    const int oi = e & 255, a = (e >> 8) % KH, tile = (e >> 8) / KH;       // tile = mo * MTT + mi
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float *p = part + ((size_t)(g * MTT * MTT + tile) * WPG) * per + (size_t)a * 256 + oi;
#pragma unroll 4
    for (int w = j; w < WPG; w += 8) {
#pragma unroll
        for (int f = 0; f < 6; ++f) s[f] += p[(size_t)w * per + f * KH * 256];
    }
#pragma unroll
    for (int f = 0; f < 6; ++f) slice[j][f][el] = s[f];
    __syncthreads();
    if (j != 0) return;
#pragma unroll
    for (int f = 0; f < 6; ++f)
        s[f] = ((slice[0][f][el] + slice[1][f][el]) + (slice[2][f][el] + slice[3][f][el])) +
               ((slice[4][f][el] + slice[5][f][el]) + (slice[6][f][el] + slice[7][f][el]));
    const int oc = 16 * (tile / MTT) + (oi >> 4), ic = 16 * (tile % MTT) + (oi & 15);
    if (oc >= Cq || ic >= Cq) return;
    const float s12 = s[1] + s[2], d12 = s[1] - s[2], s34 = s[3] + s[4], d34 = s[3] - s[4];
    float k[KW];
    if constexpr (KW == 3) {
        k[0] = s[0] * (1.f / 2.25f) - 0.4f * s12 + (8.f / 45.f) * s34;
        k[1] = -0.4f * d12 + (4.f / 15.f) * d34;
        k[2] = 0.4f * (s34 - s12) + s[5];
    } else {
        const float s6 = s12 * (1.f / 6.f), d6 = d12 * (1.f / 6.f);
        k[0] = s[0] + s6 + (4.f / 3.f) * s34;
        k[1] = d6 + (2.f / 3.f) * d34;
        k[2] = s6 + (1.f / 3.f) * s34;
        k[3] = d6 + (1.f / 6.f) * d34;
        k[4] = s6 + (1.f / 12.f) * s34 + 0.25f * s[5];
    }
    if (a == 0 && ic >= oc) k[KW - 1] = 0.f;                // the corner tap's mask (b = 0; PaddedConv2d.reset_gradients)
    float *o = gw + (((size_t)(g * Cq + oc) * Cq + ic) * KH + (KH - 1 - a)) * KW;
#pragma unroll
    for (int t = 0; t < KW; ++t) o[t] = k[t];
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
// the Winograd form: 3x3 banks of 16 (one wave holds all six frequencies), 24 and 32 channels (two waves, three each)
struct GradwWinoInst {
    int cqp, kh, fs;
    gradw_fn fn, fn32;      // strips of 16 columns / of 32 (maps at least 32 wide: 128-byte rows per request, half the steps)
};
const GradwWinoInst g_gradw_wino[] = {
    {16, 3, 1, finc_gradw_wino_kernel<16, 3, 1, 1>, finc_gradw_wino_kernel<16, 3, 1, 2>},
    {24, 3, 2, finc_gradw_wino_kernel<24, 3, 2, 1>, finc_gradw_wino_kernel<24, 3, 2, 2>},
    {32, 3, 2, finc_gradw_wino_kernel<32, 3, 2, 1>, finc_gradw_wino_kernel<32, 3, 2, 2>},
};
static int gradw_wino_strip(const FincShape &s)
{
    static const char *force = finc_env("FINC_GRADW_WINO_STRIP");             // experiment switch: 16 / 32
    if (force) return atoi(force) == 32 ? 32 : 16;
    return s.W >= 32 ? 32 : 16;
}
const GradwWinoInst *find_gradw_wino(const FincShape &s)
{
    static const bool off = finc_env("FINC_GRADW_NO_WINO") != nullptr;      // experiment switch
    static const char *pmax = finc_env("FINC_GRADW_WINO_PAIR_MAX");         // experiment switch: larger banks go to the tile-pair form
    if (off || s.KW != 3 || s.W % 4 != 0 || s.Cq <= 12 || (pmax && s.Cq > atoi(pmax))) return nullptr;
    const GradwWinoInst *best = nullptr;
    for (const GradwWinoInst &i : g_gradw_wino)
        if (i.cqp >= s.Cq && i.kh == s.KH && (!best || i.cqp < best->cqp)) best = &i;
    return best;
}
// the smallest compiled bank that holds Cq channels (every kernel here tests `channel < CQ` per lane: any padding is fine)
// the tile-pair kernels take any tile count: the filter shapes they are compiled for
gradw_tiled_fn tiled_for(int KH, int KW)
{
    switch (KH * 8 + KW) {
    case 2 * 8 + 2: return finc_gradw_tiled_kernel<2, 2>;
    case 3 * 8 + 3: return finc_gradw_tiled_kernel<3, 3>;
    case 4 * 8 + 4: return finc_gradw_tiled_kernel<4, 4>;
    case 5 * 8 + 5: return finc_gradw_tiled_kernel<5, 5>;
    case 2 * 8 + 3: return finc_gradw_tiled_kernel<2, 3>;
    case 3 * 8 + 2: return finc_gradw_tiled_kernel<3, 2>;
    case 3 * 8 + 5: return finc_gradw_tiled_kernel<3, 5>;
    case 5 * 8 + 3: return finc_gradw_tiled_kernel<5, 3>;
    }
    return nullptr;
}

const GradwInst *find_gradw(int Cq, int KH, int KW)
{
    const GradwInst *best = nullptr;
    for (const GradwInst &i : g_gradw)
        if (i.cqp >= Cq && i.kh == KH && i.kw == KW && (!best || i.cqp < best->cqp)) best = &i;
    if (best) return best;
    // banks beyond the table (the forward / grad-input of these run on the streaming-bank kernel, finc_stream.hip): one
    // (o, i) tile pair per workgroup, whatever the number of tiles
    const gradw_tiled_fn t = tiled_for(KH, KW);
    if (!t || Cq < 1 || Cq > FINC_MAX_CQ) return nullptr;
    static thread_local GradwInst gen;
    const int cqp = (Cq + 15) / 16 * 16;
    gen = GradwInst{cqp, KH, KW, cqp / 16, nullptr, nullptr, t};
    return &gen;
}

} // namespace

static int gradw_wpg(const FincShape &s)
{
    const int units = s.B * ((s.W + 15) / 16);
    int w = 1024 / s.G;                                   // one wave per SIMD over all groups (G = 4: 256 per group)
    if (w < 1) w = 1;
    return units < w ? units : w;
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
// Winograd form: G * FS * WPG workgroups of one wave, one per SIMD
static int gradw_wpg_wino(const FincShape &s, int fs)
{
    const int sw = gradw_wino_strip(s);
    const int units = s.B * ((s.W + sw - 1) / sw);
    int w = 1024 / (s.G * fs);                            // one wave per SIMD over all groups (G = 4, two waves per strip: 128)
    if (w < 1) w = 1;
    return units < w ? units : w;
}
static size_t gradw_wino_bytes(const FincShape &s, const GradwWinoInst *w)
{
    return (size_t)s.G * w->fs * gradw_wpg_wino(s, w->fs) * (6 / w->fs) * s.KH * w->cqp * w->cqp * sizeof(float);
}
static bool gradw_use_tiled(const GradwInst *i, const FincShape &s) { return i && !i->gw && i->gw_tiled && s.W % 4 == 0; }
// ... and its Winograd form: 3x3 (strips of 32 columns) and 5x5 (strips of 16)
static int gradw_winot_strip(const FincShape &s) { return s.KW == 3 ? 32 : 16; }
static bool gradw_use_winot(const GradwInst *i, const FincShape &s)
{
    static const bool off = finc_env("FINC_GRADW_NO_WINO") != nullptr;
    if (off || !i || s.W % 4 != 0 || s.W < gradw_winot_strip(s)) return false;
    if (s.KH == 5 && s.KW == 5 && s.Cq > 12 && s.Cq <= 16) return true;   // one full tile: beats the staged direct kernel too
    if (s.KH == 3 && s.KW == 3 && s.Cq > 12 && !find_gradw_wino(s)) return true;   // (only with FINC_GRADW_WINO_PAIR_MAX)
    // 9 .. 12 channels on one (3/4 full) tile pair: from a chip's worth of strips on (C = 48, 64x64, B = 256: 193 -> 125 us; c2's 64
    // images stay on the staged direct kernel: 26 against 30 us)
    if (s.KH == 3 && s.KW == 3 && s.Cq > 8 && s.Cq <= 12 && (long long)s.B * ((s.W + 31) / 32) >= 256) return true;
    return gradw_use_tiled(i, s) && ((s.KH == 3 && s.KW == 3) || (s.KH == 5 && s.KW == 5));
}
static int gradw_wpg_winot(const FincShape &s, int mtt)
{
    const int sw = gradw_winot_strip(s);
    const int units = s.B * ((s.W + sw - 1) / sw);
    int w = 2048 / (s.G * mtt * mtt);
    if (w < 1) w = 1;
    if (w > 256) w = 256;
    return units < w ? units : w;
}

size_t finc_gradw_workspace_bytes(const FincShape &s)
{
    const GradwInst *i = find_gradw(s.Cq, s.KH, s.KW);
    if (!i || !finc_conv_supported(s.Cq, s.H, s.W, s.KH, s.KW)) return 0;
    // room for the form the launch picks AND the one it falls back to on unaligned activations
    size_t need = 0;
    auto grow = [&](size_t n) { need = n > need ? n : need; };
    if (const GradwWinoInst *w = find_gradw_wino(s)) grow(gradw_wino_bytes(s, w));
    if (gradw_use_winot(i, s)) grow((size_t)s.G * gradw_wpg_winot(s, i->mtg) * 6 * s.KH * i->mtg * i->mtg * 256 * sizeof(float));
    if (gradw_use_tiled(i, s)) grow((size_t)s.G * gradw_wpg_tiled(s, i->mtg) * s.KH * s.KW * i->mtg * i->mtg * 256 * sizeof(float));
    else if (i->gw) grow((size_t)s.G * gradw_wpg(s) * s.KH * s.KW * i->mtg * i->mtg * 256 * sizeof(float));
    return need;
}

// FINC_ERR_UNSUPPORTED: no MFMA grad-weight kernel for this call (the caller falls back to the direct kernel)
int finc_gradw_launch(const float *gz, const float *x, float *gw, void *workspace, const FincShape &s, hipStream_t st)
{
    const GradwInst *i = find_gradw(s.Cq, s.KH, s.KW);
    if (!i) return FINC_ERR_UNSUPPORTED;
    const int NS = (s.W + 15) / 16;
    const bool aligned16 = (((uintptr_t)gz | (uintptr_t)x) & 15) == 0;
    int WPG;
    if (const GradwWinoInst *w = aligned16 ? find_gradw_wino(s) : nullptr) {
        WPG = gradw_wpg_wino(s, w->fs);
        const int sw = gradw_wino_strip(s);
        hipLaunchKernelGGL(sw == 32 ? w->fn32 : w->fn, dim3(s.G * WPG), dim3(64 * w->fs), 0, st, gz, x, (float *)workspace, s.G,
                           s.Cq, s.H, s.W, (s.W + sw - 1) / sw, s.B, WPG, s.orient);
        FINC_CHECK_LAUNCH();
        hipLaunchKernelGGL(gradw_wino_reduce_kernel, dim3(s.KH * w->cqp * w->cqp / 32, s.G), dim3(256), 0, st,
                           (const float *)workspace, gw, s.Cq, w->cqp, s.KH, w->fs, WPG);
        FINC_CHECK_LAUNCH();
        return FINC_OK;
    }
    if (aligned16 && gradw_use_winot(i, s)) {
        WPG = gradw_wpg_winot(s, i->mtg);
        const int sw = gradw_winot_strip(s);
        const dim3 grid(s.G * i->mtg * i->mtg * WPG);
        const int entries = i->mtg * i->mtg * s.KH * 256;
        if (s.KW == 3) {
            hipLaunchKernelGGL((finc_gradw_winot_kernel<3, 3>), grid, dim3(64), 0, st, gz, x, (float *)workspace, s.G, s.Cq, s.H, s.W,
                               (s.W + sw - 1) / sw, s.B, WPG, s.orient, i->mtg);
            FINC_CHECK_LAUNCH();
            hipLaunchKernelGGL(gradw_winot_reduce_kernel<3>, dim3(entries / 32, s.G), dim3(256), 0, st, (const float *)workspace, gw, s.Cq,
                               s.KH, i->mtg, WPG);
        } else {
            hipLaunchKernelGGL((finc_gradw_winot_kernel<5, 5>), grid, dim3(64), 0, st, gz, x, (float *)workspace, s.G, s.Cq, s.H, s.W,
                               (s.W + sw - 1) / sw, s.B, WPG, s.orient, i->mtg);
            FINC_CHECK_LAUNCH();
            hipLaunchKernelGGL(gradw_winot_reduce_kernel<5>, dim3(entries / 32, s.G), dim3(256), 0, st, (const float *)workspace, gw, s.Cq,
                               s.KH, i->mtg, WPG);
        }
        FINC_CHECK_LAUNCH();
        return FINC_OK;
    }
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
// 0 direct (no MFMA instantiation), 1 dword MFMA kernel, 2 staged (16-byte pieces through LDS), 3 tiled (one tile pair per workgroup),
// 4 Winograd (transposed F(4,3): half the multiplies), 5 Winograd on one tile pair per wave (3x3: F(4,3), 5x5: F(2,5))
int finc_gradw_variant(const FincShape &s)
{
    const GradwInst *i = find_gradw(s.Cq, s.KH, s.KW);
    if (!i || finc_gradw_workspace_bytes(s) == 0) return 0;
    if (find_gradw_wino(s)) return 4;
    if (gradw_use_winot(i, s)) return 5;
    if (gradw_use_tiled(i, s)) return 3;
    if (!i->gw) return 0;
    static const bool no_staged = finc_env("FINC_GRADW_NO_STAGED") != nullptr;
    return (i->gw_staged && s.W % 4 == 0 && !no_staged) ? 2 : 1;
}

unsigned finc_build_flags_gradw() { return FINC_BUILD_FLAGS; }
