// The streaming-bank kernels: the MFMA path of every bank that no register-resident kernel of this library holds
// (3x3 above 96 channels per group -- CINCFlowUnit at C = 192, cinc_flow.py:9-30 --, 5x5 above 48, 2x2 above 32, 4x4, 6x6,
// 7x7 and the non-square filters above 16 channels: layers/conv.py:30-36 takes any tuple).  Until round 5 these ran the
// reference-order scalar kernels of finc_generic.hip.
//
// Same visitation as everywhere in this library (cinc_cuda_kernel_level2.cu:49-56,98-111 made band-wise): lane n of a
// 16-row band trails its upper neighbour by one column, so the 16 pixels of a step lie on one anti-diagonal and only need
// pixels of earlier steps.  What differs is where the bank lives: 9 x 192 x 192 floats are 1.3 MB, more than a compute
// unit's registers and LDS together, so the bank STREAMS from the L2 once per step, in the order the MFMAs eat it
// (16-byte pieces per lane, one 1 KB fragment quad per load instruction, SPD units in flight per wave), and the solved
// pixels of the last KH+KW-2 steps sit in an LDS ring laid out [row][channel], so that a tap (a, b) is the ring slot of
// step t-a-b read a rows higher: no lane shifts, no masks.  Out-of-image taps read zeros: every band is followed by
// Wp - W ghost columns whose "solution" is forced to zero, and the rows above a band (the last KH-1 rows of the band
// before it, solved at least Wp - 16 >= 4 steps earlier by this workgroup) come back from memory into the slot's halo rows.
//
//   inverse:  x_t = Linv z_t - sum_{(a,b) != (0,0)} (Linv W_ab) x_{t-a-b, rows - a}      (bank premultiplied by the inverse
//             of the unit lower triangular corner tap, as in finc_mfma.hip; solved in fp64 while packing)
//   forward:  z_t = sum_{(a,b)} W_ab x_{t-a-b, rows - a}                                   (and grad-input: transposed bank
//             on the flipped image, finc_abi.hip)
//
// NW waves per problem, wave w owns the output channels [16 MT w, 16 MT (w+1)); one barrier per step.  Banks of up to 48
// channels are one-wave problems (NW = 1, MT = 1..3: four problems per compute unit, one per SIMD), wider ones take a
// workgroup of four waves (MT = 1..4 tiles each: 64, 128, 192, 256 padded channels).
#include "finc_common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

// FINC_STREAM_ABLATE (timing-only builds, wrong results): 1 = the bank is not re-read (the fragments of the first units stay),
// 2 = no MFMAs, 4 = no operand loads (z / halo rows), 8 = every wave starts the bank stream at block 0, 16 = no stores;
// FINC_STREAM_SPD overrides the prefetch depth
#ifndef FINC_STREAM_ABLATE
#define FINC_STREAM_ABLATE 0
#endif
#ifndef FINC_STREAM_SPD
#define FINC_STREAM_SPD 0
#endif
// units (one B quad x MT fragment quads) of bank a wave keeps in flight: 12 .. 16 KB per wave at two tiles and more (measured at
// 192 channels: 8 -> 12 units -5 %), in counts that divide a tap's units (no padding units at 3x3) and keep the two-tile
// kernel within 256 registers (two workgroups per compute unit)
constexpr int spd_of(int MT) { return FINC_STREAM_SPD ? FINC_STREAM_SPD : MT == 1 ? 8 : MT == 4 ? 16 : 12; }
// (the dword-operand form of the four-tile kernel holds 8: its scattered accesses leave no registers for 16; a stream is padded
// to spd_of units, which 8 divides)
constexpr int spd_kernel(int MT, bool vec) { return (!FINC_STREAM_SPD && MT == 4 && !vec) ? 8 : spd_of(MT); }
constexpr int SMAXK = 7;     // KH, KW <= 7
constexpr int SMINWP = 20;   // period of a band in steps: >= 16 + 4 (the halo rows' distance to their producer)
constexpr int SMAXCQ = 256;

struct Geo {
    int Cqp, MT, NW, NKQ, NT, HALO, RS, SLOTF, NRING, NSLOT, U, Wp, SPD;
    size_t lds, lds_vec;   // (lds_vec: with the forward's ring of outputs for its 16-byte stores)
};

Geo make_geo(int Cq, int W, int KH, int KW, bool inv)
{
    Geo q;
    q.NW = Cq <= 48 ? 1 : 4;
    q.MT = (Cq + 16 * q.NW - 1) / (16 * q.NW);
    q.Cqp = 16 * q.NW * q.MT;
    q.NKQ = q.Cqp / 16;
    q.NT = KH * KW;
    q.HALO = KH - 1;
    q.RS = q.Cqp + 4;
    q.SLOTF = (16 + q.HALO) * q.RS;
    const int smax = KH + KW - 2;
    q.NRING = inv ? (smax + 1 > 5 ? smax + 1 : 5) : smax + 2;   // (inverse: at least the four steps a row's 16-byte store gathers)
    q.NSLOT = q.NRING + (inv ? 2 : 0);
    q.SPD = spd_of(q.MT);
    q.U = (q.NT * q.NKQ + q.SPD - 1) / q.SPD * q.SPD;
    q.Wp = W + KW - 1 > SMINWP ? W + KW - 1 : SMINWP;
    q.lds = (size_t)q.NSLOT * q.SLOTF * sizeof(float);
    q.lds_vec = q.lds + (inv ? 0 : (size_t)5 * 16 * q.RS * sizeof(float));
    return q;
}

struct Pos {
    int col, band;
};
__device__ inline void pos_init(Pos &p, int q, int Wp)
{
    if (q >= 0) { p.band = q / Wp; p.col = q - p.band * Wp; }
    else { p.band = 0; p.col = q; }
}
__device__ inline void pos_step(Pos &p, int Wp)
{
    if (++p.col == Wp) { p.col = 0; ++p.band; }
}
// offset of the pixel (band, row rho, col) inside one plane, or -1 when it is not a pixel of the image
__device__ inline int pos_pix(const Pos &p, int rho, int H, int W, int NB, unsigned o)
{
    const int h = p.band * 16 + rho;
    if (p.col < 0 || p.col >= W || p.band >= NB || h < 0 || h >= H) return -1;
    return finc_pix(H, W, o, h, p.col);
}

// Every wave starts the bank stream at its own block: the 32 compute units of an XCD (workgroups are dealt round-robin to
// the XCDs) times NW waves, spread evenly over the stream, so that they do not walk the same L2 lines at the same time
// (measured: -7 % at 192 channels, -11 % at 128).  The order of a pixel's sum therefore depends on the workgroup: results
// are reproducible launch to launch, and equal across batch positions only to rounding.
__device__ inline int rot_of(int nblk, int wave, int nw)
{
    if ((FINC_STREAM_ABLATE & 8) || nw == 1) return 0;          // (one-wave problems: the bank is a few KB, it sits in the L1)
    const unsigned idx = ((blockIdx.x >> 3) & 31u) * (unsigned)nw + (unsigned)wave;
    return (int)(idx * (unsigned)nblk / (32u * (unsigned)nw));
}

__device__ inline void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int MT, int NW, bool INV, bool VEC>
__global__ __attribute__((amdgpu_flat_work_group_size(64 * NW, 64 * NW), amdgpu_waves_per_eu(1, 1))) void
finc_stream_kernel(const float *__restrict__ in, const float *__restrict__ bank, const float *__restrict__ biasv, float *out,
                   int G, int Cq, int H, int W, int KH, int KW, unsigned orient, int U, int Wp, int xcdmap)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NTHR = 64 * NW, Cqp = 16 * NW * MT, NKQ = NW * MT, RS = Cqp + 4;
    constexpr int AS = MT == 1 ? 2 : 1;     // accumulators per tile: no MFMA waits on the one before it
    constexpr int SPD = spd_kernel(MT, VEC);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane >> 4, n = lane & 15;
    const int NT = KH * KW, HALO = KH - 1, SLOTF = (16 + HALO) * RS;
    const int smax = KH + KW - 2;
    const int NRING = INV ? (smax + 1 > 5 ? smax + 1 : 5) : smax + 2;
    const int ZOFF = NRING * SLOTF;                    // (INV) the two z slots behind the ring; (forward, VEC) five slots of 16 output rows

    int b, g;
    {
        const int bid = blockIdx.x;
        if (xcdmap) {                                  // workgroups of one group on one XCD: its L2 holds that group's bank
            const int per = 8 / G, xcd = bid & 7, k = bid >> 3;
            g = xcd % G;
            b = k * per + xcd / G;
        } else {
            g = bid % G;
            b = bid / G;
        }
    }
    const unsigned o = finc_group_orient(orient, g);
    const int HW = H * W;
    const size_t poff = ((size_t)b * G + g) * Cq * HW;
    const float *src = in + poff;
    float *dst = out + poff;
    const int NB = (H + 15) >> 4;
    const int Tend = (NB - 1) * Wp + W + 15;

    for (int e = tid; e < (NRING + (INV ? 2 : 0)) * SLOTF + ((VEC && !INV) ? 5 * 16 * RS : 0); e += NTHR) lds[e] = 0.f;

    // ---- loaders: the 16 rows of a slot (thread -> row, 4 MT channels), its halo rows (thread -> halo row, channels) ----
    const int mrow = tid & 15, mch = tid >> 4;          // channels mch + 4 NW i
    const int CHT = HALO > 0 ? NTHR / HALO : NTHR;
    const int hidx = tid / CHT, hch = tid - hidx * CHT;
    const bool hact = HALO > 0 && hidx < HALO;
    Pos pm, ph, pc;
    pos_init(pm, -mrow, Wp);           // slot 0
    pos_init(ph, 1 + hidx, Wp);        // halo row -1-hidx of slot 0
    pos_init(pc, -n, Wp);              // this lane's pixel at step 0
    float mv[4 * MT], hv[2 * MT];

    auto load_main = [&]() {
        const int pix = pos_pix(pm, mrow, H, W, NB, o);
#pragma unroll
        for (int i = 0; i < 4 * MT; ++i) {
            const int ch = mch + 4 * NW * i;
            mv[i] = (pix >= 0 && ch < Cq && !(FINC_STREAM_ABLATE & 4)) ? src[(size_t)ch * HW + pix] : 0.f;
        }
    };
    auto store_main = [&](int slot_off) {
#pragma unroll
        for (int i = 0; i < 4 * MT; ++i) lds[slot_off + (mrow + HALO) * RS + mch + 4 * NW * i] = mv[i];
    };
    // the same 16 rows in 16-byte pieces (W % 4 == 0, aligned activations): lanes are channels, wave w owns the rows
    // RPW w .. RPW w + RPW - 1, and a row requests its next four columns on the step its column count passes a multiple of
    // four (uniform per wave: a scalar branch) -- a quarter of the requests of the dword form and every line fetched
    // 8 times instead of 32 (the activations of 32 compute units do not stay in an XCD's L2 beside the bank)
    constexpr int RPW = 16 / NW, CG = (Cqp + 63) / 64;
    Pos pv[RPW];
    v4f zp[RPW][CG];
    if constexpr (VEC) {
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            pos_init(pv[k], -(RPW * wave + k) - 1, Wp);                        // "slot -1": the first request steps to slot 0
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) zp[k][cg] = v4f{0.f, 0.f, 0.f, 0.f};
        }
    }
    // one-wave problems: the same 16-byte pieces per LANE.  Lane (row mrow, block mch) keeps the dword loader's channels mch + 4 i
    // and requests four columns when ITS row's column count passes a multiple of four -- an exec mask, not a branch: 16 rows at 16
    // phases, four of them per step -- and stores a row's last four columns the same way.  (Lanes = channels, as above, would leave
    // 12 .. 48 of 64 lanes without one.)  At a full chip the dword accesses were 2.7x .. 5.4x the kernel's floor
    // (profiles/r05/stream/one_wave_ablations.txt).
    Pos pl, pq;
    v4f zq[4 * MT];
    if constexpr (VEC && NW == 1) {
        pos_init(pl, -mrow - 1, Wp);
        pos_init(pq, -mrow - 1, Wp);
#pragma unroll
        for (int i = 0; i < 4 * MT; ++i) zq[i] = v4f{0.f, 0.f, 0.f, 0.f};
    }
    auto fetch_main_vec = [&]() {
        if constexpr (NW == 1) {
            pos_step(pl, Wp);
            if ((pl.col & 3) == 0) {
                const int h = pl.band * 16 + mrow;
                const bool ok = pl.col >= 0 && pl.col < W && pl.band < NB && h < H;
                const int hh = (o & FINC_FLIP_H) ? H - 1 - h : h;
                const int wc = (o & FINC_FLIP_W) ? W - 4 - pl.col : pl.col;
#pragma unroll
                for (int i = 0; i < 4 * MT; ++i) {
                    const int ch = mch + 4 * i;
                    zq[i] = (ok && ch < Cq && !(FINC_STREAM_ABLATE & 4)) ? *(const v4f *)(src + (size_t)ch * HW + hh * W + wc)
                                                                        : v4f{0.f, 0.f, 0.f, 0.f};
                }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            pos_step(pv[k], Wp);
            if ((pv[k].col & 3) == 0) {
                const int h = pv[k].band * 16 + RPW * wave + k;
                const bool ok = pv[k].col >= 0 && pv[k].col < W && pv[k].band < NB && h < H;
                const int hh = (o & FINC_FLIP_H) ? H - 1 - h : h;
                const int wc = (o & FINC_FLIP_W) ? W - 4 - pv[k].col : pv[k].col;
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) {
                    const int ch = lane + 64 * cg;
                    zp[k][cg] = (ok && ch < Cq && !(FINC_STREAM_ABLATE & 4)) ? *(const v4f *)(src + (size_t)ch * HW + hh * W + wc)
                                                                            : v4f{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    };
    auto store_main_vec = [&](int slot_off) {
        if constexpr (NW == 1) {
            int comp = pl.col & 3;                                             // (before a row's first column the pieces are zeros)
            if (o & FINC_FLIP_W) comp = 3 - comp;
#pragma unroll
            for (int i = 0; i < 4 * MT; ++i) {
                const v4f pcs = zq[i];
                const float v = comp == 0 ? pcs.x : comp == 1 ? pcs.y : comp == 2 ? pcs.z : pcs.w;
                lds[slot_off + (mrow + HALO) * RS + mch + 4 * i] = v;
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            int comp = pv[k].col & 3;                                          // (before a row's first column the pieces are zeros)
            if (o & FINC_FLIP_W) comp = 3 - comp;
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
                const int ch = lane + 64 * cg;
                const v4f pcs = zp[k][cg];
                const float v = comp == 0 ? pcs.x : comp == 1 ? pcs.y : comp == 2 ? pcs.z : pcs.w;
                if (ch < Cqp) lds[slot_off + (RPW * wave + k + HALO) * RS + ch] = v;
            }
        }
    };
    // the inverse's stores in the same form: a row's last four solved columns sit in the ring's last four slots; the wave that
    // owns the row writes them as one 16-byte piece per channel the step after the group's last column was solved (a scalar
    // branch per row; 4 MT stores of 1 KB per wave and step instead of 4 MT scattered dword stores that leave every line
    // of the output 32 times)
    Pos ps[RPW];
    if constexpr (VEC) {
#pragma unroll
        for (int k = 0; k < RPW; ++k) pos_init(ps[k], -(RPW * wave + k) - 1, Wp);   // the step BEFORE the current one
    }
    auto store_rows_vec = [&](int cur_slot) {
        if constexpr (NW == 1) {
            const int c = pq.col, h = pq.band * 16 + mrow;
            if ((c & 3) == 3 && c >= 0 && c < W && pq.band < NB && h < H) {
                const int hh = (o & FINC_FLIP_H) ? H - 1 - h : h;
                const int wc = (o & FINC_FLIP_W) ? W - 1 - c : c - 3;
                int sl[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    int t_ = cur_slot - 4 + d;
                    if constexpr (INV) sl[d] = (t_ < 0 ? t_ + NRING : t_) * SLOTF + (mrow + HALO) * RS;
                    else sl[d] = ZOFF + (t_ < 0 ? t_ + 5 : t_) * (16 * RS) + mrow * RS;
                }
#pragma unroll
                for (int i = 0; i < 4 * MT; ++i) {
                    const int ch = mch + 4 * i;
                    if (ch < Cq && !(FINC_STREAM_ABLATE & 16)) {
                        const float x0 = lds[sl[0] + ch], x1 = lds[sl[1] + ch], x2 = lds[sl[2] + ch], x3 = lds[sl[3] + ch];
                        *(v4f *)(dst + (size_t)ch * HW + hh * W + wc) = (o & FINC_FLIP_W) ? v4f{x3, x2, x1, x0} : v4f{x0, x1, x2, x3};
                    }
                }
            }
            pos_step(pq, Wp);
            return;
        }
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const int rho = RPW * wave + k;
            const int c = ps[k].col, h = ps[k].band * 16 + rho;
            if ((c & 3) == 3 && c >= 0 && c < W && ps[k].band < NB && h < H) {
                const int hh = (o & FINC_FLIP_H) ? H - 1 - h : h;
                const int wc = (o & FINC_FLIP_W) ? W - 1 - c : c - 3;
                int sl[4];                                                    // ring slots of the steps s-4 .. s-1 (columns c-3 .. c)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    int t_ = cur_slot - 4 + d;
                    if constexpr (INV) sl[d] = (t_ < 0 ? t_ + NRING : t_) * SLOTF + (rho + HALO) * RS;
                    else sl[d] = ZOFF + (t_ < 0 ? t_ + 5 : t_) * (16 * RS) + rho * RS;          // (the forward's own ring of outputs)
                }
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) {
                    const int ch = lane + 64 * cg;
                    if (ch < Cq && !(FINC_STREAM_ABLATE & 16)) {
                        const float x0 = lds[sl[0] + ch], x1 = lds[sl[1] + ch], x2 = lds[sl[2] + ch], x3 = lds[sl[3] + ch];
                        *(v4f *)(dst + (size_t)ch * HW + hh * W + wc) = (o & FINC_FLIP_W) ? v4f{x3, x2, x1, x0} : v4f{x0, x1, x2, x3};
                    }
                }
            }
            pos_step(ps[k], Wp);
        }
    };
    auto load_halo = [&]() {
        const int pix = hact ? pos_pix(ph, -1 - hidx, H, W, NB, o) : -1;
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) {
            hv[i] = 0.f;
            if (i * CHT < Cqp) {
                const int ch = hch + CHT * i;
                if (pix >= 0 && ch < Cq && !(FINC_STREAM_ABLATE & 4)) {
                    if constexpr (INV)      // rows this workgroup stored a few steps ago: read at the L2, never a stale L1 line
                        hv[i] = __hip_atomic_load(dst + (size_t)ch * HW + pix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        hv[i] = src[(size_t)ch * HW + pix];
                }
            }
        }
    };
    auto store_halo = [&](int slot_off) {
        if (!hact) return;
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) {
            const int ch = hch + CHT * i;
            if (i * CHT < Cqp && ch < Cqp) lds[slot_off + (HALO - 1 - hidx) * RS + ch] = hv[i];
        }
    };

    __syncthreads();
    if constexpr (VEC) {
        fetch_main_vec();
        store_main_vec(INV ? ZOFF : 0);
    } else {
        load_main();
        store_main(INV ? ZOFF : 0);
    }
    if constexpr (!INV) {
        load_halo();
        store_halo(0);
    }

    // ---- the bank stream of this wave: [U units][MT tiles][64 lanes][4 floats], SPD units in flight ----
    const char *astream = (const char *)(bank + ((size_t)g * NW + wave) * (size_t)U * MT * 256);
    const unsigned aoff = (unsigned)lane * 16u;
    v4f ar[SPD][MT];
#pragma unroll
    for (int d = 0; d < SPD; ++d)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ar[d][mt] = *(const v4f *)(astream + ((size_t)rot_of(U / SPD, wave, NW) * SPD * MT + d * MT + mt) * 1024 + aoff);
    const int NBLK = U / SPD;
    const int rot = rot_of(NBLK, wave, NW);
    const int tau0 = rot * SPD / NKQ, kq0 = rot * SPD - tau0 * NKQ, ta0 = tau0 / KW, tb0 = tau0 - ta0 * KW;
    v4f bias[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bias[mt] = *(const v4f *)(biasv + (size_t)g * Cqp + (wave * MT + mt) * 16 + 4 * j);
    const v4f *lds4 = (const v4f *)lds;
    const int blane4 = n * (RS / 4) + j;
    lds_barrier();

    int cur = 0, ocur = 0;                             // ring slot of step t (ocur: in the forward's ring of outputs)
    for (int t = 0; t < Tend; ++t) {
        if constexpr (VEC) store_rows_vec(INV ? cur : ocur);      // the groups of four columns the last step completed
        // requests of the next step's operands
        if constexpr (VEC) {
            fetch_main_vec();                          // slot t+1
        } else {
            pos_step(pm, Wp);                          // slot t+1
            load_main();
        }
        if constexpr (INV) {
            load_halo();                               // halo rows of slot t (needed from step t+1 on)
            pos_step(ph, Wp);
        } else {
            pos_step(ph, Wp);                          // halo rows of slot t+1
            load_halo();
        }

        v4f acc[MT][AS];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[mt][0] = bias[mt];
            if constexpr (AS == 2) acc[mt][1] = v4f{0.f, 0.f, 0.f, 0.f};
        }
        // unit bookkeeping (uniform, branch-free: the unrolled block must stay ONE straight-line body, or the compiler
        // threads the tap boundary into copies of the loop that rotate the ring through moves and drain it):
        // unit = (tap, quad of k-steps); tap 0 of the inverse is the z-term.  LDS offsets in 16-byte units.
        const int base0 = (INV ? ZOFF + (t & 1) * SLOTF + HALO * RS : cur * SLOTF + HALO * RS) >> 2;
        int tau = tau0, kq = kq0, ta = ta0, tb = tb0;
        v4f bcur;
        {
            int slot = cur - ta - tb;
            slot += slot < 0 ? NRING : 0;
            const int ub = tau < NT ? ((slot * SLOTF + (HALO - ta) * RS) >> 2) : base0;
            bcur = lds4[((INV && tau == 0) ? base0 : ub) + 4 * kq + blane4];
        }
        for (int bi = 0; bi < NBLK; ++bi) {
            const int blk = bi + rot >= NBLK ? bi + rot - NBLK : bi + rot;
            const int nxt = blk + 1 == NBLK ? 0 : blk + 1;                     // (a step's last block prefetches the next step's first)
            const char *pf = astream + (size_t)nxt * (SPD * MT * 1024);
#pragma unroll
            for (int d = 0; d < SPD; ++d) {
                const bool over = d == SPD - 1 && nxt == 0;                    // the unit behind the stream's last one is its first
                const int wrap = (kq + 1 == NKQ) ? 1 : 0;
                kq = (wrap || over) ? 0 : kq + 1;
                tau = over ? 0 : tau + wrap;
                const int wrap2 = (tb + wrap == KW) ? 1 : 0;
                tb = (wrap2 || over) ? 0 : tb + wrap;
                ta = over ? 0 : ta + wrap2;
                int slot = cur - ta - tb;
                slot += slot < 0 ? NRING : 0;
                const int ub = tau < NT ? ((slot * SLOTF + (HALO - ta) * RS) >> 2) : base0;   // (padding units: zero fragments on finite operands)
                const int ub0 = (INV && tau == 0) ? base0 : ub;
                const v4f bnext = lds4[ub0 + 4 * kq + blane4];
                v4f a[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    a[mt] = ar[d][mt];
                    if constexpr (!(FINC_STREAM_ABLATE & 1)) ar[d][mt] = *(const v4f *)(pf + (size_t)(d * MT + mt) * 1024 + aoff);
                }
                __builtin_amdgcn_sched_barrier(0);     // the requests of later units go out BEFORE this unit's MFMAs
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        if constexpr (FINC_STREAM_ABLATE & 2) acc[mt][i % AS][i] += a[mt][i] * bcur[i];
                        else acc[mt][i % AS] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][i], bcur[i], acc[mt][i % AS], 0, 0, 0);
                    }
                bcur = bnext;
            }
        }

        // ---- this step's pixels: lane (j, n) holds channels 16 tile + 4 j .. + 3 of row n ----
        const int pix = pos_pix(pc, n, H, W, NB, o);
        const int nslot = cur + 1 == NRING ? 0 : cur + 1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            v4f v = acc[mt][0];
            if constexpr (AS == 2) v += acc[mt][1];
            const int ch = (wave * MT + mt) * 16 + 4 * j;
            if (pix < 0) v = v4f{0.f, 0.f, 0.f, 0.f};
            if constexpr (INV) ((v4f *)lds)[((cur * SLOTF + (n + HALO) * RS) >> 2) + (ch >> 2)] = v;
            else if constexpr (VEC) ((v4f *)lds)[((ZOFF + ocur * (16 * RS) + n * RS) >> 2) + (ch >> 2)] = v;
            if (pix >= 0 && !VEC) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (ch + i < Cq && !(FINC_STREAM_ABLATE & 16)) dst[(size_t)(ch + i) * HW + pix] = v[i];
            }
        }
        pos_step(pc, Wp);
        const int moff = INV ? ZOFF + ((t + 1) & 1) * SLOTF : nslot * SLOTF;
        if constexpr (VEC) store_main_vec(moff);
        else store_main(moff);
        store_halo(INV ? cur * SLOTF : nslot * SLOTF);
        lds_barrier();
        cur = nslot;
        ocur = ocur == 4 ? 0 : ocur + 1;
    }
    if constexpr (VEC) store_rows_vec(INV ? cur : ocur);    // the groups the last step completed
}

// ---- packing ----
// inverse: y = Linv v in fp64, one thread per (group, column): v = a column of -W_ab (tau >= 1), of diag(scale) (tau = 0:
// the z-term) or the shift (the bias).  scratch: [G][NT*Cq*Cq + Cq] doubles, [tau][r][c] then the bias.
__global__ void stream_solve_kernel(const float *__restrict__ wc, const float *__restrict__ scale, const float *__restrict__ shift,
                                    double *scratch, int G, int Cq, int KH, int KW)
{
    const int NT = KH * KW;
    const int per_g = NT * Cq + 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= G * per_g) return;
    const int g = idx / per_g, e = idx - g * per_g;
    const bool isb = e == NT * Cq;
    const int tau = isb ? NT : e / Cq, c = isb ? 0 : e - tau * Cq;
    const float *wg = wc + (size_t)g * Cq * Cq * NT;
    double *y = scratch + (size_t)g * ((size_t)NT * Cq * Cq + Cq) + (isb ? (size_t)NT * Cq * Cq : (size_t)tau * Cq * Cq + c);
    const int ys = isb ? 1 : Cq;
    const int ta = isb ? 0 : tau / KW, tb = isb ? 0 : tau % KW;
    const int corner = (KH - 1) * KW + (KW - 1);
    for (int r = 0; r < Cq; ++r) {
        double v;
        if (isb) v = shift ? (double)shift[g * Cq + r] : 0.0;
        else if (tau == 0) v = r == c ? (scale ? (double)scale[g * Cq + c] : 1.0) : 0.0;
        else v = -(double)wg[((size_t)r * Cq + c) * NT + (KH - 1 - ta) * KW + (KW - 1 - tb)];
        const int k0 = (!isb && tau == 0) ? c : 0;           // (a column of the identity is zero above its diagonal entry)
        for (int k = k0; k < r; ++k) v -= (double)wg[((size_t)r * Cq + k) * NT + corner] * y[(size_t)k * ys];
        y[(size_t)r * ys] = v;
    }
}

// fragments: [g][wave][unit][tile][lane (j, m)][i] = M_tau[row 16 tile + m][col 16 kq + 4 j + i]; bias [g][Cqp]
__global__ void stream_frag_kernel(const float *__restrict__ wc, const double *__restrict__ scratch, const float *__restrict__ scale,
                                   const float *__restrict__ shift, float *__restrict__ bankp, float *__restrict__ biasp, int G, int Cq,
                                   int KH, int KW, int MT, int NW, int U, int inverse, int transpose)
{
    const int NT = KH * KW, NKQ = NW * MT, Cqp = 16 * NW * MT;
    const size_t per_g = (size_t)NW * U * MT * 256;
    const size_t total = per_g * G;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(idx / per_g);
        size_t e = idx - (size_t)g * per_g;
        const int i = (int)(e & 3); e >>= 2;
        const int lane = (int)(e & 63); e >>= 6;
        const int mt = (int)(e % MT); e /= MT;
        const int u = (int)(e % U);
        const int wave = (int)(e / U);
        const int tau = u / NKQ, kq = u - tau * NKQ;
        const int row = (wave * MT + mt) * 16 + (lane & 15), col = 16 * kq + 4 * (lane >> 4) + i;
        float v = 0.f;
        if (tau < NT && row < Cq && col < Cq) {
            if (inverse) {
                v = (float)scratch[(size_t)g * ((size_t)NT * Cq * Cq + Cq) + ((size_t)tau * Cq + row) * Cq + col];
            } else {
                const int ta = tau / KW, tb = tau - ta * KW;
                const int oc = transpose ? col : row, ic = transpose ? row : col;
                v = wc[(((size_t)g * Cq + oc) * Cq + ic) * NT + (KH - 1 - ta) * KW + (KW - 1 - tb)];
                if (scale) v *= scale[g * Cq + row];
            }
        }
        bankp[idx] = v;
    }
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < G * Cqp; idx += gridDim.x * blockDim.x) {
        const int g = idx / Cqp, r = idx - g * Cqp;
        float v = 0.f;
        if (r < Cq) {
            if (inverse) v = (float)scratch[(size_t)g * ((size_t)NT * Cq * Cq + Cq) + (size_t)NT * Cq * Cq + r];
            else if (shift) v = shift[g * Cq + r];
        }
        biasp[idx] = v;
    }
}

typedef void (*stream_fn)(const float *, const float *, const float *, float *, int, int, int, int, int, int, unsigned, int, int, int);
template <int MT, int NW>
stream_fn pick4(bool inv, bool vec)
{
    return inv ? (vec ? finc_stream_kernel<MT, NW, true, true> : finc_stream_kernel<MT, NW, true, false>)
               : (vec ? finc_stream_kernel<MT, NW, false, true> : finc_stream_kernel<MT, NW, false, false>);
}
stream_fn pick(int MT, int NW, bool inv, bool vec)
{
    if (NW == 1) {
        switch (MT) {
        case 1: return pick4<1, 1>(inv, vec);
        case 2: return pick4<2, 1>(inv, vec);
        case 3: return pick4<3, 1>(inv, vec);
        }
        return nullptr;
    }
    switch (MT) {
    case 1: return pick4<1, 4>(inv, vec);
    case 2: return pick4<2, 4>(inv, vec);
    case 3: return pick4<3, 4>(inv, vec);
    case 4: return pick4<4, 4>(inv, vec);
    }
    return nullptr;
}

size_t align256(size_t n) { return (n + 255) / 256 * 256; }
size_t bank_bytes(const Geo &q, int G) { return align256((size_t)G * q.NW * q.U * q.MT * 256 * sizeof(float)); }
size_t bias_bytes(const Geo &q, int G) { return align256((size_t)G * q.Cqp * sizeof(float)); }

} // namespace

bool finc_stream_bank_ok(int Cq, int KH, int KW)
{
    return Cq >= 1 && Cq <= SMAXCQ && KH >= 1 && KW >= 1 && KH <= SMAXK && KW <= SMAXK;
}

bool finc_stream_supported(int Cq, int H, int W, int KH, int KW, bool inverse)
{
    if (!finc_stream_bank_ok(Cq, KH, KW) || H < 1 || W < 1) return false;
    const Geo q = make_geo(Cq, W, KH, KW, inverse);
    if (q.lds > 160 * 1024) return false;
    if ((size_t)Cq * H * W >= ((size_t)1 << 30)) return false;
    if ((long long)((H + 15) / 16) * q.Wp + 64 >= (1LL << 30)) return false;
    return true;
}

size_t finc_stream_packed_bytes(int G, int Cq, int KH, int KW, bool inverse)
{
    if (!finc_stream_bank_ok(Cq, KH, KW)) return 0;
    const Geo q = make_geo(Cq, 16, KH, KW, inverse);
    size_t n = bank_bytes(q, G) + bias_bytes(q, G);
    if (inverse) n += align256((size_t)G * ((size_t)q.NT * Cq * Cq + Cq) * sizeof(double));
    return n;
}

int finc_stream_pack(const float *wc, const float *scale, const float *shift, void *packed, int G, int Cq, int KH, int KW, bool inverse,
                     bool transpose, hipStream_t st)
{
    if (!finc_stream_bank_ok(Cq, KH, KW)) return FINC_ERR_UNSUPPORTED;
    if (((uintptr_t)packed & 15u) != 0) return FINC_ERR_ALIGNMENT;
    const Geo q = make_geo(Cq, 16, KH, KW, inverse);
    float *bankp = (float *)packed;
    float *biasp = (float *)((char *)packed + bank_bytes(q, G));
    double *scratch = (double *)((char *)packed + bank_bytes(q, G) + bias_bytes(q, G));
    if (inverse) {
        const int threads = G * (q.NT * Cq + 1);
        hipLaunchKernelGGL(stream_solve_kernel, dim3((threads + 63) / 64), dim3(64), 0, st, wc, scale, shift, scratch, G, Cq, KH, KW);
        FINC_CHECK_LAUNCH();
    }
    const size_t total = (size_t)G * q.NW * q.U * q.MT * 256;
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(stream_frag_kernel, dim3((unsigned)blocks), dim3(256), 0, st, wc, (const double *)scratch, scale, shift, bankp, biasp,
                       G, Cq, KH, KW, q.MT, q.NW, q.U, inverse ? 1 : 0, transpose ? 1 : 0);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

// One-wave problems (banks of up to 48 channels): which I/O form the launch takes.  The dword form (lanes = pixels) is latency-bound
// and wins while the chip has few problems; its time grows with problems x channels, the per-lane 16-byte form's hardly at all.
// Measured crossovers at 32x32, G = 4 (profiles/r05/stream/one_wave_crossover.txt; 256 .. 1,024 problems in both forms): a 4x4 and a
// 7x7 bank of 12 channels (one tile) between 640 and 768 problems, a 2x2 bank of 40 channels (three tiles) between 256 and 384 -- the
// rule below is the product problems x padded channels that separates all 36 timings, never below one problem per compute unit.
// (Two-tile banks: interpolated, not measured.)
static bool one_wave_vec_form(long long problems, int Cqp)
{
    static const char *force1 = finc_env("FINC_STREAM_ONE_WAVE_VEC");   // experiment switch ("0" / "1"): the one-wave form whatever the count
    if (force1) return force1[0] == '1';
    return problems > 256 && problems * Cqp > 10240;
}

int finc_stream_info(const FincShape &s, bool inverse, int *cqp, int *lds, int *steps, int *waves, int *one_wave_vec)
{
    if (!finc_stream_supported(s.Cq, s.H, s.W, s.KH, s.KW, inverse)) return FINC_ERR_UNSUPPORTED;
    const Geo q = make_geo(s.Cq, s.W, s.KH, s.KW, inverse);
    if (cqp) *cqp = q.Cqp;
    if (waves) *waves = q.NW;
    if (lds) *lds = (int)q.lds;
    if (steps) *steps = ((s.H + 15) / 16 - 1) * q.Wp + s.W + 15;
    // (what the launch does with 16-byte aligned tensors)
    if (one_wave_vec) *one_wave_vec = q.NW == 1 && s.W % 4 == 0 && q.lds_vec <= 160 * 1024 && one_wave_vec_form((long long)s.B * s.G, q.Cqp);
    return FINC_OK;
}

int finc_stream_launch(const float *in, const void *packed, float *out, const FincShape &s, bool inverse, hipStream_t st)
{
    if (!finc_stream_supported(s.Cq, s.H, s.W, s.KH, s.KW, inverse)) return FINC_ERR_UNSUPPORTED;
    if (((uintptr_t)packed & 15u) != 0) return FINC_ERR_ALIGNMENT;
    const Geo q = make_geo(s.Cq, s.W, s.KH, s.KW, inverse);
    // the 16-byte operand loader needs whole groups of four columns and aligned rows
    const bool vec = s.W % 4 == 0 && (((uintptr_t)in | (uintptr_t)out) & 15u) == 0 && q.lds_vec <= 160 * 1024 &&
                     (q.NW == 4 || one_wave_vec_form((long long)s.B * s.G, q.Cqp));
    const size_t lds_bytes = vec ? q.lds_vec : q.lds;   // (one-wave problems: 12 of 64 lanes would hold a channel)
    const stream_fn fn = pick(q.MT, q.NW, inverse, vec);
    if (!fn) return FINC_ERR_UNSUPPORTED;
    if (int e = finc_ensure_dynamic_lds((const void *)fn, lds_bytes)) return e;
    const float *bankp = (const float *)packed;
    const float *biasp = (const float *)((const char *)packed + bank_bytes(q, s.G));
    const int xcdmap = (s.G <= 8 && 8 % s.G == 0 && s.B % (8 / s.G) == 0) ? 1 : 0;
    hipLaunchKernelGGL(fn, dim3(s.B * s.G), dim3(64 * q.NW), lds_bytes, st, in, bankp, biasp, out, s.G, s.Cq, s.H, s.W, s.KH, s.KW, s.orient, q.U,
                       q.Wp, xcdmap);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

unsigned finc_build_flags_stream() { return FINC_BUILD_FLAGS; }
