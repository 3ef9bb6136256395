// Inverse for the BIG banks (gfx950 / CDNA4 only): 3x3 filters with 64 < Cq <= 96 -- CINCFlowUnit (fastflow/cinc_flow.py:9-30,
// one group over all channels) at C = 96, FastFlowUnit at C = 260 .. 384 -- which no wave and no K-split of finc_mfma.hip can
// hold: a 96-channel 3x3 bank is 1,296 16x16x4 fragments = 331 KB per group.
//
// M-split on the role-split kernel's machinery (finc_split.hip).  A workgroup of NWV = 8 waves solves one problem, lane p of
// every wave owns the rows p, P+p, ... of the image and trails lane p-1 by one step (the reference's anti-diagonal order,
// cinc_cuda_kernel_level2.cu:49-56,98-111, band by band).  Wave w owns the OUTPUT channels 12w .. 12w+11 -- three 4-row blocks
// on v_mfma_f32_4x4x1_16B_f32 -- for ALL nine taps and all 24 k-steps: 9 x 24 x 3 = 648 block fragments, four to a register
// (finc_tile.h) = 162 AGPRs, resident.  No wave keeps any operand history: the pixels solved in the last steps live in a shared
// x ring in LDS ([k-step][time slot][lane]; a solved pixel's channels come out of the blocks' reduce in natural order, channel
// 4j+q in k-slot q of k-step j, so wave w simply writes k-steps 3w .. 3w+2), tap (a,b) of step u reads the slot of step u-a-b at
// lane p-a (the rows of the band above: the hand-over FIFO), a tap's column mask is applied to the address (an invalid lane
// reads a zero word), and z waits in a shared z ring.  Nothing is exchanged but the 12 channels each wave publishes, and ONE
// workgroup barrier per step orders it all: what step u reads was written in step u-1 or earlier.
//
// A step is 648 block MFMAs per wave (the z-term is computed in full although Linv is triangular: one code path for all
// waves), two waves per SIMD: 10.4 k cycles of MFMA per step and SIMD, the floor of the shape.  A per-channel SCALE in front of
// the inverse folds into the z-term as everywhere; a folded SHIFT is not supported here (no register left for it: the caller
// keeps the affine layer as its own launch).  Each wave also owns the HBM side of its three k-steps:
// 16-byte pieces, one window (4 steps) ahead -- a step lasts microseconds here -- landed in / collected from the rings.
//
// LDS (one workgroup per CU): per k-step the x ring (2 KB) and the FIFO block behind it (1,704 B: 53 slots), then the z ring
// (3 KB per k-step): 24 x 6,824 = 163,776 of the 163,840 bytes: maps up to 64 wide.
#include "finc_common.h"
#include "finc_tile.h"

#include <stdlib.h>

#include <type_traits>
#include <utility>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr unsigned OFF_INVALID = 0x80000000u;     // voffset beyond any slab: buffer loads return 0, stores are dropped
constexpr unsigned OFF_BAD_CHANNEL = 0x40000000u;  // added to a valid offset it still lands beyond the slab (< 1 GiB)

template <int I>
using IC = std::integral_constant<int, I>;
#define FINC_SB() __builtin_amdgcn_sched_barrier(0)

constexpr int XSLOTS = 8;        // x ring: the pixels of the last 8 steps (taps reach back KH + KW - 2 = 4, the stores 7)
constexpr int RING_J = 2048;     // a k-step's x ring (8 slots x 64 lanes) ...
constexpr int FIFO_J = 1704;     // ... and its FIFO block behind it: 53 slots x 32 bytes + the zero word (W <= 64)
constexpr int ABOVE_J = 516;     // ... or, for wider maps, 16 slots of the rows above re-read from the OUTPUT + the zero word
constexpr int ZJSTRIDE = 3072;   // bytes between the k-steps of the z ring (12 slots x 64 lanes)
constexpr int UNROLL = 8;        // steps per iteration of the loop: two I/O windows

template <int CQP, int KH, int KW, int NWV, bool HBMF = false>
struct BCfg {
    // bytes between the k-steps of the x ring AND of the block behind it: a lane's k-step stride is an immediate whichever it reads
    static constexpr int JSTRIDE = RING_J + (HBMF ? ABOVE_J : FIFO_J);
    static constexpr int NK = CQP / 4;                    // k-steps of an operand (k-slot q <-> channel 4j+q)
    static constexpr int MO = CQP / NWV;                  // output channels of a wave
    static constexpr int NB = MO / 4;                     // ... as 4-row blocks = the k-steps of the solved pixel it writes
    static_assert(CQP % (4 * NWV) == 0, "every wave owns whole 4-row blocks");
    static constexpr int NITEM = KH * KW;                 // item 0 = the z-term (tap (0,0): Linv), item a*KW+b = tap (a,b)
    static constexpr int NFR = NITEM * NK * NB;           // block fragments of a wave
    static constexpr int NREG = (NFR + 3) / 4;            // ... four to a register
    static constexpr int JS = 4 * (KH - 1);               // FIFO: floats per slot and k-step (4 k-slots x (KH-1) lanes)
    static constexpr int RING_B = 0, FIFO_B = RING_J, ZR_B = NK * JSTRIDE, LDS_BYTES = ZR_B + NK * ZJSTRIDE;
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
    // operands are read in batches of BS k-steps, one batch ahead of their MFMAs; a DS immediate reaches 64 KB: a second base
    // register for the upper half of an operand that is longer
    // (HBMF: the stream of the rows above costs a few registers more than the FIFO's bookkeeping: smaller batches pay for them)
    static constexpr int BS = HBMF && NK % 8 == 0 ? 4 : NK % 6 == 0 ? 6 : NK % 4 == 0 ? 4 : NK;
    static constexpr int NBAT = NK / BS;
    static_assert(NK % BS == 0 && BS <= 8, "batches tile the k-steps");
    static_assert(NK * ZJSTRIDE < 65536 || (NK % 2 == 0 && (NK / 2) % BS == 0 && (NK / 2) * JSTRIDE < 65536), "DS immediates reach half an operand");
    // issue order of the items: the z-term, then the taps by falling a + b (oldest operand first) -- the two taps that need the
    // pixel of the step before come last
    static constexpr int order(int k)
    {
        if (k == 0) return 0;
        int n = 0;
        for (int sum = KH + KW - 2; sum >= 1; --sum)
            for (int a = 0; a < KH; ++a)
                for (int b = 0; b < KW; ++b)
                    if (a + b == sum && ++n == k) return a * KW + b;
        return 0;
    }
    static_assert(KH + KW - 2 <= XSLOTS - 1, "the x ring must reach back to the farthest tap");
};

// -----------------------------------------------------------------------------------------------
// grid = B*G workgroups of NWV waves.  W % 4 == 0 (16-byte pieces); P = 16.
//   HBMF = false: 16 <= W <= 64; the last KH-1 rows of a band wait for the band below in the FIFO (DF slots).
//   HBMF = true ("hand-over through memory"): W >= 48, any width.  Two rows of a 256-wide map are 96 KB at 48 channels -- no
//          FIFO in LDS holds that -- but they are in the OUTPUT: the band below re-reads them from there, where this very
//          workgroup stored them W - P steps (hundreds of microseconds) earlier.  Each wave fetches, for its own k-steps, the
//          16-byte pieces of those two rows that lane 0 reaches two windows later (lanes (q, r): k-slot q, row P-2+r) -- with
//          sc0 sc1, past the vector cache: a cached line would also hold columns that are not stored yet -- and lands them in a
//          ring of 16 positions where the FIFO block used to be, indexed by lane 0's position (= its step): tap (a,b) of a
//          lane p < a reads position u - p - b.  The pieces ride in the registers of the z pieces, alternating with them: z is
//          requested at step 0 of a window and lands at step 2, the rows above are requested at step 2 and land at step 0.
// -----------------------------------------------------------------------------------------------
template <int CQP, int KH, int KW, int NWV, bool HBMF>
__global__ __launch_bounds__(64 * NWV) void finc_big_kernel(const float *__restrict__ in, const float *__restrict__ packed,
                                                             float *__restrict__ out, int G, int CQ, int H, int W, int T,
                                                             unsigned orient, int DF)
{
    using C = BCfg<CQP, KH, KW, NWV, HBMF>;
    constexpr int NK = C::NK, NB = C::NB, NJ = C::NB, JS = C::JS, P = 16, JSTRIDE = C::JSTRIDE;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char *const ldsb = reinterpret_cast<char *>(lds);
    // wave -> the block of MO output channels it owns.  Linv is lower triangular: block c needs the z-term's k-steps up to
    // NJ*c + NJ-1 only, so the low blocks have fewer MFMAs.  The waves w and w + NWV/2 share a SIMD (dispatch order): pairing
    // block c with block NWV-1-c gives every SIMD the same load
    const int wvid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wv = wvid < NWV / 2 ? wvid : NWV + NWV / 2 - 1 - wvid;
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, p = lane & 15;
    const int bg = blockIdx.x, g = bg % G;
    const unsigned o = finc_group_orient(orient, g);
    const bool fh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 4u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);

    for (int i = threadIdx.x; i < C::LDS_BYTES / 4; i += 64 * NWV) lds[i] = 0.f;

    auto ld = [&](int byte_off) { return *reinterpret_cast<const float *>(ldsb + byte_off); };
    auto st = [&](int byte_off, float v) { *reinterpret_cast<float *>(ldsb + byte_off) = v; };

    // ---- this wave's fragments -> AGPRs (they are only ever MFMA A operands); the loads must have landed before the loop: all
    // VMEM instructions of the loop are inline asm that the kernel counts itself
    float fr[C::NREG];
    {
        const float *pk = packed + ((size_t)(g * NWV + wv) * C::NREG) * 64 + lane;
#pragma unroll
        for (int r = 0; r < C::NREG; ++r) fr[r] = pk[r * 64];
        // (hipcc splits the 256 registers of a wave at two waves per SIMD 128 : 128 between VGPRs and AGPRs: the first 128
        // fragment registers are pinned to AGPRs, the rest stay with the VGPRs)
#pragma unroll
        for (int r = 0; r < C::NREG; ++r) {
            if (r < 128) asm volatile("" : "+a"(fr[r]));
            else asm volatile("" : "+v"(fr[r]));
        }
    }
    // fragment f = (item * NK + j) * NB + sb sits in register f >> 2 as ABID f & 3
    auto mma = [&](v4f &acc, int f, float b) { finc_mma_small(acc, fr[f >> 2], b, f & 3); };

    // ---- HBM side of this wave's k-steps JLO .. JLO+NJ-1 (finc_split.hip: 16-byte pieces = groups of 4 canonical columns of one
    // row; group gi of lane p covers its positions n = 4gi .. 4gi+3, n = step - p; the z ring holds 3 groups, slot of position
    // n = n mod 12, a W-flipped group is mirrored when it lands).  In window w (steps 4w .. 4w+3) a lane reads z of the groups
    // w+f and w+f+1, f = floor(-p/4); it LANDS group w+f+2, requested ONE window earlier, and requests group w+f+3.  Stores: in
    // window w the group w+fs, fs = floor((-3-p)/4), is collected from the x ring's time slots and leaves as one piece.
    const int JLO = NJ * wv;
    const int zlast = NJ * wv + NJ - 1;        // last k-step of the z-term with a nonzero fragment in this wave's rows
    // channel 4(JLO+j)+q: the natural order serves loads and stores alike.  (HBMF: recomputed at every use -- three times per
    // window -- instead of kept: the variant is three registers short, and a spill is not an option beside hand-counted loads)
    unsigned cmask_k[HBMF ? 1 : NJ];
    if constexpr (!HBMF) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) cmask_k[j] = (4 * (JLO + j) + q) < CQ ? (unsigned)((4 * (JLO + j) + q) * HW * 4) : OFF_BAD_CHANNEL;
    }
    struct CMask { unsigned v[NJ]; };
    auto cmask_now = [&]() {
        CMask c;
        if constexpr (HBMF) {
            int l = lane;
            asm volatile("" : "+v"(l));        // (opaque: keeps the computation here instead of hoisted out of the loop)
#pragma unroll
            for (int j = 0; j < NJ; ++j) c.v[j] = (4 * (JLO + j) + (l >> 4)) < CQ ? (unsigned)((4 * (JLO + j) + (l >> 4)) * HW * 4) : OFF_BAD_CHANNEL;
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) c.v[j] = cmask_k[j];
        }
        return c;
    };
    const int f4 = -((p + 3) >> 2), fs4 = -((p + 3 + 3) >> 2);             // floor(-p / 4), floor((-3 - p) / 4)
    const int dgrp = fw ? -16 : 16;                                        // bytes from a group to the next one of the row
    const int drow = (fh ? -P : P) * W * 4 - (dgrp / 4) * W;               // ... and from the end of a row to the start of row + P
    auto piece_off = [&](int row, int col0) { return ((fh ? H - 1 - row : row) * W + (fw ? W - 4 - col0 : col0)) * 4; };
    int lcol = 4 * f4, lrow = p, loff = piece_off(p, 0) + f4 * dgrp;
    int scol = 4 * fs4, srow = p, soff = piece_off(p, 0) + fs4 * dgrp;
    v4f zin[NJ];                               // in flight HBM -> z ring
    auto zreq = [&](v4f (&dst)[NJ]) {
        const bool ok = lcol >= 0 && lrow < H;
        const unsigned base = ok ? (unsigned)loff : OFF_INVALID;
        const CMask cmask = cmask_now();
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst[j]) : "v"(base + cmask.v[j]), "s"(rin) : "memory");
        lcol += 4; loff += dgrp;
        if (lcol == W) { lcol = 0; lrow += P; loff += drow; }
    };
    int gland = ((f4 % 3) + 3) % 3;            // ring group of the next landing
    const int zr_w = C::ZR_B + JLO * ZJSTRIDE + lane * 4;
    const int e0 = fw ? 3 : 0, e1 = fw ? 2 : 1, e2 = fw ? 1 : 2, e3 = fw ? 0 : 3;
    auto zland = [&](v4f (&src)[NJ], auto vm_c) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(vm_c)::value) : "memory");
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(src[j]));      // (ties the reads below to the wait)
        const int base = zr_w + gland * 1024;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float v0 = src[j].x, v1 = src[j].y, v2 = src[j].z, v3 = src[j].w;
            st(base + e0 * 256 + j * ZJSTRIDE, v0);
            st(base + e1 * 256 + j * ZJSTRIDE, v1);
            st(base + e2 * 256 + j * ZJSTRIDE, v2);
            st(base + e3 * 256 + j * ZJSTRIDE, v3);
        }
        gland = gland == 2 ? 0 : gland + 1;
    };
    // stores: element k of the group was solved at step 4gs + k + p: time slot ((p + k) & 7) ^ (4 * (gs & 1))
    const int xbase = C::RING_B + JLO * JSTRIDE;   // (not a multiple of 2 KB: the toggle below works on the slot part alone)
    // (the lane's share of the toggle -- the parity of its first group -- sits in the constants: what toggles is wave-uniform)
    const int xs0 = ((((p + e0) & 7) * 64 + lane) * 4) ^ ((fs4 & 1) * 1024), xs1 = ((((p + e1) & 7) * 64 + lane) * 4) ^ ((fs4 & 1) * 1024);
    const int xs2 = ((((p + e2) & 7) * 64 + lane) * 4) ^ ((fs4 & 1) * 1024), xs3 = ((((p + e3) & 7) * 64 + lane) * 4) ^ ((fs4 & 1) * 1024);
    int stog = 0;                              // toggles with the group
    auto xstore = [&]() {
        const bool ok = scol >= 0 && srow < H;
        const unsigned base = ok ? (unsigned)soff : OFF_INVALID;
        const CMask cmask = cmask_now();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            v4f v;
            v.x = ld(xbase + (xs0 ^ stog) + j * JSTRIDE);
            v.y = ld(xbase + (xs1 ^ stog) + j * JSTRIDE);
            v.z = ld(xbase + (xs2 ^ stog) + j * JSTRIDE);
            v.w = ld(xbase + (xs3 ^ stog) + j * JSTRIDE);
            // (s_nop: a store of more than 8 bytes reads its data one wait state after issue, and the hazard recognizer does not
            // see inline asm)
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(base + cmask.v[j]), "s"(rout) : "memory");
        }
        stog ^= 1024;
        scol += 4; soff += dgrp;
        if (scol == W) { scol = 0; srow += P; soff += drow; }
    };
    {   // prologue: groups f, f+1 land now (window 0 reads them); f+2 waits in flight (HBMF: is requested in window 0)
        v4f tmp0[NJ], tmp1[NJ];
        zreq(tmp0); zreq(tmp1);
        if constexpr (!HBMF) zreq(zin);
        zland(tmp0, IC<HBMF ? 0 : NJ>{}); zland(tmp1, IC<HBMF ? 0 : NJ>{});
    }
    // ---- HBMF: the rows above the band, re-read from the output.  Piece ww covers lane 0's positions 4ww .. 4ww+3 (band
    // abnd = position / W, columns acol ..): requested in window ww - 2, landed in window ww - 1, read from window ww on.
    int acol = 8 % W, abnd = 8 / W;            // (the first request, in window 0, is piece 2: positions 8 .. 11)
    int aslot = 2;                             // ring group (piece & 3) of the next landing
    auto areq = [&](v4f (&dst)[NJ]) {
        const int row = P * abnd - (KH - 1) + p;                           // lanes p < KH-1: rows P-2, P-1 of the band above
        const bool ok = abnd >= 1 && p < KH - 1 && row < H;
        const unsigned base = ok ? (unsigned)piece_off(row, acol) : OFF_INVALID;
        const CMask cmask = cmask_now();
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen sc0 sc1" : "=v"(dst[j]) : "v"(base + cmask.v[j]), "s"(rout) : "memory");
        acol += 4;
        if (acol >= W) { acol -= W; ++abnd; }
    };
    auto aland = [&](v4f (&src)[NJ]) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(src[j]));
        if (p < KH - 1) {
            int l = lane;                      // (recomputed here, once per window: hoisted out of the loop the address costs a register
            asm volatile("" : "+v"(l));        // the kernel does not have -- it would be spilled)
            const int base = C::FIFO_B + JLO * JSTRIDE + ((l >> 4) * (KH - 1) + (l & 15)) * 4 + aslot * (4 * JS * 4);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float v0 = src[j].x, v1 = src[j].y, v2 = src[j].z, v3 = src[j].w;
                st(base + e0 * (JS * 4) + j * JSTRIDE, v0);
                st(base + e1 * (JS * 4) + j * JSTRIDE, v1);
                st(base + e2 * (JS * 4) + j * JSTRIDE, v2);
                st(base + e3 * (JS * 4) + j * JSTRIDE, v3);
            }
        }
        aslot = (aslot + 1) & 3;
    };
    int zn = ((-p) % 12 + 12) % 12;            // z read slot: position n mod 12 of this lane

    // ---- operands of step u.  S_a(u-a-b): the x ring at lane p - a (lanes p >= a) / the FIFO (lanes p < a: the band above);
    // a tap's column mask (b > 0: column c - b must exist) is applied to the ADDRESS -- the zero word of the k-step's FIFO block
    int cb = 0 - p;                            // col of this lane at step u (negative: not started)
    int fs2 = ((-2 - (W - P)) % DF + DF) % DF; // FIFO slot of the push of step u - 2 - (W - P)
    int fpush = 0;                             // FIFO slot of this step's push
    const bool pusher = p >= P - (KH - 1);
    const int ring_w = C::RING_B + JLO * JSTRIDE + lane * 4;
    const int push_w = C::FIFO_B + JLO * JSTRIDE + (q * (KH - 1) + (p - (P - (KH - 1)))) * 4;
    auto item_addr = [&](auto item_c, int u) {
        constexpr int item = decltype(item_c)::value;
        constexpr int a = item / KW, b = item % KW;
        if constexpr (item == 0) return C::ZR_B + zn * 256 + lane * 4;
        const int tau = u - a - b;
        int fs = fs2 - (a + b - 2);            // slot of push step tau - (W - P)
        if (fs < 0) fs += DF;
        if (fs >= DF) fs -= DF;
        const int ring = C::RING_B + ((tau & (XSLOTS - 1)) * 64 + lane - a) * 4;
        const int fifo = C::FIFO_B + ((HBMF ? ((u - p - b) & 15) : fs) * JS + q * (KH - 1) + (KH - 1 - a + p)) * 4;
        int addr = (a == 0 || p >= a) ? ring : fifo;
        if constexpr (b > 0) addr = cb >= b ? addr : JSTRIDE - 4;          // (the zero word: the end of the k-step's FIFO block)
        return addr;
    };
    __syncthreads();
    // one step: pixel u of every lane.  Items in the order oldest operand first (the z-term, then the taps by falling a + b):
    // the two taps that need the pixel of the step before come last.
    constexpr int NITEM = C::NITEM;
    constexpr int HALF = C::BS, NBATCH = C::NBAT;   // (batch size, batches per item)
    auto step = [&](auto k_c, int u) {
        constexpr int KU = decltype(k_c)::value;                            // u % UNROLL
        constexpr int PH = KU & 3;
        // ---- HBM side of the window
        if constexpr (!HBMF) {
            if constexpr (PH == 0) {
                if (u == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no stores yet behind the first request)
                zland(zin, IC<NJ>{});          // younger than the loads that land: the stores of the window in between
                zreq(zin);
            }
            if constexpr (PH == 2) xstore();
        } else {
            if constexpr (PH == 0) {
                if (u != 0) aland(zin);        // the rows above, requested two steps ago (everything older is complete)
                zreq(zin);
            }
            if constexpr (PH == 2) {
                zland(zin, IC<0>{});
                xstore();
                areq(zin);
            }
        }
        v4f acc[NB];
#pragma unroll
        for (int sb = 0; sb < NB; ++sb) acc[sb] = (v4f){0.f, 0.f, 0.f, 0.f};
        float v[2][HALF];
        auto read_half = [&](auto i_c, auto h_c, float (&dst)[HALF]) {
            constexpr int I = decltype(i_c)::value, HH = decltype(h_c)::value;
            constexpr int item = C::order(I);
            constexpr int SJ = item == 0 ? ZJSTRIDE : JSTRIDE;
            constexpr bool FAR = NK * SJ > 65535;
            constexpr int J0 = FAR && HH * HALF >= NK / 2 ? NK / 2 : 0;    // (k-step of the base register)
            const int addr = item_addr(IC<item>{}, u) + J0 * SJ;
#ifndef FINC_BIG_ABLATE   // timing-only bit mask (experiment builds): 1 = the operands are not read from LDS; results wrong
#define FINC_BIG_ABLATE 0
#endif
#pragma unroll
            for (int j = 0; j < HALF; ++j) dst[j] = (FINC_BIG_ABLATE & 1) ? __builtin_bit_cast(float, addr + j) : ld(addr + (HH * HALF + j - J0) * SJ);
        };
        auto mma_half = [&](auto i_c, auto h_c, const float (&src)[HALF]) {
            constexpr int I = decltype(i_c)::value, HH = decltype(h_c)::value;
            constexpr int item = C::order(I);
#pragma unroll
            for (int j = 0; j < HALF; ++j)
#pragma unroll
                for (int sb = 0; sb < NB; ++sb) mma(acc[sb], (item * NK + HH * HALF + j) * NB + sb, src[j]);
        };
        read_half(IC<0>{}, IC<0>{}, v[0]);
        [&]<int... S>(std::integer_sequence<int, S...>) {
            (([&] {
                 constexpr int I = S / NBATCH, HH = S % NBATCH;             // batch S of the NBATCH * NITEM
                 if constexpr (S + 1 < NBATCH * NITEM) read_half(IC<(S + 1) / NBATCH>{}, IC<(S + 1) % NBATCH>{}, v[(S + 1) & 1]);
                 FINC_SB();
                 // (z-term: a batch whose k-steps all lie right of this wave's rows multiplies zeros -- skipped, wave-uniformly)
                 if (C::order(I) != 0 || HH * HALF <= zlast) mma_half(IC<I>{}, IC<HH>{}, v[S & 1]);
                 FINC_SB();
             }()), ...);
        }(std::make_integer_sequence<int, NBATCH * NITEM>{});
        // ---- the pixel: channel 4(JLO+sb)+q in lane row q; a lane that has not started yields exact zeros
        float xq[NB];
#pragma unroll
        for (int sb = 0; sb < NB; ++sb) xq[sb] = finc_block_reduce(acc[sb]);
        if (__builtin_expect(u < P - 1, 0)) {
            const bool started = cb >= 0;
#pragma unroll
            for (int sb = 0; sb < NB; ++sb) xq[sb] = started ? xq[sb] : 0.f;
        }
        const int slot = u & (XSLOTS - 1);
#pragma unroll
        for (int sb = 0; sb < NB; ++sb) st(ring_w + slot * 256 + sb * JSTRIDE, xq[sb]);
        if constexpr (!HBMF) {
            if (pusher) {
#pragma unroll
                for (int sb = 0; sb < NB; ++sb) st(push_w + fpush * (JS * 4) + sb * JSTRIDE, xq[sb]);
            }
        }
        ++cb; if (cb == W) cb = 0;
        ++fs2; if (fs2 == DF) fs2 = 0;
        ++fpush; if (fpush == DF) fpush = 0;
        zn = zn == 11 ? 0 : zn + 1;
        __syncthreads();
    };
    // steps u = 0 .. T, unrolled by UNROLL: the host rounds T up so that T + 1 is a multiple of it (the extra steps solve rows
    // below the image: nothing is stored)
    for (int u0 = 0; u0 <= T; u0 += UNROLL) {
        [&]<int... K>(std::integer_sequence<int, K...>) { ((step(IC<K>{}, u0 + K)), ...); }(std::make_integer_sequence<int, UNROLL>{});
    }
}

// -----------------------------------------------------------------------------------------------
// Fragment packing (fp64 math, one workgroup per group): Linv = L^-1 by forward substitution;
// item 0 (z-term) = Linv * diag(scale); item a*KW+b = -(Linv * Wc[:,:,KH-1-a,KW-1-b]).  Wave w, fragment f = (item*NK + j)*NB + sb: value (row i, k-slot q) = M_item[MO*w + 4sb + i][4j + q], in register
// f >> 2, lane (q, 4*(f & 3) + i)  (finc_tile.h: four block fragments to a register).
// -----------------------------------------------------------------------------------------------
__global__ void big_pack_kernel(const float *__restrict__ wc, const float *__restrict__ scale, float *__restrict__ packed, int Cq, int KH, int KW, int NWV, int NK, int NB, int NREG)
{
    extern __shared__ __attribute__((aligned(16))) double sm[]; // Linv [Cq][Cq]
    const int g = blockIdx.x;
    const float *wg = wc + (size_t)g * Cq * Cq * KH * KW;
    const int KK = KH * KW, MO = 4 * NB, NITEM = KK;
    double *Linv = sm;
    for (int j = threadIdx.x; j < Cq; j += blockDim.x) {      // column j of Linv: solve L y = e_j (L unit lower triangular)
        for (int r = 0; r < Cq; ++r) {
            double s = (r == j) ? 1.0 : 0.0;
            for (int k = j; k < r; ++k) s -= (double)wg[((size_t)r * Cq + k) * KK + (KK - 1)] * Linv[k * Cq + j];
            Linv[r * Cq + j] = (r < j) ? 0.0 : s;
        }
    }
    __syncthreads();
    const int total = NWV * NREG * 64;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int lane = e & 63, r = (e >> 6) % NREG, w = e / (64 * NREG);
        const int q = lane >> 4, a4 = (lane & 15) >> 2, i = lane & 3;
        const int f = 4 * r + a4;
        double v = 0.0;
        if (f < NITEM * NK * NB) {
            const int sb = f % NB;
            const int row = MO * w + 4 * sb + i;
            const int j = (f / NB) % NK, item = f / (NB * NK);
            const int col = 4 * j + q;
            if (row < Cq && col < Cq) {
                if (item == 0) {
                    v = Linv[row * Cq + col] * (scale ? (double)scale[g * Cq + col] : 1.0);
                } else {
                    const int a = item / KW, b = item % KW;
                    const int widx = (KH - 1 - a) * KW + (KW - 1 - b);
                    double s = 0.0;
                    for (int k = 0; k <= row; ++k) s += Linv[row * Cq + k] * (double)wg[((size_t)k * Cq + col) * KK + widx];
                    v = -s;
                }
            }
        }
        packed[((size_t)(g * NWV + w) * NREG + r) * 64 + lane] = (float)v;
    }
}

// -----------------------------------------------------------------------------------------------
// The FORWARD (and grad-input: the same kernel on transposed fragments and the toggled orientation) of the big banks, on the same
// split.  A workgroup of NWV waves walks one strip of 16 canonical columns of one (image, group) row by row; wave w owns the
// output channels MO*w .. MO*w + MO-1 (NB blocks) for all taps and k-steps, fragments resident as above.  The input rows are
// shared: a ring of 4 rows in LDS, [slot][k-step][column 0..19][k-slot] -- the strip's 16 columns behind 4 columns of halo --, so
// that the B operand of tap (a,b) is one conflict-free ds_read_b32 (lane (q,p) <- row h-a, column p-b, channel 4j+q).  Every
// wave fetches the NEXT row of its own NB k-steps (4*NB channel rows x 5 pieces of 16 bytes: one dwordx4 of the first lanes)
// while it multiplies the current one, and writes it into the ring behind the MFMAs; one barrier per row.  The outputs leave as
// one dword per lane (16 lanes = 64 contiguous bytes of a channel row).  W % 4 == 0 (16-byte pieces; the last strip may be partial).
// An affine map behind the conv (the ActNorm that follows the unit in the model: layers/actnorm.py:39-46) folds in as everywhere:
// the scale in the fragments' rows, the shift added to the finished pixel.
// -----------------------------------------------------------------------------------------------
template <int CQP, int KH, int KW, int NWV>
__global__ __launch_bounds__(64 * NWV) void finc_bigfwd_kernel(const float *__restrict__ in, const float *__restrict__ packed,
                                                                const float *__restrict__ shift, float *__restrict__ out, int G,
                                                                int CQ, int H, int W, int NS, unsigned orient)
{
    using C = BCfg<CQP, KH, KW, NWV>;
    constexpr int NK = C::NK, NB = C::NB, NITEM = C::NITEM, BS = C::BS, NBAT = C::NBAT;
    constexpr int COLS = 20, KSTEP_B = COLS * 16, SLOT_B = NK * KSTEP_B, NSLOT = 4;   // bytes
    static_assert(KH <= NSLOT - 1 && KW - 1 <= 4, "the ring holds the rows of a filter plus the one being fetched; the halo is one piece");
    constexpr int NF = (20 * NB + 63) / 64;    // dwordx4 loads per lane that fetch a wave's share of a row (20 * NB pieces)
    __shared__ __attribute__((aligned(16))) float lds[NSLOT * SLOT_B / 4];
    char *const ldsb = reinterpret_cast<char *>(lds);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, p = lane & 15;
    const int strip = blockIdx.x % NS, bg = blockIdx.x / NS, g = bg % G;
    const unsigned o = finc_group_orient(orient, g);
    const bool fh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 4u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    for (int i = threadIdx.x; i < NSLOT * SLOT_B / 4; i += 64 * NWV) lds[i] = 0.f;

    float fr[C::NREG];
    {
        const float *pk = packed + ((size_t)(g * NWV + wv) * C::NREG) * 64 + lane;
#pragma unroll
        for (int r = 0; r < C::NREG; ++r) fr[r] = pk[r * 64];
#pragma unroll
        for (int r = 0; r < C::NREG; ++r) {
            if (r < 128) asm volatile("" : "+a"(fr[r]));
            else asm volatile("" : "+v"(fr[r]));
        }
    }
    auto mma = [&](v4f &acc, int f, float b) { finc_mma_small(acc, fr[f >> 2], b, f & 3); };
    // the finished pixel of block sb: channel MO*w + 4sb + q in lane row q
    float sh[NB];
    unsigned ochan[NB];
#pragma unroll
    for (int sb = 0; sb < NB; ++sb) {
        const int ch = C::MO * wv + 4 * sb + q;
        sh[sb] = (shift && ch < CQ) ? shift[g * CQ + ch] : 0.f;
        ochan[sb] = ch < CQ ? (unsigned)(ch * HW * 4) : OFF_BAD_CHANNEL;
    }
    // ---- the fetch of one row: lane l < 20*NB takes piece l % 5 (canonical columns 16*strip - 4 + 4*piece ..) of channel row
    // l / 5 of this wave's 4*NB input channels
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    struct Row { v4u v[NF]; };
    bool fetcher[NF];
    unsigned fbase[NF];
    int fdst[NF];
#pragma unroll
    for (int r = 0; r < NF; ++r) {
        const int idx = lane + 64 * r;
        const int fpiece = idx % 5, frow = idx / 5;                        // frow = 4 * (k-step of the wave) + k-slot
        fetcher[r] = idx < 20 * NB;
        const int fch = 4 * NB * wv + frow;                                // input channel: k-step NB*wv + frow/4, k-slot frow%4
        const int fcol = 16 * strip - 4 + 4 * fpiece;                      // first canonical column of the piece
        const bool fok = fetcher[r] && fch < CQ && fcol >= 0 && fcol < W;    // (W % 4 == 0: a piece is inside the row or outside)
        fbase[r] = fok ? (unsigned)(fch * HW * 4 + (fw ? W - 4 - fcol : fcol) * 4) : OFF_INVALID;
        fdst[r] = ((NB * wv + frow / 4) * COLS + 4 * fpiece) * 16 + (frow % 4) * 4;   // [k-step][column][k-slot] of the piece's first column
    }
    auto fetch = [&](int h) {
        Row row;
#pragma unroll
        for (int r = 0; r < NF; ++r) {
            const unsigned off = (h < H && fbase[r] != OFF_INVALID) ? fbase[r] + (unsigned)((fh ? H - 1 - h : h) * W * 4) : OFF_INVALID;
            row.v[r] = __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0);
        }
        return row;
    };
    auto land = [&](const Row &row, int h) {
#pragma unroll
        for (int r = 0; r < NF; ++r) {
            if (fetcher[r]) {
                char *d = ldsb + (h & (NSLOT - 1)) * SLOT_B + fdst[r];
                const v4u v = row.v[r];
                const unsigned e0 = fw ? v.w : v.x, e1 = fw ? v.z : v.y, e2 = fw ? v.y : v.z, e3 = fw ? v.x : v.w;
                *reinterpret_cast<unsigned *>(d + 0) = e0;
                *reinterpret_cast<unsigned *>(d + 16) = e1;
                *reinterpret_cast<unsigned *>(d + 32) = e2;
                *reinterpret_cast<unsigned *>(d + 48) = e3;
            }
        }
    };
    __syncthreads();
    land(fetch(0), 0);
    __syncthreads();
    const int rd = ((p + 4) * 4 + q) * 4;                                  // this lane's operand of tap (a, 0): column p, k-slot q
    // (the last strip of a row that is not a multiple of 16 wide: its columns >= W are computed from zeros and not stored)
    const unsigned ocol = 16 * strip + p < W ? (unsigned)((fw ? W - 1 - (16 * strip + p) : 16 * strip + p) * 4) : OFF_INVALID;
    for (int h = 0; h < H; ++h) {
        const Row nxt = fetch(h + 1);
        v4f acc[NB];
#pragma unroll
        for (int sb = 0; sb < NB; ++sb) acc[sb] = (v4f){0.f, 0.f, 0.f, 0.f};
        float v[2][BS];
        auto read_b = [&](auto i_c, auto h_c, float (&dst)[BS]) {
            constexpr int item = decltype(i_c)::value, HH = decltype(h_c)::value;
            constexpr int a = item / KW, b = item % KW;
            // row h - a: above the image the ring still holds zeros only for h < a at the start -- mask by address: slot of a
            // row that does not exist reads the zeroed slot NSLOT-1 of the prologue?  No: rows -1, -2 map to slots 3, 2, which the
            // zero fill left untouched until rows 3, 2 land -- and those land after rows 1, 0 are done with them.
            const int base = ((h - a) & (NSLOT - 1)) * SLOT_B + rd - b * 16 + HH * BS * KSTEP_B;
#pragma unroll
            for (int j = 0; j < BS; ++j) dst[j] = *reinterpret_cast<const float *>(ldsb + base + j * KSTEP_B);
        };
        auto mma_b = [&](auto i_c, auto h_c, const float (&src)[BS]) {
            constexpr int item = decltype(i_c)::value, HH = decltype(h_c)::value;
#pragma unroll
            for (int j = 0; j < BS; ++j)
#pragma unroll
                for (int sb = 0; sb < NB; ++sb) mma(acc[sb], (item * NK + HH * BS + j) * NB + sb, src[j]);
        };
        read_b(IC<0>{}, IC<0>{}, v[0]);
        [&]<int... S>(std::integer_sequence<int, S...>) {
            (([&] {
                 constexpr int I = S / NBAT, HH = S % NBAT;
                 if constexpr (S + 1 < NBAT * NITEM) read_b(IC<(S + 1) / NBAT>{}, IC<(S + 1) % NBAT>{}, v[(S + 1) & 1]);
                 FINC_SB();
                 mma_b(IC<I>{}, IC<HH>{}, v[S & 1]);
                 FINC_SB();
             }()), ...);
        }(std::make_integer_sequence<int, NBAT * NITEM>{});
        const unsigned orow = (unsigned)((fh ? H - 1 - h : h) * W * 4) + ocol;
#pragma unroll
        for (int sb = 0; sb < NB; ++sb) {
            const float zq = finc_block_reduce(acc[sb]) + sh[sb];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, zq), rout, orow + ochan[sb], 0, 0);
        }
        land(nxt, h + 1);
        __syncthreads();
    }
}

// forward bank of a wave: fragment f = (item*NK + j)*NB + sb, value (row i, k-slot q) = scale[row] * Wc[row][4j+q][KH-1-a][KW-1-b]
// (transpose: Wc[4j+q][row], the grad-input's bank), row = MO*w + 4sb + i; four block fragments to a register
__global__ void bigfwd_pack_kernel(const float *__restrict__ wc, const float *__restrict__ scale, float *__restrict__ packed,
                                   int Cq, int KH, int KW, int NWV, int NK, int NB, int NREG, int transpose)
{
    const int g = blockIdx.y;
    const float *wg = wc + (size_t)g * Cq * Cq * KH * KW;
    const int KK = KH * KW, MO = 4 * NB;
    const int total = NWV * NREG * 64;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int lane = e & 63, r = (e >> 6) % NREG, w = e / (64 * NREG);
        const int q = lane >> 4, a4 = (lane & 15) >> 2, i = lane & 3;
        const int f = 4 * r + a4;
        float v = 0.f;
        if (f < KK * NK * NB) {
            const int sb = f % NB, j = (f / NB) % NK, item = f / (NB * NK);
            const int row = MO * w + 4 * sb + i, col = 4 * j + q;
            if (row < Cq && col < Cq) {
                const int a = item / KW, b = item % KW;
                const int oc = transpose ? col : row, ic = transpose ? row : col;
                v = wg[((size_t)oc * Cq + ic) * KK + (KH - 1 - a) * KW + (KW - 1 - b)];
                if (scale) v *= scale[g * Cq + row];
            }
        }
        packed[((size_t)(g * NWV + w) * NREG + r) * 64 + lane] = v;
    }
}

typedef void (*big_fn)(const float *, const float *, float *, int, int, int, int, int, unsigned, int);
typedef void (*bigfwd_fn)(const float *, const float *, const float *, float *, int, int, int, int, int, unsigned);
struct BInst {
    int cqp, cq_lo, kh, kw, nwv, nk, nb, nreg, lds_bytes, lds_bytes_wide;   // serves cq_lo < Cq <= cqp
    big_fn fn;       // FIFO in LDS: 16 <= W <= 64
    big_fn fn_wide;  // hand-over through memory: W >= 48
};
template <int CQP, int KH, int KW, int NWV, int CQLO, bool NARROW = true>
constexpr BInst make_binst()
{
    using C = BCfg<CQP, KH, KW, NWV, false>;
    using CW = BCfg<CQP, KH, KW, NWV, true>;
    big_fn narrow = nullptr;
    if constexpr (NARROW) narrow = finc_big_kernel<CQP, KH, KW, NWV, false>;
    return BInst{CQP, CQLO, KH, KW, NWV, C::NK, C::NB, C::NREG, C::LDS_BYTES, CW::LDS_BYTES, narrow, finc_big_kernel<CQP, KH, KW, NWV, true>};
}
// <96>: the banks beyond the wavefront kernel's table, any width.  <64>: the banks of 33 .. 64 channels on maps too WIDE for
// the wavefront kernel's K-split forms (their FIFOs: from about 130 columns at 64 channels, 250 at 48 -- the reference's
// shape sweep has a 50-channel 256 x 256 layer, fastflow/test_examples.py:218-222), hand-over through memory only.
const BInst g_binsts[] = {make_binst<96, 3, 3, 8, 64>(), make_binst<64, 3, 3, 8, 32, false>()};

// the forward's own table: (cq_lo, cqp] = the channel counts a row serves
struct BFInst {
    int cqp, kh, kw, nwv, nk, nb, nreg, cq_lo;
    bigfwd_fn fn;
};
template <int CQP, int KH, int KW, int NWV, int CQLO>
constexpr BFInst make_bfinst()
{
    using C = BCfg<CQP, KH, KW, NWV>;
    return BFInst{CQP, KH, KW, NWV, C::NK, C::NB, C::NREG, CQLO, finc_bigfwd_kernel<CQP, KH, KW, NWV>};
}
// (The same kernel on the mid-size banks -- <64,3,3> and <48,3,3> as 4 waves -- was measured against the strip kernel's K-split
// rows: 104 against 99 TFLOP/s at Cq = 64, 108 against 112 at Cq = 48, 77 against 96 at Cq = 40 on the 48-channel bank: no case
// for a second path there.  profiles/r03/notes/msplit_forward_on_mid_banks.txt)
const BFInst g_bfinsts[] = {make_bfinst<96, 3, 3, 8, 64>()};
const BFInst *find_bfinst(int Cq, int KH, int KW)
{
    for (const BFInst &i : g_bfinsts)
        if (i.kh == KH && i.kw == KW && Cq > i.cq_lo && Cq <= i.cqp) return &i;
    return nullptr;
}
// the big banks start where the wavefront kernel's table ends (finc_mfma.hip: 64 channels at 3x3).  (The same kernel on the
// SMALL banks -- <24,3,3> as 6 waves, <12,3,3> as 3 -- was measured against the role-split kernel for the under-filled chip:
// 0.93 against 0.60 us per step at c3, 0.59 against 0.47 at c2: a step is then its fixed costs, not its MFMAs.
// profiles/r03/notes/msplit_on_small_banks.txt)
const BInst *find_binst(int Cq, int KH, int KW)
{
    for (const BInst &i : g_binsts)
        if (i.kh == KH && i.kw == KW && Cq > i.cq_lo && Cq <= i.cqp) return &i;
    return nullptr;
}

int big_fifo_depth(int W, int KH, int KW) { return W - 16 + KH + KW - 2 + 1; }   // (+1: a step pushes the slot its farthest tap read)

} // namespace

// the banks this file owns outright (no row in the wavefront kernel's table) ...
bool finc_big_bank(int Cq, int KH, int KW)
{
    const BInst *i = find_binst(Cq, KH, KW);
    return i && i->fn != nullptr;
}
// ... and the ones it takes over from that table on maps too wide for it (their bank rides behind the wavefront kernel's)
bool finc_big_wide_bank(int Cq, int KH, int KW)
{
    const BInst *i = find_binst(Cq, KH, KW);
    return i && i->fn == nullptr;
}

// maps up to 64 wide keep the hand-over rows in LDS; wider ones re-read them from the output (FINC_BIG_WIDE=1: from 48 columns
// on, so that both forms can be compared on one shape)
static bool big_wide(int W, int KH, int KW)
{
    static const bool force = finc_env("FINC_BIG_WIDE") != nullptr;
    const bool fits = (size_t)big_fifo_depth(W, KH, KW) * 4 * (KH - 1) * 4 <= (size_t)FIFO_J - 4;   // (+ the zero word)
    return !fits || (force && W >= 48);
}

bool finc_big_supported(int Cq, int H, int W, int KH, int KW)
{
    const BInst *i = find_binst(Cq, KH, KW);
    if (!i || H < 1 || W < 16 || W % 4 != 0) return false;
    if (!i->fn && W < 48) return false;                                    // (a wide-only row)
    return (size_t)Cq * H * W * 4 < ((size_t)1 << 30);                     // buffer-offset range marks (OFF_BAD_CHANNEL)
}

size_t finc_big_packed_bytes(int G, int Cq, int KH, int KW)
{
    const BInst *i = find_binst(Cq, KH, KW);
    return i ? (size_t)G * i->nwv * i->nreg * 64 * sizeof(float) : 0;
}

int finc_big_pack(const float *wc, const float *scale, const float *shift, void *packed, int G, int Cq, int KH, int KW, hipStream_t st)
{
    const BInst *i = find_binst(Cq, KH, KW);
    if (!i || shift) return FINC_ERR_UNSUPPORTED;              // (a folded shift: not in this kernel)
    const size_t lds = sizeof(double) * (size_t)Cq * Cq;
    if (int e = finc_ensure_dynamic_lds((const void *)big_pack_kernel, lds)) return e;
    hipLaunchKernelGGL(big_pack_kernel, dim3(G), dim3(256), lds, st, wc, scale, (float *)packed, Cq, KH, KW, i->nwv, i->nk,
                       i->nb, i->nreg);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

int finc_big_info(const FincShape &s, int *waves, int *lds, int *cqp)
{
    const BInst *i = find_binst(s.Cq, s.KH, s.KW);
    if (!i) return FINC_ERR_UNSUPPORTED;
    *waves = i->nwv; *lds = (big_wide(s.W, s.KH, s.KW) || !i->fn) ? i->lds_bytes_wide : i->lds_bytes; *cqp = i->cqp;
    return FINC_OK;
}

int finc_big_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st)
{
    const BInst *i = find_binst(s.Cq, s.KH, s.KW);
    if (!i || !finc_big_supported(s.Cq, s.H, s.W, s.KH, s.KW)) return FINC_ERR_UNSUPPORTED;
    if (((uintptr_t)in & 15u) || ((uintptr_t)out & 15u)) return FINC_ERR_ALIGNMENT;
    const int P = 16;
    // steps of one problem: NB*W + P - 1; the last group of lane P-1 leaves in window NB*W/4 + 4, at step NB*W + 18; the loop is
    // unrolled by UNROLL and runs u = 0 .. Tr
    const int T = ((s.H + P - 1) / P) * s.W + P - 1;
    const int Tr = (T + 4 + UNROLL - 1) / UNROLL * UNROLL - 1;
    const bool wide = big_wide(s.W, s.KH, s.KW) || !i->fn;
    const big_fn fn = wide ? i->fn_wide : i->fn;
    const size_t lds = (size_t)(wide ? i->lds_bytes_wide : i->lds_bytes);
    if (int e = finc_ensure_dynamic_lds((const void *)fn, lds)) return e;
    hipLaunchKernelGGL(fn, dim3(s.B * s.G), dim3(64 * i->nwv), lds, st, in, (const float *)packed, out, s.G, s.Cq, s.H, s.W, Tr,
                       s.orient, big_fifo_depth(s.W, s.KH, s.KW));
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

// ---- forward / grad-input of the big banks: the bank sits behind the strip kernel's in the packed buffer (finc_conv.hip), with
// the G*Cq shifts of a folded affine map behind it
size_t finc_bigfwd_packed_bytes(int G, int Cq, int KH, int KW)
{
    const BFInst *i = find_bfinst(Cq, KH, KW);
    return i ? ((size_t)G * i->nwv * i->nreg * 64 + (size_t)G * Cq) * sizeof(float) : 0;
}

bool finc_bigfwd_takes(const float *in, const float *out, const FincShape &s)
{
    static const bool off = finc_env("FINC_NO_BIGFWD") != nullptr;           // A/B switch: the 8-wave K-split row of the strip kernel
    const BFInst *i = find_bfinst(s.Cq, s.KH, s.KW);
    if (!i || off || s.W % 4 != 0 || s.W < 4 || s.H < 1) return false;       // (16-byte pieces)
    if (((uintptr_t)in | (uintptr_t)out) & 15u) return false;
    return (size_t)s.Cq * s.H * s.W * 4 < ((size_t)1 << 30);
}

int finc_bigfwd_pack(const float *wc, void *packed, int G, int Cq, int KH, int KW, bool transpose, hipStream_t st, const float *scale,
                     const float *shift)
{
    const BFInst *i = find_bfinst(Cq, KH, KW);
    if (!i) return FINC_ERR_UNSUPPORTED;
    const int total = i->nwv * i->nreg * 64;
    hipLaunchKernelGGL(bigfwd_pack_kernel, dim3((total + 255) / 256, G), dim3(256), 0, st, wc, scale, (float *)packed, Cq, KH, KW, i->nwv,
                       i->nk, i->nb, i->nreg, transpose ? 1 : 0);
    FINC_CHECK_LAUNCH();
    float *sh = (float *)packed + (size_t)G * i->nwv * i->nreg * 64;
    if (shift) FINC_HIP_TRY(hipMemcpyAsync(sh, shift, sizeof(float) * (size_t)G * Cq, hipMemcpyDeviceToDevice, st));
    else FINC_HIP_TRY(hipMemsetAsync(sh, 0, sizeof(float) * (size_t)G * Cq, st));
    return FINC_OK;
}

int finc_bigfwd_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st)
{
    const BFInst *i = find_bfinst(s.Cq, s.KH, s.KW);
    if (!i || !finc_bigfwd_takes(in, out, s)) return FINC_ERR_UNSUPPORTED;
    const int NS = (s.W + 15) / 16;
    const float *sh = (const float *)packed + (size_t)s.G * i->nwv * i->nreg * 64;
    hipLaunchKernelGGL(i->fn, dim3(s.B * s.G * NS), dim3(64 * i->nwv), 0, st, in, (const float *)packed, sh, out, s.G, s.Cq, s.H, s.W,
                       NS, s.orient);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

unsigned finc_build_flags_big() { return FINC_BUILD_FLAGS; }
