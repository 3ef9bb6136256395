// MFMA wavefront kernels for the FInC Flow hot path (gfx950 / CDNA4 only).
//
// ONE WAVEFRONT OWNS ONE (image, group) PROBLEM -- no workgroup barrier, no
// inter-workgroup traffic, one launch per layer call (the reference needs
// (H+W-1)*Cq launches + device syncs, cinc_cuda_kernel_level2.cu:98-130).
//
// Mapping (inverse; DESIGN.md has the derivation):
//   * the 64 lanes are 4 rows `q` of 16 lanes `p`.  Lane p owns image rows
//     p, P+p, 2P+p, ... (P = min(16, W)) and walks each of them left to right,
//     one pixel per step, lagging lane p-1 by exactly one step.  Every step is
//     therefore one anti-diagonal segment of 16 pixels: the reference's
//     visitation order (cinc_cuda_kernel_level2.cu:49-56) restricted to a band
//     of P rows, with the bands chained back to back so the pipeline never
//     drains.
//   * per step the wave evaluates, with v_mfma_f32_16x16x4_f32 for every full 16-channel tile of the output and
//     v_mfma_f32_4x4x1_16B_f32 for every remaining 4-channel block (finc_tile.h),
//         x_new[Cq x 16px] = Linv * z  -  sum_{(a,b)!=(0,0)} (Linv * W_ab) * x[(h-a, w-b)]
//     where L = W_00 is the unit-lower-triangular corner tap.  Folding Linv
//     into the filter bank (finc_mfma_pack, fp64) removes the sequential
//     in-pixel channel substitution of the reference (the `kc < c` terms of
//     cinc_cuda_kernel_level2.cu:64-69) from the critical path.
//   * ALL filter fragments live in AGPRs for the whole kernel (162 registers
//     at Cq=24, 3x3).  The MFMA D layout (lane (q,p), reg r -> channel 4q+r,
//     pixel p) is directly a B operand of the next step once the K order is
//     permuted to match, so solved pixels never leave registers on the
//     critical path: neighbours in the same row age in place, neighbours in
//     row-a come from lane p-a by DPP row_shr:a.  Only the band hand-over
//     (row P-a of the previous band -> lanes p<a) goes through a small LDS FIFO.
//   * only the taps with a+b == 1 depend on the pixel solved in the previous
//     step; every other MFMA of step t+1 is issued while step t's result is
//     still being post-processed.
//   * HBM traffic: 16-byte loads / stores staged through per-lane LDS rings
//     (lane p is (t-p) mod 4 into its 4-column group, so per-lane LDS
//     addressing does the de-skew).  When W % 8 == 0 the unit is the aligned
//     32-byte piece and every lane moves one half of a due piece per window,
//     for its own row or for a partner row ("lane-pair I/O", below).
//
// The forward has no recurrence and uses a different, simpler mapping: finc_conv.hip.
#include "finc_common.h"
#include "finc_tile.h"

#include <stdlib.h>

#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int CQP_, int KH_, int KW_, int NW_ = 1>
struct Cfg {
    static constexpr int CQP = CQP_, KH = KH_, KW = KW_, NW = NW_;
    // Output channels: MTB full 16-row tiles (v_mfma_f32_16x16x4_f32) + NSM 4-row blocks for the rest
    // (v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4 outer products = 4 k-slots x 4 pixel quads of ONE 4-channel block;
    // its B operand is the very same register, and 2 passes instead of 8 -- so Cq = 24 costs 1 + 2*1/4 tiles of MFMA
    // time instead of 2, and neither M nor K carries padding).
    static constexpr int MTB = CQP / 16;
    static constexpr int NSM = (CQP % 16) / 4;
    static constexpr int MT = MTB + NSM;                  // accumulators / fragments per (tap, k-step)
    static constexpr int NKZT = CQP / 4;                  // k-steps of a streamed operand (k-slot q <-> channel 4j+q)
    static constexpr int NKDT = CQP / 4;                  // regs of a solved pixel as operand / for store
    // NW > 1 ("K-split"): wave w of the workgroup owns k-steps [w*NKZ, (w+1)*NKZ) of the z-term and
    // [w*NKD, (w+1)*NKD) of every tap, i.e. 1/NW of the fragments; the output registers it finalises after the
    // per-step exchange are exactly the D registers that are its own operands.
    static_assert(NW == 1 || (NKZT % NW == 0 && NKDT % NW == 0), "K-split must divide the k-steps and the output registers");
    static constexpr int NKZ = NKZT / NW;                 // per wave
    static constexpr int NKD = NKDT / NW;
    static constexpr int NK = NKD;                        // k-steps per neighbour tap (per wave)
    static constexpr int NTAP = KH * KW;
    static constexpr int NFRAG = (NKZ + (NTAP - 1) * NKD) * MT;   // fragments a wave holds
    static constexpr int NFRAGT = (NKZT + (NTAP - 1) * NKDT) * MT; // filter fragments packed per group
    // then, per group: 4*MT registers of BIAS in accumulator layout (an affine map in front of the inverse,
    // z = s*y + t, is folded into the bank -- Linv*diag(s) as the z-term, Linv*t as the accumulators' initial value) and
    // 4*MT registers of zeros (what the waves w > 0 of a K-split start from)
    static constexpr int NPACK = NFRAGT + 8 * MT;
    static constexpr int ZSLOTS = 12, XSLOTS = 8;
    static constexpr int ZRING = NKZ * ZSLOTS * 64;       // floats
    static constexpr int XRING = NKD * XSLOTS * 64;
    // K-split exchange.  Two waves: a wave leaves its share of the partner's registers in the PARTNER's x-ring slot of
    // this step -- the slot the partner is about to fill with the finished pixel anyway and that nobody reads before
    // (the oldest element a store-side read touches is 7 steps old, the end-of-step barrier orders the rest) -- so the
    // exchange needs no LDS of its own: 2 x (2 waves x 20.4 KB) = 81.7 KB lets TWO 4-wave workgroups share a CU.
    // More waves: a buffer [dst][src][reg][lane].
    static constexpr bool XALIAS = NW == 2;
    static constexpr int XCH = (NW > 1 && !XALIAS) ? NW * NW * NKD * 64 : 0;
};

// channel held by k-slot q of k-step j of an operand made from solved pixels: registers of the 16-row tiles first
// (D layout: reg r of tile mt, lane row q = channel 16mt + 4q + r), then one register per 4-row block (channel
// 16*MTB + 4sb + q after the reduce-transpose of pack_d)
__host__ __device__ inline int chan_d(int MTB, int j, int q)
{
    if (j < 4 * MTB) return 16 * (j >> 2) + 4 * q + (j & 3);
    return 16 * MTB + 4 * (j - 4 * MTB) + q;
}

template <int N>
__device__ inline float row_shr(float old, float src)
{
    // lane i of each 16-lane row <- lane i-N; lanes i < N keep `old`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                 __builtin_bit_cast(int, src), 0x110 + N, 0xf, 0xf,
                                                                 false));
}

template <int A>
struct ShiftOp {
    template <int NK>
    static __device__ inline void apply(float (&dst)[NK], const float (&fifo)[NK], const float (&src)[NK])
    {
#pragma unroll
        for (int j = 0; j < NK; ++j) dst[j] = row_shr<A>(fifo[j], src[j]);
    }
};

// accumulators -> operand/store registers.  A 16-row tile's 4 registers are operands as they are.  A 4-row block's
// register i holds, in lane row q', the k-slot-q' PARTIAL sum of channel base+i: sum over the 4 lane rows and leave
// channel base+q in lane row q (a 4x4 transpose-reduce: rows two apart by v_permlane32_swap + add, rows one apart by
// v_permlane16_swap + add).
template <class C>
__device__ inline void pack_d(const v4f (&acc)[C::MT], float (&xpk)[C::NKDT])
{
#pragma unroll
    for (int mt = 0; mt < C::MTB; ++mt) {
        xpk[4 * mt + 0] = acc[mt].x;
        xpk[4 * mt + 1] = acc[mt].y;
        xpk[4 * mt + 2] = acc[mt].z;
        xpk[4 * mt + 3] = acc[mt].w;
    }
#pragma unroll
    for (int sb = 0; sb < C::NSM; ++sb) xpk[4 * C::MTB + sb] = finc_block_reduce(acc[C::MTB + sb]);
}

// taps of the inverse's phase B (a+b >= 2), row-major
template <int KH, int KW>
struct BTaps {
    static constexpr int count()
    {
        int n = 0;
        for (int a = 0; a < KH; ++a)
            for (int b = 0; b < KW; ++b) n += (a + b >= 2);
        return n;
    }
    static constexpr int a_of(int i)
    {
        for (int a = 0; a < KH; ++a)
            for (int b = 0; b < KW; ++b)
                if (a + b >= 2 && i-- == 0) return a;
        return 0;
    }
    static constexpr int b_of(int i)
    {
        for (int a = 0; a < KH; ++a)
            for (int b = 0; b < KW; ++b)
                if (a + b >= 2 && i-- == 0) return b;
        return 0;
    }
};

#define FINC_SB() __builtin_amdgcn_sched_barrier(0)
// Put at the top of a rarely taken, wave-uniform branch body: an opaque volatile asm cannot be speculated, so the
// compiler keeps a real (scalar) branch instead of if-converting the body into per-step v_cndmask work.
#define FINC_COLD() asm volatile("; cold path")
// Timing-only ablation builds (scripts/ablate.sh): 1 = no HBM I/O, 2 = also no post-processing of the solved
// pixel, 3 = also no operand ageing / z reads.  Results are wrong for any value but 0; never shipped.
#ifndef FINC_ABLATE
#define FINC_ABLATE 0
#endif
#ifndef FINC_ABLATE_IO   // timing-only bit mask: 1 no loads, 2 no stores, 4 no landing, 8 no x-ring read
#define FINC_ABLATE_IO 0
#endif
template <int I>
using IC = std::integral_constant<int, I>;

// Diagnostic build only (-DFINC_STAMP, scripts/stamp.sh): s_memtime stamps at the region boundaries of a step,
// accumulated per (step & 3, segment) for ONE wave over the steady-state steps and left in a buffer of their own.
// The stamp's lgkmcnt(0) drains the LDS queue, so read the SHARES of such a build, never its run time.
#ifdef FINC_STAMP
__device__ unsigned long long finc_stamp_buf[64];
#define FINC_STAMP_AT(k)                                                                                              \
    do {                                                                                                              \
        FINC_SB();                                                                                                    \
        unsigned long long t_;                                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                     \
        if (stamp_on) stamp_acc[PH * 10 + (k)] += (unsigned)(t_ - stamp_prev);                                         \
        stamp_prev = t_;                                                                                              \
        FINC_SB();                                                                                                    \
    } while (0)
#else
#define FINC_STAMP_AT(k) do { } while (0)
#endif

__device__ unsigned finc_hlp_timeouts = 0;       // helper-wave protocol: waits that gave up (must stay 0)
// A give-up must not pass unnoticed (the launch would return FINC_OK with garbage in its output): besides the counter the
// kernel sets a word in HOST memory (pinned, mapped; armed once per device by the first helper-wave launch), which the next
// finc_* call on that device reads without any synchronisation and turns into FINC_ERR_LAUNCH -- sticky until
// finc_clear_fault().
__device__ unsigned *finc_fault_word = nullptr;
#ifndef FINC_HLP_BUDGET_LOG2   // test-only builds shorten the spin budget (2^20 polls ~ 0.1 s) so that an injected fault ends quickly
#define FINC_HLP_BUDGET_LOG2 20
#endif
#ifdef FINC_HLP_COUNT    // diagnostic build: how often the compute wave finds its partner late (word 0: landing, 1: x-ring reads)
__device__ unsigned finc_hlp_late[2] = {0, 0};
#define FINC_HLP_LATE(k) do { if ((threadIdx.x & 63) == 0) atomicAdd(&finc_hlp_late[k], 1u); } while (0)
extern "C" int finc_debug_hlp_late(unsigned *h) { return (int)hipMemcpyFromSymbol(h, HIP_SYMBOL(finc_hlp_late), 8); }
#else
#define FINC_HLP_LATE(k) do { } while (0)
#endif

constexpr unsigned OFF_INVALID = 0x80000000u;    // voffset beyond any slab: buffer loads return 0, stores are dropped
constexpr unsigned OFF_BAD_CHANNEL = 0x40000000u; // added to a valid offset it still lands beyond the slab (< 1 GiB)

// -----------------------------------------------------------------------------------------------
// The kernel.  grid = B*G workgroups of one wavefront.
//
// A step is cut into scheduling regions by sched_barrier(0); every region holds a block of
// independent MFMAs plus side work (LDS traffic, DPP shifts, masks, HBM loads/stores) whose inputs
// were produced at least one region earlier, so the in-order wave never waits on them.  The loop is
// unrolled x4 (x8 with 32-byte I/O) so the 4-step I/O cadence (read x ring / store / land z and issue
// loads) and the alternation of the I/O register sets fall on fixed steps.
// -----------------------------------------------------------------------------------------------
// NW waves work on one problem (K-split); NPW problems share a workgroup.  NPW exists because the waves of a small
// workgroup do not spread over the CU: with 2-wave workgroups only two of the four SIMDs ever get work (measured: the
// time of a 2-wave K-split doubles as soon as a CU holds two workgroups), so two such problems are packed into one
// 4-wave workgroup.  The packed problems share nothing but the barrier.
// S64 ("sector pairing", W % 16 == 0, one wave per problem): HBM is touched in whole 64-byte sectors.  The memory system
// behind L2 pays per request, and a 32-byte write is a partial write of the 64-byte memory granule: with 32-byte pieces
// the c3 inverse hits a wall as soon as all four SIMDs of every CU stream (B = 256: 0.48 ms against 0.41 at B = 192;
// profiles/r02/notes).  A sector's two pieces are therefore moved by the SAME lanes in two instructions back to back
// (the second is an L2 hit / completes the line the first one opened):
//   loads   at the event of a LOWER piece a lane asks for the lower piece (-> ZL, lands one event later, as before) and
//           the upper piece (-> ZU, lands two events later -- exactly when the old schedule landed it); nothing is
//           requested at the event of an upper piece.
//   stores  a completed LOWER piece is parked (PK) instead of stored; one event later the upper piece is complete and
//           both leave together.
// S64 is a bit mask: 1 = stores paired, 2 = loads paired too.
// ZU and PK live in accumulation registers -- the one resource this kernel has to spare (VGPRs: 234 of 256, LDS:
// 40.6 of 40.96 KB per wave, AGPRs: 159 of 256) -- and are read where they are: VMEM loads write AGPRs, DS and VMEM
// stores read them, so the parked data costs no VALU instruction.  The price is that these loads are inline asm, which
// hipcc does not count: the kernel waits for them itself (one s_waitcnt vmcnt per event; the instruction order of a
// window is fixed, so the count is a constant) and ALL z loads of this variant are asm so that hipcc has no VMEM load of
// its own to mis-count around them.
// HLP ("helper waves"; with S64 = 3): the workgroup is 8 waves = 4 problems.  Waves 0-3 only COMPUTE (z-ring reads, MFMAs,
// post-processing, x-ring / FIFO writes: no HBM instruction at all); waves 4-7 -- by the dispatch order of a workgroup's
// waves the SIMD partners of waves 0-3 -- run the whole HBM side of "their" problem (loads, landings, x-ring reads, parking,
// stores) in the original order.  A second wave's VALU / LDS / VMEM work does not slow an fp32 MFMA stream on the same
// SIMD (scripts/micro/mfma_helper.hip: 0.0 %), whereas the same instructions issued by the MFMA wave itself cost their
// full issue time: the I/O side was 63 us of the 438 us kernel (profiles/r02/notes/ab12).  The two
// waves of a problem meet through four monotonic words in LDS (no workgroup barrier: the four problems stay independent):
//   A1  compute -> helper   A1 >= w: the z read of step 4w-1 is in the LDS queue: the landing of window w-1 may go ahead
//   A   compute -> helper   A >= w: the compute wave is past the x-ring write of step 4w-1: window w's x-ring reads may
//   B   helper -> compute   B >= w: the landing of window w-1 is done -- checked before the z read of step 4w
//   B2  helper -> compute   B2 >= w+1: the x-ring reads of window w are done -- checked before the x-ring write of step 4w+1
// LDS executes a wave's operations in order and has no cache, so a flag written after the data is seen after the data.
// Every spin is bounded.
// ZPRE ("z premultiplied", HLP form only): the input is z' = Linv * z already -- in a flow stack the channel mix in front of
// the unit applies blockdiag(Linv) for free (fincflow_amd/glow.py) -- so the z-term's MFMAs (15 of the 162 of a
// c3 step) disappear from the compute wave.  The helper fetches channel chan_d(j, q) instead of 4j+q into k-slot q of
// register j (the addressing of the store side), so what the compute wave reads from the z ring IS the accumulators' start
// (16-row tiles) resp. the term to add behind the 4-row blocks' reduce; the folded shift of the bank is not applied (the
// caller's mix carries it).
template <int CQP, int KH, int KW, bool SEC, int NW = 1, int NPW = 1, int S64 = 0, int HLP = 0, bool ZPRE = false>
__global__ __launch_bounds__(HLP ? 512 : 64 * NW * NPW) void finc_wave_kernel(const float *__restrict__ in,
                                                            const float *__restrict__ packed, float *__restrict__ out,
                                                            int G, int CQ, int H, int W, int P, int Tend,
                                                            unsigned orient)
{
    using C = Cfg<CQP, KH, KW, NW>;
    constexpr int MT = C::MT, NKZ = C::NKZ, NKD = C::NKD, NK = C::NK, NFRAG = C::NFRAG;
    constexpr int JS = 4 * (KH - 1);          // FIFO: floats per k-step (4 k-slots x (KH-1) source lanes)
    constexpr int SS = NK * JS;               // FIFO: floats per step slot
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert(!HLP || (S64 == 3 && NW == 1 && NPW == 1), "helper waves: the sector-pairing kernel, one compute wave per problem");
    static_assert(!ZPRE || HLP, "premultiplied input: the helper-wave form only");
    const int wvt = (HLP || NW * NPW > 1) ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;  // wave of the workgroup
    const bool helper = HLP && wvt >= 4;      // waves 4-7: the HBM side of problems 0-3
    const int wv = HLP ? 0 : wvt % NW;        // wave of the problem (K-split)
    const int prob = HLP ? (wvt & 3) : wvt / NW;   // problem of the workgroup
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, p = lane & 15;
    const int bg = blockIdx.x * (HLP ? 4 : NPW) + prob;
    const int g = bg % G;
    const unsigned o = finc_group_orient(orient, g);
    const bool fh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 4u;
    const int bg_in = bg, bg_out = bg;
    const __amdgpu_buffer_rsrc_t rin =
        __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)bg_in * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout =
        __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)bg_out * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    const int D = W - P + 1;                  // FIFO depth (steps between a band's last rows and the next band's first)
    const int fifo_n = D * SS;
    // LDS: [exchange buffer (K-split only)] then per wave: z ring | x ring | FIFO + trash
    const int wave_lds = C::ZRING + C::XRING + fifo_n + SS + 64;
    float *const lds_p = lds + prob * (C::XCH + NW * wave_lds);
    float *xch = lds_p;
    float *zring = lds_p + C::XCH + wv * wave_lds;
    float *xring = zring + C::ZRING;
    // z ring: per k-step j, 12 column slots as 3 groups of 4; a group is [cell 0..63][4 columns], one 16-byte cell per
    // (lane row q, image row r) -- so a landing is ONE ds_write_b128 of the piece as it came from memory (a W-flipped
    // group reads its elements in reverse instead), and it can be written straight from an AGPR tuple.  The cell of (q, r)
    // is swizzled for both sides: the per-step reads (lane (q,p) takes element (t-p)&3 of its own cell) hit 32 different
    // banks, and the 8 lanes that ds_write_b128 serves together -- they land the rows {0,5,6,7}, {8,13,14,15}, {1,2,3,4} or
    // {9,10,11,12}, each row twice (its two groups) -- spread over 4 bank sets (2-way, the minimum; unswizzled: 6-way, and
    // the landings of a window cost 5 % of the kernel).  Low bits of the cell = (r >> 2) ^ T[r & 3], T = {0, 0, 2, 3}:
    // found by search over the Latin squares that separate both families.
    auto zcell = [](int qq, int rr) {
        const int lo = ((rr >> 2) ^ ((0x3200 >> (4 * (rr & 3))) & 3)) | ((qq & 1) << 2);
        return lo | (((rr & 3) | ((qq >> 1) << 2)) << 3);
    };
    float *fifo = xring + C::XRING;
    const int trash = fifo_n + lane;          // per-lane scratch word(s): lanes that neither push nor pop point here
    // HLP: the three progress words of this problem, behind the four problems' rings
    const unsigned flag_a = (unsigned)(uintptr_t)(lds + 4 * wave_lds + 4 * prob);   // +0: A, +4: B, +8: B2, +12: A1 (bytes)
    auto flag_set = [flag_a](auto word_c, int v) {
        constexpr int WORD = decltype(word_c)::value;
        (void)flag_a;
#ifdef FINC_HLP_INJECT_TIMEOUT   // test-only build: the helper stops announcing its landings -> the compute wave's wait gives up
        if (WORD == 1 && v >= 8) return;
#endif
        if constexpr (HLP) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(flag_a), "v"(v), "n"(4 * WORD) : "memory");
    };
    // two-stage wait for the compute role: `peek` only issues the read (no wait); `check` -- a few hundred cycles later, when
    // the value has long arrived -- tests it and falls back to the polling loop only if the partner really is late.  A wait
    // that reads and tests in one go stalls the MFMA wave for a whole LDS round trip twice per window (2 % of the kernel).
    auto flag_peek = [flag_a](auto word_c, int &reg) {
        constexpr int WORD = decltype(word_c)::value;
        (void)flag_a;
        if constexpr (HLP) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(reg) : "v"(flag_a), "n"(4 * WORD) : "memory");
    };
    auto flag_wait = [flag_a](auto word_c, int target) {
        constexpr int WORD = decltype(word_c)::value;
#ifndef FINC_HLP_NOWAIT     // timing-only bit mask of flag words whose waits are skipped (0 A, 1 B, 2 B2); results wrong
#define FINC_HLP_NOWAIT 0
#endif
        if constexpr (HLP && !((FINC_HLP_NOWAIT >> WORD) & 1)) {
            int budget = 1 << FINC_HLP_BUDGET_LOG2;   // bounded (~0.1 s): a protocol bug must not hang the GPU ...
            for (; budget > 0; --budget) {
                int v;
                asm volatile("ds_read_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(flag_a), "n"(4 * WORD) : "memory");
                if (__builtin_amdgcn_readfirstlane(v) >= target) break;
                __builtin_amdgcn_s_sleep(2);
            }
            // ... and must not pass unnoticed: finc_debug_hlp_timeouts() reads this counter (tests assert it stays 0)
            if (budget == 0 && (threadIdx.x & 63) == 0) {
                atomicAdd(&finc_hlp_timeouts, 1u);
                if (finc_fault_word) __hip_atomic_store(finc_fault_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };

    // ---- filter fragments -> registers -------------------------------------------------------
    // 16-row-tile fragments: one register each (af[f], f = the fragment's index); 4-row-block fragments: four to a
    // register (afs[s >> 2], ABID = s & 3, s = the block fragment's ordinal in issue order -- finc_tile.h).  The packed
    // bank in memory keeps one 64-lane fragment per (tap, k-step, tile): a lane of a packed register simply reads the
    // fragment its pixel quad stands for.
    constexpr int NSMALL = (NFRAG / MT) * C::NSM, NSR = (NSMALL + 3) / 4;
    float af[NFRAG];
    float afs[NSR > 0 ? NSR : 1];
    if (!helper) {
        // packed index: z-term (j*MT + mt), then taps ((tap-1)*NKDT + j)*MT + mt, j global; this wave's j = wv*N + jl
        const float *pk = packed + (size_t)g * C::NPACK * 64 + lane;
        auto gindex = [&](int f) {
            if (f < NKZ * MT) return (wv * NKZ + f / MT) * MT + f % MT;
            const int ff = f - NKZ * MT;
            return C::NKZT * MT + ((ff / (MT * NKD)) * C::NKDT + wv * NKD + (ff / MT) % NKD) * MT + ff % MT;
        };
#pragma unroll
        for (int f = 0; f < NFRAG; ++f) {
            if (f % MT >= C::MTB) continue;    // a 4-row block: packed below
            if (ZPRE && f < NKZ * MT) { af[f] = 0.f; continue; }   // (no z-term)
            if (NW == 1 && f < NKZ * MT && finc_zterm_is_zero(C::MTB, f / MT, f % MT)) { af[f] = 0.f; continue; }
            af[f] = pk[gindex(f) * 64];
        }
        const int quad = (lane & 15) >> 2;
#pragma unroll
        for (int r = 0; r < NSR; ++r) {
            int gi = 0;
#pragma unroll
            for (int a = 3; a >= 0; --a) {
                const int sfr = 4 * r + a < NSMALL ? 4 * r + a : NSMALL - 1;
                constexpr int NSMD = C::NSM > 0 ? C::NSM : 1;      // (NSR == 0 when there are no 4-row blocks: loop is empty)
                const int f = (sfr / NSMD) * MT + C::MTB + sfr % NSMD;
                const int ga = gindex(f);
                gi = (a == 3 || quad == a) ? ga : gi;
            }
            afs[r] = pk[gi * 64];
        }
        // The fragments are only ever MFMA A operands, which may be AGPRs; everything the VALU touches must be a
        // VGPR and there are only 256 of each.  Pin the fragments to AGPRs so the allocator does not shuffle
        // operands between the two files (v_accvgpr_* moves are VALU issue that f32 MFMAs do not hide).
#pragma unroll
        for (int f = 0; f < NFRAG; ++f) {
            if (f % MT >= C::MTB) continue;
            if (ZPRE && f < NKZ * MT) continue;
            if (NW == 1 && f < NKZ * MT && finc_zterm_is_zero(C::MTB, f / MT, f % MT)) continue;   // never read
            asm volatile("" : "+a"(af[f]));
        }
#pragma unroll
        for (int r = 0; r < NSR; ++r) asm volatile("" : "+a"(afs[r]));
    }
    // one accumulator update with fragment f (tile mt = f % MT)
    auto mma = [&](v4f &acc_, int f, float b) {
        const int mt = f % MT;
        if (mt < C::MTB) acc_ = __builtin_amdgcn_mfma_f32_16x16x4f32(af[f], b, acc_, 0, 0, 0);
        else {
            const int sfr = (f / MT) * C::NSM + (mt - C::MTB);
            finc_mma_small(acc_, afs[sfr >> 2], b, sfr & 3);
        }
    };
    v4f bias[MT];                             // initial value of a pixel's accumulators (zero unless an affine map is folded in)
    {
        const float *pb = packed + ((size_t)g * C::NPACK + C::NFRAGT + (wv == 0 ? 0 : 4 * MT)) * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            bias[mt] = (v4f){pb[(4 * mt + 0) * 64], pb[(4 * mt + 1) * 64], pb[(4 * mt + 2) * 64], pb[(4 * mt + 3) * 64]};
    }
    for (int i = lane; i < fifo_n + SS + 64; i += 64) fifo[i] = 0.f;

    // ---- per-lane HBM stream state -------------------------------------------------------------
    const int fl4 = -((p + 3) >> 2);          // floor(-p/4): first 4-column group this lane ever needs
    // A W-flipped group (TR/BR) is mirrored by choosing the ring SLOT of each element, never by moving data:
    // a select on a loaded value would drag the s_waitcnt for the whole HBM latency up to the issue point.
    const int k0 = fw ? 3 : 0, k1 = fw ? 2 : 1, k2 = fw ? 1 : 2, k3 = fw ? 0 : 3;
    // Offsets: lane part (row/column of the piece + the lane row's share of the channel) in the VGPR offset, the uniform
    // share of the channel in the instruction's scalar offset.  Register jj of a solved pixel holds channel
    // 16(jj>>2) + 4q + (jj&3) (a 16-row tile's D register) or 16*MTB + 4(jj-4MTB) + q (a reduced 4-row block): two lane
    // parts.  Channels >= CQ exist only in the last group of four, so only the registers that can hold one carry a
    // mask (validity must sit in the VGPR offset: the scalar offset is not range checked).
    const unsigned lane_t = 4u * (unsigned)q * HW * 4u, lane_b = (unsigned)q * HW * 4u;
    unsigned cmask[NKD];
#pragma unroll
    for (int j = 0; j < NKD; ++j) cmask[j] = chan_d(C::MTB, wv * NKD + j, q) < CQ ? 0u : OFF_BAD_CHANNEL;
    const unsigned zlast = (4 * (wv * NKZ + NKZ - 1) + q) < CQ ? 0u : OFF_BAD_CHANNEL;  // only the last k-step can hold a padded channel ...
    // ... in the one-wave kernels (the table has every multiple of 4 there).  The K-split banks also serve channel counts well
    // below CQP (Cq = 50 on the 64-channel bank): any k-step of the last waves may be padded, so each has its own mark.
    unsigned zpad[NW > 1 ? NKZ : 1];
    if constexpr (NW > 1) {
#pragma unroll
        for (int j = 0; j < NKZ; ++j) zpad[j] = (4 * (wv * NKZ + j) + q) < CQ ? 0u : OFF_BAD_CHANNEL;
    }
    auto zmark = [&](int j) { return NW > 1 ? zpad[j] : (j == NKZ - 1 ? zlast : 0u); };
    const int rowstep = (fh ? -P : P) * W * 4;                               // bytes from a row to the same column of row+P
    const int dirw = fw ? -1 : 1;

    // ================= 32-byte pieces (SEC, W % 8 == 0): lane-pair I/O =================
    // HBM is touched in aligned 32-byte pieces (two 4-column groups), the write atom of the fabric.  (16-byte pieces cost
    // 4 fetches per 64-byte sector and a 32-byte partial write per store: FETCH_SIZE 4x, WRITE_SIZE 2x.)  The rows of a
    // band fall into two classes by the parity of ceil(row/4); a row's piece is due every other window, class c rows in
    // the windows of parity c.  In such a window EVERY lane moves one 16-byte half of a due piece: the lanes of a class-c
    // row the first group, the lanes of its partner row (the i-th row of the other class) the second group.  So every
    // memory instruction has all 64 lanes busy on 32 whole pieces (half as many instructions as one piece per lane),
    // and nothing is selected or copied per lane: the two register sets (loads Z[parity], stores XS[parity]) simply
    // alternate with the window -- the main loop is unrolled x8 (two windows).  LDS does the redistribution: a landing
    // writes the OWNER lane's z-ring column, a store reads the owner's x-ring column.
    // Loads: in its window ("event", step 3) a lane lands the group it issued two windows ago and re-issues at once:
    // that is the one step at which the ring slots of both groups of a piece are free, and every load gets 8 steps.
    // Stores: at step 0 the lanes of the class that does NOT fire read their own row's first group (held one window in
    // XS[parity ^ 1]) and their partner's just completed second group (into XS[parity]); steps 1-3 store XS[parity].
    constexpr unsigned long long PARTNER = 0xCBAFED8943276501ull;           // nibble p = partner row of row p
    const int cls = ((p + 3) >> 2) & 1;
    const int part = (int)((PARTNER >> (4 * p)) & 15);
    int srvrow[2], srvhalf[2];                // window parity wp: the class-wp row this lane serves and which group of its piece
#pragma unroll
    for (int wp = 0; wp < 2; ++wp) {
        const bool own = cls == wp || p >= P; // (lanes beyond the band serve themselves: masked loads, their own ring column)
        srvrow[wp] = own ? p : part;
        srvhalf[wp] = own ? 0 : 1;
    }
    // load stream wp: piece (groups Ga, Ga+1) of row srvrow[wp] with Ga = e + 2 + fl4(row) at event window e; first
    // events: window -4 (class 0) / -3 (class 1); the piece issued at e lands at e + 2
    int lcolS[2], lrowS[2], loffS[2], lslotS[2], ldst[2];
    // store stream wp: the pair (gs-1, gs) of row srvrow[wp], gs = u - 1 + fl4(row) at fire window u; first fire
    // windows: 0 (class 0) / -1 (class 1, the prologue)
    int scolS[2], srowS[2], soffS[2];
#pragma unroll
    for (int wp = 0; wp < 2; ++wp) {
        const int R = srvrow[wp], h = srvhalf[wp];
        const int f = -((R + 3) >> 2);
        lcolS[wp] = 4 * ((wp ? -3 : -4) + 4 + f);
        lrowS[wp] = R;
        lslotS[wp] = ((4 * (((wp ? -3 : -4) + 2 + f + h) % 3 + 3)) % 12);
        ldst[wp] = 4 * zcell(q, R);
        const int c = lcolS[wp] + 4 * h;
        loffS[wp] = ((fh ? H - 1 - R : R) * W + (fw ? W - 4 - c : c)) * 4 + (4 * wv * NKZ + q) * HW * 4;
        scolS[wp] = 4 * ((wp ? -1 : 0) - 2 + f);
        srowS[wp] = R;
        const int sc = scolS[wp] + 4 * h;
        soffS[wp] = ((fh ? H - 1 - R : R) * W + (fw ? W - 4 - sc : sc)) * 4;
    }
    v4u Z[2][NKZ];                            // in flight HBM -> z ring (raw: nothing may touch it until it lands)
#pragma unroll
    for (int wp = 0; wp < 2; ++wp)
#pragma unroll
        for (int j = 0; j < NKZ; ++j) Z[wp][j] = (v4u){0u, 0u, 0u, 0u};
    float XS[2][NKD][4];                      // x ring -> HBM
#pragma unroll
    for (int wp = 0; wp < 2; ++wp)
#pragma unroll
        for (int j = 0; j < NKD; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) XS[wp][j][k] = 0.f;
    // x-ring read indices (the ring is indexed by TIME: element k of group g of row r was written at step 4g + k + r,
    // slot = step & 7): a row's first group of a pair is an even group, the second an odd one, so the slots are constants
    const int xown[4] = {((p + k0) & 7) * 64 + lane, ((p + k1) & 7) * 64 + lane, ((p + k2) & 7) * 64 + lane,
                         ((p + k3) & 7) * 64 + lane};
    const int xprt[4] = {(((part + k0) & 7) ^ 4) * 64 + q * 16 + part, (((part + k1) & 7) ^ 4) * 64 + q * 16 + part,
                         (((part + k2) & 7) ^ 4) * 64 + q * 16 + part, (((part + k3) & 7) ^ 4) * 64 + q * 16 + part};
    unsigned st_off = OFF_INVALID;
    // ---- sector pairing state (S64: bit 0 stores, bit 1 loads)
    static_assert(!S64 || (SEC && NW == 1 && NPW == 1), "sector pairing: one wave per problem, 32-byte pieces");
    constexpr bool S64S = (S64 & 1) != 0, S64L = (S64 & 2) != 0;
    // the lower pieces of the load side stay in Z (VGPRs, written by asm loads in this variant); the upper pieces (ZU) and
    // the parked store pieces (PK) are the new state and live in AGPRs
    constexpr bool PK_AGPR = S64S;
    // HLP: two waves per SIMD share the register file, and hipcc splits a wave's 256 registers 128 : 128 between VGPRs and
    // AGPRs.  The helper role fits only if the lower pieces (ZA) wait in AGPRs too and one parked set stays in VGPRs; it
    // MUST fit: a spilled in-flight load destination is saved before the data has arrived (hipcc takes an asm output for
    // complete).
    constexpr bool ZA_AGPR = HLP != 0;
    v4u ZU[2][S64L ? NKZ : 1], ZA[2][ZA_AGPR ? NKZ : 1];
    v4f PK[2][S64S ? NKD : 1];                // parked lower pieces of the store side
    int lphi[2] = {0, 0};                     // phase (0 lower / 1 upper) of the piece of the stream's previous event
    bool spark = false;                       // this window's piece is a lower one: park it
    unsigned st_off2 = OFF_INVALID;           // where the parked lower piece goes when its upper piece leaves
    auto s64_init = [&]() {                    // zeros in the waiting sets (what lands before anything was requested)
        if constexpr (S64L) {
#pragma unroll
            for (int wp = 0; wp < 2; ++wp)
#pragma unroll
                for (int j = 0; j < NKZ; ++j) {
                    ZU[wp][j] = (v4u){0u, 0u, 0u, 0u}; asm volatile("" : "+a"(ZU[wp][j]));
                    if constexpr (ZA_AGPR) { ZA[wp][j] = (v4u){0u, 0u, 0u, 0u}; asm volatile("" : "+a"(ZA[wp][j])); }
                }
        }
        if constexpr (S64S) {
#pragma unroll
            for (int wp = 0; wp < 2; ++wp)
#pragma unroll
                for (int j = 0; j < NKD; ++j) {
                    PK[wp][j] = (v4f){0.f, 0.f, 0.f, 0.f};
                    if constexpr (PK_AGPR) { if (!HLP || wp == 0) asm volatile("" : "+a"(PK[wp][j])); }
                }
        }
    };
    if constexpr (!HLP) s64_init();            // (HLP: at the entry of the helper role -- the compute role never touches these)
    // S64 landing: `vm` = the s_waitcnt vmcnt immediate that covers the lower pieces issued two windows ago (everything
    // older, the upper pieces of four windows ago included, is then complete as well: the counter is in order)
    // (part_c: 0 = the whole event; 1 = only the landing, 2 = only the requests -- the helper role does its urgent x-ring
    // reads between the two)
    auto s64_part = [&](auto wp_c, auto vm_c, int pub_b, auto part_c) {
        constexpr int WP = decltype(wp_c)::value;
        constexpr int VM = decltype(vm_c)::value;
        constexpr int PART = decltype(part_c)::value;
        if constexpr (S64L && PART != 2) {
            float *dst = zring + lslotS[WP] * 64 + ldst[WP];
#ifndef FINC_S64_ABLATE   // timing-only bits: 1 no vmcnt wait, 2 no landing writes, 4 no load issue
#define FINC_S64_ABLATE 0
#endif
            if constexpr (!(FINC_S64_ABLATE & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
            if constexpr (FINC_S64_ABLATE & 2) {
            } else if (lphi[WP] == 0) {        // (divergent; both sides always have lanes: W % 16 == 0 keeps the rows' phases apart)
                if constexpr (ZA_AGPR) {
                    const unsigned ba = (unsigned)(uintptr_t)dst;
#pragma unroll
                    for (int j = 0; j < NKZ; ++j)      // (the comment keeps the two sides' asm different: identical asm is merged over a select of the tuples)
                        asm volatile("ds_write_b128 %0, %1 offset:%2 ; lower piece" ::"v"(ba), "a"(ZA[WP][j]), "n"(j * C::ZSLOTS * 256) : "memory");
                } else {
#pragma unroll
                    for (int j = 0; j < NKZ; ++j) *reinterpret_cast<v4u *>(dst + j * C::ZSLOTS * 64) = Z[WP][j];
                }
            } else {                           // the upper piece goes from its AGPR tuple to the ring as it is
                const unsigned ba = (unsigned)(uintptr_t)dst;
#pragma unroll
                for (int j = 0; j < NKZ; ++j)
                    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(ba), "a"(ZU[WP][j]), "n"(j * C::ZSLOTS * 256) : "memory");
            }
            if constexpr (HLP) {               // the landing is in the LDS queue: tell the compute wave (HLP)
                if (pub_b != -0x7fffffff) flag_set(IC<1>{}, pub_b);
            }
            lslotS[WP] = lslotS[WP] >= 4 ? lslotS[WP] - 4 : lslotS[WP] + 8;   // + 8 mod 12
        }
        if constexpr (S64L && PART != 1) {
            const int phi = ((lcolS[WP] + 64) >> 3) & 1;
            lphi[WP] = phi;
            if ((FINC_S64_ABLATE & 4) == 0 && phi == 0) {   // a lower piece: ask for the whole sector (exec-masked: the other lanes' ZU is waiting to land)
                const bool ok = lcolS[WP] >= 0 && lrowS[WP] < H && p < P;
                const unsigned vl = ok ? (unsigned)loffS[WP] : OFF_INVALID;
                const unsigned vu = ok ? (unsigned)(loffS[WP] + 32 * dirw) : OFF_INVALID;
#pragma unroll
                for (int j = 0; j < NKZ; ++j) {
                    unsigned ol = j == NKZ - 1 ? vl + zlast : vl, ou = j == NKZ - 1 ? vu + zlast : vu;
                    int so = __builtin_amdgcn_readfirstlane(j * 16 * HW);
                    if constexpr (ZPRE) {      // register j = D register j of the pixel: the store side's channel map
                        const bool tile = j < 4 * C::MTB;
                        const bool last_group = tile ? (C::NSM == 0 && (j >> 2) == C::MTB - 1) : j == C::NKDT - 1;
                        const unsigned add = (tile ? lane_t - lane_b : 0u) + ((last_group && C::CQP != CQ) ? cmask[j] : 0u);
                        ol = vl + add; ou = vu + add;       // (loffS carries lane_b = q*HW*4, the k-slot's share of a block register)
                        so = __builtin_amdgcn_readfirstlane((tile ? 16 * (j >> 2) + (j & 3) : 16 * C::MTB + 4 * (j - 4 * C::MTB)) * HW * 4);
                    }
                    if constexpr (ZA_AGPR)
                        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=a"(ZA[WP][j]) : "v"(ol), "s"(rin), "s"(so) : "memory");
                    else
                        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(Z[WP][j]) : "v"(ol), "s"(rin), "s"(so) : "memory");
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=a"(ZU[WP][j]) : "v"(ou), "s"(rin), "s"(so) : "memory");
                }
            }
            lcolS[WP] += 8;
            loffS[WP] += 32 * dirw;
            if (lcolS[WP] == W) { lcolS[WP] = 0; lrowS[WP] += P; loffS[WP] += rowstep - dirw * W * 4; }
        }
    };

    auto s64_event = [&](auto wp_c, auto vm_c, int pub_b = -0x7fffffff) { s64_part(wp_c, vm_c, pub_b, IC<0>{}); };

    auto sec_event = [&](auto wp_c) {
        constexpr int WP = decltype(wp_c)::value;
        if constexpr (!SEC) return;
        if constexpr (S64L) return;            // (s64_event is called in its place)
        float *dst = zring + lslotS[WP] * 64 + ldst[WP];
#pragma unroll
        for (int j = 0; j < NKZ; ++j) *reinterpret_cast<v4u *>(dst + j * C::ZSLOTS * 64) = Z[WP][j];
        lslotS[WP] = lslotS[WP] >= 4 ? lslotS[WP] - 4 : lslotS[WP] + 8;   // + 8 mod 12
        const bool ok = lcolS[WP] >= 0 && lrowS[WP] < H && p < P;
        const unsigned vb = ok ? (unsigned)loffS[WP] : OFF_INVALID;
#pragma unroll
        for (int j = 0; j < NKZ; ++j)
            Z[WP][j] = __builtin_amdgcn_raw_buffer_load_b128(rin, (NW > 1 || j == NKZ - 1) ? vb + zmark(j) : vb, j * 16 * HW, 0);
        lcolS[WP] += 8;
        loffS[WP] += 32 * dirw;
        if (lcolS[WP] == W) { lcolS[WP] = 0; lrowS[WP] += P; loffS[WP] += rowstep - dirw * W * 4; }
    };
    auto sec_sread = [&](auto wp_c, int pub_b2 = -0x7fffffff) {   // window of parity WP: class WP fires
        constexpr int WP = decltype(wp_c)::value;
        if constexpr (!SEC) return;
        if (cls != WP && p < P) {              // (divergent) own row: first group of its next pair; partner: second group, due now
#pragma unroll
            for (int j = 0; j < NKD; ++j) {
                const float *xo = xring + j * C::XSLOTS * 64, *xp = xring + j * C::XSLOTS * 64;
                XS[WP ^ 1][j][0] = xo[xown[0]];
                XS[WP ^ 1][j][1] = xo[xown[1]];
                XS[WP ^ 1][j][2] = xo[xown[2]];
                XS[WP ^ 1][j][3] = xo[xown[3]];
                XS[WP][j][0] = xp[xprt[0]];
                XS[WP][j][1] = xp[xprt[1]];
                XS[WP][j][2] = xp[xprt[2]];
                XS[WP][j][3] = xp[xprt[3]];
            }
        }
        if constexpr (HLP) {                   // the reads are in the LDS queue: once they are back, tell the compute wave
            if (pub_b2 != -0x7fffffff) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                flag_set(IC<2>{}, pub_b2);
            }
        }
        const bool ok = scolS[WP] >= 0 && srowS[WP] < H && p < P;
        st_off = ok ? (unsigned)soffS[WP] : OFF_INVALID;
        if constexpr (S64S) {                  // lower piece: park it; upper piece: leaves with the parked lower one
            const int phi = ((scolS[WP] + 64) >> 3) & 1;
            spark = phi == 0;
            st_off2 = (ok && phi == 1) ? (unsigned)(soffS[WP] - 32 * dirw) : OFF_INVALID;
            if (phi == 0) st_off = OFF_INVALID;
        }
        scolS[WP] += 8;
        soffS[WP] += 32 * dirw;
        if (scolS[WP] == W) { scolS[WP] = 0; srowS[WP] += P; soffS[WP] += rowstep - dirw * W * 4; }
    };
    auto sec_swrite = [&](auto wp_c, int j0, int j1) {
        constexpr int WP = decltype(wp_c)::value;
        if constexpr (!SEC) return;
        const unsigned vt = st_off == OFF_INVALID ? OFF_INVALID : st_off + lane_t;
        const unsigned vb = st_off == OFF_INVALID ? OFF_INVALID : st_off + lane_b;
        const int cpad = C::CQP - CQ;                           // padded channels (uniform; 0..3 in the one-wave kernels)
#pragma unroll
        for (int j = 0; j < NKD; ++j) {
            if (j < j0 || j >= j1) continue;
            const int jj = wv * NKD + j;
            const bool tile = jj < 4 * C::MTB;
            // (readfirstlane: tells the compiler this is wave-uniform, or it wraps every store in a waterfall loop)
            const int uni = __builtin_amdgcn_readfirstlane((tile ? 16 * (jj >> 2) + (jj & 3) : 16 * C::MTB + 4 * (jj - 4 * C::MTB)) * HW * 4);
            unsigned vo = tile ? vt : vb;
            // only a register of the last channel group can hold a padded channel
            const bool last_group = tile ? (C::NSM == 0 && (jj >> 2) == C::MTB - 1) : jj == C::NKDT - 1;
            if ((NW > 1 || last_group) && cpad != 0) vo += cmask[j];   // (K-split banks: any register may hold a padded channel)
            v4u v;
            v.x = __builtin_bit_cast(unsigned, XS[WP][j][0]);
            v.y = __builtin_bit_cast(unsigned, XS[WP][j][1]);
            v.z = __builtin_bit_cast(unsigned, XS[WP][j][2]);
            v.w = __builtin_bit_cast(unsigned, XS[WP][j][3]);
            if constexpr (S64S) {              // the parked lower piece first: two instructions, one whole sector
                const unsigned vo2 = st_off2 == OFF_INVALID ? OFF_INVALID : vo - st_off + st_off2;
                if constexpr (PK_AGPR && (!HLP || WP == 0)) {
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" ::"a"(PK[WP][j]), "v"(vo2), "s"(rout), "s"(uni) : "memory");
                } else {
                    const float e0 = PK[WP][j].x, e1 = PK[WP][j].y, e2 = PK[WP][j].z, e3 = PK[WP][j].w;
                    v4u w;
                    w.x = __builtin_bit_cast(unsigned, e0);
                    w.y = __builtin_bit_cast(unsigned, e1);
                    w.z = __builtin_bit_cast(unsigned, e2);
                    w.w = __builtin_bit_cast(unsigned, e3);
                    __builtin_amdgcn_raw_buffer_store_b128(w, rout, vo2 + (unsigned)uni, 0, 0);   // (offset in the vector operand: scripts/check_store_hazard.py)
                }
            }
            __builtin_amdgcn_raw_buffer_store_b128(v, rout, vo + (unsigned)uni, 0, 0);     // (offset in the vector operand: scripts/check_store_hazard.py)
        }
        if constexpr (S64S) {
            if (spark) {                       // (divergent) park this window's lower piece where the upper one will find it
                asm volatile("; park");
#pragma unroll
                for (int j = 0; j < NKD; ++j) {
                    if (j < j0 || j >= j1) continue;
                    PK[WP][j] = (v4f){XS[WP][j][0], XS[WP][j][1], XS[WP][j][2], XS[WP][j][3]};
                    if constexpr (PK_AGPR && (!HLP || WP == 0)) asm volatile("" : "+a"(PK[WP][j]));
                }
            }
        }
    };

    // ================= 16-byte pieces (W % 4 == 0 only): one group per lane and window =================
    int lcol = 4 * fl4, lrow = p;
    int scol = 4 * (fl4 - 2), srow = p;       // next group to store
    int lslot = ((4 * fl4) % 12 + 12) % 12;   // z-ring slot of the next landing
    v4u zb[NKZ];
#pragma unroll
    for (int j = 0; j < NKZ; ++j) zb[j] = (v4u){0u, 0u, 0u, 0u};
    float sv[NKD][4];                         // group just read from the x ring
    // x-ring read indices of the group to store: slot of element k = (4gs + k + p) & 7 = ((p+k)&7) ^ 4*(gs&1)
    int stog = 256 * (fl4 & 1);               // 4 slots x 64 floats, toggled once per window (gs parity)
    // Load offsets are kept incrementally: canonical column +4 is a constant byte step in memory, a row change another
    // constant; the per-channel part j*4*HW*4 is uniform and rides in the buffer instruction's scalar offset.
    int loff = ((fh ? H - 1 - lrow : lrow) * W + (fw ? W - 4 - lcol : lcol)) * 4 + (4 * wv * NKZ + q) * HW * 4;
    auto io_issue = [&](int j0, int j1, bool advance) {
        const bool ok = lcol >= 0 && lrow < H && p < P;
        const unsigned vb = ok ? (unsigned)loff : OFF_INVALID;
#pragma unroll
        for (int j = 0; j < NKZ; ++j) {
            if (j < j0 || j >= j1) continue;
            zb[j] = __builtin_amdgcn_raw_buffer_load_b128(rin, (NW > 1 || j == NKZ - 1) ? vb + zmark(j) : vb, j * 16 * HW, 0);
        }
        if (advance) {
            lcol += 4;
            loff += 16 * dirw;
            if (lcol == W) { lcol = 0; lrow += P; loff += rowstep - dirw * W * 4; }
        }
    };
    auto io_land = [&]() {
        float *dst = zring + lslot * 64 + 4 * zcell(q, p);
#pragma unroll
        for (int j = 0; j < NKZ; ++j) *reinterpret_cast<v4u *>(dst + j * C::ZSLOTS * 64) = zb[j];
        lslot = lslot == 8 ? 0 : lslot + 4;
    };
    auto io_sread = [&]() {
        const bool ok = scol >= 0 && srow < H && p < P;
        const int mrow = fh ? H - 1 - srow : srow;
        const int mcol = fw ? W - 4 - scol : scol;
        st_off = ok ? (unsigned)(mrow * W + mcol) * 4u : OFF_INVALID;
        // the x ring is indexed by TIME (slot = step & 7): element k of this group was written at step 4gs+k+p
        const float *b0 = xring + (xown[0] ^ stog), *b1 = xring + (xown[1] ^ stog);
        const float *b2 = xring + (xown[2] ^ stog), *b3 = xring + (xown[3] ^ stog);
#pragma unroll
        for (int j = 0; j < NKD; ++j) {
            sv[j][0] = b0[j * C::XSLOTS * 64];
            sv[j][1] = b1[j * C::XSLOTS * 64];
            sv[j][2] = b2[j * C::XSLOTS * 64];
            sv[j][3] = b3[j * C::XSLOTS * 64];
        }
        stog ^= 256;
        scol += 4;
        if (scol == W) { scol = 0; srow += P; }
    };
    auto io_swrite = [&](int j0, int j1) {
        const unsigned vt = st_off == OFF_INVALID ? OFF_INVALID : st_off + lane_t;
        const unsigned vb = st_off == OFF_INVALID ? OFF_INVALID : st_off + lane_b;
        const int cpad = C::CQP - CQ;
#pragma unroll
        for (int j = 0; j < NKD; ++j) {
            if (j < j0 || j >= j1) continue;
            const int jj = wv * NKD + j;
            const bool tile = jj < 4 * C::MTB;
            const int uni = __builtin_amdgcn_readfirstlane((tile ? 16 * (jj >> 2) + (jj & 3) : 16 * C::MTB + 4 * (jj - 4 * C::MTB)) * HW * 4);
            unsigned vo = tile ? vt : vb;
            const bool last_group = tile ? (C::NSM == 0 && (jj >> 2) == C::MTB - 1) : jj == C::NKDT - 1;
            if ((NW > 1 || last_group) && cpad != 0) vo += cmask[j];
            v4u v;
            v.x = __builtin_bit_cast(unsigned, sv[j][0]);
            v.y = __builtin_bit_cast(unsigned, sv[j][1]);
            v.z = __builtin_bit_cast(unsigned, sv[j][2]);
            v.w = __builtin_bit_cast(unsigned, sv[j][3]);
            __builtin_amdgcn_raw_buffer_store_b128(v, rout, vo + (unsigned)uni, 0, 0);     // (offset in the vector operand: scripts/check_store_hazard.py)
        }
    };
    // the window's HBM work, spread over its 4 steps: step 0 reads the x ring; steps 1-3 store a third of the registers
    // each; step 3 (32-byte pieces) lands and re-issues / steps 2-3 (16-byte pieces) land and issue the two halves
    constexpr int S1 = (NKD + 2) / 3, S2 = (2 * NKD + 2) / 3, L1 = (NKZ + 1) / 2;
    auto io_phase = [&](auto ph_c, auto wp_c) {
        constexpr int PH = decltype(ph_c)::value;
        if constexpr (FINC_ABLATE >= 1) return;
        constexpr int AB = FINC_ABLATE_IO;
        if constexpr (SEC) {
            if constexpr (PH == 0 && !(AB & 8)) sec_sread(wp_c);
            if constexpr (PH == 1 && !(AB & 2)) sec_swrite(wp_c, 0, S1);
            if constexpr (PH == 2 && !(AB & 2)) sec_swrite(wp_c, S1, S2);
            if constexpr (PH == 3 && !(AB & 2)) sec_swrite(wp_c, S2, NKD);
            if constexpr (PH == 3 && !(AB & 5)) {
                sec_event(wp_c);
                // VMEM instructions younger than the lower-piece loads of two windows ago: that event's last upper load
                // (1) + a whole window (2*NKD stores, 2*NKZ loads) + this window's 2*NKD stores
                s64_event(wp_c, IC<1 + (S64 & 1 ? 2 : 1) * NKD + 2 * NKZ + (S64 & 1 ? 2 : 1) * NKD>{});
            }
        } else {
            if constexpr (PH == 0 && !(AB & 8)) io_sread();
            if constexpr (PH == 1 && !(AB & 2)) io_swrite(0, S1);
            if constexpr (PH == 2) {
                if constexpr (!(AB & 2)) io_swrite(S1, S2);
                if constexpr (!(AB & 4)) io_land();
                if constexpr (!(AB & 1)) io_issue(0, L1, false);
            }
            if constexpr (PH == 3) {
                if constexpr (!(AB & 2)) io_swrite(S2, NKD);
                if constexpr (!(AB & 1)) io_issue(L1, NKZ, true);
            }
        }
    };

    // ---- neighbour operands --------------------------------------------------------------------
    // S_a(tau) = row_shr:a of the pixels solved at step tau (S_0 = the pixels themselves).  Tap (a,b) of step t reads
    // S_a(t-a-b).  ROT (every operand dies within 4 steps, i.e. KH-1+KW < 6): the S_a live in period-4 rings indexed
    // by (step & 3), which is a compile-time constant in the x4-unrolled loop, so ageing an operand costs NO
    // instruction.  Otherwise: explicit ageing arrays R (copy per step) and delay lines DL.
    constexpr bool ROT = (KH - 1 + KW) < 6;
    float Q[KH][4][NK];                       // ROT: Q[a][tau & 3] = S_a(tau)
    float R[KH][KW][NK];                      // !ROT: R[a][b] = operand of tap (a,b) for the current step
    float DL[KH][KH][NK];                     // !ROT: DL[a][k]: row_shr:a copies that are k+1 steps old
#pragma unroll
    for (int a = 0; a < KH; ++a) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < NK; ++j) Q[a][k][j] = 0.f;
#pragma unroll
        for (int b = 0; b < KW; ++b)
#pragma unroll
            for (int j = 0; j < NK; ++j) R[a][b][j] = 0.f;
#pragma unroll
        for (int k = 0; k < KH; ++k)
#pragma unroll
            for (int j = 0; j < NK; ++j) DL[a][k][j] = 0.f;
    }
    float fv[KH][NK];                         // FIFO pops (row a of the previous band) waiting for their DPP merge
#pragma unroll
    for (int a = 0; a < KH; ++a)
#pragma unroll
        for (int j = 0; j < NK; ++j) fv[a][j] = 0.f;

    // FIFO: lanes P-(KH-1)..P-1 push their operand regs every step; lanes p < a pop lane P-a+p of D-1 steps ago.
    // pointer = base + slot * stride with per-lane constants (stride 0 / base = trash word for lanes not taking
    // part): one v_mad per pointer per step.
    const int push_l = p - (P - (KH - 1));
    const bool do_push = KH > 1 && push_l >= 0 && p < P;
    const int push_base = do_push ? q * (KH - 1) + push_l : trash;
    const int push_stride = do_push ? SS : 0;
    int pop_base[KH], pop_stride[KH];
#pragma unroll
    for (int a = 1; a < KH; ++a) {
        pop_base[a] = p < a ? q * (KH - 1) + (KH - 1 - a + p) : trash;
        pop_stride[a] = p < a ? SS : 0;
    }
    int fslot = 0;
    float *push_p = fifo + push_base;         // = fifo + push_base + fslot * push_stride, kept incrementally (bytes)
    float *pop_p[KH];                         // = fifo + pop_base + ((fslot+1) % D) * pop_stride
#pragma unroll
    for (int a = 1; a < KH; ++a) pop_p[a] = fifo + pop_base[a] + (D > 1 ? pop_stride[a] : 0);
    // (exec-masked pushes / pops instead of the trash words: 1.2 % slower, profiles/r02/notes/ab30)
    auto fifo_push = [&](const float (&v)[NK]) {
#pragma unroll
        for (int j = 0; j < NK; ++j) push_p[j * JS] = v[j];
    };
    auto fifo_pop_all = [&]() {
#pragma unroll
        for (int a = 1; a < KH; ++a) {
#pragma unroll
            for (int j = 0; j < NK; ++j) fv[a][j] = pop_p[a][j * JS];
        }
    };
    auto fifo_advance = [&]() {               // one v_add per pointer; the rewinds are real scalar branches (1 in D steps)
        ++fslot;
        push_p += push_stride;
#pragma unroll
        for (int a = 1; a < KH; ++a) pop_p[a] += pop_stride[a];
        if (__builtin_expect(fslot == D, 0)) {
            FINC_COLD();
            fslot = 0;
            push_p = fifo + push_base;
        }
        if (__builtin_expect(fslot + 1 == D, 0)) {
            FINC_COLD();
#pragma unroll
            for (int a = 1; a < KH; ++a) pop_p[a] = fifo + pop_base[a];
        }
    };
    // S_a(t) for a >= 1: row_shr:a(src), lanes p < a take the FIFO value.
    auto shift_all = [&](const float (&src)[NK], auto ph_c) {
        constexpr int PHX = decltype(ph_c)::value;
        if constexpr (KH > 1) {
            float sn[KH][NK];
            ShiftOp<1>::apply(sn[1], fv[1], src);
            if constexpr (KH > 2) ShiftOp<2>::apply(sn[2], fv[2], src);
            if constexpr (KH > 3) ShiftOp<3>::apply(sn[3], fv[3], src);
            if constexpr (KH > 4) ShiftOp<4>::apply(sn[4], fv[4], src);
            if constexpr (KH > 5) ShiftOp<5>::apply(sn[5], fv[5], src);
            if constexpr (KH > 6) ShiftOp<6>::apply(sn[6], fv[6], src);
            if constexpr (ROT) {
#pragma unroll
                for (int a = 1; a < KH; ++a)
#pragma unroll
                    for (int j = 0; j < NK; ++j) Q[a][PHX][j] = sn[a][j];
            } else {                           // R[a][0](t+1) = S_a(t+1-a): a-1 steps of delay line
#pragma unroll
                for (int j = 0; j < NK; ++j) R[1][0][j] = sn[1][j];
#pragma unroll
                for (int a = 2; a < KH; ++a) {
#pragma unroll
                    for (int k = a - 2; k >= 1; --k)
#pragma unroll
                        for (int j = 0; j < NK; ++j) DL[a][k][j] = DL[a][k - 1][j];
#pragma unroll
                    for (int j = 0; j < NK; ++j) DL[a][0][j] = sn[a][j];
                }
            }
        }
    };

    if constexpr (HLP) {                      // A = 0, B = -1, B2 = 0
        if (!helper) { flag_set(IC<0>{}, 0); flag_set(IC<1>{}, -1); flag_set(IC<2>{}, 0); flag_set(IC<3>{}, 0); }
    }
    __syncthreads(); // single wave: orders the FIFO zero-fill before use (HLP: and the flags' initial values)

    if constexpr (HLP) {
        if (helper) {
            // ===== the HBM side of one problem, in the order the single-wave kernel issues it =====
            // (Priority: at equal priority the helper gets only the bubbles of its partner's MFMA stream -- ~60 cycles per VALU
            // instruction, scripts/micro/mfma_helper.hip -- and its landing is late for two checks out of three; raised, it is
            // late for one in seven, but every instruction it then issues first delays the MFMA wave and the kernel is 2 %
            // SLOWER (profiles/r02/notes/ab18): the partner's slack is the pole wave's cover.  It stays at 0.)
            s64_init();
            const int wlast = Tend >> 2;       // the window after the last computed one (even: Tend % 8 == 0)
            s64_event(IC<0>{}, IC<0>{});       // window -4: the first pieces of the class-0 rows leave
            s64_event(IC<1>{}, IC<0>{});       // window -3: class 1
            s64_event(IC<0>{}, IC<0>{});       // window -2: the first pieces land, the second ones leave
            sec_sread(IC<1>{});                // window -1
            sec_swrite(IC<1>{}, 0, NKD);
            s64_event(IC<1>{}, IC<0>{}, 0);    // its landing is what the compute wave's first z read waits for: B = 0
            sec_sread(IC<0>{});                // window 0 (nothing has been written to the x ring yet: no wait)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            flag_set(IC<2>{}, 1);
            sec_swrite(IC<0>{}, 0, NKD);
            constexpr int VMS = 1 + 2 * NKD + 2 * NKZ + 2 * NKD;
            for (int w = 1; w < wlast; w += 2) {
                // odd window w: the event of window w-1 (parity 0) and window w's stores (parity 1).  What the compute wave
                // waits for goes first -- the landing, then the x-ring reads; the requests and the stores have all the time
                flag_wait(IC<3>{}, w);
                s64_part(IC<0>{}, IC<VMS>{}, w, IC<1>{});
                flag_wait(IC<0>{}, w);
                sec_sread(IC<1>{}, w + 1);
                s64_part(IC<0>{}, IC<VMS>{}, 0, IC<2>{});
                sec_swrite(IC<1>{}, 0, NKD);
                flag_wait(IC<3>{}, w + 1);     // even window w+1
                s64_part(IC<1>{}, IC<VMS>{}, w + 1, IC<1>{});
                flag_wait(IC<0>{}, w + 1);
                sec_sread(IC<0>{}, w + 2);
                s64_part(IC<1>{}, IC<VMS>{}, 0, IC<2>{});
                sec_swrite(IC<0>{}, 0, NKD);
            }
            return;
        }
    }

    // pre-loop = the HBM side of the two windows before the first computed one
    if constexpr (HLP) {
    } else if constexpr (SEC) {
        sec_event(IC<0>{});  // window -4: the first pieces of the class-0 rows leave
        sec_event(IC<1>{});  // window -3: class 1
        sec_event(IC<0>{});  // window -2: the first pieces land, the second ones leave
        s64_event(IC<0>{}, IC<0>{});  // (S64: the same three events; before the loop they simply wait for everything)
        s64_event(IC<1>{}, IC<0>{});
        s64_event(IC<0>{}, IC<0>{});
    } else {
        io_issue(0, NKZ, true);
        io_land();
        io_issue(0, NKZ, true);  // left in flight, lands in window -1
    }

    int nslot = ((-3 - p) % 12 + 12) % 12;    // z-ring slot of the position of step t+1 ...
    // ... and its address: group (slot >> 2), own cell, element (slot & 3) -- counted from the other end in a W-flipped group
    const float *zrd = zring + (nslot >> 2) * 256 + 4 * zcell(q, p) + (fw ? 3 - (nslot & 3) : (nslot & 3));
    // per-step advance of zrd: +-1 inside a group, a jump to the next group when the new slot starts one.  After the advance
    // of step t the slot is (t + 2 - p) mod 12, so which lanes jump is a function of t & 3: four per-lane constants.
    int zstep[4];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) zstep[ph] = (((ph + 2 - p) & 3) == 0) ? 256 - 3 * dirw : dirw;
    int tp1 = -3;                             // t+1 (scalar).  Lane p has started its chain iff p <= t+1 ...
    int tm = -3;                              // ... and its NEXT position starts a row iff p == (t+1) mod W =: tm
    int xwin = 256 + lane;                    // x-ring write index of this window: 256*((t>>2)&1) + lane, t = -4
    int win = 0;                              // HLP: index of the window being computed (main loop)
    int peek_b = -1, peek_b2 = 0;             // HLP: flag values read ahead of their checks

    {
        // =========================== inverse ===========================
        constexpr int FZ = 0;                          // z-term fragments: j*MT + mt
        constexpr int FT = NKZ * MT;                   // tap fragments: FT + (((a*KW+b)-1)*NK + j)*MT + mt
        using BT = BTaps<KH, KW>;
        constexpr int NCH = BT::count();
        v4f acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f){0.f, 0.f, 0.f, 0.f};   // set in the prologue (step -1)
        // ZPRE: z' of the 4-row blocks' channels (operand layout: channel base + q in lane row q) joins behind the reduce
        float zqc[C::NSM > 0 ? C::NSM : 1], zqn[C::NSM > 0 ? C::NSM : 1];
#pragma unroll
        for (int sb = 0; sb < (C::NSM > 0 ? C::NSM : 1); ++sb) zqc[sb] = zqn[sb] = 0.f;

        auto phase_a = [&](auto pha_c, int j0, int j1) {
            constexpr int PHA = decltype(pha_c)::value;
#pragma unroll
            for (int j = j0; j < j1; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if constexpr (KW > 1)
                        mma(acc[mt], FT + ((0 * KW + 1 - 1) * NK + j) * MT + mt, ROT ? Q[0][(PHA + 3) & 3][j] : R[0][1][j]);
                    if constexpr (KH > 1)
                        mma(acc[mt], FT + ((1 * KW + 0 - 1) * NK + j) * MT + mt, ROT ? Q[1][(PHA + 3) & 3][j] : R[1][0][j]);
                }
        };

#ifdef FINC_STAMP
        unsigned stamp_acc[40];
#pragma unroll
        for (int i = 0; i < 40; ++i) stamp_acc[i] = 0u;
        unsigned long long stamp_prev = 0;
        bool stamp_on = false;
#endif
        auto step = [&](auto ph_c, auto wp_c) {
            constexpr int PH = decltype(ph_c)::value;   // == t & 3; wp_c: parity of the window (32-byte I/O)
            FINC_STAMP_AT(9);                           // segment 9: loop latch (between two steps)
            // the masks cost VALU issue that f32 MFMAs do not hide: apply them only on the steps where a lane
            // wraps (P of every W steps) / has not started yet (the first P steps); both tests are scalar
            const bool any_wrap = tm >= 0 && tm < P;
            const bool any_idle = tp1 < P - 1;
            float zraw[NKZ], zv[NKZ], xpk[NKD];
            float xown[NKD];                          // K-split: this wave's own share of the registers it finalises
            v4f accn[MT];
            if constexpr (HLP && PH == 0) {            // the landing of the previous window must be in the ring (peeked in step 3)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (__builtin_expect(__builtin_amdgcn_readfirstlane(peek_b) < win, 0)) { FINC_COLD(); FINC_HLP_LATE(0); flag_wait(IC<1>{}, win); }
            }
            if constexpr (HLP && PH == 1) flag_peek(IC<2>{}, peek_b2);     // (for the check before this step's x-ring write)

            // ---- RA1: z of the next position is requested; operands that do not depend on this step age
#pragma unroll
            for (int j = 0; j < NKZ; ++j) zraw[j] = FINC_ABLATE >= 3 ? af[j] : zrd[j * C::ZSLOTS * 64];
            if constexpr (!ROT) {
#pragma unroll
                for (int a = 0; a < KH; ++a) {
#pragma unroll
                    for (int b = KW - 1; b >= 1; --b) {
                        if (a + b >= 2) {
#pragma unroll
                            for (int j = 0; j < NK; ++j) R[a][b][j] = R[a][b - 1][j];
                        }
                    }
                    if (a >= 2) {
#pragma unroll
                        for (int j = 0; j < NK; ++j) R[a][0][j] = DL[a][a - 2][j];
                    }
                }
                if (__builtin_expect(any_wrap, 0)) {
                    FINC_COLD();
                    const bool wrapn = p == tm;
#pragma unroll
                    for (int a = 0; a < KH; ++a)
#pragma unroll
                        for (int b = 1; b < KW; ++b) {
                            if (a + b >= 2) {
#pragma unroll
                                for (int j = 0; j < NK; ++j) R[a][b][j] = wrapn ? 0.f : R[a][b][j];
                            }
                        }
                }
            }
            phase_a(ph_c, 0, NK / 2);
            // HLP: this step's z read is in the LDS queue -- the last one before the slots of this window's landing are
            // free -- so the helper may land (A1), a whole step before the next read
            if constexpr (HLP && PH == 3) flag_set(IC<3>{}, win + 1);
            FINC_SB();
            // ---- RA2
#pragma unroll
            for (int j = 0; j < NKZ; ++j) zv[j] = zraw[j];
            if (__builtin_expect(any_idle, 0)) {
                FINC_COLD();
                const bool started = p <= tp1;
#pragma unroll
                for (int j = 0; j < NKZ; ++j) zv[j] = started ? zv[j] : 0.f;
            }
            phase_a(ph_c, NK / 2, NK);
            FINC_STAMP_AT(0);                           // segment 0: phase A (+ z read, idle mask)
            if constexpr (ROT) {
                // the operands phase A just read unmasked (b = 0 / 1 of this step) become b >= 1 taps of a step that
                // starts a row: zero them, for the wrapping lanes only, after phase A has issued and before phase B
                if (__builtin_expect(any_wrap, 0)) {
                    FINC_COLD();
                    const bool wrapn = p == tm;
#pragma unroll
                    for (int a = 0; a < KH; ++a)
#pragma unroll
                        for (int b = 1; b < KW; ++b) {
                            if (a + b >= 2) {
#pragma unroll
                                for (int j = 0; j < NK; ++j)
                                    Q[a][(PH + 9 - a - b) & 3][j] = wrapn ? 0.f : Q[a][(PH + 9 - a - b) & 3][j];
                            }
                        }
                }
            }
            FINC_SB();
            // ---- RB0: z-term of the next step
            // The accumulators of the next pixel start from the bias: it is the C operand of the first z-term MFMA of
            // every tile (k-step 0 is never a structural zero), so the start costs no instruction.  Both branches define
            // accn by that MFMA -- a select on accn itself would make the common path copy the bias every step.
            if constexpr (ZPRE) {
                // no z-term: the ring holds the accumulators' start (idle lanes: zv was zeroed above, so they keep producing
                // exact zeros); the 4-row blocks start from zero and take their z' behind the reduce of the NEXT step
#pragma unroll
                for (int mt = 0; mt < C::MTB; ++mt) accn[mt] = (v4f){zv[4 * mt], zv[4 * mt + 1], zv[4 * mt + 2], zv[4 * mt + 3]};
#pragma unroll
                for (int sb = 0; sb < C::NSM; ++sb) {
                    accn[C::MTB + sb] = (v4f){0.f, 0.f, 0.f, 0.f};
                    zqn[sb] = zv[4 * C::MTB + sb];
                }
            } else if (__builtin_expect(any_idle, 0)) {
                // a lane that has not started must keep producing exact zeros: its "pixels" are what lane p+1 and, through
                // the FIFO, the first band's lanes 0..KH-2 read as the (non-existent) rows above the image
                FINC_COLD();
                const bool started = p <= tp1;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float b0 = bias[mt].x, b1 = bias[mt].y, b2 = bias[mt].z, b3 = bias[mt].w;
                    accn[mt] = (v4f){started ? b0 : 0.f, started ? b1 : 0.f, started ? b2 : 0.f, started ? b3 : 0.f};
                    mma(accn[mt], FZ + mt, zv[0]);
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    accn[mt] = bias[mt];
                    mma(accn[mt], FZ + mt, zv[0]);
                }
            }
#pragma unroll
            for (int j = 1; j < NKZ; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (ZPRE) continue;
                    if (NW == 1 && finc_zterm_is_zero(C::MTB, j, mt)) continue;   // Linv is lower triangular
                    mma(accn[mt], FZ + j * MT + mt, zv[j]);
                }
            FINC_STAMP_AT(1);                           // segment 1: z-term
            if constexpr (NW > 1) {
                // K-split exchange: acc holds this wave's share of ALL output registers.  Ship the registers other waves
                // own, keep ours; after the barrier post1 adds the NW-1 shares it received.
                float vv[C::NKDT];                     // partial sums in operand order (4-row blocks already reduced)
                pack_d<C>(acc, vv);
#pragma unroll
                for (int d = 0; d < C::NKDT; ++d) {
                    const int dst = d / NKD;
                    if (dst != wv) {
                        if constexpr (C::XALIAS)
                            (lds_p + dst * wave_lds + C::ZRING)[(d % NKD) * C::XSLOTS * 64 + PH * 64 + xwin] = vv[d];
                        else
                            xch[((dst * NW + wv) * NKD + d % NKD) * 64 + lane] = vv[d];
                    }
                    else xown[d % NKD] = vv[d];
                }
                FINC_SB();
                __syncthreads();
            }
            FINC_SB();

            auto post1 = [&]() {                       // the pixel solved this step leaves the accumulators
                if constexpr (FINC_ABLATE >= 2 && NW == 1) { // keep every accumulator alive, do nothing else
                    pack_d<C>(acc, xpk);
#pragma unroll
                    for (int j = 0; j < NKD; ++j) asm volatile("" ::"v"(xpk[j]));
                    return;
                }
                if constexpr (NW == 1) {
                    pack_d<C>(acc, xpk);
                    if constexpr (ZPRE) {
#pragma unroll
                        for (int sb = 0; sb < C::NSM; ++sb) xpk[4 * C::MTB + sb] += zqc[sb];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NKD; ++j) {
                        float sum = xown[j];
#pragma unroll
                        for (int src = 0; src < NW; ++src)
                            if (src != wv)
                                sum += C::XALIAS ? xring[j * C::XSLOTS * 64 + PH * 64 + xwin]
                                                 : xch[((wv * NW + src) * NKD + j) * 64 + lane];
                        xpk[j] = sum;
                    }
                }
#pragma unroll
                for (int j = 0; j < NKD; ++j) xring[j * C::XSLOTS * 64 + PH * 64 + xwin] = xpk[j]; // slot = t & 7
                if constexpr (KH > 1) {
                    fifo_push(xpk);
                    fifo_pop_all();
                }
                if constexpr (KW > 1) {                // S_0(t): tap (0,1) of the next step
                    if (__builtin_expect(any_wrap, 0)) {
                        FINC_COLD();
                        const bool wrapn = p == tm;
#pragma unroll
                        for (int j = 0; j < NK; ++j) (ROT ? Q[0][PH][j] : R[0][1][j]) = wrapn ? 0.f : xpk[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < NK; ++j) (ROT ? Q[0][PH][j] : R[0][1][j]) = xpk[j];
                    }
                }
            };
            auto post2 = [&]() { if constexpr (FINC_ABLATE < 2) shift_all(xpk, ph_c); };

            // ---- RB1..: one region per remaining tap, side work attached to the first three
            auto chunk = [&](auto ci_c) {
                constexpr int CI = decltype(ci_c)::value;
                constexpr int a = BT::a_of(CI), b = BT::b_of(CI);
#pragma unroll
                for (int j = 0; j < NK; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        mma(accn[mt], FT + ((a * KW + b - 1) * NK + j) * MT + mt, ROT ? Q[a][(PH + 9 - a - b) & 3][j] : R[a][b][j]);
                if constexpr (CI == 0) {
                    if constexpr (HLP && PH == 1) {    // the helper must have read what this write replaces (peeked in step 0)
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (__builtin_expect(__builtin_amdgcn_readfirstlane(peek_b2) < win + 1, 0)) { FINC_COLD(); FINC_HLP_LATE(1); flag_wait(IC<2>{}, win + 1); }
                    }
                    post1();
                    if constexpr (HLP && PH == 3) flag_set(IC<0>{}, win + 1);    // window done as far as the helper cares
                }
                if constexpr (!HLP && CI == (NCH > 1 ? 1 : 0)) io_phase(ph_c, wp_c);
                if constexpr (CI == (NCH > 2 ? 2 : NCH - 1)) {
                    post2();
                    if constexpr (HLP && PH == 3) flag_peek(IC<1>{}, peek_b);   // (for the check at the top of the next step)
                }
                FINC_SB();
                if constexpr (CI < 6) FINC_STAMP_AT(2 + CI); // segments 2.. : phase-B chunks (0: post1, 1: HBM I/O, 2: post2)
            };
            if constexpr (NCH == 0) {
                post1();
                io_phase(ph_c, wp_c);
                post2();
                FINC_SB();
            } else {
                [&]<int... I>(std::integer_sequence<int, I...>) { (chunk(IC<I>{}), ...); }
                (std::make_integer_sequence<int, NCH>{});
            }
            // ---- advance
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = accn[mt];
            if constexpr (ZPRE) {
#pragma unroll
                for (int sb = 0; sb < C::NSM; ++sb) zqc[sb] = zqn[sb];
            }
            if constexpr (NW > 1) __syncthreads();     // every wave has read its shares: the buffer may be rewritten
            ++tp1;
            ++tm; if (tm == W) tm = 0;
            ++nslot; zrd += zstep[PH];
            if (nslot == 12) { nslot = 0; zrd -= 12 * 64; }
            if constexpr (PH == 3) { xwin ^= 256; ++win; }
            fifo_advance();
            FINC_STAMP_AT(8);                           // segment 8: advance
        };

        // Window -4 (steps -4..-1) solves nothing: every lane is still before its first pixel.  Only its HBM side
        // and the z-term of lane 0's first pixel (phase B of step -1) matter, so it runs without the other 3.9 steps.
        if constexpr (HLP) {
        } else if constexpr (SEC) {                    // window -1 has parity 1
            sec_sread(IC<1>{});
            sec_swrite(IC<1>{}, 0, NKD);
            sec_event(IC<1>{});
            s64_event(IC<1>{}, IC<0>{});
        } else {
            io_sread();
            io_swrite(0, NKD);
            io_land();
            io_issue(0, NKZ, true);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {                  // steps -4, -3, -2: bookkeeping only (the FIFO ring is zero)
            ++tp1; ++tm;
            ++nslot; zrd += zstep[k];                  // (t = -4 + k: t & 3 == k)
            if (nslot == 12) { nslot = 0; zrd -= 12 * 64; }
            fifo_advance();
        }
        {                                              // step -1: acc = Linv * z of the position lanes with p == 0 start at
            const bool started = p <= tp1;             // tp1 == 0 here
            flag_wait(IC<1>{}, 0);                     // (HLP) the first landings
            peek_b = 0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float b0 = bias[mt].x, b1 = bias[mt].y, b2 = bias[mt].z, b3 = bias[mt].w;
                acc[mt] = (v4f){started ? b0 : 0.f, started ? b1 : 0.f, started ? b2 : 0.f, started ? b3 : 0.f};
                if constexpr (ZPRE) acc[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < NKZ; ++j) {
                const float v = zrd[j * C::ZSLOTS * 64];
                const float zvj = started ? v : 0.f;
                if constexpr (ZPRE) {
                    if (j < 4 * C::MTB) acc[j >> 2][j & 3] = zvj;
                    else zqc[j - 4 * C::MTB] = zvj;
                    continue;
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (NW == 1 && finc_zterm_is_zero(C::MTB, j, mt)) continue;
                    mma(acc[mt], FZ + j * MT + mt, zvj);
                }
            }
            ++tp1; ++tm;
            ++nslot; zrd += zstep[3];
            if (nslot == 12) { nslot = 0; zrd -= 12 * 64; }
            xwin ^= 256;
            fifo_advance();
        }
        // 32-byte I/O alternates two register sets with the window: two windows per iteration (Tend % 8 == 0)
        for (int t0 = 0; t0 < Tend; t0 += SEC ? 8 : 4) {
#ifdef FINC_STAMP
            stamp_on = t0 >= 64 && t0 < 192;
#endif
            step(IC<0>{}, IC<0>{});
            step(IC<1>{}, IC<0>{});
            step(IC<2>{}, IC<0>{});
            step(IC<3>{}, IC<0>{});
            if constexpr (SEC) {
                step(IC<0>{}, IC<1>{});
                step(IC<1>{}, IC<1>{});
                step(IC<2>{}, IC<1>{});
                step(IC<3>{}, IC<1>{});
            }
        }
        if constexpr (HLP) {
        } else if constexpr (SEC) {                    // window Tend/4 (parity 0): the last pieces leave
            sec_sread(IC<0>{});
            sec_swrite(IC<0>{}, 0, NKD);
        } else {
            io_sread();
            io_swrite(0, NKD);
        }
#ifdef FINC_STAMP
        if (blockIdx.x == 517 && threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < 40; ++i) finc_stamp_buf[i] = stamp_acc[i];
        }
#endif
    }
}

// -----------------------------------------------------------------------------------------------
// Fragment packing (fp64 math, one workgroup per group): Linv = L^-1 by forward substitution;
// z-term fragment = Linv * diag(scale); tap (a,b) fragment = -(Linv * Wc[:,:,KH-1-a,KW-1-b]); bias = Linv * shift
// (scale / shift: the affine map z = scale*y + shift folded in front of the inverse, per channel of the group;
// nullptr = identity).  After the nfrag filter fragments of a group come 4*MT bias registers in accumulator layout
// (16-row tile: lane (q,p), register r = row 16mt+4q+r; 4-row block: register i = row base+i, in lane row 0 only,
// because the block's 4 lane rows are summed) and 4*MT registers of zeros.
// Lane (q,i) of fragment (tap, j, mt) holds row 16mt+i for a 16-row tile (mt < MTB), row 16*MTB + 4(mt-MTB) + (i&3)
// for a 4-row block (the 4x4x1 A operand: lane 4*blk+i' = row i' of block blk, the same for all 4 pixel quads);
// column = channel of k-slot q of k-step j (z-term: 4j+q; taps: chan_d, the order in which the solved pixels leave
// the accumulators).
// -----------------------------------------------------------------------------------------------
__global__ void pack_kernel(const float *__restrict__ wc, const float *__restrict__ scale,
                            const float *__restrict__ shift, float *__restrict__ packed, int Cq, int KH, int KW, int MT,
                            int NKZ, int NKD, int MTB, int nfrag)
{
    extern __shared__ __attribute__((aligned(16))) double sm[]; // Linv [Cq][Cq]
    const int g = blockIdx.x;
    const float *wg = wc + (size_t)g * Cq * Cq * KH * KW;
    const int KK = KH * KW;
    double *Linv = sm;
    const int npack = nfrag + 8 * MT;
    // column j of Linv: solve L y = e_j  (L unit lower triangular)
    for (int j = threadIdx.x; j < Cq; j += blockDim.x) {
        for (int r = 0; r < Cq; ++r) {
            double s = (r == j) ? 1.0 : 0.0;
            for (int k = j; k < r; ++k) s -= (double)wg[((size_t)r * Cq + k) * KK + (KK - 1)] * Linv[k * Cq + j];
            Linv[r * Cq + j] = (r < j) ? 0.0 : s;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nfrag * 64; e += blockDim.x) {
        const int lane = e & 63, f = e >> 6;
        const int q = lane >> 4, i = lane & 15;
        int tap, j, mt;
        const bool zterm = f < NKZ * MT;
        if (zterm) {
            mt = f % MT; j = f / MT; tap = 0;
        } else {
            const int ff = f - NKZ * MT;
            mt = ff % MT; j = (ff / MT) % NKD; tap = 1 + ff / (MT * NKD);
        }
        const int row = finc_tile_row(MTB, mt, i);
        const int col = zterm ? 4 * j + q : chan_d(MTB, j, q);
        double v = 0.0;
        if (row < Cq && col < Cq) {
            if (zterm) {
                v = Linv[row * Cq + col] * (scale ? (double)scale[g * Cq + col] : 1.0);
            } else {
                const int a = tap / KW, b = tap % KW;
                const int widx = (KH - 1 - a) * KW + (KW - 1 - b);
                double s = 0.0;
                for (int k = 0; k <= row; ++k) s += Linv[row * Cq + k] * (double)wg[((size_t)k * Cq + col) * KK + widx];
                v = -s;
            }
        }
        packed[((size_t)g * npack + f) * 64 + lane] = (float)v;
    }
    for (int e = threadIdx.x; e < 8 * MT * 64; e += blockDim.x) {
        const int lane = e & 63, f = e >> 6;           // f < 4*MT: bias register r of tile mt; then the zero block
        const int q = lane >> 4, mt = (f % (4 * MT)) >> 2, r = f & 3;
        const int row = mt < MTB ? 16 * mt + 4 * q + r : (q == 0 ? 16 * MTB + 4 * (mt - MTB) + r : Cq);
        double v = 0.0;
        if (f < 4 * MT && shift && row < Cq)
            for (int k = 0; k <= row; ++k) v += Linv[row * Cq + k] * (double)shift[g * Cq + k];
        packed[((size_t)g * npack + nfrag + f) * 64 + lane] = (float)v;
    }
}

// -----------------------------------------------------------------------------------------------
// Instantiation table
// -----------------------------------------------------------------------------------------------
typedef void (*wave_fn)(const float *, const float *, float *, int, int, int, int, int, int, unsigned);

struct Inst {
    int cqp, kh, kw;
    wave_fn fn;      // 16-byte-group I/O (any W % 4 == 0)
    wave_fn fn_sec;  // 32-byte-piece I/O (W % 8 == 0)
    wave_fn fn_s64;  // 64-byte sector pairing (W % 16 == 0); nullptr where the variant does not exist
    wave_fn fn_hlp;  // sector pairing + helper waves: 8-wave workgroups of 4 problems (problem count % 4 == 0); or nullptr
    wave_fn fn_zpre; // fn_hlp for a premultiplied input (ZPRE); or nullptr
    int nkz, nkd, nk, mt, nfrag, mtb;  // nkz/nkd/nfrag: per GROUP (packing); nk: per wave; mt = mtb tiles + 4-row blocks
    int nw, wnkz, wnkd;                // K-split: waves per problem, per-wave k-steps
    int npw;                           // problems packed into one workgroup (the launch needs B*G % npw == 0)
    int max_problems;                  // > 0: a small-batch variant, used only while B*G <= max_problems
};

// The packed bank depends on (CQP, KH, KW) only, so variants of one shape (different NW / NPW) share it.
template <int CQP, int KH, int KW, int NW = 1, int NPW = 1, int MAXP = 0>
constexpr Inst make_inst()
{
    using C = Cfg<CQP, KH, KW, NW>;
    // sector pairing parks 8 registers per k-step of z and of x in AGPRs, next to the pinned fragments
    constexpr int zskip = [] {               // (16-row-tile fragments of the triangular z-term that are never loaded)
        int n = 0;
        for (int j = 0; j < C::NKZ; ++j)
            for (int mt = 0; mt < C::MTB; ++mt) n += finc_zterm_is_zero(C::MTB, j, mt) ? 1 : 0;
        return n;
    }();
    // AGPR budget next to the pinned fragments: loads paired park 16 registers per k-step of z
    constexpr bool one = NW == 1 && NPW == 1;
    constexpr int pinned = (C::NFRAG / C::MT) * C::MTB - zskip + ((C::NFRAG / C::MT) * C::NSM + 3) / 4;
    constexpr int mode = !one ? 0 : (pinned + 8 * C::NKZ + 8 * C::NKD <= 256) ? 3 : (pinned + 8 * C::NKD <= 256) ? 1 : 0;
    wave_fn f64 = nullptr, fhl = nullptr, fzp = nullptr;
    if constexpr (mode != 0) f64 = finc_wave_kernel<CQP, KH, KW, true, NW, NPW, mode>;
#ifndef FINC_HLP_MODE
#define FINC_HLP_MODE 1
#endif
    // helper waves: both roles must fit hipcc's 128 : 128 split of a 256-register wave WITHOUT spilling (an in-flight load
    // destination that is spilled is saved before its data has arrived): helper AGPRs = lower + upper pieces + one parked
    // set, helper VGPRs = two store sets + the other parked set + bookkeeping, compute AGPRs = the pinned fragments
    // (filters whose operands do not rotate in place -- 5x5 -- keep explicit ageing copies: too many VGPRs in the compute role)
    constexpr bool hlp_fits = 16 * C::NKZ + 4 * C::NKD <= 124 && 12 * C::NKD + 48 <= 124 && pinned <= 124 && (KH - 1 + KW) < 6;
    if constexpr (mode == 3 && FINC_HLP_MODE && hlp_fits) {
        fhl = finc_wave_kernel<CQP, KH, KW, true, NW, NPW, 3, 1>;
        fzp = finc_wave_kernel<CQP, KH, KW, true, NW, NPW, 3, 1, true>;
    }
    return Inst{CQP, KH, KW, finc_wave_kernel<CQP, KH, KW, false, NW, NPW>, finc_wave_kernel<CQP, KH, KW, true, NW, NPW>, f64, fhl, fzp,
                C::NKZT, C::NKDT, C::NK, C::MT, C::NFRAGT, C::MTB, NW, C::NKZ, C::NKD, NPW, MAXP};
}

#define FINC_BOTH(cqp, kh, kw) make_inst<cqp, kh, kw>()

#ifdef FINC_ONLY_C3   // experiment builds (scripts/build_variant.sh): only the c3 kernels, compiles in seconds
const Inst g_insts[] = {make_inst<24, 3, 3, 2, 2, 512>(), FINC_BOTH(24, 3, 3), FINC_BOTH(12, 3, 3)};
#else
const Inst g_insts[] = {
    // 3x3: every Cq % 4 == 0 up to 32, then K-split (2 / 4 waves per problem) for the banks one wave cannot hold.
    // Variants of one shape are tried in table order.  <24,3,3> first lists its small-batch variant: while the problems
    // number no more than the SIMD pairs of the chip (B*G <= 512), splitting each over 2 waves is 27 % faster
    // (289 vs 395 us at B <= 128); 2-wave problems are packed in pairs (NPW = 2) so that all four SIMDs of a CU get work.
    // (While the problems do not outnumber the CUs -- B*G <= 256 -- the 2x2 and 3x3 banks up to Cq = 32 do not come here at
    // all: they run on the role-split kernel, finc_split.hip.  Its predecessors in this table, three-wave K-splits of
    // <24,3,3> and <12,3,3>, took 245 / 48 us at c3 / c2 where it takes 165 / 2x us: profiles/r03.)
    FINC_BOTH(4, 3, 3),  FINC_BOTH(8, 3, 3),  FINC_BOTH(12, 3, 3), FINC_BOTH(16, 3, 3),
    FINC_BOTH(20, 3, 3),
    make_inst<24, 3, 3, 2, 2, 512>(), FINC_BOTH(24, 3, 3), FINC_BOTH(28, 3, 3),
    // Cq = 32: one wave's rings (53 KB at W = 64) let only 2 problems onto a CU; split over 2 waves and packed in pairs
    // the same 2 problems keep all 4 SIMDs busy (1.25 -> 0.89 ms at B = 256, 64x64)
    make_inst<32, 3, 3, 2, 2>(), FINC_BOTH(32, 3, 3),
    make_inst<40, 3, 3, 2, 2>(), make_inst<40, 3, 3, 2>(), make_inst<48, 3, 3, 4>(), make_inst<64, 3, 3, 4>(),
    FINC_BOTH(4, 2, 2),  FINC_BOTH(8, 2, 2),  FINC_BOTH(12, 2, 2), FINC_BOTH(16, 2, 2), FINC_BOTH(24, 2, 2),
    // (Cq = 32 at 2x2: as at 3x3, one wave's rings let 3 problems onto a CU; 2 waves per problem, packed in pairs: 685 -> 608 us
    // at B = 256, 64x64)
    make_inst<32, 2, 2, 2, 2>(), FINC_BOTH(32, 2, 2),
    FINC_BOTH(4, 5, 5),  FINC_BOTH(8, 5, 5),  FINC_BOTH(12, 5, 5), FINC_BOTH(16, 5, 5),
    // (Cq = 17 .. 24 at 5x5 -- the 20-channel 5x5 layers of fastflow/test_examples.py:218-222 -- on two waves: the operands
    // of a 5x5 filter do not rotate in place, and their ageing copies beside 450 fragments do not fit one wave)
    make_inst<24, 5, 5, 2>(), make_inst<32, 5, 5, 4>(), make_inst<48, 5, 5, 4>(),
    // non-square filters (PaddedConv2d takes a (K_H, K_W) tuple, layers/conv.py:30-36; the reference's fixtures have 3x5 and 2x3)
    FINC_BOTH(4, 3, 5),  FINC_BOTH(8, 3, 5),  FINC_BOTH(16, 3, 5),
    FINC_BOTH(4, 2, 3),  FINC_BOTH(8, 2, 3),  FINC_BOTH(16, 2, 3),
};
#endif

size_t lds_bytes(const Inst &i, int W, int P);

// FINC_NO_HLP=1 keeps the helper-wave variant off (A/B timing, tests of the single-wave sector-pairing path)
bool finc_no_hlp()
{
    static const bool off = [] { const char *e = finc_env("FINC_NO_HLP"); return e && e[0] == '1'; }();
    return off;
}

// FINC_NO_S64=1 in the environment keeps W % 16 == 0 shapes on the 32-byte-piece kernel (A/B timing, tests of that path)
bool finc_no_s64()
{
    static const bool off = [] { const char *e = finc_env("FINC_NO_S64"); return e && e[0] == '1'; }();
    return off;
}

// first variant of the shape (any: they share the packed layout); with a problem count and a width, the first variant
// that may run them
// Cq padded to the smallest compiled bank that holds it (0: none).  One-wave kernels mask the last group of four only -- the
// table has every multiple of 4 up to their largest bank --, the K-split banks any number of padded channels.
int padded_cq(int Cq, int KH, int KW)
{
    int best = 0, nw = 1;
    for (const Inst &i : g_insts)
        if (i.cqp >= Cq && i.kh == KH && i.kw == KW && (best == 0 || i.cqp < best)) { best = i.cqp; nw = i.nw; }
    if (best && best - Cq > 3) {               // needs a K-split bank: every variant of that bank must be one
        for (const Inst &i : g_insts)
            if (i.cqp == best && i.kh == KH && i.kw == KW && i.nw == 1) return 0;
    }
    (void)nw;
    return best;
}

// A bank without a two-wave form of its own borrows the next bank's: the 28-channel 3x3 bank (7 k-steps do not split over two
// waves) the packed two-wave kernel of the 32-channel one, whose K-split takes any number of padded channels.  The borrowed bank is
// packed BEHIND the bank's own (finc_mfma_pack), so which kernel runs stays a per-launch choice.
int borrowed_cqp(int cqp, int KH, int KW) { return cqp == 28 && KH == 3 && KW == 3 ? 32 : 0; }

// One-wave problems run n1 to a compute unit (four SIMDs; fewer once their rings outgrow a quarter of the LDS), borrowed two-wave
// problems two (a packed pair = four waves), and the chip works through a problem set in rounds of that many per unit.  Measured at
// C = 112 against C = 128 (profiles/r05/notes/c28_borrowed_bank.txt): a round of one-wave 28-channel problems takes 540 us at 64x64
// (170 at 32x32), a round of two-wave problems 437 (138) -- 0.81 of it, whatever the map.  Up to 512 problems the borrowed form is one
// round against one: -19 %.  Beyond, with three or four one-wave problems to a unit, the bank's own kernel has the better rate (0.70
// against 0.85 us per problem) and hands a remainder to the role-split kernel (finc_mfma_launch: 1,024 problems = 768 + 256 in
// 540 + 205 us, where two borrowed rounds take 876); with two or fewer to a unit (maps of about 100 columns and more) the borrowed
// form's rounds are never the longer sum.
static bool borrowed_form_wins(long long problems, size_t lds_one_wave)
{
    const long long CUS = 256;                                  // MI355X
    long long n1 = (long long)((160 * 1024 - 64) / lds_one_wave);
    n1 = n1 > 4 ? 4 : n1 < 1 ? 1 : n1;
    if (n1 >= 3 && problems > 2 * CUS) return false;
    const long long r1 = (problems + n1 * CUS - 1) / (n1 * CUS), r2 = (problems + 2 * CUS - 1) / (2 * CUS);
    return r2 * 13 < r1 * 16;                                   // (437 / 540 = 0.81 = 13 / 16)
}

const Inst *find_inst(int Cq, int KH, int KW, long long problems = -1, int W = 0)
{
    const int cqp = padded_cq(Cq, KH, KW);
    if (cqp == 0) return nullptr;
    for (const Inst &i : g_insts) {
        if (i.cqp != cqp || i.kh != KH || i.kw != KW) continue;
        if (problems >= 0) {
            if (i.max_problems > 0 && problems > i.max_problems) continue;
            if (problems % i.npw != 0) continue;
            const int P = W < 16 ? W : 16;
            if (lds_bytes(i, W, P) > 160 * 1024) continue;
        }
        if (problems >= 0 && W > 0 && i.nw == 1 && i.npw == 1 && problems % 2 == 0 && borrowed_cqp(cqp, KH, KW)) {
            const int P = W < 16 ? W : 16;
            if (borrowed_form_wins(problems, lds_bytes(i, W, P)))
                for (const Inst &k : g_insts)
                    if (k.cqp == borrowed_cqp(cqp, KH, KW) && k.kh == KH && k.kw == KW && k.nw == 2 && k.npw == 2 && lds_bytes(k, W, P) <= 160 * 1024)
                        return &k;
        }
        // Wide maps: the band hand-over FIFO grows with W, and once four one-wave problems no longer fit a CU's LDS (W >= 80
        // at Cq = 24) the helper-wave form is out and only three SIMDs of a CU have a problem.  The packed two-wave form
        // -- two problems per workgroup, every SIMD busy -- is then the faster one at ANY problem count (B = 256:
        // 64x80 1,033 -> 764 us, 64x96 1,225 -> 903, 128x128 3,163 -> 2,302; profiles/r02/notes/ab36), although it loses
        // at 64x64 where four one-wave problems do fit (521 vs 413 us).
        if (problems >= 0 && W > 0 && i.nw == 1 && i.npw == 1) {
            const int P = W < 16 ? W : 16;
            if (4 * lds_bytes(i, W, P) + 64 > 160 * 1024 && problems % 2 == 0) {
                for (const Inst &k : g_insts)
                    if (k.cqp == cqp && k.kh == KH && k.kw == KW && k.nw == 2 && k.npw == 2 && lds_bytes(k, W, P) <= 160 * 1024)
                        return &k;
            }
        }
        return &i;
    }
    return nullptr;
}

size_t lds_bytes(const Inst &i, int W, int P)
{
    const size_t D = (size_t)(W - P + 1);
    const size_t ss = (size_t)i.nk * 4 * (i.kh - 1);
    const size_t per_wave = (size_t)i.wnkz * 12 * 64 + (size_t)i.wnkd * 8 * 64 + D * ss + ss + 64;
    const size_t xch = i.nw > 2 ? (size_t)i.nw * i.nw * i.wnkd * 64 : 0;   // 2 waves exchange through their x rings
    return sizeof(float) * (size_t)i.npw * (xch + (size_t)i.nw * per_wave);
}

} // namespace

#ifdef FINC_STAMP
extern "C" int finc_debug_stamps(unsigned long long *host_out, int n)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(finc_stamp_buf), sizeof(unsigned long long) * (size_t)n);
}
#endif

int finc_mfma_arm_fault_word(unsigned *device_ptr_to_host_word)
{
    FINC_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(finc_fault_word), &device_ptr_to_host_word, sizeof(device_ptr_to_host_word)));
    return FINC_OK;
}

int finc_mfma_hlp_timeouts(unsigned *count)
{
    FINC_HIP_TRY(hipMemcpyFromSymbol(count, HIP_SYMBOL(finc_hlp_timeouts), sizeof(unsigned)));
    return FINC_OK;
}

// banks that neither this table nor finc_big.hip holds in registers: the streaming-bank kernel (finc_stream.hip)
static bool stream_bank(int Cq, int KH, int KW)
{
    return !finc_big_bank(Cq, KH, KW) && padded_cq(Cq, KH, KW) == 0 && finc_stream_bank_ok(Cq, KH, KW);
}

bool finc_mfma_supported(int Cq, int H, int W, int KH, int KW)
{
    if (finc_big_bank(Cq, KH, KW)) return finc_big_supported(Cq, H, W, KH, KW);   // beyond this table: finc_big.hip
    if (stream_bank(Cq, KH, KW)) return finc_stream_supported(Cq, H, W, KH, KW, true);
    if (W % 4 != 0 || W < 4 || H < 1) return false;
    const int P = W < 16 ? W : 16;
    if (P < KH - 1) return false;
    // the shape is supported if a variant exists that takes ANY problem count (a huge odd one rules out the packed and
    // the small-batch variants) and whose rings fit the LDS at this width -- or, on maps too wide for that, finc_big.hip does
    if (!find_inst(Cq, KH, KW, (1LL << 40) + 1, W)) return finc_big_wide_bank(Cq, KH, KW) && finc_big_supported(Cq, H, W, KH, KW);
    if ((size_t)Cq * H * W * 4 >= ((size_t)1 << 30)) return false; // buffer-offset range marks (OFF_BAD_CHANNEL)
    return true;
}

int finc_mfma_packed_cqp(int Cq, int KH, int KW)
{
    if (finc_big_bank(Cq, KH, KW)) { int w, l, c; return finc_big_info(FincShape{1, 1, Cq, 16, 16, KH, KW, 0}, &w, &l, &c) ? 0 : c; }
    if (stream_bank(Cq, KH, KW)) { int c = 0; return finc_stream_info(FincShape{1, 1, Cq, 16, 16, KH, KW, 0}, true, &c, nullptr, nullptr) ? 0 : c; }
    const Inst *a = find_inst(Cq, KH, KW);
    return a ? a->cqp : 0;
}

// bytes of this table's own bank
static size_t own_bank_bytes(int G, int Cq, int KH, int KW)
{
    const Inst *a = find_inst(Cq, KH, KW);
    return a ? (size_t)(a->nfrag + 8 * a->mt) * 64 * sizeof(float) * (size_t)G : 0;
}
// the borrowed bank's row of the table (nullptr: the bank borrows none) and its bytes
static const Inst *borrowed_inst(int Cq, int KH, int KW)
{
    const int b = borrowed_cqp(padded_cq(Cq, KH, KW), KH, KW);
    if (b)
        for (const Inst &k : g_insts)
            if (k.cqp == b && k.kh == KH && k.kw == KW && k.nw == 2 && k.npw == 2) return &k;
    return nullptr;
}
static size_t borrowed_bank_bytes(int G, int Cq, int KH, int KW)
{
    const Inst *k = borrowed_inst(Cq, KH, KW);
    return k ? (size_t)(k->nfrag + 8 * k->mt) * 64 * sizeof(float) * (size_t)G : 0;
}
// bytes of this table's banks: its own, then the borrowed one (behind them: the bank of finc_big.hip for the banks it takes over on
// wide maps)
static size_t wave_bank_bytes(int G, int Cq, int KH, int KW)
{
    return own_bank_bytes(G, Cq, KH, KW) + borrowed_bank_bytes(G, Cq, KH, KW);
}
// (which problem sets of such a bank go to finc_big.hip: those no variant of this table can hold)
static bool wide_takeover(const FincShape &s)
{
    return finc_big_wide_bank(s.Cq, s.KH, s.KW) && !find_inst(s.Cq, s.KH, s.KW, (1LL << 40) + 1, s.W) &&
           finc_big_supported(s.Cq, s.H, s.W, s.KH, s.KW);
}

// Banks packed WITH a folded shift have no wide-map form (finc_big.hip carries a scale only).  Such a packed buffer is
// remembered here (device, address) and its big-bank region is filled with NaNs, so a wide-map launch on it is refused
// (FINC_ERR_UNSUPPORTED) -- and could not pass for a result even if the table were bypassed.  Re-packing the address
// without a shift takes it out again.
static std::mutex g_dead_mutex;
static std::vector<std::pair<int, const void *>> g_dead_wide;
static void dead_wide_set(const void *packed, bool dead)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_dead_mutex);
    for (size_t i = 0; i < g_dead_wide.size(); ++i)
        if (g_dead_wide[i].first == dev && g_dead_wide[i].second == packed) {
            if (!dead) { g_dead_wide[i] = g_dead_wide.back(); g_dead_wide.pop_back(); }
            return;
        }
    if (dead) g_dead_wide.emplace_back(dev, packed);
}
static bool dead_wide_has(const void *packed)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(g_dead_mutex);
    for (const auto &e : g_dead_wide)
        if (e.first == dev && e.second == packed) return true;
    return false;
}

bool finc_mfma_affine_takes(const FincShape &s)
{
    if (!finc_mfma_supported(s.Cq, s.H, s.W, s.KH, s.KW)) return false;
    if (stream_bank(s.Cq, s.KH, s.KW)) return true;       // (scale into the z-term's columns, Linv * shift as the accumulators' start)
    return !finc_big_bank(s.Cq, s.KH, s.KW) && !wide_takeover(s);
}

size_t finc_mfma_packed_bytes(int G, int Cq, int KH, int KW)
{
    if (finc_big_bank(Cq, KH, KW)) return finc_big_packed_bytes(G, Cq, KH, KW);
    if (stream_bank(Cq, KH, KW)) return finc_stream_packed_bytes(G, Cq, KH, KW, true);
    const size_t own = wave_bank_bytes(G, Cq, KH, KW);
    return own && finc_big_wide_bank(Cq, KH, KW) ? own + finc_big_packed_bytes(G, Cq, KH, KW) : own;
}

int finc_mfma_pack(const float *wc, const float *scale, const float *shift, void *packed, int G, int Cq, int KH, int KW,
                   hipStream_t st)
{
    (void)finc_fault_gate(true, st);           // arm the device's fault word here, outside any capture of the launches
    (void)finc_split_prepare(st);              // ... and the band split's progress words (finc_split.hip)
    if (finc_big_bank(Cq, KH, KW)) return finc_big_pack(wc, scale, shift, packed, G, Cq, KH, KW, st);
    if (stream_bank(Cq, KH, KW)) return finc_stream_pack(wc, scale, shift, packed, G, Cq, KH, KW, true, false, st);
    const Inst *i = find_inst(Cq, KH, KW);
    if (!i) return FINC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pack_kernel, dim3(G), dim3(256), sizeof(double) * Cq * Cq, st, wc, scale, shift, (float *)packed, Cq,
                       KH, KW, i->mt, i->nkz, i->nkd, i->mtb, i->nfrag);
    FINC_CHECK_LAUNCH();
    if (const Inst *k = borrowed_inst(Cq, KH, KW)) {      // the borrowed two-wave bank, same fold, behind the bank's own
        float *behind = (float *)((char *)packed + own_bank_bytes(G, Cq, KH, KW));
        hipLaunchKernelGGL(pack_kernel, dim3(G), dim3(256), sizeof(double) * Cq * Cq, st, wc, scale, shift, behind, Cq, KH, KW, k->mt,
                           k->nkz, k->nkd, k->mtb, k->nfrag);
        FINC_CHECK_LAUNCH();
    }
    // (a folded shift is the one thing finc_big.hip does not carry: such a bank has no wide-map form -- the launch refuses
    // it, finc_inverse_affine_supported() says so beforehand, and the region holds NaNs, never a plausible bank)
    if (finc_big_wide_bank(Cq, KH, KW)) {
        char *behind = (char *)packed + wave_bank_bytes(G, Cq, KH, KW);
        dead_wide_set(packed, shift != nullptr);
        if (shift) FINC_HIP_TRY(hipMemsetAsync(behind, 0xFF, finc_big_packed_bytes(G, Cq, KH, KW), st));
        else return finc_big_pack(wc, scale, nullptr, behind, G, Cq, KH, KW, st);
    }
    return FINC_OK;
}

int finc_mfma_variant(int B, int G, int Cq, int H, int W, int KH, int KW, int *info)
{
    if (!finc_mfma_supported(Cq, H, W, KH, KW)) return FINC_ERR_UNSUPPORTED;
    if (stream_bank(Cq, KH, KW)) {             // form 7: the streaming-bank kernel, one workgroup of info[1] waves per problem
        int cqp = 0, lds = 0, waves = 0, owv = 0;
        if (int e = finc_stream_info(FincShape{B, G, Cq, H, W, KH, KW, 0}, true, &cqp, &lds, nullptr, &waves, &owv)) return e;
        info[0] = cqp; info[1] = waves; info[2] = 1; info[3] = 7; info[4] = lds; info[5] = B * G;
        info[6] = owv ? -4 : -3;               // (-4: one-wave problems in their per-lane 16-byte form) info[7] = (int)(sizeof(g_insts) / sizeof(g_insts[0]));
        return FINC_OK;
    }
    if (finc_big_bank(Cq, KH, KW) || wide_takeover(FincShape{B, G, Cq, H, W, KH, KW, 0})) {   // form 5: the big-bank kernel, one workgroup of info[1] waves per problem
        int waves = 0, lds = 0, cqp = 0;
        if (int e = finc_big_info(FincShape{B, G, Cq, H, W, KH, KW, 0}, &waves, &lds, &cqp)) return e;
        info[0] = cqp; info[1] = waves; info[2] = 1; info[3] = 5; info[4] = lds; info[5] = B * G;
        info[6] = -2; info[7] = (int)(sizeof(g_insts) / sizeof(g_insts[0]));
        return FINC_OK;
    }
    const Inst *i = find_inst(Cq, KH, KW, (long long)B * G, W);
    if (!i) return FINC_ERR_UNSUPPORTED;
    const int P = W < 16 ? W : 16;
    const FincShape fs{B, G, Cq, H, W, KH, KW, 0};
    if (finc_split_takes(fs)) {                // form 4: the role-split kernel, info[5] / (B*G) workgroups of info[1] waves per problem
        int waves = 0, lds = 0, steps = 0, nwg = 1;
        if (int e = finc_split_info(fs, &waves, &lds, &steps, &nwg)) return e;
        info[0] = i->cqp; info[1] = waves; info[2] = 1; info[3] = finc_split_uses_chain(fs) ? 6 : 4; info[4] = lds; info[5] = B * G * nwg;   // (6: its short-step form for the small banks, finc_chain.hip)
        info[6] = -1; info[7] = (int)(sizeof(g_insts) / sizeof(g_insts[0]));
        return FINC_OK;
    }
    info[0] = i->cqp;
    info[1] = i->nw;
    info[2] = i->npw;
    const bool s64 = W % 16 == 0 && i->fn_s64 && !finc_no_s64();
    const bool hlp = s64 && i->fn_hlp && ((long long)B * G) % 4 == 0 && 4 * lds_bytes(*i, W, P) + 64 <= 160 * 1024 && !finc_no_hlp();
    info[3] = hlp ? 3 : s64 ? 2 : W % 8 == 0 ? 1 : 0;
    info[4] = hlp ? (int)(4 * lds_bytes(*i, W, P) + 64) : (int)lds_bytes(*i, W, P);
    info[5] = hlp ? B * G / 4 : B * G / i->npw;
    info[6] = (int)(i - g_insts);
    info[7] = (int)(sizeof(g_insts) / sizeof(g_insts[0]));
    return FINC_OK;
}

int finc_mfma_table_row(int row, int *info)
{
    const int rows = (int)(sizeof(g_insts) / sizeof(g_insts[0]));
    if (row < 0 || row >= rows) return FINC_ERR_BAD_DIMS;
    const Inst &i = g_insts[row];
    info[0] = i.cqp; info[1] = i.kh; info[2] = i.kw; info[3] = i.nw; info[4] = i.npw; info[5] = i.max_problems;
    return FINC_OK;
}

// A problem set that is not a whole number of ROUNDS.  One-wave problems run n1 to a compute unit and every problem is the same chain
// of steps, so 1,025 problems take as long as 2,048 -- unless the remainder goes to the kernel the library would pick for it on its
// own (role-split / short-step kernel up to 256 / 512 problems, the two-wave variants), which is faster than a round of this one or
// it would not be picked.  The images are independent: the remainder is a second launch on the images behind the whole rounds
// (c3, 64x64: B = 320 = 1,024 + 256 problems 765 -> 591 us, B = 264 761 -> 522; profiles/r05/notes/remainder_launch.txt).
// The packed two-wave kernels -- a pair of problems on a unit's four SIMDs -- work in rounds of 512 the same way (32 channels,
// B = 160: 865 -> 642 us).  Returns the number of images of the remainder launch (0: one launch).
static int remainder_images(const FincShape &s, const Inst *i, bool hlp, size_t lds)
{
    static const bool no_remainder = finc_env("FINC_NO_REMAINDER_LAUNCH") != nullptr;   // experiment switch (A/B timing)
    const bool one_wave = i->nw == 1 && i->npw == 1, packed_pair = i->nw == 2 && i->npw == 2;
    if (no_remainder || !(one_wave || packed_pair)) return 0;
    const long long problems = (long long)s.B * s.G;
    long long n1 = packed_pair ? 2 : hlp ? 4 : (long long)((160 * 1024 - 64) / lds);
    n1 = n1 > 4 ? 4 : n1 < 1 ? 1 : n1;
    const long long round = n1 * 256, r = problems % round;
    if (problems <= round || r == 0 || r > 512 || r % s.G != 0) return 0;
    FincShape tail = s;
    tail.B = (int)(r / s.G);
    return (finc_split_takes(tail) || find_inst(s.Cq, s.KH, s.KW, r, s.W) != i) ? tail.B : 0;
}

// the premultiplied-input form exists for the shapes the helper-wave kernel takes (a full chip: the role-split kernel's
// small problem sets and the forms without helper waves keep their z-term) -- in ONE launch: a problem set with a remainder launch
// (above) keeps the plain chain, whose remainder runs on kernels without that form (c3, B = 264: 522 us against 0.93 x 761)
bool finc_mfma_zpre_takes(const FincShape &s)
{
    if (!finc_mfma_supported(s.Cq, s.H, s.W, s.KH, s.KW) || stream_bank(s.Cq, s.KH, s.KW) || finc_split_takes(s)) return false;
    const Inst *i = find_inst(s.Cq, s.KH, s.KW, (long long)s.B * s.G, s.W);
    if (!i || !i->fn_zpre || s.W % 16 != 0 || finc_no_s64() || finc_no_hlp()) return false;
    const int P = 16;
    if (!(((long long)s.B * s.G) % 4 == 0 && 4 * lds_bytes(*i, s.W, P) + 64 <= 160 * 1024)) return false;
    return remainder_images(s, i, true, lds_bytes(*i, s.W, P)) == 0;
}

// images of the remainder launch of an inverse call (0: the call is one launch) -- the launch's own decision, for tests and bench.py
int finc_mfma_remainder_images(const FincShape &s)
{
    if (!finc_mfma_supported(s.Cq, s.H, s.W, s.KH, s.KW) || finc_big_bank(s.Cq, s.KH, s.KW) || stream_bank(s.Cq, s.KH, s.KW) || wide_takeover(s) ||
        finc_split_takes(s))
        return 0;
    const Inst *i = find_inst(s.Cq, s.KH, s.KW, (long long)s.B * s.G, s.W);
    if (!i) return 0;
    const int P = s.W < 16 ? s.W : 16;
    const size_t lds = lds_bytes(*i, s.W, P);
    const bool s64 = s.W % 16 == 0 && i->fn_s64 && !finc_no_s64();
    const bool hlp = s64 && i->fn_hlp && ((long long)s.B * s.G) % 4 == 0 && 4 * lds + 64 <= 160 * 1024 && !finc_no_hlp();
    return remainder_images(s, i, hlp, lds);
}

int finc_mfma_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st, bool zpre)
{
    if (!finc_mfma_supported(s.Cq, s.H, s.W, s.KH, s.KW)) return FINC_ERR_UNSUPPORTED;
    if (zpre && !finc_mfma_zpre_takes(s)) return FINC_ERR_UNSUPPORTED;
    if (int e = finc_fault_gate(false)) return e;          // an earlier launch on this device gave up a protocol wait
    if (finc_big_bank(s.Cq, s.KH, s.KW)) return finc_big_launch(in, packed, out, s, st);   // beyond this table (finc_big.hip)
    if (stream_bank(s.Cq, s.KH, s.KW)) return finc_stream_launch(in, packed, out, s, true, st);   // beyond both (finc_stream.hip)
    if (wide_takeover(s) && dead_wide_has(packed)) return FINC_ERR_UNSUPPORTED;            // packed with a folded shift: no wide-map form
    if (wide_takeover(s))                                                                  // too wide for this table's forms
        return finc_big_launch(in, (const char *)packed + wave_bank_bytes(s.G, s.Cq, s.KH, s.KW), out, s, st);
    if (finc_split_takes(s)) return finc_split_launch(in, packed, out, s, st);   // the under-filled chip (finc_split.hip)
    const Inst *i = find_inst(s.Cq, s.KH, s.KW, (long long)s.B * s.G, s.W);
    if (!i) return FINC_ERR_UNSUPPORTED;
    if (((uintptr_t)in & 15u) || ((uintptr_t)out & 15u)) return FINC_ERR_ALIGNMENT;
    const int P = s.W < 16 ? s.W : 16;
    const int NB = (s.H + P - 1) / P;
    const int Tend = s.W % 8 == 0 ? (NB * s.W + P - 1 + 7) / 8 * 8 : (NB * s.W + P - 1 + 3) / 4 * 4;  // 32-byte I/O: x8 loop
    const size_t lds = lds_bytes(*i, s.W, P);
    const bool s64 = s.W % 16 == 0 && i->fn_s64 && !finc_no_s64();
    // helper waves: 4 problems per 8-wave workgroup; their rings + 3 progress words each must fit one CU's LDS
    const size_t lds_hlp = 4 * lds + 64;
    const bool hlp = s64 && i->fn_hlp && ((long long)s.B * s.G) % 4 == 0 && lds_hlp <= 160 * 1024 && !finc_no_hlp();
    const wave_fn fn = zpre ? i->fn_zpre : hlp ? i->fn_hlp : s64 ? i->fn_s64 : (s.W % 8 == 0) ? i->fn_sec : i->fn;
    if (!zpre) {                                            // whole rounds + a remainder on the remainder's own kernel (remainder_images)
        if (const int tb = remainder_images(s, i, hlp, lds)) {
            FincShape head = s, tail = s;
            tail.B = tb;
            head.B = s.B - tb;
            const size_t off = (size_t)head.B * s.G * s.Cq * s.H * s.W;
            if (int e = finc_mfma_launch(in, packed, out, head, st, false)) return e;
            return finc_mfma_launch(in + off, packed, out + off, tail, st, false);
        }
    }
    if (i == borrowed_inst(s.Cq, s.KH, s.KW))              // the borrowed two-wave form reads the bank packed behind the bank's own
        packed = (const char *)packed + own_bank_bytes(s.G, s.Cq, s.KH, s.KW);
    if (int e = finc_ensure_dynamic_lds((const void *)fn, hlp ? lds_hlp : lds)) return e;
    if (hlp) {
        if (int e = finc_fault_gate(true, st)) return e;   // (arms the device's fault word if no packing call has: never inside a capture)
    }
    if (hlp)
        hipLaunchKernelGGL(fn, dim3(s.B * s.G / 4), dim3(512), lds_hlp, st, in, (const float *)packed, out, s.G, s.Cq, s.H, s.W, P,
                           Tend, s.orient);
    else
        hipLaunchKernelGGL(fn, dim3(s.B * s.G / i->npw), dim3(64 * i->nw * i->npw), lds, st, in, (const float *)packed, out, s.G, s.Cq, s.H, s.W, P,
                           Tend, s.orient);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

unsigned finc_build_flags_mfma() { return FINC_BUILD_FLAGS; }
