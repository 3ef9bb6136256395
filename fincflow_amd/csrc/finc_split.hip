// Role-split inverse for the UNDER-FILLED chip (gfx950 / CDNA4 only): B*G problems that do not outnumber the compute
// units a few times over (c2; c3 at the batch a GPU gets when 256 images are split 4 or 8 ways; the c4 units).
//
// There the wavefront kernel (finc_mfma.hip) is bound by the LENGTH OF A STEP, not by throughput: a problem is a chain of
// NB*W + P - 1 dependent steps (the reference's anti-diagonal order, cinc_cuda_kernel_level2.cu:49-56,98-111, band by
// band), and a step of one wave is its whole instruction stream -- 162 MFMAs at c3 although only the 36 of the two taps
// with a + b == 1 need the pixel solved in the previous step.  Splitting the reduction over waves (K-split) shortens the
// MFMA part but adds an exchange and two barriers to every step of every wave: 0.9 us per step whatever the split
// (profiles/r03/notes/band_pipeline.md).  So the step is split BY DEPENDENCE instead:
//
//   wave 0 ("A")     carries the recurrence and nothing else: acc = prepared part + taps (0,1) and (1,0) applied to the
//                    pixel of the previous step (registers / DPP row_shr:1, as in the wavefront kernel), pack, publish the
//                    pixel in the x ring (LDS) and the hand-over FIFO.
//   waves 1..NBW ("B") prepare, ONE STEP AHEAD, everything that does not need the newest pixel: Linv*z (+ the folded
//                    bias) and the taps with a + b >= 2, shared out tap by tap.  Their operands are older pixels read
//                    straight from the x ring -- S_a(tau) is the ring slot of step tau at lane p - a, the rows of the
//                    band above (lanes p < a) come from the FIFO -- so they keep no operand history at all.  Each leaves
//                    its partial accumulators in LDS; A adds them up.  The B waves also own the HBM side: one loads z
//                    (dword per lane and k-step, PF steps ahead), one stores the pixels a step after they were solved.
//
// One workgroup barrier per step keeps the four waves in lockstep: what B reads in step t was written in step t-1 or
// earlier, what A reads in step t was written by B in step t-1 (the partial buffers alternate with the parity of the
// step).  A's step is 2*NK*MT MFMAs + pack + one LDS round trip + the barrier: ~1,000 cycles at c3 against ~2,200.
//
// Same packed bank as the wavefront kernel (finc_mfma_pack), same lanes (lane p owns the rows p, P+p, ... and trails
// lane p-1 by one step), same arithmetic (exact fp32 MFMA; only the order in which the partial sums of a pixel are added
// differs).  Any Cq <= CQP (padded channels are masked per k-step) and any W (dword I/O: no alignment rule).
//
// BSP ("bands split over workgroups", round 4; jobs since round 5): with problems to spare CUs for -- B*G <= half the compute
// units: c3 at the 32 images an 8-way strong split leaves a GPU -- ONE problem still is a chain of NB*W + P - 1 steps on one CU
// while the other half of the chip idles.  The bands of 16 rows are therefore JOBS: a workgroup (one per compute unit, the grid
// is min(jobs, CUs)) draws a ticket, band-major, claims that band of that problem, solves it and draws again; the one thing a
// band needs from the band above -- its last KH-1 rows -- comes through memory: the workgroup above stores those rows like every
// other row (write-through), says how far it got in the BAND's progress word, and the consumer's B waves fetch the pieces (past
// the caches) into the very FIFO slots the chained form pushes them to, two windows before the first lane needs them.  A ticket
// is band-major, so the owner of band k-1 drew its ticket before the owner of band k: a waiter's producer runs or is done,
// whatever the residency.  When bands outnumber the workgroups, a workgroup that holds band k goes on into band k+2 without a
// restart if band k+1 is claimed already.  Nothing else changes: same lanes, same visitation inside a band
// (cinc_cuda_kernel_level2.cu:49-56), same arithmetic.  The dependent chain of c3 shrinks from 271 steps to
// 64 + 15 + 3 x (the hand-over lag); the lag is what the memory round trip makes it, ~34 steps.
// The words (ticket, claims, progress) live in a per-device area the library owns, in slots: a STREAM owns its slot (another
// stream takes it over only behind the event of the slot's last launch), and the last workgroup of a launch to finish zeroes
// the launch's words -- so nothing is cleared between launches and a captured launch can be replayed.  Every wait is bounded;
// one that gives up sets the device's fault word (include/finc.h) exactly like the helper-wave protocol of the wavefront kernel.
#include "finc_common.h"
#include "finc_tile.h"

#include <stdlib.h>

#include <mutex>

#include <type_traits>
#include <utility>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr unsigned OFF_INVALID = 0x80000000u;     // voffset beyond any slab: buffer loads return 0, stores are dropped
constexpr unsigned OFF_BAD_CHANNEL = 0x40000000u;  // added to a valid offset it still lands beyond the slab (< 1 GiB)

template <int I>
using IC = std::integral_constant<int, I>;
#define FINC_SB() __builtin_amdgcn_sched_barrier(0)

// channel held by k-slot q of register j of a solved pixel (finc_mfma.hip chan_d: the D layout of the 16-row tiles, then one
// register per reduced 4-row block)
__host__ __device__ inline int chan_d(int MTB, int j, int q)
{
    if (j < 4 * MTB) return 16 * (j >> 2) + 4 * q + (j & 3);
    return 16 * MTB + 4 * (j - 4 * MTB) + q;
}

__device__ inline float row_shr1(float old, float src)   // lane i of each 16-lane row <- lane i-1; lane 0 keeps `old`
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                 0x111, 0xf, 0xf, false));
}

// the taps B prepares (a + b >= 2), ordered by a + b (then row-major): the taps with a + b == 2 come first -- they are the
// only ones that need the pixel solved in the step before, so shared out round-robin every B wave gets one of them
template <int KH, int KW>
struct BTaps {
    static constexpr int count()
    {
        int n = 0;
        for (int a = 0; a < KH; ++a)
            for (int b = 0; b < KW; ++b) n += (a + b >= 2);
        return n;
    }
    static constexpr int find(int i, bool want_a)
    {
        for (int sum = 2; sum <= KH + KW - 2; ++sum)
            for (int a = 0; a < KH; ++a)
                for (int b = 0; b < KW; ++b)
                    if (a + b == sum && i-- == 0) return want_a ? a : b;
        return 0;
    }
    static constexpr int a_of(int i) { return find(i, true); }
    static constexpr int b_of(int i) { return find(i, false); }
};

#ifdef FINC_BSP_TRACE     // diagnostic build: one record per band-split job (scripts/bsp_trace.py)
__device__ unsigned long long finc_bsp_trace[1 + 4 * 4096];   // [0] = records; then {job, blockIdx | xcc << 16 | bands << 24, start, end (s_memrealtime)}
#endif
#ifdef FINC_SPLIT_STAMP   // diagnostic build: busy cycles (barrier exit -> next barrier arrival) per wave of workgroup 0, summed over the steps
__device__ unsigned long long finc_split_stamps[16];
#define FINC_ST_BEGIN() unsigned long long st_b_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_b_)::"memory")
#define FINC_ST_END()                                                                                                     \
    do {                                                                                                                  \
        unsigned long long st_e_;                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_e_)::"memory");             \
        st_busy += st_e_ - st_b_;                                                                                         \
    } while (0)
#else
#define FINC_ST_BEGIN() do { } while (0)
#define FINC_ST_END() do { } while (0)
#endif

constexpr int XSLOTS = 8;        // x ring: the pixels of the last 8 steps (taps reach back KH + KW - 2 <= 8 steps)
constexpr int JSTRIDE = 2048;    // bytes between the k-steps of the x ring (8 slots x 64 lanes) AND of the FIFO: one immediate
constexpr int ZJSTRIDE = 3072;   // bytes between the k-steps of the z ring (12 slots x 64 lanes)
#ifndef FINC_BSP_HSC      // cache-policy bits of the loads that fetch the rows above a band (timing experiments only: anything weaker is stale)
#define FINC_BSP_HSC "sc0 sc1"
#endif
#ifndef FINC_BSP_ABL      // timing-only builds (bit mask; results are garbage): 1 no wait for the producer, 2 no early stores of the handed-over rows,
#define FINC_BSP_ABL 0    // 4 no fetch of the rows above, 8 no publish, 16 the fetched pieces are not written to the FIFO
#endif
#ifndef FINC_BSP_TRIES
#define FINC_BSP_TRIES 6
#endif
constexpr int UNROLL = 8;        // steps per iteration of the B waves' loop: two I/O windows (the in-flight sets alternate)

template <int CQP, int KH, int KW, int NBW>
struct SCfg {
    static constexpr int MTB = CQP / 16, NSM = (CQP % 16) / 4, MT = MTB + NSM, NK = CQP / 4, NTAP = KH * KW;
    static constexpr int NCH = BTaps<KH, KW>::count();
    static constexpr int JS = 4 * (KH - 1);                       // FIFO: floats per slot and k-step (4 k-slots x (KH-1) lanes)
    static constexpr int NPACK = (NK + (NTAP - 1) * NK) * MT + 8 * MT;   // finc_mfma.hip Cfg::NPACK (NW = 1)
    // LDS (bytes): x ring | FIFO | z ring | partial accumulators [parity][B wave][tile][lane] (v4f)
    static constexpr int RING_B = 0, FIFO_B = NK * JSTRIDE, ZR_B = 2 * NK * JSTRIDE, PART_B = ZR_B + NK * ZJSTRIDE;
    static constexpr int LDS_BYTES = PART_B + 2 * NBW * MT * 1024;
    static_assert(KH + KW - 2 <= XSLOTS, "the x ring must reach back to the farthest tap");
    // k-steps whose z a B wave loads and whose z-term it adds; registers of the solved pixel it stores
    static constexpr int jlo(int bi) { return bi * NK / NBW; }
    static constexpr int jhi(int bi) { return (bi + 1) * NK / NBW; }
};

// -----------------------------------------------------------------------------------------------
// grid = B*G workgroups of (1 + NBW) waves.  W % 4 == 0 (16-byte pieces), DF * JS * 4 <= JSTRIDE.
// -----------------------------------------------------------------------------------------------
__device__ unsigned finc_split_timeouts = 0;     // BSP: progress waits that gave up (must stay 0)

template <int CQP, int KH, int KW, int NBW, bool BSP>
__global__ __launch_bounds__(64 * (1 + NBW)) void finc_split_kernel(const float *__restrict__ a_in, const float *__restrict__ a_packed,
                                                                   float *__restrict__ a_out, int a_G, int a_CQ, int a_H, int a_W, int a_P,
                                                                   int a_T, unsigned a_orient, int a_DF, int a_nwg, int a_nprob,
                                                                   unsigned *__restrict__ a_sync, unsigned *a_fault_word)
{
    using C = SCfg<CQP, KH, KW, NBW>;
    constexpr int MT = C::MT, MTB = C::MTB, NK = C::NK, NCH = C::NCH, JS = C::JS;
    using BT = BTaps<KH, KW>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char *const ldsb = reinterpret_cast<char *>(lds);
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // 0: A, 1..NBW: B
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, p = lane & 15;
    // BSP (round 5): a problem's bands are separate jobs, one band each (nwg = the number of bands), handed out by a TICKET counter,
    // band-major; a workgroup draws a ticket, solves that band, draws again.  The producer of a band -- the band above, an earlier
    // ticket -- is therefore running or done whenever a consumer exists, whatever order and placement the dispatcher chose and
    // however few workgroups are resident: a workgroup only ever waits for one that makes progress.  The
    // launch's words (ticket and done counters included) start at zero: the previous launch on the slot left them so.
    // (BSP: the arguments are read from the kernel-argument segment again for every band -- scalar loads, once per job -- instead of
    // being kept in registers across the whole body: 16 more live scalars spilled 40-48 SGPRs in every instantiation)
    typedef const volatile __attribute__((address_space(4))) unsigned *karg_ptr;
    const karg_ptr ka = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    auto karg_p = [&](int dw) { return (unsigned long long)ka[dw] | ((unsigned long long)ka[dw + 1] << 32); };
    for (;;) {
    unsigned *const sync = BSP ? (unsigned *)karg_p(16) : a_sync;
    const int nwg = BSP ? (int)ka[14] : a_nwg, nprob = BSP ? (int)ka[15] : a_nprob;
    const float *const in = BSP ? (const float *)karg_p(0) : a_in;
    const float *const packed = BSP ? (const float *)karg_p(2) : a_packed;
    float *const out = BSP ? (float *)karg_p(4) : a_out;
    const int G = BSP ? (int)ka[6] : a_G, CQ = BSP ? (int)ka[7] : a_CQ, H = BSP ? (int)ka[8] : a_H, W = BSP ? (int)ka[9] : a_W;
    const int P = BSP ? (int)ka[10] : a_P, T = BSP ? (int)ka[11] : a_T, DF = BSP ? (int)ka[13] : a_DF;
    const unsigned orient = BSP ? ka[12] : a_orient;
    unsigned *const fault_word = BSP ? (unsigned *)karg_p(18) : a_fault_word;
    int slot_id = (int)blockIdx.x;
    float *const ctrl = lds + C::LDS_BYTES / 4;   // BSP: two words behind the kernel's own LDS (the launch allocates 82 KB): ticket, bands of this job
    if constexpr (BSP) {
        // a job's CLAIM word ([2 + job]) says a running workgroup owns it: set by the workgroup that drew its ticket, or by the one
        // that extended its own job to it (below).  A drawn ticket whose job is already claimed is skipped.
        if (threadIdx.x == 0) {
            // Ticket t is band t / nprob of problem (t % nprob + band) % nprob: band-major as the order of claims requires, and rotated by
            // one problem per band.  Workgroups tend to draw their tickets in the order of their indices, and the dispatcher deals
            // indices round-robin over the XCDs: unrotated, ticket j (band 0 of problem j) and ticket nprob + j (its band 1) would go to
            // the same XCD whenever nprob % 8 == 0 -- and an XCD whose 32 workgroups all hand over to a neighbour on the same L2 runs
            // 10-15 % slower (profiles/r05/notes/band_split_jobs.txt: job traces).  Rotated, neighbours in a problem sit on neighbouring XCDs.
            int tk;
            for (;;) {
                tk = (int)__hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tk >= nwg * nprob) break;
                const int band = tk / nprob, pr = (tk - band * nprob + band) % nprob;
                tk = band * nprob + pr;                                    // (from here on: the job's index = band * nprob + problem)
                unsigned expect = 0u;
                if (__hip_atomic_compare_exchange_strong(sync + 2 + tk, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            }
            ctrl[0] = __builtin_bit_cast(float, tk);
        }
        __syncthreads();
        slot_id = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ctrl[0]));
        __syncthreads();
        if (slot_id >= nwg * nprob) break;     // no band left
    }
    const int wg = BSP ? slot_id / nprob : 0;  // BSP: the image band of this job (its first one)
    const int bg = BSP ? slot_id - wg * nprob : slot_id, g = bg % G;
    constexpr int BSTRIDE = 2;                 // BSP: a job that is extended continues with band wg + BSTRIDE
    const int band_rows = BSP ? BSTRIDE * P : P;   // rows from a band of this workgroup to its next one
    const int row0 = BSP ? wg * P : 0;         // first row of its first band
    const unsigned o = finc_group_orient(orient, g);
    const bool fh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 4u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);

    for (int i = threadIdx.x; i < C::LDS_BYTES / 4; i += 64 * (1 + NBW)) lds[i] = 0.f;

    const float *pk = packed + (size_t)g * C::NPACK * 64 + lane;
    auto frag_z = [&](int j, int mt) { return pk[(j * MT + mt) * 64]; };
    auto frag_tap = [&](int a, int b, int j, int mt) { return pk[(NK * MT + ((a * KW + b - 1) * NK + j) * MT + mt) * 64]; };
    auto ld = [&](int byte_off) { return *reinterpret_cast<const float *>(ldsb + byte_off); };
    auto st = [&](int byte_off, float v) { *reinterpret_cast<float *>(ldsb + byte_off) = v; };

    __syncthreads();

    if (role == 0) {
        // =================================== A: the recurrence ===================================
        float f01[KW > 1 ? NK : 1][MT], f10[KH > 1 ? NK : 1][MT];
#pragma unroll
        for (int j = 0; j < NK; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (KW > 1) f01[j][mt] = frag_tap(0, 1, j, mt);
                if constexpr (KH > 1) f10[j][mt] = frag_tap(1, 0, j, mt);
            }
        // (the empty asm makes the compiler wait for the loads HERE, not in the loop)
#pragma unroll
        for (int j = 0; j < NK; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (KW > 1) asm volatile("" : "+v"(f01[j][mt]));
                if constexpr (KH > 1) asm volatile("" : "+v"(f10[j][mt]));
            }
        float q0[NK], q1[NK];                  // S_0(t-1) (masked at a row start) and S_1(t-1)
#pragma unroll
        for (int j = 0; j < NK; ++j) q0[j] = q1[j] = 0.f;
        int ca = -p;                           // col of this lane at step t (negative: not started)
        const bool pusher = KH > 1 && !BSP && p >= P - (KH - 1) && p < P;   // (BSP: the FIFO is filled from memory by the B waves)
        // byte addresses inside one k-step's block (the k-step is the instruction's immediate offset)
        const int ring_w = C::RING_B + lane * 4;
        const int push_w = C::FIFO_B + (q * (KH - 1) + (p - (P - (KH - 1)))) * 4;
        const int pop_r = C::FIFO_B + (q * (KH - 1) + (KH - 2)) * 4;       // lane P-1 of the band above: S_1 of lane 0
        const int part_r = C::PART_B + lane * 16;
        // FIFO slots: the push of step s goes to slot s % DF; S_a(tau) of the lanes p < a is the push of step tau - (W - P)
        int fpush = 0;
        int fpop = ((-(W - P)) % DF + DF) % DF;
        int tm = 0;                            // t % W
        int Tj = T;                            // steps of this job
        if constexpr (BSP && KH > 1) {
            // EXTENSION.  A job is one band; a workgroup that went from band k to band k + 2 without a restart would save the restart
            // (ticket, LDS clear, fragment and first-z loads, the rounding of the step count: ~17 us) -- the round-4 form did, by
            // dealing the bands cyclically, and paid with a wait on a workgroup that might not be running.  Here the workgroup takes
            // band k + 2 as well only if band k + 1 is CLAIMED: then whoever owns it runs, and every wait of this job is on a running
            // workgroup (band k + 1's owner needs band k, which this workgroup solves first).  At a launch on an idle chip every
            // ticket is drawn within microseconds, so the pairs {k, k + 2} form; beside another tenant the jobs stay single bands.
            if (lane == 0) {
                int nbl = 1;
                const int NBi = (H + P - 1) / P;
                // (only when the bands outnumber the workgroups: with every band resident at once four single bands trail each other by
                // 32 steps, a pair's second band trails the other pair's first by 44 -- c3 at 16 images: 130 against 137 us)
                if (wg + BSTRIDE < NBi && nwg * nprob > (int)gridDim.x) {
                    bool claimed = false;
                    for (int tries = 0; tries < FINC_BSP_TRIES && !claimed; ++tries) {
                        claimed = __hip_atomic_load(sync + 2 + (wg + 1) * nprob + bg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                        if (!claimed) __builtin_amdgcn_s_sleep(8);
                    }
                    unsigned expect = 0u;
                    if (claimed && __hip_atomic_compare_exchange_strong(sync + 2 + (wg + BSTRIDE) * nprob + bg, &expect, 1u, __ATOMIC_RELAXED,
                                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        nbl = 2;
                }
                ctrl[1] = __builtin_bit_cast(float, nbl);
            }
            // the first band of this workgroup may have real rows above it (image band wg >= 1): the B waves have just landed their
            // first pieces in the FIFO -- S_1(-1) of lane 0 is the pixel above column 0, the push of virtual step -1
            __syncthreads();
            {
                const int nbl = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ctrl[1]));
                Tj = nbl * W + P - 1 + 2;     // (finc_split_launch: the iterations u = 0 .. chain + 3)
            }
            float fv0[NK];
            const int fp0 = fpop == 0 ? DF - 1 : fpop - 1;
#pragma unroll
            for (int j = 0; j < NK; ++j) fv0[j] = ld(pop_r + fp0 * (JS * 4) + j * JSTRIDE);
#pragma unroll
            for (int j = 0; j < NK; ++j) q1[j] = row_shr1(fv0[j], 0.f);
        }
        __syncthreads();                       // (iteration t = -1: the B waves prepare step 0)
        unsigned long long st_busy = 0;
#ifdef FINC_BSP_TRACE
        const unsigned long long tr_start = __builtin_amdgcn_s_memrealtime();
#endif
        for (int t = 0; t <= Tj; ++t) {
            FINC_ST_BEGIN();
            const int par = t & 1, slot = t & (XSLOTS - 1);
            float fv[NK];
            if constexpr (KH > 1) {
                if (W > P) {                   // (W == P: the pop is this very step's push -- below, after it)
#pragma unroll
                    for (int j = 0; j < NK; ++j) fv[j] = ld(pop_r + fpop * (JS * 4) + j * JSTRIDE);
                }
            }
            // what the B waves prepared for this step: read now, added after the MFMAs (the round trip hides behind them)
            v4f prep[NBW][MT];
#pragma unroll
            for (int i = 0; i < NBW; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    prep[i][mt] = *reinterpret_cast<const v4f *>(ldsb + part_r + ((par * NBW + i) * MT + mt) * 1024);
            FINC_SB();                         // (the reads are issued here, not next to their uses)
            // the (0,1) tap reads the pixel left of this one: none at a row start (scalar test: does any lane start a row?)
            if constexpr (KW > 1) {
                if (__builtin_expect(tm < P, 0)) {
                    const bool rowstart = ca == 0;
#pragma unroll
                    for (int j = 0; j < NK; ++j) q0[j] = rowstart ? 0.f : q0[j];
                }
            }
            v4f acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
            // the MFMAs in a fixed order -- per k-step the tiles of one tap, then of the other, so that two MFMAs on the same
            // accumulator are MT instructions apart (a 4x4x1 straight behind its predecessor on the same accumulator stalls) --
            // and nothing else between them: the prepared part is added afterwards, when its LDS round trip is long over
#pragma unroll
            for (int j = 0; j < NK; ++j) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if constexpr (KW > 1) finc_mma<MTB>(acc[mt], mt, f01[j][mt], q0[j]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if constexpr (KH > 1) finc_mma<MTB>(acc[mt], mt, f10[j][mt], q1[j]);
                FINC_SB();
            }
            float xpk[NK];
            {
                v4f sum[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    v4f r = prep[0][mt];
#pragma unroll
                    for (int i = 1; i < NBW; ++i) r += prep[i][mt];
                    sum[mt] = acc[mt] + r;
                }
#pragma unroll
                for (int mt = 0; mt < MTB; ++mt) {
                    xpk[4 * mt + 0] = sum[mt].x; xpk[4 * mt + 1] = sum[mt].y; xpk[4 * mt + 2] = sum[mt].z; xpk[4 * mt + 3] = sum[mt].w;
                }
#pragma unroll
                for (int sb = 0; sb < C::NSM; ++sb) xpk[4 * MTB + sb] = finc_block_reduce(sum[MTB + sb]);
            }
            if (__builtin_expect(t < P - 1 || P < 16, 0)) {                // a lane that has not started yields exact zeros
                const bool started = ca >= 0 && p < P;
#pragma unroll
                for (int j = 0; j < NK; ++j) xpk[j] = started ? xpk[j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < NK; ++j) st(ring_w + slot * 256 + j * JSTRIDE, xpk[j]);
            if constexpr (KH > 1) {
                if (pusher) {
#pragma unroll
                    for (int j = 0; j < NK; ++j) st(push_w + fpush * (JS * 4) + j * JSTRIDE, xpk[j]);
                }
                if (W == P) {
#pragma unroll
                    for (int j = 0; j < NK; ++j) fv[j] = ld(pop_r + fpop * (JS * 4) + j * JSTRIDE);
                }
#pragma unroll
                for (int j = 0; j < NK; ++j) q1[j] = row_shr1(fv[j], xpk[j]);
            }
#pragma unroll
            for (int j = 0; j < NK; ++j) q0[j] = xpk[j];
            ++ca; if (ca == W) ca = 0;
            ++tm; if (tm == W) tm = 0;
            ++fpush; if (fpush == DF) fpush = 0;
            ++fpop; if (fpop == DF) fpop = 0;
            FINC_ST_END();
            __syncthreads();
        }
#ifdef FINC_BSP_TRACE
        if constexpr (BSP) {
            if (lane == 0) {
                const unsigned long long n = __hip_atomic_fetch_add(finc_bsp_trace, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (n < 4096) {
                    unsigned xcc;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                    const int nb = (Tj - 2 - (P - 1)) / W;
                    finc_bsp_trace[1 + 4 * n] = (unsigned long long)slot_id;
                    finc_bsp_trace[2 + 4 * n] = (unsigned long long)blockIdx.x | ((unsigned long long)(xcc & 15) << 16) | ((unsigned long long)nb << 24);
                    finc_bsp_trace[3 + 4 * n] = tr_start;
                    finc_bsp_trace[4 + 4 * n] = __builtin_amdgcn_s_memrealtime();
                }
            }
        }
#endif
#ifdef FINC_SPLIT_STAMP
        if (blockIdx.x == 0 && lane == 0) { finc_split_stamps[0] = st_busy; finc_split_stamps[9] = T + 1; }
#else
        (void)st_busy;
#endif
    } else {

    // =================================== B: everything that can be prepared ===================================
    // One copy of the code per B wave (`bi` is a compile-time constant inside): a step must not contain role branches.
    auto run_b = [&](auto bi_c) {
    constexpr int bi = decltype(bi_c)::value;
    constexpr int JLO = C::jlo(bi), JHI = C::jhi(bi), NJ = JHI - JLO;      // this wave's share of the z-term, of the loads and stores
    constexpr int NT = (NCH + NBW - 1 - bi) / NBW;                          // its taps: items bi, bi + NBW, ...
    float fz[NJ > 0 ? NJ : 1][MT];
    float ft[NT > 0 ? NT : 1][NK][MT];
    v4f bias[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bias[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fz[j][mt] = finc_zterm_is_zero(MTB, JLO + j, mt) ? 0.f : frag_z(JLO + j, mt);
    if constexpr (bi == 0) {                   // the folded affine map's shift: the accumulators' start, added once
        const float *pb = packed + ((size_t)g * C::NPACK + (C::NPACK - 8 * MT)) * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            bias[mt] = (v4f){pb[(4 * mt + 0) * 64], pb[(4 * mt + 1) * 64], pb[(4 * mt + 2) * 64], pb[(4 * mt + 3) * 64]};
    }
    [&]<int... I>(std::integer_sequence<int, I...>) {
        (([&] {
#pragma unroll
             for (int j = 0; j < NK; ++j)
#pragma unroll
                 for (int mt = 0; mt < MT; ++mt) ft[I][j][mt] = frag_tap(BT::a_of(bi + I * NBW), BT::b_of(bi + I * NBW), j, mt);
         }()), ...);
    }(std::make_integer_sequence<int, NT>{});
    // every fragment load must have LANDED before the loop (the empty asm makes the compiler wait here): all VMEM
    // instructions of the loop are inline asm that the kernel counts itself
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(fz[j][mt]));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(bias[mt]));
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NK; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(ft[i][j][mt]));

    // ---- HBM side: 16-byte pieces = groups of 4 canonical columns of one row; group gi of lane p covers its positions
    // n = 4gi .. 4gi+3 (n = step - p).  In window w (steps 4w .. 4w+3) a lane reads z of the groups w+f and w+f+1,
    // f = floor(-p / 4); it LANDS group w+f+2 (requested two windows earlier) into the z ring and REQUESTS group w+f+4.
    // The ring holds 3 groups: 12 slots [slot][lane] per k-step, slot of position n = n mod 12; a W-flipped group is mirrored
    // when it lands.  Stores: in window w the group w + fs, fs = floor((-3 - p) / 4) -- the last one whose four pixels are
    // all in the x ring when the wave prepares step 4w + 2 (i.e. were solved by step 4w) -- is collected from the x ring (time
    // slots) and leaves as one piece.
    unsigned zmask[NJ > 0 ? NJ : 1], xmask[NJ > 0 ? NJ : 1];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        zmask[j] = (4 * (JLO + j) + q) < CQ ? (unsigned)((4 * (JLO + j) + q) * HW * 4) : OFF_BAD_CHANNEL;
        xmask[j] = chan_d(MTB, JLO + j, q) < CQ ? (unsigned)(chan_d(MTB, JLO + j, q) * HW * 4) : OFF_BAD_CHANNEL;
    }
    const int f4 = -((p + 3) >> 2), fs4 = -((p + 3 + 3) >> 2);             // floor(-p / 4), floor((-3 - p) / 4)
    const int dgrp = fw ? -16 : 16;                                        // bytes from a group to the next one of the row
    const int drow = (fh ? -band_rows : band_rows) * W * 4 - (dgrp / 4) * W;   // ... and from the end of a row to the start of the same lane's next row
    auto piece_off = [&](int row, int col0) { return ((fh ? H - 1 - row : row) * W + (fw ? W - 4 - col0 : col0)) * 4; };
    // load walk: next group to request (starts at group f); store walk: next group to store (starts at group fs)
    int nbl = 1, row_lim = H;                  // BSP: bands of this job; rows beyond its last band belong to another workgroup
    int lcol = 4 * f4, lrow = row0 + p, loff = piece_off(row0 + p, 0) + f4 * dgrp;
    int scol = 4 * fs4, srow = row0 + p, soff = piece_off(row0 + p, 0) + fs4 * dgrp;
    v4f zin[2][NJ > 0 ? NJ : 1];               // in flight: the set of window parity wp is requested in the windows of parity wp
    auto zreq = [&](v4f (&dst)[NJ > 0 ? NJ : 1]) {
        const bool ok = lcol >= 0 && lrow < row_lim && p < P;
        const unsigned base = ok ? (unsigned)loff : OFF_INVALID;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst[j]) : "v"(base + zmask[j]), "s"(rin) : "memory");
        lcol += 4; loff += dgrp;
        if (lcol == W) { lcol = 0; lrow += band_rows; loff += drow; }
    };
    // landing: group gl (mod 3) -> slots 4*(gl % 3) + k; element k of the piece is canonical column k, or 3 - k when flipped
    int gland = ((f4 % 3) + 3) % 3;            // ring group of the next landing
    const int zr_w = C::ZR_B + lane * 4;
    const int e0 = fw ? 3 : 0, e1 = fw ? 2 : 1, e2 = fw ? 1 : 2, e3 = fw ? 0 : 3;
    auto zland = [&](v4f (&src)[NJ > 0 ? NJ : 1], auto vm_c) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(vm_c)::value) : "memory");
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(src[j]));      // (ties the reads below to the wait)
        const int base = zr_w + gland * 1024;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float v0 = src[j].x, v1 = src[j].y, v2 = src[j].z, v3 = src[j].w;
            st(base + e0 * 256 + (JLO + j) * ZJSTRIDE, v0);
            st(base + e1 * 256 + (JLO + j) * ZJSTRIDE, v1);
            st(base + e2 * 256 + (JLO + j) * ZJSTRIDE, v2);
            st(base + e3 * 256 + (JLO + j) * ZJSTRIDE, v3);
        }
        gland = gland == 2 ? 0 : gland + 1;
    };
    // stores: element k of the group was solved at step 4gs + k + p: time slot ((p + k) & 7) ^ (4 * (gs & 1))
    const int xs0 = C::RING_B + (((p + e0) & 7) * 64 + lane) * 4, xs1 = C::RING_B + (((p + e1) & 7) * 64 + lane) * 4;
    const int xs2 = C::RING_B + (((p + e2) & 7) * 64 + lane) * 4, xs3 = C::RING_B + (((p + e3) & 7) * 64 + lane) * 4;
    int stog = (fs4 & 1) * 1024;               // toggles with the group
    auto xstore = [&]() {
        const bool ok = scol >= 0 && srow < row_lim && p < P;
        const unsigned base = ok ? (unsigned)soff : OFF_INVALID;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            v4f v;
            v.x = ld((xs0 ^ stog) + (JLO + j) * JSTRIDE);
            v.y = ld((xs1 ^ stog) + (JLO + j) * JSTRIDE);
            v.z = ld((xs2 ^ stog) + (JLO + j) * JSTRIDE);
            v.w = ld((xs3 ^ stog) + (JLO + j) * JSTRIDE);
            // (s_nop: a store of more than 8 bytes reads its data one wait state after issue, and the hazard recognizer does not
            // see inline asm -- without it the next instruction may overwrite the data registers: measured, lanes 12-15 of one register)
            // (BSP: write-through -- the band below is solved on another CU, possibly behind another L2)
            if constexpr (BSP)
                asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc0 sc1\n\ts_nop 1" ::"v"(v), "v"(base + xmask[j]), "s"(rout) : "memory");
            else
                asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(base + xmask[j]), "s"(rout) : "memory");
        }
        stog ^= 1024;
        scol += 4; soff += dgrp;
        if (scol == W) { scol = 0; srow += band_rows; soff += drow; }
    };
    // BSP: the rows that are handed over leave EARLY.  The group a lane of the last KH-1 rows stores in step 2 of a window is
    // complete two steps before (lane 15 solves the last pixel of group w - 5 in step 4w - 2), so at step 0 those lanes store it
    // already (the others do nothing; step 2 stores the whole instruction again, which is harmless): its completion is then
    // covered by the window's one counted wait at step 3, and the progress word says so four steps sooner.
    auto xstore_early = [&]() {
        if constexpr (FINC_BSP_ABL & 2) return;
        const bool ok = scol >= 0 && srow < row_lim && p >= P - (KH - 1) && p < P;
        const unsigned base = ok ? (unsigned)soff : OFF_INVALID;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            v4f v;
            v.x = ld((xs0 ^ stog) + (JLO + j) * JSTRIDE);
            v.y = ld((xs1 ^ stog) + (JLO + j) * JSTRIDE);
            v.z = ld((xs2 ^ stog) + (JLO + j) * JSTRIDE);
            v.w = ld((xs3 ^ stog) + (JLO + j) * JSTRIDE);
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc0 sc1\n\ts_nop 1" ::"v"(v), "v"(base + xmask[j]), "s"(rout) : "memory");
        }
    };
    // ---- BSP: the rows above a band, fetched from the output of the workgroup that solves the band above.
    // Lanes (q, p < KH-1) carry row a' = p + 1 above the band; a piece of 4 canonical columns goes to the 4 FIFO slots the chained
    // form pushes those pixels to: pixel (row -a', column c) of this workgroup's local band i = the push of lane P - a' at local
    // step (i-1)*W + c + P - a', slot = that step mod DF -- so nothing changes for the readers (A's pop, the B waves' taps).  The
    // slot of the previous content (DF steps earlier) was read by local step (i-1)*W + c + P + KH + KW - 2 at the latest and the
    // new one is first read at i*W + c - 1: the piece lands one window before that, two after it was requested.
    // Progress: producer B wave `bi` of workgroup wgp = (wg - 1) mod nwg publishes, in its word, the number of its local windows
    // whose HANDED-OVER stores are complete; group gq of its local band i' leaves (early, step 0) in its window i'*W/4 + gq + 5
    // (rows P-2, P-1: fs4 = -5) and is complete by that window's step 3: word >= i'*W/4 + gq + 6.
    const bool hl = BSP && p < KH - 1;
    int hcol = 0, hband = 0;
    int hoff = BSP ? piece_off(row0 - 1 - p, 0) : 0;
    // FIFO cell (byte address, k-step 0) of each of the four elements of the next piece to LAND: element e of the piece in memory order
    // is canonical column e (3 - e on a W-flipped group) and goes to slot (first + column) % DF; a window moves all four on by 4 slots
    int ha[4] = {0, 0, 0, 0};
    const int hcell = C::FIFO_B + (q * (KH - 1) + (KH - 2 - p)) * 4;
    const int ha_wrap = hcell + DF * (JS * 4);                             // (this lane's cell in slot DF: back to slot 0)
    if constexpr (BSP) {
        const int first = (((-W + P - 1 - p) % DF) + DF) % DF;
#pragma unroll
        for (int e = 0; e < 4; ++e) ha[e] = hcell + ((first + (fw ? 3 - e : e)) % DF) * (JS * 4);
    }
    v4f hin[NJ > 0 ? NJ : 1];
    int seen = 0;                                                          // producer progress read so far
    const int NBimg = (H + P - 1) / P;
    // progress words: one per (problem, image band, B wave) behind the claim words = that wave's store windows complete, counted from
    // the start of THAT band
    const int nwords = BSP ? 2 + nprob * nwg + nprob * nwg * NBW : 0;
    const __amdgpu_buffer_rsrc_t rsync = __builtin_amdgcn_make_buffer_rsrc((void *)sync, 0, nwords * 4, 0x00020000);
    auto band_word = [&](int band) { return (unsigned)(2 + nprob * nwg + (bg * nwg + band) * NBW + bi) * 4u; };
    unsigned its_flag = BSP && wg >= 1 ? band_word(wg - 1) : 0u;
    unsigned seen_raw = 0;                     // the progress word as last fetched (asynchronously, once per window)
    auto progress_fetch = [&]() {              // (no wait: the value is read by progress_take, behind the window's one vmcnt wait)
        asm volatile("buffer_load_dword %0, %1, %2, 0 offen sc0 sc1" : "=v"(seen_raw) : "v"(its_flag), "s"(rsync) : "memory");
    };
    auto progress_take = [&]() {
        asm volatile("" : "+v"(seen_raw));
        const unsigned val = __builtin_amdgcn_readfirstlane(seen_raw);
        if ((int)val > seen) seen = (int)val;
    };
    auto progress_wait = [&](int need) {
        if constexpr (FINC_BSP_ABL & 1) return;
        if (seen >= need) return;
        int budget = 1 << 21;                  // bounded (seconds): a protocol bug must not hang the GPU, and must not pass unnoticed;
                                               // long enough for a producer workgroup that another tenant of the chip keeps waiting
        for (; budget > 0; --budget) {
            unsigned v;
            asm volatile("buffer_load_dword %0, %1, %2, 0 offen sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(its_flag), "s"(rsync) : "memory");
            seen = (int)__builtin_amdgcn_readfirstlane(v);
            if (seen >= need) break;
            __builtin_amdgcn_s_sleep(8);
        }
        if (budget == 0 && lane == 0) {
            atomicAdd(&finc_split_timeouts, 1u);
            if (fault_word) __hip_atomic_store(fault_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    // windows_done counts this job's windows; local band i began at window i * W / 4: lane 0 says it for the band that has begun last,
    // lane 1 for the one before (whose last rows are still leaving); `all`: every band of the job is complete
    auto publish = [&](int windows_done, bool all) {
        const int GR = W >> 2;
        int i1 = windows_done / GR;
        i1 = i1 > nbl - 1 ? nbl - 1 : i1;
        const int bi_l = i1 - lane;                                        // lane 0: band i1, lane 1: band i1 - 1
        const bool act = lane < 2 && bi_l >= 0;
        const unsigned v = all ? 0x7FFFFFFFu : (unsigned)(windows_done - bi_l * GR);
        const unsigned off = act && !(FINC_BSP_ABL & 8) ? band_word(wg + bi_l * BSTRIDE) : OFF_INVALID;
        asm volatile("buffer_store_dword %0, %1, %2, 0 offen sc0 sc1" ::"v"(v), "v"(off), "s"(rsync) : "memory");
    };
    auto hreq = [&]() {
        const int kb = wg + hband * BSTRIDE;                               // the image band these rows sit above
        const bool live = kb >= 1 && kb < NBimg && hband < nbl;            // (band 0: the zero rows above the image; beyond: nothing)
        if (live) progress_wait((hcol >> 2) + 6);
        const unsigned base = (live && hl && !(FINC_BSP_ABL & 4)) ? (unsigned)hoff : OFF_INVALID;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen " FINC_BSP_HSC : "=v"(hin[j]) : "v"(base + xmask[j]), "s"(rout) : "memory");
        hcol += 4; hoff += dgrp;
        if (hcol == W) {                       // on to the rows above this job's next band: another producer, another word
            hcol = 0; ++hband; hoff += drow;
            seen = 0;
            its_flag = band_word(wg + hband * BSTRIDE - 1);
        }
    };
    auto hland = [&]() {                       // (the caller has waited for the loads)
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(hin[j]));
        if (hl && !(FINC_BSP_ABL & 16)) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float v0 = hin[j].x, v1 = hin[j].y, v2 = hin[j].z, v3 = hin[j].w;
                st(ha[0] + (JLO + j) * JSTRIDE, v0);
                st(ha[1] + (JLO + j) * JSTRIDE, v1);
                st(ha[2] + (JLO + j) * JSTRIDE, v2);
                st(ha[3] + (JLO + j) * JSTRIDE, v3);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ha[e] += 4 * (JS * 4);
            ha[e] = ha[e] >= ha_wrap ? ha[e] - DF * (JS * 4) : ha[e];
        }
    };
    // prologue: groups f, f+1 land now (window 0 reads them), f+2 and f+3 wait in the two sets
    if constexpr (NJ > 0) {
        v4f tmp0[NJ], tmp1[NJ];
        zreq(tmp0); zreq(tmp1); zreq(zin[0]); zreq(zin[1]);
        zland(tmp0, IC<0>{}); zland(tmp1, IC<0>{});
        if constexpr (BSP) {                   // the rows above: piece 0 lands before the loop (window w requests AND lands piece w + 1)
            hreq();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            hland();
        }
    }
    int Tj = T;                                // steps of this job
    if constexpr (BSP) {
        __syncthreads();                       // every B wave's first piece of the rows above is in the FIFO (the taps read all k-steps)
        nbl = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ctrl[1]));   // (A's decision: see there)
        const int last_row = (wg + (nbl - 1) * BSTRIDE + 1) * P;
        row_lim = last_row < H ? last_row : H;
        Tj = nbl * W + P - 1 + 2;
    }
    // z read address: slot (n mod 12) of this lane
    int zn = ((-p) % 12 + 12) % 12;

    float vz[NJ > 0 ? NJ : 1], vt[NT > 0 ? NT : 1][NK];                    // operands (see bstep)
    // S_a(u - a - b) of tap I for the step u: the x ring at lane p - a (lanes p >= a) / the FIFO (lanes p < a: the band above)
    auto tap_read = [&](auto i_c, int u, int cbu, int fs2u) {
        constexpr int I = decltype(i_c)::value;
        constexpr int item = bi + I * NBW;
        constexpr int a = BT::a_of(item), b = BT::b_of(item);
        const int tau = u - a - b;
        int fs = fs2u - (a + b - 2);           // slot of push step tau - (W - P)
        if (fs < 0) fs += DF;
        const int ring = C::RING_B + ((tau & (XSLOTS - 1)) * 64 + lane - a) * 4;
        const int fifo = C::FIFO_B + (fs * JS + q * (KH - 1) + (KH - 1 - a + p)) * 4;
        int addr = (a == 0 || p >= a) ? ring : fifo;
        if constexpr (b > 0) addr = cbu >= b ? addr : C::FIFO_B + JSTRIDE - 4;   // (the zero word)
#pragma unroll
        for (int j = 0; j < NK; ++j) vt[I][j] = ld(addr + j * JSTRIDE);
    };
    int cb = 0 - p;                            // col of this lane at the step u being prepared (u = t + 1)
    int fs2 = ((-2 - (W - P)) % DF + DF) % DF; // FIFO slot of S_a(tau) for tau = u - 2: push step tau - (W - P)
    const int part_w = C::PART_B + lane * 16;
    unsigned long long st_busy = 0;

    // the operands of step u that are already in LDS one step early: z and the taps with a + b >= 3
    auto prefetch = [&](int u) {
        if constexpr (NJ > 0) {
            const int za = zr_w + zn * 256;
#pragma unroll
            for (int j = 0; j < NJ; ++j) vz[j] = ld(za + (JLO + j) * ZJSTRIDE);
        }
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (([&] {
                 constexpr int item = bi + I * NBW;
                 if constexpr (BT::a_of(item) + BT::b_of(item) > 2) tap_read(IC<I>{}, u, cb, fs2);
             }()), ...);
        }(std::make_integer_sequence<int, NT>{});
    };
    prefetch(0);
    auto bstep = [&](auto k_c, int t) {
        constexpr int KU = decltype(k_c)::value;                            // u % UNROLL
        constexpr int PH = KU & 3, WP = (KU >> 2) & 1;
        FINC_ST_BEGIN();
        const int u = t + 1;
        constexpr int par = KU & 1;
        v4f acc[MT];
        // ---- HBM side of the window
        if constexpr (NJ > 0 && PH == 0 && BSP) {
            // everything this wave has in memory is waited for here, once per window: last window's stores (two steps old),
            // requests and progress word.  Then: land z and the piece of the rows above requested a window ago, say how far the
            // stores got (windows 0 .. w-1 are complete: w of them), request the next pieces.
            // Window w (BSP).  Step 0: land z (requested two windows ago: long complete, see step 3), ask for the producer's progress
            // word (no wait), request piece w + 1 of the rows above -- its first reader is step 3 of THIS window's successor... no:
            // of this window + 1, i.e. four steps from now -- and the next z.  Step 2: stores.  Step 3: the window's ONE wait,
            // vmcnt(2 NJ): everything up to the requests of step 0 is back (this step's z requests and stores stay in flight), so
            // the piece lands, the progress word is read, and the stores of the windows 0 .. w-1 are complete: w windows.
            zland(zin[WP], IC<63>{});
            xstore_early();                    // (the rows handed over: this window's group of the last KH-1 lanes)
            hreq();                            // (a progress wait that has to spin does so on vmcnt(0): only stores are in flight)
            progress_fetch();
            zreq(zin[WP]);
        }
        if constexpr (NJ > 0 && PH == 3 && BSP) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NJ) : "memory");
            progress_take();
            hland();
            publish(((t + 1) >> 2) + 1, false); // the handed-over stores of the windows 0 .. w are complete: w + 1 windows
        }
        if constexpr (NJ > 0 && PH == 0 && !BSP) {
            // younger than the set that lands: the other set's requests (NJ) and the stores of the two windows in between
            zland(zin[WP], IC<NJ + 2 * NJ>{});
            zreq(zin[WP]);
        }
        if constexpr (NJ > 0 && PH == 2) xstore();
        // ---- z-term (+ bias) of step u
        if constexpr (bi == 0) {
            if (__builtin_expect(u < P - 1 || P < 16, 0)) {
                const bool started = cb >= 0 && p < P;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float b0 = bias[mt].x, b1 = bias[mt].y, b2 = bias[mt].z, b3 = bias[mt].w;
                    acc[mt] = (v4f){started ? b0 : 0.f, started ? b1 : 0.f, started ? b2 : 0.f, started ? b3 : 0.f};
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = bias[mt];
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
        }
        // ---- operands.  What does not need the pixel of the step before -- z and the taps with a + b >= 3 -- was read at the
        // end of the previous step, BEFORE the barrier (vz / vt of those taps arrive as loop state); only the taps with
        // a + b == 2 are read now, and their round trip hides behind the MFMAs of the others.  A tap's column mask (b > 0:
        // column c - b must exist) is applied to the ADDRESS -- an invalid lane reads the zero word of the k-step's FIFO block
        // -- so nothing stands between an LDS result and its MFMA.
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (([&] {
                 constexpr int item = bi + I * NBW;
                 if constexpr (BT::a_of(item) + BT::b_of(item) == 2) tap_read(IC<I>{}, u, cb, fs2);
             }()), ...);
        }(std::make_integer_sequence<int, NT>{});
        FINC_SB();
        if constexpr (NJ > 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (finc_zterm_is_zero(MTB, JLO + j, mt)) continue;
                    finc_mma<MTB>(acc[mt], mt, fz[j][mt], vz[j]);
                }
            FINC_SB();
        }
        // fixed order: the prefetched taps first, one k-step of all of them after the other (MFMAs on the same accumulator
        // stay apart), then the taps read in this step
        auto tap_pass = [&](auto pass_c) {
            constexpr int PASS = decltype(pass_c)::value;
#pragma unroll
            for (int j = 0; j < NK; ++j) {
                [&]<int... I>(std::integer_sequence<int, I...>) {
                    (([&] {
                         constexpr int item = bi + I * NBW;
                         if constexpr ((BT::a_of(item) + BT::b_of(item) == 2) == (PASS == 1)) {
#pragma unroll
                             for (int mt = 0; mt < MT; ++mt) finc_mma<MTB>(acc[mt], mt, ft[I][j][mt], vt[I][j]);
                         }
                     }()), ...);
                }(std::make_integer_sequence<int, NT>{});
                FINC_SB();
            }
        };
        tap_pass(IC<0>{});
        tap_pass(IC<1>{});
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<v4f *>(ldsb + part_w + ((par * NBW + bi) * MT + mt) * 1024) = acc[mt];
        ++cb; if (cb == W) cb = 0;
        ++fs2; if (fs2 == DF) fs2 = 0;
        zn = zn == 11 ? 0 : zn + 1;
        prefetch(u + 1);
        FINC_ST_END();
        __syncthreads();
    };
    // iterations t = -1 .. Tj (u = t + 1 = 0 .. Tj + 1), unrolled by UNROLL and left at the exact step: the last store -- lane P - 1's
    // last group, solved by step chain - 1 -- leaves in the store phase (u = 2 mod 4) of window chain / 4 + 1, i.e. in iteration
    // chain + 3 = Tj + 1; the body is left after its 3rd or 7th step (the extra steps solve rows below the band: nothing is stored)
    for (int t0 = -1; t0 < Tj; t0 += UNROLL) {
        const bool last = [&]<int... K>(std::integer_sequence<int, K...>) {
            return ((bstep(IC<K>{}, t0 + K), (K & 3) == 2 && t0 + K == Tj) || ...);
        }(std::make_integer_sequence<int, UNROLL>{});
        if (last) break;
    }
    if constexpr (BSP && NJ > 0) {
        // the stores of the last windows are not yet accounted for in the progress word (a window's stores are said complete two
        // windows later, and the loop ends with them): the consumer of this workgroup's LAST band waits for exactly those
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish((nbl - 1) * (W >> 2), true);
    }
    if constexpr (BSP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (its last word is out before the workgroup moves on or says "done")
#ifdef FINC_SPLIT_STAMP
    if (blockIdx.x == 0 && lane == 0) finc_split_stamps[1 + bi] = st_busy;
#else
    (void)st_busy;
#endif
    };   // run_b
    [&]<int... BI>(std::integer_sequence<int, BI...>) {
        (([&] {
             if (role - 1 == BI) run_b(IC<BI>{});
         }()), ...);
    }(std::make_integer_sequence<int, NBW>{});
    }   // role
    if constexpr (!BSP) break;
    __syncthreads();                           // (every wave is done with the LDS -- and the B waves with their last words -- before the next band)
    }   // for (;;)
    if constexpr (BSP) {
        unsigned *const sync = a_sync;
        const int nwg = a_nwg, nprob = a_nprob;
        if (role == 0) {
            // The launch leaves its words as it found them: zero.  Every workgroup counts itself done when it finds no band left (all
            // its waves are then past their last access to the words: the barrier above; the B waves drained their stores in front of
            // it); the last one to do so -- every band is then solved and every word final -- zeroes the words of the launch, ticket
            // and done counters last.  So the next launch on this slot, or the next replay of a captured one, starts clean without a
            // memset in front of it (a captured hipMemsetAsync node did not reset the words on replays: ROCm 7.2,
            // scripts/debug_graph_bands.py).
            unsigned done = 0;
            if (lane == 0) done = __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            done = __builtin_amdgcn_readfirstlane(done);
            if (done == gridDim.x - 1) {
                const int nwords = 2 + nprob * nwg + nprob * nwg * NBW;
                for (int i = 2 + lane; i < nwords; i += 64) __hip_atomic_store(sync + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (lane == 0) {
                    __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(sync, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

// -----------------------------------------------------------------------------------------------
// Instantiations: every bank of the wavefront kernel's table with a 2x2 or 3x3 filter that one wave holds
// -----------------------------------------------------------------------------------------------
typedef void (*split_fn)(const float *, const float *, float *, int, int, int, int, int, int, unsigned, int, int, int, unsigned *, unsigned *);
struct SInst {
    int cqp, kh, kw, nbw, lds_bytes;
    split_fn fn;      // one workgroup per problem, its bands chained
    split_fn fn_bsp;  // bands dealt out to several workgroups per problem (BSP)
};
template <int CQP, int KH, int KW, int NBW = 3>
constexpr SInst make_sinst()
{
    return SInst{CQP, KH, KW, NBW, SCfg<CQP, KH, KW, NBW>::LDS_BYTES, finc_split_kernel<CQP, KH, KW, NBW, false>,
                 finc_split_kernel<CQP, KH, KW, NBW, true>};
}

#ifdef FINC_ONLY_C3
const SInst g_sinsts[] = {make_sinst<24, 3, 3>(), make_sinst<12, 3, 3>()};
#else
const SInst g_sinsts[] = {
    make_sinst<4, 3, 3>(),  make_sinst<8, 3, 3>(),  make_sinst<12, 3, 3>(), make_sinst<16, 3, 3>(), make_sinst<20, 3, 3>(),
    make_sinst<24, 3, 3>(), make_sinst<28, 3, 3>(), make_sinst<32, 3, 3>(),
    make_sinst<4, 2, 2>(),  make_sinst<8, 2, 2>(),  make_sinst<12, 2, 2>(), make_sinst<16, 2, 2>(), make_sinst<24, 2, 2>(),
    make_sinst<32, 2, 2>(),
};
#endif

const SInst *find_sinst(int Cq, int KH, int KW)
{
    const int cqp = finc_mfma_packed_cqp(Cq, KH, KW);   // the bank is the wavefront kernel's: same padding rule
    if (cqp == 0) return nullptr;
    for (const SInst &i : g_sinsts)
        if (i.cqp == cqp && i.kh == KH && i.kw == KW) return &i;
    return nullptr;
}

int fifo_depth(int W, int P, int KH, int KW) { return W - P + KH + KW - 2; }

// the FIFO of a k-step must fit its JSTRIDE bytes
bool fifo_fits(const SInst &i, int W, int P) { return (size_t)fifo_depth(W, P, i.kh, i.kw) * 4 * (i.kh - 1) * 4 <= (size_t)JSTRIDE - 4; }   // (+ the zero word)

// problems (B*G) up to which the role-split kernel is the faster one; FINC_SPLIT_MAX overrides (0 turns it off: A/B timing)
long long split_max_problems()
{
    static const long long v = [] { const char *e = finc_env("FINC_SPLIT_MAX"); return e ? atoll(e) : 256LL; }();
    return v;
}

// ---- BSP: where the progress words live.  One area per device, BSP_SLOTS + BSP_GRAPH_SLOTS slots of BSP_SLOT_WORDS words: [0] the
// ticket counter, [1] workgroups done, [2 ..] one word per (problem, band, B wave) = the number of that wave's store windows that are
// complete.  A launch finds its words zero and leaves them zero (its last workgroup to finish clears them), so nothing of an earlier
// launch or replay can be taken for progress.  Launches that may EXECUTE at the same time must not share a slot: a slot
// belongs to a STREAM -- launches of one stream run one after the other -- for as long as that stream keeps launching; a stream
// that is new to a full table takes over the least recently used slot whose last launch has completed (an event per slot says so),
// or runs the chained form.  A launch inside a stream capture takes a slot of its own from a second pool and keeps it for good (a
// replay runs on whatever stream the graph is launched on; replays of one executable graph are ordered among themselves -- two
// executable graphs instantiated from ONE capture must not run concurrently); the BSP_GRAPH_SLOTS + 1st captured band-split launch of
// a process runs the chained form, and so does hipStreamPerThread (one handle, many streams).
constexpr int BSP_SLOTS = 24, BSP_GRAPH_SLOTS = 8, BSP_SLOT_WORDS = 8192;
constexpr int BSP_MAX_DEV = 64;
struct BspSlot {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;                 // recorded behind the slot's last launch
    bool used = false;
    unsigned long long last_use = 0;
};
struct BspDevice {
    unsigned *area = nullptr;                  // (BSP_SLOTS + BSP_GRAPH_SLOTS) * BSP_SLOT_WORDS words
    BspSlot slots[BSP_SLOTS];
    unsigned long long clock = 0;
    int graph_next = 0;
};
BspDevice g_bsp[BSP_MAX_DEV];
std::mutex g_bsp_mutex;

int device_cus()
{
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            return 256;                         // (no device: host-side queries answer for an MI355X)
        }
        return n;
    }();
    return cus;
}

// FINC_SPLIT_BANDS=0 keeps every problem on one workgroup (A/B timing, tests of the chained form)
bool bsp_off()
{
    static const bool off = [] { const char *e = finc_env("FINC_SPLIT_BANDS"); return e && e[0] == '0'; }();
    return off;
}

// workgroups per problem the band split would use for this problem set (1: the chained form): one per band (round 5; each waits
// for the one above only, in ticket order -- no map is too narrow to complete, but below ~48 columns the hand-over lag of ~44 steps
// per band exceeds what chaining costs)
int bsp_nwg(const SInst &i, const FincShape &s)
{
    const int P = s.W < 16 ? s.W : 16;
    const int NB = (s.H + P - 1) / P;
    const long long problems = (long long)s.B * s.G;
    if (bsp_off() || i.kh < 2 || NB < 2 || s.W < 64 || 2 * problems > device_cus()) return 1;
    if (2 + problems * NB * (1 + i.nbw) > BSP_SLOT_WORDS) return 1;          // ticket, done, one claim word and NBW progress words per band
    return NB;
}

} // namespace

int finc_split_prepare(hipStream_t st)
{
    int dev = 0;
    FINC_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= BSP_MAX_DEV) return FINC_ERR_BAD_DIMS;
    if (g_bsp[dev].area) return FINC_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    if (cs != hipStreamCaptureStatusNone) return FINC_OK;     // (not now: the launches of this capture run the chained form)
    std::lock_guard<std::mutex> lk(g_bsp_mutex);
    if (g_bsp[dev].area) return FINC_OK;
    unsigned *a = nullptr;
    const size_t bytes = sizeof(unsigned) * (BSP_SLOTS + BSP_GRAPH_SLOTS) * BSP_SLOT_WORDS;
    FINC_HIP_TRY(hipMalloc((void **)&a, bytes));
    if (hipError_t e = hipMemset(a, 0, bytes); e != hipSuccess) { finc_set_hip_error(e); (void)hipFree(a); return FINC_ERR_LAUNCH; }
    g_bsp[dev].area = a;
    return FINC_OK;
}

// a slot of progress words for one launch on `st`, or nullptr (table full of busy streams, no area, a capture beyond its pool, the
// per-thread stream handle): the chained form
static unsigned *bsp_take_slot(hipStream_t st, int *slot_index)
{
    *slot_index = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= BSP_MAX_DEV) { (void)hipGetLastError(); return nullptr; }
    BspDevice &d = g_bsp[dev];
    if (!d.area || st == hipStreamPerThread) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    std::lock_guard<std::mutex> lk(g_bsp_mutex);
    if (cs != hipStreamCaptureStatusNone) {
        if (d.graph_next >= BSP_GRAPH_SLOTS) return nullptr;
        return d.area + (size_t)(BSP_SLOTS + d.graph_next++) * BSP_SLOT_WORDS;
    }
    int pick = -1;
    for (int i = 0; i < BSP_SLOTS && pick < 0; ++i)
        if (d.slots[i].used && d.slots[i].stream == st) pick = i;           // this stream's own slot
    for (int i = 0; i < BSP_SLOTS && pick < 0; ++i)
        if (!d.slots[i].used) pick = i;                                      // a fresh one
    if (pick < 0) {                                                          // the least recently used slot whose last launch is over
        for (int i = 0; i < BSP_SLOTS; ++i) {
            if (pick >= 0 && d.slots[i].last_use >= d.slots[pick].last_use) continue;
            if (hipEventQuery(d.slots[i].done) != hipSuccess) { (void)hipGetLastError(); continue; }
            pick = i;
        }
        if (pick < 0) return nullptr;
    }
    BspSlot &sl = d.slots[pick];
    if (!sl.done && hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); sl.done = nullptr; return nullptr; }
    sl.used = true;
    sl.stream = st;
    sl.last_use = ++d.clock;
    *slot_index = pick;
    return d.area + (size_t)pick * BSP_SLOT_WORDS;
}

static void bsp_record_slot(hipStream_t st, int slot_index)
{
    if (slot_index < 0) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= BSP_MAX_DEV) { (void)hipGetLastError(); return; }
    std::lock_guard<std::mutex> lk(g_bsp_mutex);
    // (an event that cannot be recorded would let another stream take the slot over while this launch runs: then the slot stays
    // with this stream for good -- its event is dropped, and a slot without an event is never taken over)
    BspSlot &sl = g_bsp[dev].slots[slot_index];
    if (sl.done && hipEventRecord(sl.done, st) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipEventDestroy(sl.done);
        sl.done = nullptr;
    }
}

int finc_split_timeouts_count(unsigned *count)
{
    FINC_HIP_TRY(hipMemcpyFromSymbol(count, HIP_SYMBOL(finc_split_timeouts), sizeof(unsigned)));
    return FINC_OK;
}

#ifdef FINC_BSP_TRACE
extern "C" int finc_debug_bsp_trace(unsigned long long *h, int reset)
{
    int e = (int)hipMemcpyFromSymbol(h, HIP_SYMBOL(finc_bsp_trace), sizeof(finc_bsp_trace));
    if (!e && reset) { const unsigned long long z = 0; e = (int)hipMemcpyToSymbol(HIP_SYMBOL(finc_bsp_trace), &z, sizeof(z)); }
    return e;
}
#endif
#ifdef FINC_SPLIT_STAMP
extern "C" int finc_debug_split_stamps(unsigned long long *h) { return (int)hipMemcpyFromSymbol(h, HIP_SYMBOL(finc_split_stamps), sizeof(finc_split_stamps)); }
#endif

// the short-step form (finc_chain.hip) runs the banks of up to 16 channels -- except where this kernel would deal the bands of a
// 16-channel problem out to two workgroups: there the band split is the faster one (C = 64, 64x64: 85 against 90 us; 128x64: 141
// against 168; profiles/r05/notes/chain_vs_split.txt), while the 12-channel banks win on one workgroup (64x64: 76 against 80 us)
bool finc_split_uses_chain(const FincShape &s)
{
    if (!finc_chain_takes(s)) return false;
    const SInst *i = find_sinst(s.Cq, s.KH, s.KW);
    const int P = s.W < 16 ? s.W : 16;
    if ((long long)s.B * s.G > split_max_problems()) return true;          // (only it takes two problems per compute unit: finc_split_takes)
    if (i && i->cqp == 16 && fifo_fits(*i, s.W, P) && bsp_nwg(*i, s) > 1) return false;
    return true;
}

bool finc_split_takes(const FincShape &s)
{
    const long long problems = (long long)s.B * s.G;
    if (problems > split_max_problems()) {
        // two problems per compute unit: the short-step form still beats the wavefront kernel's table (its workgroup is 45 KB of LDS
        // and five waves) -- C = 48, 32x32, B = 128: 40.3 against 57.3 us; the maps of the CIFAR stack at its sampling batch: 16x16
        // 12.6 against 15.1, 8x8 10.3 against 11.7, 4x4 9.0 against 9.6 (profiles/r05/tiny_maps_kernel_time.txt)
        return problems <= 2 * split_max_problems() && finc_chain_takes(s);
    }
    if (finc_split_uses_chain(s)) return true;                             // (the small banks' short-step form: no FIFO-width limit)
    const SInst *i = find_sinst(s.Cq, s.KH, s.KW);
    if (!i || s.H < 1 || s.W < 1) return false;
    const int P = s.W < 16 ? s.W : 16;
    if (P < s.KH - 1) return false;
    if (s.W % 4 != 0) return false;                                        // 16-byte pieces
    if ((size_t)s.Cq * s.H * s.W * 4 >= ((size_t)1 << 30)) return false;   // buffer-offset range marks (OFF_BAD_CHANNEL)
    return fifo_fits(*i, s.W, P);
}

int finc_split_info(const FincShape &s, int *waves, int *lds, int *steps, int *nwg)
{
    if (finc_split_uses_chain(s)) {
        if (nwg) *nwg = 1;
        return finc_chain_info(s, waves, lds, steps);
    }
    const SInst *i = find_sinst(s.Cq, s.KH, s.KW);
    if (!i) return FINC_ERR_UNSUPPORTED;
    const int P = s.W < 16 ? s.W : 16;
    const int NB = (s.H + P - 1) / P;
    const int n = bsp_nwg(*i, s);
    *waves = 1 + i->nbw;
    *lds = i->lds_bytes;
    *steps = ((NB + n - 1) / n) * s.W + P - 1;         // steps of one workgroup (BSP: + the hand-over lag between workgroups)
    if (nwg) *nwg = n;
    return FINC_OK;
}

int finc_split_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st)
{
    if (finc_split_uses_chain(s)) return finc_chain_launch(in, packed, out, s, st);   // the small banks' short-step form (finc_chain.hip)
    const SInst *i = find_sinst(s.Cq, s.KH, s.KW);
    if (!i) return FINC_ERR_UNSUPPORTED;
    const int P = s.W < 16 ? s.W : 16;
    const int NB = (s.H + P - 1) / P;
    const size_t lds = (size_t)i->lds_bytes;
    int nwg = bsp_nwg(*i, s);
    unsigned *slot = nullptr;
    int slot_index = -1;
    if (nwg > 1) {
        (void)finc_split_prepare(st);                          // (allocates unless a capture is going on)
        slot = bsp_take_slot(st, &slot_index);
        if (!slot) nwg = 1;
    }
    const int T = ((NB + nwg - 1) / nwg) * s.W + P - 1;        // steps of one workgroup
    const int Tr = T + 2;                                      // iterations u = 0 .. T + 3: the last store leaves in u = T + 3 (see the B waves' loop)
    const int DF = fifo_depth(s.W, P, s.KH, s.KW);
    if (nwg > 1) {
        // One workgroup per compute unit, enforced by its LDS size: two chains on one unit run at half the pace each (c3 at 32
        // images with one workgroup per band and no such rule: 216 against 136 us).
        const size_t lds_one_per_cu = lds > 82 * 1024 ? lds : 82 * 1024;
        if (int e = finc_ensure_dynamic_lds((const void *)i->fn_bsp, lds_one_per_cu)) { bsp_record_slot(st, slot_index); return e; }
        // (a workgroup draws bands until none is left: no more workgroups than compute units)
        const long long jobs = (long long)s.B * s.G * nwg;
        const int grid = (int)(jobs < device_cus() ? jobs : device_cus());
        hipLaunchKernelGGL(i->fn_bsp, dim3(grid), dim3(64 * (1 + i->nbw)), lds_one_per_cu, st, in, (const float *)packed, out, s.G, s.Cq, s.H,
                           s.W, P, Tr, s.orient, DF, nwg, s.B * s.G, slot, finc_fault_device_word());
        bsp_record_slot(st, slot_index);
    } else {
        if (int e = finc_ensure_dynamic_lds((const void *)i->fn, lds)) return e;
        hipLaunchKernelGGL(i->fn, dim3(s.B * s.G), dim3(64 * (1 + i->nbw)), lds, st, in, (const float *)packed, out, s.G, s.Cq, s.H, s.W, P,
                           Tr, s.orient, DF, 1, s.B * s.G, (unsigned *)nullptr, (unsigned *)nullptr);
    }
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

unsigned finc_build_flags_split() { return FINC_BUILD_FLAGS; }
