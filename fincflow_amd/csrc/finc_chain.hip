// Short-step inverse for the SMALL banks on the under-filled chip (gfx950 / CDNA4 only): Cq <= 16, B*G problems that do
// not outnumber the compute units -- configs[1] (C = 48, 32x32), the units of the CIFAR Glow stack (Cq = 3, 6, 12).
//
// There the inverse is a chain of NB*W + P - 1 dependent steps (the reference's anti-diagonal order,
// cinc_cuda_kernel_level2.cu:49-56,98-111, band by band) and its time is the LENGTH OF A STEP.  The role-split kernel
// (finc_split.hip) cut the step by dependence -- one wave carries the recurrence, three prepare everything else -- and
// its step still took 1,100 cycles at Cq = 12, of which 150 were the recurrence wave's MFMAs (profiles/r05/notes): the
// three preparing waves split the work BY TAP, so every step the recurrence wave read nine partial accumulators, added
// them up, ran three 4x4 transpose-reduces and moved three registers per ring.  This kernel keeps the roles and
// changes the cut (profiles/r05/notes/chain_kernel.md has the measurements that led here: an LDS hand-over -- write,
// acknowledge, barrier, read -- costs 220 cycles before any arithmetic, a 4x4 transpose-reduce 80, every instruction of a wave adds):
//
//   wave 0 ("A")       the recurrence on ONE 16-row tile (v_mfma_f32_16x16x4_f32), its rows permuted so that register r of
//                      the result, lane row q, IS channel chan_d(r, q) of the pixel -- the very register the next step's
//                      MFMAs take as their B operand and the very 16-byte cell the other waves read.  Per step: NBW
//                      16-byte LDS reads (what the B waves prepared), 2*NK MFMAs, a few adds, one 16-byte LDS write.  No
//                      transpose-reduce.
//   waves 1..NBW ("B") split the remaining taps -- wave i takes the i-th tap with a + b == 2 (the only ones that need the pixel
//                      solved in the step before), the i-th with a + b >= 3 and every NBW-th k-step of Linv*z -- on the SAME
//                      permuted tile: 2*NK + 1 MFMAs, no reduce either, and the result is one 16-byte cell that A adds.
//   wave NBW + 1       the HBM side and the rows handed over between bands ("I/O wave"): not one instruction of it is on
//                      the path of a step.
//
// A pixel is a 16-byte cell [lane][register] everywhere (x ring, rows handed over between bands, what B prepares), so a
// tap's operand is ONE ds_read_b128 and a solved pixel ONE ds_write_b128.  The rows above a band are copied from the
// hand-over FIFO into eight "halo" cells of the ring slot they belong to in time, so a tap's address is a per-lane
// constant plus an immediate (the B loop is unrolled by the ring's 8 slots): no address arithmetic per step.
//
// z comes in by LDS-DMA (buffer_load_dwordx4 ... lds): a window's pieces go straight from memory into a ring of NG slabs,
// DPF windows ahead -- no registers, no landing instructions, one counted vmcnt per window -- because with steps this
// short a piece needs 20-40 steps to arrive.
//
// Same packed bank as the wavefront kernel (finc_mfma_pack), same lanes (lane p owns rows p, P+p, ... and trails lane
// p-1 by one step), same visitation, exact fp32 MFMA; only the order in which a pixel's partial sums are added differs.
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

// channel held by k-slot q of register j of a solved pixel (finc_mfma.hip chan_d)
__host__ __device__ constexpr int chan_d(int MTB, int j, int q)
{
    if (j < 4 * MTB) return 16 * (j >> 2) + 4 * q + (j & 3);
    return 16 * MTB + 4 * (j - 4 * MTB) + q;
}

__device__ inline float row_shr1(float old, float src)   // lane i of each 16-lane row <- lane i-1; lane 0 keeps `old`
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                 0x111, 0xf, 0xf, false));
}

// the taps B prepares (a + b >= 2), ordered by a + b (then row-major)
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

#ifdef FINC_SPLIT_STAMP   // diagnostic build: busy cycles (barrier exit -> next barrier arrival) per wave of workgroup 0
__device__ unsigned long long finc_chain_stamps[32];
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
#if defined(FINC_SPLIT_STAMP) && FINC_SPLIT_STAMP == 2   // fine stamps: raw s_memtime values at three points of a step, summed per segment
#undef FINC_ST_BEGIN
#undef FINC_ST_END
#define FINC_ST_BEGIN() unsigned long long fs0_, fs1_ = 0, fs2_ = 0; asm volatile("s_memtime %0" : "=s"(fs0_)::"memory")
#define FINC_ST_MID1() asm volatile("s_memtime %0" : "=s"(fs1_)::"memory")
#define FINC_ST_MID2() asm volatile("s_memtime %0" : "=s"(fs2_)::"memory")
#define FINC_ST_END()                                                                                                     \
    do {                                                                                                                  \
        unsigned long long fs3_;                                                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(fs3_), "+s"(fs0_), "+s"(fs1_), "+s"(fs2_)::"memory");     \
        seg[0] += fs1_ - fs0_; seg[1] += fs2_ - fs1_; seg[2] += fs3_ - fs2_; st_busy += fs3_ - fs0_;                      \
    } while (0)
#define FINC_ST_DECL() unsigned long long seg[3] = {0, 0, 0}
#define FINC_ST_OUT(base) do { if (blockIdx.x == 0 && lane == 0) { finc_chain_stamps[base] = seg[0]; finc_chain_stamps[base + 1] = seg[1]; finc_chain_stamps[base + 2] = seg[2]; } } while (0)
#else
#define FINC_ST_MID1() do { } while (0)
#define FINC_ST_MID2() do { } while (0)
#define FINC_ST_DECL() do { } while (0)
#define FINC_ST_OUT(base) do { } while (0)
#endif

#ifndef FINC_CHAIN_ABLATE   // timing-only builds (bit mask): 1 B without MFMAs, 2 B without the reduce, 4 no HBM side in the loop, 8 A without
#define FINC_CHAIN_ABLATE 0 // MFMAs, 16 no copy of the rows above, 32 B without its prefetched operands, 64 B without the taps read in the step
#endif
constexpr int ABL = FINC_CHAIN_ABLATE;

constexpr int XS = 8;             // x ring: the pixels of the last 8 steps (taps reach back KH + KW - 2 <= 4; the store side 7)
constexpr int HALO = 8;           // cells per slot for the rows above the band: 4 k-slots x (KH - 1 <= 2) rows
constexpr int SLOT_B = (64 + HALO + 1 + 8) * 16;   // a ring slot: 64 lane cells, the halo cells, one ZERO cell (column masks read it), 8 trash cells
constexpr int ZERO_CELL = (64 + HALO) * 16, TRASH_CELL = (64 + HALO + 1) * 16;
constexpr int NG = 8;             // z ring: slabs (one per window) per k-step
constexpr int DPF = 6;            // ... requested this many windows ahead
constexpr int FSLOT_B = HALO * 16;   // bytes per slot of the hand-over FIFO
constexpr int UNROLL = 8;         // B loop: the ring's slots become immediates; two I/O windows
static_assert(NG > DPF && (NG & (NG - 1)) == 0, "a slab is overwritten only after its last reader");

template <int CQP, int KH, int KW, int NBW>
struct LCfg {
    static_assert(CQP % 4 == 0 && CQP <= 16, "one 16-row tile carries the recurrence");
    static constexpr int MTB = CQP / 16;          // the packed bank's tiling (finc_mfma.hip Cfg): one 16-row tile, or 4-row blocks
    static constexpr int NK = CQP / 4;            // k-steps = registers of a solved pixel = output blocks
    static constexpr int MT = MTB ? 1 : NK;
    static constexpr int NTAP = KH * KW, NCH = BTaps<KH, KW>::count();
    static constexpr int NPACK = (NK + (NTAP - 1) * NK) * MT + 8 * MT;   // finc_mfma.hip Cfg::NPACK (NW = 1)
    static_assert(NBW >= 1 && NBW <= 3, "B waves");
    // LDS (bytes): ring slots | what the B waves prepared [parity][wave][lane] | z ring [k-step][slab][lane] | FIFO [slot][halo cell]
    static constexpr int RING_B = 0, PREP_B = XS * SLOT_B, ZR_B = PREP_B + 2 * NBW * 1024, FIFO_B = ZR_B + NK * NG * 1024;
    static constexpr int lds_bytes(int DF) { return FIFO_B + 2 * DF * FSLOT_B; }   // (FIFO + its trash copy: A's branch-free push)
    static_assert(KH + KW - 2 <= 4 && KH - 1 <= 2, "taps reach back at most 4 steps, at most 2 rows");
};

// element of the packed bank: row channel `c`, k-step j, k-slot q of fragment group `f0` (= index of the group's first fragment)
template <int MTB, int MT>
__device__ inline float bank_elem(const float *pk, int f0, int j, int q, int c)
{
    // 16-row tile: fragment (j, 0), lane (q, c); 4-row blocks: fragment (j, c >> 2), lane (q, c & 3)
    const int f = f0 + j * MT + (MTB ? 0 : (c >> 2));
    return pk[(size_t)f * 64 + q * 16 + (MTB ? c : (c & 3))];
}

// -----------------------------------------------------------------------------------------------
// grid = B*G workgroups of (2 + NBW) waves.  W % 4 == 0 (16-byte pieces).
// -----------------------------------------------------------------------------------------------
template <int CQP, int KH, int KW, int NBW>
__global__ __launch_bounds__(64 * (2 + NBW)) void finc_chain_kernel(const float *__restrict__ in, const float *__restrict__ packed,
                                                                   float *__restrict__ out, int G, int CQ, int H, int W, int P,
                                                                   int T, unsigned orient, int DF)
{
    using C = LCfg<CQP, KH, KW, NBW>;
    constexpr int MTB = C::MTB, MT = C::MT, NK = C::NK, NCH = C::NCH;
    using BT = BTaps<KH, KW>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char *const ldsb = reinterpret_cast<char *>(lds);
    // roles: 0 = A, 1..NBW = B, NBW + 1 = the I/O wave.  A workgroup's waves go to the four SIMDs in turn, so with five waves the
    // last one shares its SIMD with the first: wave 0 is the I/O wave (no MFMA, never on a step's path), wave 1 is A, then the B waves
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wv == 0 ? NBW + 1 : wv - 1;
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, p = lane & 15;
    const int bg = (int)blockIdx.x, g = bg % G;
    const unsigned o = finc_group_orient(orient, g);
    const bool fh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 4u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    const bool direct = W == P;                 // the band below starts the very step a row above it is solved: A writes the halo itself
#ifdef FINC_SPLIT_STAMP
    const unsigned long long st_entry = __builtin_amdgcn_s_memtime();
#endif

    // Everything but the z ring starts as zeros (a slab of the z ring is written by its request before anything reads it -- lanes
    // whose piece lies outside the image get zeros from the buffer unit), so the I/O wave takes no part in the clear: its first
    // requests are under way while the other waves clear and gather their fragments; barrier (2) closes both.
    auto clear_lds = [&]() {                    // (called by the A and B waves once their fragment loads are in flight)
        constexpr int Z0 = C::ZR_B / 16, Z1 = C::FIFO_B / 16;
        const int n16 = C::lds_bytes(DF) / 16 - (Z1 - Z0);
        for (int i = threadIdx.x - 64; i < n16; i += 64 * (1 + NBW)) {
            reinterpret_cast<v4f *>(ldsb)[i < Z0 ? i : i + (Z1 - Z0)] = (v4f){0.f, 0.f, 0.f, 0.f};
        }
    };
    const float *pk = packed + (size_t)g * C::NPACK * 64;
    auto tap_f0 = [&](int a, int b) { return NK * MT + (a * KW + b - 1) * NK * MT; };
    auto ld4 = [&](int byte_off) { return *reinterpret_cast<const v4f *>(ldsb + byte_off); };
    auto st4 = [&](int byte_off, v4f v) { *reinterpret_cast<v4f *>(ldsb + byte_off) = v; };
    auto ld1 = [&](int byte_off) { return *reinterpret_cast<const float *>(ldsb + byte_off); };

    if (role == 0) {
        // =================================== A: the recurrence ===================================
        // tile row i = 4q' + r  <->  channel chan_d(r, q'): register r of the result, lane row q', is the B operand of k-step r
        const int arow = chan_d(MTB, p & 3, p >> 2);
        const bool arow_ok = MTB ? true : (p & 3) < NK;
        float f01[NK], f10[NK];
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            f01[j] = (KW > 1 && arow_ok) ? bank_elem<MTB, MT>(pk, tap_f0(0, 1), j, q, arow) : 0.f;
            f10[j] = (KH > 1 && arow_ok) ? bank_elem<MTB, MT>(pk, tap_f0(1, 0), j, q, arow) : 0.f;
        }
        clear_lds();
#pragma unroll
        for (int j = 0; j < NK; ++j) {          // (the empty asm makes the compiler wait for the loads HERE, not in the loop)
            asm volatile("" : "+v"(f01[j]));
            asm volatile("" : "+v"(f10[j]));
        }
        const bool pusher = KH > 1 && p >= P - (KH - 1) && p < P;
        const int cell_w = C::RING_B + lane * 16;
        const int prep_r = C::PREP_B + lane * 16;
        const int hcell = (q * (KH - 1) + (P - 1 - p)) * 16;                  // halo / FIFO cell of a pusher lane: row a' = P - p above the next band
        const int pop_r = C::RING_B + 64 * 16 + (q * (KH - 1)) * 16;          // halo cell a' = 1: S_1 of lane 0
        // the push is branch-free: every lane writes, the lanes that hand nothing over into a trash copy of the FIFO behind it
        const int push_w = pusher ? C::FIFO_B + hcell : C::FIFO_B + DF * FSLOT_B + (lane & 7) * 16;
        const int push_d = pusher ? C::RING_B + 64 * 16 + hcell : C::RING_B + TRASH_CELL + (lane & 7) * 16;   // (W == P: straight into the slot's halo)
        __syncthreads();                        // (2) LDS is zero, the I/O wave has the first slabs
        __syncthreads();                        // (3) iteration t = -1: the B waves prepare step 0
        unsigned long long st_busy = 0;
#ifdef FINC_SPLIT_STAMP
        const unsigned long long st_loop = __builtin_amdgcn_s_memtime();
#endif
        // (a lane that has not started yields exact zeros by itself: its z is zero -- the request lies outside the slab -- and so is
        // every pixel its taps read; a folded shift is masked where it enters, in B wave 0)
        auto loop_a = [&](auto direct_c) {
            constexpr bool DIRECT = decltype(direct_c)::value != 0;
            float q0[NK], q1[NK];               // S_0(t-1) (zero at a row start) and S_1(t-1)
#pragma unroll
            for (int j = 0; j < NK; ++j) q0[j] = q1[j] = 0.f;
            int fpush = 0;                      // FIFO slot of this step's push: t % DF, in bytes
            int tm = 1 % W;                     // (t + 1) % W
            int soff = 0;                       // (t & 7) * SLOT_B
            FINC_ST_DECL();
            for (int t = 0; t <= T; ++t) {
                FINC_ST_BEGIN();
                const int par = t & 1;
                v4f prep[NBW];
#pragma unroll
                for (int i = 0; i < NBW; ++i) prep[i] = ld4(prep_r + (par * NBW + i) * 1024);
                v4f fv = (v4f){0.f, 0.f, 0.f, 0.f};
                if constexpr (KH > 1 && !DIRECT && !(ABL & 1024)) fv = ld4(pop_r + soff);      // halo of step t (copied from the FIFO two steps ago)
                FINC_SB();
                v4f acc0 = (v4f){0.f, 0.f, 0.f, 0.f}, acc1 = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NK; ++j) {
                    if constexpr (ABL & 8) { asm volatile("" ::"v"(q0[j]), "v"(q1[j])); continue; }
                    if constexpr (KW > 1) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f01[j], q0[j], acc0, 0, 0, 0);
                    if constexpr (KH > 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f10[j], q1[j], acc1, 0, 0, 0);
                }
                // ---- in the shadow of the MFMAs: the step's bookkeeping, and what the B waves prepared
                const bool rowstart = tm == p;  // (of the NEXT step: the (0,1) tap reads the pixel left of this one -- none at a row start)
                const int soff_now = soff, fpush_now = fpush;
                ++tm; if (tm == W) tm = 0;
                fpush += FSLOT_B; if (fpush == DF * FSLOT_B) fpush = 0;
                soff += SLOT_B; if (soff == XS * SLOT_B) soff = 0;
                // what the B waves prepared enters behind an accumulator each: no wait for LDS data can then stand between the MFMAs (the
                // wave issues in order), and one add is left behind the last MFMA
                v4f x0s = acc0 + prep[0], x1s = acc1;
                if constexpr (NBW > 1) x1s += prep[1];
                if constexpr (NBW > 2) x0s += prep[2];
                const v4f x = x0s + x1s;
                FINC_ST_MID1();
                st4(cell_w + soff_now, x);
                if constexpr (KH > 1) {
                    if constexpr (DIRECT) {
                        st4(push_d + soff_now, x);
                        fv = ld4(pop_r + soff_now);                          // (W == P: the pop is this very step's push)
                    } else {
                        if constexpr (!(ABL & 512)) st4(push_w + fpush_now, x);
                    }
                }
                FINC_SB();                      // (the copies of the halo read into the DPP's registers must not travel up between the MFMAs with their wait)
                {
                    const float x0 = x.x, x1 = x.y, x2 = x.z, x3 = x.w;
                    const float xs[4] = {x0, x1, x2, x3};
#pragma unroll
                    for (int j = 0; j < NK; ++j) q0[j] = rowstart ? 0.f : xs[j];
                    if constexpr (KH > 1) {
                        const float v0 = fv.x, v1 = fv.y, v2 = fv.z, v3 = fv.w;
                        const float vs[4] = {v0, v1, v2, v3};
#pragma unroll
                        for (int j = 0; j < NK; ++j) q1[j] = row_shr1(vs[j], xs[j]);
                        // (all four registers of the read stay reserved until here: a register of the tuple that is never used -- NK < 4 --
                        // gets reused by the allocator at once, and the wait for the read that then guards it lands between the MFMAs)
                        asm volatile("" ::"v"(fv));
                    }
                }
                FINC_ST_MID2();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                FINC_ST_END();
                __syncthreads();
            }
            FINC_ST_OUT(16);
        };
        if (direct) loop_a(IC<1>{}); else loop_a(IC<0>{});
#ifdef FINC_SPLIT_STAMP
        if (blockIdx.x == 0 && lane == 0) {
            finc_chain_stamps[0] = st_busy; finc_chain_stamps[9] = T + 1;
            finc_chain_stamps[10] = st_loop - st_entry; finc_chain_stamps[11] = __builtin_amdgcn_s_memtime() - st_loop;
        }
#else
        (void)st_busy;
#endif
        return;
    }

    if (role == NBW + 1) {
    // =================================== the I/O wave: memory <-> LDS, and the rows above a band ===================================
    // ---- HBM side: 16-byte pieces = groups of 4 canonical columns of one row; group gi of lane p covers its positions
    // n = 4gi .. 4gi+3 (n = step - p).  Slab s of the z ring holds, for lane p, its group s + f4, f4 = floor(-p / 4): the position
    // n = u - p of step u lies in slab (u + e) >> 2, element (u + e) & 3, e = (-p) mod 4.  Window w (steps 4w .. 4w+3) requests slab
    // w + DPF (LDS-DMA; step 0), stores group w + fs4, fs4 = floor((-3 - p) / 4) (collected from the x ring; step 2), and makes sure
    // that slab w + 2 has landed (step 3; first read at the end of step 4w + 4).  Loads and stores alternate from the first
    // virtual window -DPF on (the stores of the windows w < 0 are dropped), so the one counted wait is the same number always.
    unsigned zmask[NK], xmask[NK];
#pragma unroll
    for (int n = 0; n < NK; ++n) {
        const int j = n;
        zmask[n] = (4 * j + q) < CQ ? (unsigned)((4 * j + q) * HW * 4) : OFF_BAD_CHANNEL;
        xmask[n] = chan_d(MTB, j, q) < CQ ? (unsigned)(chan_d(MTB, j, q) * HW * 4) : OFF_BAD_CHANNEL;
    }
    const int f4 = -((p + 3) >> 2), fs4 = -((p + 3 + 3) >> 2);             // floor(-p / 4), floor((-3 - p) / 4)
    const int dgrp = fw ? -16 : 16;                                        // bytes from a group to the next one of the row
    const int drow = (fh ? -P : P) * W * 4 - (dgrp / 4) * W;               // ... and from the end of a row to the start of the same lane's next row
    auto piece_off = [&](int row, int col0) { return ((fh ? H - 1 - row : row) * W + (fw ? W - 4 - col0 : col0)) * 4; };
    int lcol = 4 * f4, lrow = p, loff = piece_off(p, 0) + f4 * dgrp;
    int scol = 4 * (fs4 - DPF), srow = p, soff = piece_off(p, 0) + (fs4 - DPF) * dgrp;
    int zslab = 0;                              // slab of the next request
    auto zreq = [&]() {
        const bool ok = lcol >= 0 && lrow < H && p < P;
        const unsigned base = ok ? (unsigned)loff : OFF_INVALID;
#pragma unroll
        for (int n = 0; n < NK; ++n) {
            const int m0v = C::ZR_B + ((n) * NG + zslab) * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(base + zmask[n]), "s"(rin) : "memory");
        }
        zslab = (zslab + 1) & (NG - 1);
        lcol += 4; loff += dgrp;
        if (lcol == W) { lcol = 0; lrow += P; loff += drow; }
    };
    // stores: element k of group gs was solved at step 4gs + k + p: time slot ((p + k) & 7) ^ (4 * (gs & 1)); gs & 1 = (w & 1) ^ (fs4 & 1)
    const int e_of[4] = {fw ? 3 : 0, fw ? 2 : 1, fw ? 1 : 2, fw ? 0 : 3};   // element k of the piece is canonical column k, or 3 - k when flipped
    int xs[2][4];
#pragma unroll
    for (int wp = 0; wp < 2; ++wp)
#pragma unroll
        for (int k = 0; k < 4; ++k) xs[wp][k] = C::RING_B + ((((p + e_of[k]) & 7) ^ (4 * ((wp ^ fs4) & 1))) * SLOT_B) + lane * 16;
    auto xstore = [&](auto wp_c) {
        constexpr int WP = decltype(wp_c)::value;
        const bool ok = scol >= 0 && srow < H && p < P;
        const unsigned base = ok ? (unsigned)soff : OFF_INVALID;
#pragma unroll
        for (int n = 0; n < NK; ++n) {
            const int j = n;
            v4f v;
            v.x = ld1(xs[WP][0] + j * 4);
            v.y = ld1(xs[WP][1] + j * 4);
            v.z = ld1(xs[WP][2] + j * 4);
            v.w = ld1(xs[WP][3] + j * 4);
            // (s_nop: a store of more than 8 bytes reads its data one wait state after issue, and the hazard recognizer does not
            // see inline asm -- without it the next instruction may overwrite the data registers)
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(base + xmask[n]), "s"(rout) : "memory");
        }
        scol += 4; soff += dgrp;
        if (scol == W) { scol = 0; srow += P; soff += drow; }
    };
    // virtual windows -DPF .. -1, then: slabs 0 and 1 have landed
#pragma unroll 1
    for (int v = 0; v < DPF; ++v) {
        zreq();
        if (v & 1) xstore(IC<1>{}); else xstore(IC<0>{});
    }
    // (the stores above were dropped: what they read from the ring, cleared or not yet, did not matter)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NK * (2 * DPF - 3)) : "memory");
    __syncthreads();                            // (2)
    // the copy of the rows above: FIFO slot of push step (u + 1) - (W - P) -> halo of ring slot (u + 1) & 7, in iteration u - 1
    const bool copier = KH > 1 && !direct && lane < 4 * (KH - 1);
    int fcopy = ((1 - (W - P)) % DF + DF) % DF;
    const int ccell = lane * 16;
    unsigned long long st_busy = 0;
    FINC_ST_DECL();
    auto iostep = [&](auto k_c) {
        constexpr int KU = decltype(k_c)::value;                            // u % UNROLL
        constexpr int PH = KU & 3, WP = (KU >> 2) & 1;
        FINC_ST_BEGIN();
        if constexpr (KH > 1 && !(ABL & 16)) {
            if (copier) {
                const v4f hv = ld4(C::FIFO_B + ccell + fcopy * FSLOT_B);
                st4(C::RING_B + 64 * 16 + ccell + ((KU + 1) & 7) * SLOT_B, hv);
            }
            ++fcopy; if (fcopy == DF) fcopy = 0;
        }
        if constexpr (PH == 0 && !(ABL & 4)) zreq();
        if constexpr (PH == 2 && !(ABL & 4)) xstore(IC<WP>{});
        if constexpr (PH == 3 && !(ABL & 4)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NK * (2 * DPF - 3)) : "memory");
        FINC_ST_END();
        __syncthreads();
    };
    // (iterations u = 0 .. T + 1; finc_chain_launch: T + 1 = 2 mod 4, the last one is a store phase -- the unrolled body is left after its 3rd or 7th step)
    for (int t0 = -1; t0 < T; t0 += UNROLL) {
        const bool last = [&]<int... K>(std::integer_sequence<int, K...>) {
            return ((iostep(IC<K>{}), (K & 3) == 2 && t0 + K == T) || ...);
        }(std::make_integer_sequence<int, UNROLL>{});
        if (last) break;
    }
#ifdef FINC_SPLIT_STAMP
    if (blockIdx.x == 0 && lane == 0) finc_chain_stamps[1 + NBW] = st_busy;
#else
    (void)st_busy;
#endif
    return;
    }


    // =================================== B: everything that can be prepared ===================================
    // One copy of the code per B wave (`bi` is a compile-time constant inside): a step must not contain role branches.
    auto run_b = [&](auto bi_c) {
    constexpr int bi = decltype(bi_c)::value;
    constexpr int NT = (NCH + NBW - 1 - bi) / NBW;                          // its taps: items bi, bi + NBW, ... of BTaps (a + b == 2 first)
    constexpr int NZ = (NK + NBW - 1 - bi) / NBW;                           // its k-steps of the z-term: bi, bi + NBW, ...
    const int arow = chan_d(MTB, p & 3, p >> 2);                            // (A's tile: row i = 4q' + r <-> channel chan_d(r, q'))
    const bool arow_ok = MTB ? true : (p & 3) < NK;
    float fz[NZ > 0 ? NZ : 1], ft[NT > 0 ? NT : 1][NK];
#pragma unroll
    for (int n = 0; n < NZ; ++n) fz[n] = arow_ok ? bank_elem<MTB, MT>(pk, 0, bi + n * NBW, q, arow) : 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NK; ++j)
            ft[i][j] = arow_ok ? bank_elem<MTB, MT>(pk, tap_f0(BT::a_of(bi + i * NBW), BT::b_of(bi + i * NBW)), j, q, arow) : 0.f;
    // the folded shift (finc_mfma.hip pack_kernel: accumulator layout) in cell layout -- register r, lane row q = channel chan_d(r, q):
    // the start of wave 0's accumulator
    v4f bias = (v4f){0.f, 0.f, 0.f, 0.f};
    if constexpr (bi == 0) {
        const float *pb = pk + (size_t)(C::NPACK - 8 * MT) * 64;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < NK; ++r) bv[r] = MTB ? pb[r * 64 + lane] : pb[(4 * r + q) * 64 + p];
        bias = (v4f){bv[0], bv[1], bv[2], bv[3]};
    }
    clear_lds();
    asm volatile("" : "+v"(bias));              // every fragment load must have LANDED before the loop
#pragma unroll
    for (int n = 0; n < NZ; ++n) asm volatile("" : "+v"(fz[n]));
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NK; ++j) asm volatile("" : "+v"(ft[i][j]));
    __syncthreads();                            // (2) LDS is zero, the I/O wave has the first slabs

    // ---- operands.  S_a(tau) of tap (a, b): ring slot tau & 7 at lane p - a, or the slot's halo cell (row a - p above the band)
    // for the lanes p < a: a per-lane constant per a; the slot is the read's immediate.  A tap's column mask (b > 0: column c - b
    // must exist) is applied to the ADDRESS -- an invalid lane reads the slot's zero cell.
    int taddr[KH];
#pragma unroll
    for (int a = 0; a < KH; ++a)
        taddr[a] = C::RING_B + (p >= a ? (lane - a) * 16 : (64 + q * (KH - 1) + (a - 1 - p)) * 16);
    int cb = 0 - p;                             // col of this lane at the step u being prepared (negative: not started)
    int ustep = 0;                              // u
    int m4 = 4 * ((4 - (p & 3)) & 3);           // 4 * (u + e): slab and element of this lane's z for step u
    const int zr_r = C::ZR_B + lane * 16;
    auto zaddr = [&](int mm) { return zr_r + ((mm << 6) & ((NG - 1) << 10)) + ((fw ? ~mm : mm) & 12); };
    const int prep_w = C::PREP_B + bi * 1024 + lane * 16;
    unsigned long long st_busy = 0;

    float vz[NZ > 0 ? NZ : 1];
    v4f vt[NT > 0 ? NT : 1];
    auto is_near = [](int i) { return BT::a_of(bi + i * NBW) + BT::b_of(bi + i * NBW) == 2; };
    auto tap_read = [&](auto i_c, auto ku_c, int cbu) {
        constexpr int I = decltype(i_c)::value, KU = decltype(ku_c)::value;
        constexpr int a = BT::a_of(bi + I * NBW), b = BT::b_of(bi + I * NBW);
        int addr = taddr[a];
        if constexpr (b > 0 && !(ABL & 4096)) addr = cbu >= b ? addr : C::RING_B + ZERO_CELL;
        vt[I] = ld4(addr + ((KU - a - b) & 7) * SLOT_B);
    };
    // the operands of step u that are in LDS one step early: z and the taps with a + b >= 3
    auto prefetch = [&](auto ku_c) {
        if constexpr (ABL & 32) return;
        if constexpr (NZ > 0 && !(ABL & 2048)) {
            const int za = zaddr(m4);
#pragma unroll
            for (int n = 0; n < NZ; ++n) vz[n] = ld1(za + (bi + n * NBW) * NG * 1024);
        }
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (([&] { if constexpr (!is_near(I)) tap_read(IC<I>{}, ku_c, cb); }()), ...);
        }(std::make_integer_sequence<int, NT>{});
    };
    prefetch(IC<0>{});
    FINC_ST_DECL();
    auto bstep = [&](auto k_c) {
        constexpr int KU = decltype(k_c)::value;                            // u % UNROLL
        constexpr int par = KU & 1;
        FINC_ST_BEGIN();
        // ---- the tap with a + b == 2 needs the pixel solved in the step before: read now
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (([&] { if constexpr (is_near(I) && !(ABL & 64)) tap_read(IC<I>{}, k_c, cb); }()), ...);
        }(std::make_integer_sequence<int, NT>{});
        FINC_SB();
        // the z-term and the older taps on two accumulators (the first starts from the folded shift), the tap read in this step on a
        // third: its round trip hides behind the others
        v4f accf[2] = {bias, (v4f){0.f, 0.f, 0.f, 0.f}}, accn = (v4f){0.f, 0.f, 0.f, 0.f};
        if constexpr (bi == 0) {                // a lane that has not started must yield exact zeros: the shift enters only behind its start
            if (__builtin_expect(ustep < P - 1 || P < 16, 0)) {
                const bool started = cb >= 0 && p < P;
                const float b0 = bias.x, b1 = bias.y, b2 = bias.z, b3 = bias.w;
                accf[0] = (v4f){started ? b0 : 0.f, started ? b1 : 0.f, started ? b2 : 0.f, started ? b3 : 0.f};
            }
        }
        int alt = 0;
#pragma unroll
        for (int n = 0; n < NZ; ++n) {
            if constexpr (ABL & (1 | 32)) { asm volatile("" ::"v"(vz[n])); continue; }
            accf[alt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fz[n], vz[n], accf[alt], 0, 0, 0);
            alt ^= 1;
        }
        auto tap_pass = [&](auto pass_c) {
            constexpr int PASS = decltype(pass_c)::value;
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (([&] {
                     constexpr bool near = is_near(I);
                     if constexpr (near == (PASS == 1) && !((ABL & 64) && PASS == 1) && !((ABL & 32) && PASS == 0)) {
                         const float t0 = vt[I].x, t1 = vt[I].y, t2 = vt[I].z, t3 = vt[I].w;
                         const float ts[4] = {t0, t1, t2, t3};
#pragma unroll
                         for (int j = 0; j < NK; ++j) {
                             if constexpr (ABL & 1) { asm volatile("" ::"v"(ts[j])); continue; }
                             if constexpr (near) accn = __builtin_amdgcn_mfma_f32_16x16x4f32(ft[I][j], ts[j], accn, 0, 0, 0);
                             else { accf[alt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ft[I][j], ts[j], accf[alt], 0, 0, 0); alt ^= 1; }
                         }
                     }
                 }()), ...);
            }(std::make_integer_sequence<int, NT>{});
        };
        tap_pass(IC<0>{});
        FINC_SB();
        FINC_ST_MID1();
        tap_pass(IC<1>{});
        if constexpr (NT > 0) {                 // (whole tuples stay reserved until their MFMAs are issued: see A's halo read)
#pragma unroll
            for (int i = 0; i < NT; ++i) asm volatile("" ::"v"(vt[i]));
        }
        // the next step's early operands are requested while this step's last MFMAs run, BEFORE its result is written: behind the
        // write they would come back after its acknowledge and the barrier would wait for them
        ++cb; if (cb == W) cb = 0;
        ++ustep;
        m4 += 4;
        prefetch(IC<(KU + 1) & 7>{});
        FINC_SB();
        const v4f cellv = (accf[0] + accf[1]) + accn;
        FINC_ST_MID2();
        st4(prep_w + par * NBW * 1024, cellv);
        FINC_ST_END();
        __syncthreads();
    };
    // iterations t = -1 .. T (u = t + 1 = 0 .. T + 1), unrolled by UNROLL and left at the exact step (see the I/O wave)
    for (int t0 = -1; t0 < T; t0 += UNROLL) {
        const bool last = [&]<int... K>(std::integer_sequence<int, K...>) {
            return ((bstep(IC<K>{}), (K & 3) == 2 && t0 + K == T) || ...);
        }(std::make_integer_sequence<int, UNROLL>{});
        if (last) break;
    }
#ifdef FINC_SPLIT_STAMP
    if (blockIdx.x == 0 && lane == 0) finc_chain_stamps[1 + bi] = st_busy;
    if constexpr (bi == 1) FINC_ST_OUT(20);
#else
    (void)st_busy;
#endif
    };   // run_b
    [&]<int... BI>(std::integer_sequence<int, BI...>) {
        (([&] {
             if (role - 1 == BI) run_b(IC<BI>{});
         }()), ...);
    }(std::make_integer_sequence<int, NBW>{});
}

// -----------------------------------------------------------------------------------------------
// Instantiations: the banks of up to 16 channels with a 2x2 or 3x3 filter
// -----------------------------------------------------------------------------------------------
typedef void (*chain_fn)(const float *, const float *, float *, int, int, int, int, int, int, unsigned, int);
struct CInst {
    int cqp, kh, kw, nbw, lds_fixed;
    chain_fn fn;
};
constexpr int near_taps(int KH, int KW)
{
    int n = 0;
    for (int a = 0; a < KH; ++a)
        for (int b = 0; b < KW; ++b) n += (a + b == 2);
    return n;
}
template <int CQP, int KH, int KW, int NBW = near_taps(KH, KW)>
constexpr CInst make_cinst()
{
    return CInst{CQP, KH, KW, NBW, LCfg<CQP, KH, KW, NBW>::lds_bytes(0), finc_chain_kernel<CQP, KH, KW, NBW>};
}

#ifdef FINC_ONLY_C3
const CInst g_cinsts[] = {make_cinst<12, 3, 3>()};
#else
const CInst g_cinsts[] = {
    make_cinst<4, 3, 3>(), make_cinst<8, 3, 3>(), make_cinst<12, 3, 3>(), make_cinst<16, 3, 3>(),
    make_cinst<4, 2, 2>(), make_cinst<8, 2, 2>(), make_cinst<12, 2, 2>(), make_cinst<16, 2, 2>(),
};
#endif

const CInst *find_cinst(int Cq, int KH, int KW)
{
    const int cqp = finc_mfma_packed_cqp(Cq, KH, KW);   // the bank is the wavefront kernel's: same padding rule
    if (cqp == 0) return nullptr;
    for (const CInst &i : g_cinsts)
        if (i.cqp == cqp && i.kh == KH && i.kw == KW) return &i;
    return nullptr;
}

int fifo_depth(int W, int P) { return W - P + 2; }

// FINC_NO_CHAIN=1 keeps these problem sets on the role-split kernel (A/B timing)
bool chain_off()
{
    static const bool off = [] { const char *e = finc_env("FINC_NO_CHAIN"); return e && e[0] == '1'; }();
    return off;
}

} // namespace

#ifdef FINC_SPLIT_STAMP
extern "C" int finc_debug_chain_stamps(unsigned long long *h) { return (int)hipMemcpyFromSymbol(h, HIP_SYMBOL(finc_chain_stamps), sizeof(finc_chain_stamps)); }
#endif

// (the caller has established what finc_split_takes establishes: problems, W % 4 == 0, P >= KH - 1, slab < 1 GiB)
bool finc_chain_takes(const FincShape &s)
{
    if (chain_off()) return false;
    const CInst *i = find_cinst(s.Cq, s.KH, s.KW);
    if (!i || s.H < 1 || s.W < 1 || s.W % 4 != 0) return false;
    const int P = s.W < 16 ? s.W : 16;
    if (P < s.KH - 1) return false;
    if ((size_t)s.Cq * s.H * s.W * 4 >= ((size_t)1 << 30)) return false;
    return (size_t)i->lds_fixed + 2 * (size_t)fifo_depth(s.W, P) * FSLOT_B <= 160 * 1024;
}

int finc_chain_info(const FincShape &s, int *waves, int *lds, int *steps)
{
    const CInst *i = find_cinst(s.Cq, s.KH, s.KW);
    if (!i) return FINC_ERR_UNSUPPORTED;
    const int P = s.W < 16 ? s.W : 16;
    const int NB = (s.H + P - 1) / P;
    *waves = 2 + i->nbw;
    *lds = i->lds_fixed + 2 * fifo_depth(s.W, P) * FSLOT_B;
    *steps = NB * s.W + P - 1;
    return FINC_OK;
}

int finc_chain_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st)
{
    const CInst *i = find_cinst(s.Cq, s.KH, s.KW);
    if (!i) return FINC_ERR_UNSUPPORTED;
    const int P = s.W < 16 ? s.W : 16;
    const int NB = (s.H + P - 1) / P;
    const int DF = fifo_depth(s.W, P);
    const size_t lds = (size_t)i->lds_fixed + 2 * (size_t)DF * FSLOT_B;
    const int T = NB * s.W + P - 1;
    // the kernel runs the iterations u = 0 .. Tr + 1; the last store -- lane P - 1's last group, solved by step T - 1 -- leaves in the
    // store phase of window NB * W / 4 + P / 4, i.e. in iteration u = T + 3 (W, P multiples of 4: T = 3 mod 4, u = 2 mod 4)
    const int Tr = T + 2;
    if (int e = finc_ensure_dynamic_lds((const void *)i->fn, lds)) return e;
    hipLaunchKernelGGL(i->fn, dim3(s.B * s.G), dim3(64 * (2 + i->nbw)), lds, st, in, (const float *)packed, out, s.G, s.Cq, s.H, s.W, P, Tr,
                       s.orient, DF);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

unsigned finc_build_flags_chain() { return FINC_BUILD_FLAGS; }
