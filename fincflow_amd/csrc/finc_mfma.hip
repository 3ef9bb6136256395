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
//   * per step the wave evaluates, with v_mfma_f32_16x16x4_f32,
//         x_new[Cq x 16px] = Linv * z  -  sum_{(a,b)!=(0,0)} (Linv * W_ab) * x[(h-a, w-b)]
//     where L = W_00 is the unit-lower-triangular corner tap.  Folding Linv
//     into the filter bank (finc_mfma_pack, fp64) removes the sequential
//     in-pixel channel substitution of the reference (the `kc < c` terms of
//     cinc_cuda_kernel_level2.cu:64-69) from the critical path.
//   * ALL filter fragments live in VGPRs for the whole kernel (108 registers
//     at Cq=24, 3x3).  The MFMA D layout (lane (q,p), reg r -> channel 4q+r,
//     pixel p) is directly a B operand of the next step once the K order is
//     permuted to match, so solved pixels never leave registers on the
//     critical path: neighbours in the same row age in place, neighbours in
//     row-a come from lane p-a by DPP row_shr:a.  Only the band hand-over
//     (row P-a of the previous band -> lanes p<a) goes through a small LDS FIFO.
//   * only the taps with a+b == 1 depend on the pixel solved in the previous
//     step; every other MFMA of step t+1 is issued while step t's result is
//     still being post-processed.
//   * HBM traffic: every lane streams its row with 16-byte loads / stores that
//     are staged through per-lane LDS rings, because lane p is (t-p) mod 4
//     into its 4-column group.
//
// The forward kernel is the same machine with the input stream in place of the
// solved pixels (no recurrence), so it shares the I/O rings, the DPP/FIFO
// neighbour exchange and the fragment packing.
#include "finc_common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

template <int CQP_, int KH_, int KW_, bool FWD_>
struct Cfg {
    static constexpr int CQP = CQP_, KH = KH_, KW = KW_;
    static constexpr bool FWD = FWD_;
    static constexpr int MT = (CQP + 15) / 16;            // 16-row output-channel tiles
    static constexpr int NKZ = CQP / 4;                   // k-steps of a streamed operand (k-slot q <-> channel 4j+q)
    static constexpr int LASTV = (CQP - 16 * (MT - 1)) / 4; // k-slots of the last D tile that hold real channels
    static constexpr bool PACK = LASTV <= 2;              // fold the last tile's 4 half-empty regs into 2
    static constexpr int NKD = PACK ? 4 * (MT - 1) + 2 : 4 * MT; // regs of a D-layout result as operand / for store
    static constexpr int NK = FWD ? NKZ : NKD;            // k-steps per neighbour tap
    static constexpr int NTAP = KH * KW;
    static constexpr int NFRAG = FWD ? NTAP * NKZ * MT : (NKZ + (NTAP - 1) * NKD) * MT;
    static constexpr int ZSLOTS = 12, XSLOTS = 8;
    static constexpr int ZRING = NKZ * ZSLOTS * 64;       // floats
    static constexpr int XRING = NKD * XSLOTS * 64;
};

// channel held by k-slot q of k-step j of a D-layout-derived operand
__host__ __device__ inline int chan_d(int MT, bool PACK, int j, int q)
{
    const int full = PACK ? 4 * (MT - 1) : 4 * MT;
    if (j < full) return 16 * (j >> 2) + 4 * q + (j & 3);
    const int jj = j - full;
    return 16 * (MT - 1) + (q < 2 ? 4 * q + 2 * jj : 4 * (q - 2) + 2 * jj + 1);
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

// D-layout accumulators -> operand/store registers (identity or pair-packing of the last tile)
template <class C>
__device__ inline void pack_d(const v4f (&acc)[C::MT], float (&xpk)[C::NKD])
{
    constexpr int full = C::PACK ? C::MT - 1 : C::MT;
#pragma unroll
    for (int mt = 0; mt < full; ++mt) {
        xpk[4 * mt + 0] = acc[mt].x;
        xpk[4 * mt + 1] = acc[mt].y;
        xpk[4 * mt + 2] = acc[mt].z;
        xpk[4 * mt + 3] = acc[mt].w;
    }
    if constexpr (C::PACK) {
        // NB: __builtin_bit_cast applied directly to an ext-vector ELEMENT (a.y, s0.x ...) silently reads
        // element 0 with this compiler; always go through scalar temporaries.
        const float ax = acc[C::MT - 1].x, ay = acc[C::MT - 1].y, az = acc[C::MT - 1].z, aw = acc[C::MT - 1].w;
        // v_permlane32_swap: new vdst = [vdst.lo32lanes, src.lo32lanes]
        const v2u s0 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, ax), __builtin_bit_cast(unsigned, ay),
                                                        false, false);
        const v2u s1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, az), __builtin_bit_cast(unsigned, aw),
                                                        false, false);
        const unsigned u0 = s0.x, u1 = s1.x;
        xpk[4 * (C::MT - 1) + 0] = __builtin_bit_cast(float, u0);
        xpk[4 * (C::MT - 1) + 1] = __builtin_bit_cast(float, u1);
    }
}

struct IoState {
    int lcol, lrow; // next group to load (canonical col may be negative = not yet)
    int scol, srow; // next group to store
    int lslot;      // z-ring slot (0,4,8) where the in-flight group lands
    int sslot;      // x-ring slot (0,4) of the group to store
};

// -----------------------------------------------------------------------------------------------
// The kernel.  grid = B*G workgroups of one wavefront.
// -----------------------------------------------------------------------------------------------
template <int CQP, int KH, int KW, bool FWD>
__global__ __launch_bounds__(64) void finc_wave_kernel(const float *__restrict__ in, const float *__restrict__ packed,
                                                       float *__restrict__ out, int G, int CQ, int H, int W, int P,
                                                       int Tend, unsigned orient)
{
    using C = Cfg<CQP, KH, KW, FWD>;
    constexpr int MT = C::MT, NKZ = C::NKZ, NKD = C::NKD, NK = C::NK, NFRAG = C::NFRAG;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *zring = lds;
    float *xring = lds + C::ZRING;
    float *fifo = xring + C::XRING;

    const int lane = threadIdx.x;
    const int q = lane >> 4, p = lane & 15;
    const int bg = blockIdx.x;
    const int g = bg % G;
    const unsigned o = finc_group_orient(orient, g);
    const bool fh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    const int HW = H * W;
    const float *ing = in + (size_t)bg * CQ * HW;
    float *outg = out + (size_t)bg * CQ * HW;
    const int D = W - P + 1;                 // FIFO depth (steps between a band's last rows and the next band's first)
    const int fifo_n = D * NK * 4 * (KH - 1);

    // ---- filter fragments -> registers -------------------------------------------------------
    float af[NFRAG];
    {
        const float *pk = packed + (size_t)g * NFRAG * 64 + lane;
#pragma unroll
        for (int f = 0; f < NFRAG; ++f) af[f] = pk[f * 64];
    }
    for (int i = lane; i < fifo_n; i += 64) fifo[i] = 0.f;

    // ---- per-lane stream state -----------------------------------------------------------------
    const int fl4 = -((p + 3) >> 2);         // floor(-p/4)
    IoState io;
    io.lcol = 4 * fl4; io.lrow = p;          // io(-12) loads group floor((-12-p)/4)+3 = floor(-p/4)
    io.scol = 4 * (fl4 - 1); io.srow = p;    // io(0) stores group floor(-p/4)-1
    io.lslot = ((4 * fl4) % 12 + 12) % 12;
    io.sslot = (4 * (fl4 - 1)) & 7;
    float zin[NKZ][4];
#pragma unroll
    for (int j = 0; j < NKZ; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) zin[j][k] = 0.f;
    bool zin_valid = false;                  // nothing in flight yet

    auto io_land = [&]() {
        if (zin_valid) {
#pragma unroll
            for (int j = 0; j < NKZ; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) zring[(j * C::ZSLOTS + io.lslot + k) * 64 + lane] = zin[j][k];
            io.lslot = io.lslot == 8 ? 0 : io.lslot + 4;
        }
    };
    auto io_issue = [&]() {
        const bool ok = io.lcol >= 0 && io.lrow < H && p < P;
        const int mrow = fh ? H - 1 - io.lrow : io.lrow;
        const int mcol = fw ? W - 4 - io.lcol : io.lcol;
        const float *src = ing + mrow * W + mcol;
#pragma unroll
        for (int j = 0; j < NKZ; ++j) {
            const int c = 4 * j + q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && c < CQ) v = *reinterpret_cast<const float4 *>(src + c * HW);
            zin[j][0] = fw ? v.w : v.x;
            zin[j][1] = fw ? v.z : v.y;
            zin[j][2] = fw ? v.y : v.z;
            zin[j][3] = fw ? v.x : v.w;
        }
        zin_valid = true;
        io.lcol += 4;
        if (io.lcol == W) { io.lcol = 0; io.lrow += P; }
    };
    auto io_store = [&]() {
        const bool ok = io.scol >= 0 && io.srow < H && p < P;
        const int mrow = fh ? H - 1 - io.srow : io.srow;
        const int mcol = fw ? W - 4 - io.scol : io.scol;
        float *dst = outg + mrow * W + mcol;
#pragma unroll
        for (int j = 0; j < NKD; ++j) {
            const int c = chan_d(MT, C::PACK, j, q);
            const float *r = xring + (j * C::XSLOTS + io.sslot) * 64 + lane;
            float4 v;
            v.x = r[fw ? 192 : 0];
            v.y = r[fw ? 128 : 64];
            v.z = r[fw ? 64 : 128];
            v.w = r[fw ? 0 : 192];
            if (ok && c < CQ) *reinterpret_cast<float4 *>(dst + c * HW) = v;
        }
        io.sslot ^= 4;
        io.scol += 4;
        if (io.scol == W) { io.scol = 0; io.srow += P; }
    };

    // position bookkeeping: at step t this lane is at position u = t - p of its row chain
    // (inverse starts at t = -1 so that position 0 of lane 0 gets its z-term; forward starts at t = 0)
    int cn = -p;                              // inverse: column of position u+1 (negative: not started)
    int zs = ((-p) % 12 + 12) % 12;           // inverse: z-ring slot of position u+1
    int xs = FWD ? ((-p) & 7) : ((-1 - p) & 7); // x-ring slot of position u
    int fslot = 0;                            // FIFO slot written this step

    // neighbour operands.  R[a][b] = operand of tap (a,b) for the current step.
    float R[KH][KW][NK];
    float DL[KH][KH][NK];                     // DL[a][k]: row_shr:a copies waiting k+1 more steps (a >= 2; a >= 1 fwd)
#pragma unroll
    for (int a = 0; a < KH; ++a) {
#pragma unroll
        for (int b = 0; b < KW; ++b)
#pragma unroll
            for (int j = 0; j < NK; ++j) R[a][b][j] = 0.f;
#pragma unroll
        for (int k = 0; k < KH; ++k)
#pragma unroll
            for (int j = 0; j < NK; ++j) DL[a][k][j] = 0.f;
    }
    v4f acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f){0.f, 0.f, 0.f, 0.f};

    // FIFO: lanes P-(KH-1)..P-1 push their operand regs; lanes p < a pop lane P-a+p of D-1 steps ago
    const int push_l = p - (P - (KH - 1));
    const bool do_push = KH > 1 && push_l >= 0 && p < P;
    auto fifo_push = [&](const float (&v)[NK]) {
        if (do_push) {
#pragma unroll
            for (int j = 0; j < NK; ++j) fifo[((fslot * NK + j) * 4 + q) * (KH - 1) + push_l] = v[j];
        }
    };
    auto fifo_pop = [&](int a, float (&v)[NK]) {
        const int ps = fslot + 1 == D ? 0 : fslot + 1;
        if (p < a) {
#pragma unroll
            for (int j = 0; j < NK; ++j) v[j] = fifo[((ps * NK + j) * 4 + q) * (KH - 1) + (KH - 1 - a + p)];
        } else {
#pragma unroll
            for (int j = 0; j < NK; ++j) v[j] = 0.f;
        }
    };

    __syncthreads(); // single wave: orders the FIFO zero-fill before use

    if constexpr (!FWD) {
        // =========================== inverse ===========================
        constexpr int FZ = 0;                          // z-term fragments: (j*MT + mt)
        constexpr int FT = NKZ * MT;                   // tap fragments: FT + (((a*KW+b)-1)*NK + j)*MT + mt
        io_issue(); io_land(); io_issue(); io_land(); io_issue(); // virtual t = -12, -8, -4
        for (int t = -1; t < Tend; ++t) {
            if ((t & 3) == 0) { io_land(); io_issue(); io_store(); }

            // (1) z of the NEXT position
            float zv[NKZ];
#pragma unroll
            for (int j = 0; j < NKZ; ++j) {
                float v = zring[(j * C::ZSLOTS + zs) * 64 + lane];
                zv[j] = cn >= 0 ? v : 0.f;
            }
            const bool wrapn = cn == 0;                // the next position starts a row

            // (2) phase A: the two taps that need the pixel solved last step
#pragma unroll
            for (int j = 0; j < NK; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if constexpr (KW > 1)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[FT + ((0 * KW + 1 - 1) * NK + j) * MT + mt],
                                                                       R[0][1][j], acc[mt], 0, 0, 0);
                    if constexpr (KH > 1)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[FT + ((1 * KW + 0 - 1) * NK + j) * MT + mt],
                                                                       R[1][0][j], acc[mt], 0, 0, 0);
                }

            // (3) age the operands that do not depend on this step's result
#pragma unroll
            for (int a = 0; a < KH; ++a) {
#pragma unroll
                for (int b = KW - 1; b >= 1; --b) {
                    if (a + b >= 2) {
#pragma unroll
                        for (int j = 0; j < NK; ++j) R[a][b][j] = wrapn ? 0.f : R[a][b - 1][j];
                    }
                }
                if (a >= 2) {
#pragma unroll
                    for (int j = 0; j < NK; ++j) R[a][0][j] = DL[a][a - 2][j];
                }
            }

            // (4) phase B: everything of the next step that is already known
            v4f accn[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) accn[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NKZ; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    accn[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[FZ + j * MT + mt], zv[j], accn[mt], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < KH; ++a)
#pragma unroll
                for (int b = 0; b < KW; ++b) {
                    if (a + b >= 2) {
#pragma unroll
                        for (int j = 0; j < NK; ++j)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                accn[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    af[FT + ((a * KW + b - 1) * NK + j) * MT + mt], R[a][b][j], accn[mt], 0, 0, 0);
                    }
                }

            // (5) post-process the pixel just solved
            float xpk[NKD];
            pack_d<C>(acc, xpk);
#pragma unroll
            for (int j = 0; j < NKD; ++j) xring[(j * C::XSLOTS + xs) * 64 + lane] = xpk[j];
            if constexpr (KH > 1) {
                fifo_push(xpk);
                float fv[NK];
                fifo_pop(1, fv);
                ShiftOp<1>::apply(R[1][0], fv, xpk);
                if constexpr (KH > 2) {
#pragma unroll
                    for (int a = 2; a < KH; ++a) {
#pragma unroll
                        for (int k = a - 2; k >= 1; --k)
#pragma unroll
                            for (int j = 0; j < NK; ++j) DL[a][k][j] = DL[a][k - 1][j];
                    }
                    if constexpr (KH > 2) { fifo_pop(2, fv); ShiftOp<2>::apply(DL[2][0], fv, xpk); }
                    if constexpr (KH > 3) { fifo_pop(3, fv); ShiftOp<3>::apply(DL[3][0], fv, xpk); }
                    if constexpr (KH > 4) { fifo_pop(4, fv); ShiftOp<4>::apply(DL[4][0], fv, xpk); }
                    if constexpr (KH > 5) { fifo_pop(5, fv); ShiftOp<5>::apply(DL[5][0], fv, xpk); }
                    if constexpr (KH > 6) { fifo_pop(6, fv); ShiftOp<6>::apply(DL[6][0], fv, xpk); }
                }
            }
            if constexpr (KW > 1) {
#pragma unroll
                for (int j = 0; j < NK; ++j) R[0][1][j] = wrapn ? 0.f : xpk[j];
            }

            // (6) advance
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = accn[mt];
            ++cn; if (cn == W) cn = 0;
            ++zs; if (zs == 12) zs = 0;
            xs = (xs + 1) & 7;
            ++fslot; if (fslot == D) fslot = 0;
        }
        io_store(); // flush: t == Tend
    } else {
        // =========================== forward ===========================
        // fragments: ((a*KW+b)*NK + j)*MT + mt
        int cc = -p;                                  // column of position u
        int zc = ((-p) % 12 + 12) % 12;
        io_issue(); io_land(); io_issue(); io_land(); io_issue();
        for (int t = 0; t < Tend; ++t) {
            if ((t & 3) == 0) { io_land(); io_issue(); io_store(); }
            const bool wrap = cc == 0;
            // age first (uses the previous step's operands), then insert the new column
#pragma unroll
            for (int a = 0; a < KH; ++a)
#pragma unroll
                for (int b = KW - 1; b >= 1; --b)
#pragma unroll
                    for (int j = 0; j < NK; ++j) R[a][b][j] = wrap ? 0.f : R[a][b - 1][j];
#pragma unroll
            for (int j = 0; j < NKZ; ++j) {
                float v = zring[(j * C::ZSLOTS + zc) * 64 + lane];
                R[0][0][j] = cc >= 0 ? v : 0.f;
            }
            if constexpr (KH > 1) {
                // R[a][0](t) = row_shr:a of the stream a steps ago
#pragma unroll
                for (int a = 1; a < KH; ++a) {
#pragma unroll
                    for (int j = 0; j < NK; ++j) R[a][0][j] = DL[a][a - 1][j];
#pragma unroll
                    for (int k = a - 1; k >= 1; --k)
#pragma unroll
                        for (int j = 0; j < NK; ++j) DL[a][k][j] = DL[a][k - 1][j];
                }
                fifo_push(R[0][0]);
                float fv[NK];
                fifo_pop(1, fv); ShiftOp<1>::apply(DL[1][0], fv, R[0][0]);
                if constexpr (KH > 2) { fifo_pop(2, fv); ShiftOp<2>::apply(DL[2][0], fv, R[0][0]); }
                if constexpr (KH > 3) { fifo_pop(3, fv); ShiftOp<3>::apply(DL[3][0], fv, R[0][0]); }
                if constexpr (KH > 4) { fifo_pop(4, fv); ShiftOp<4>::apply(DL[4][0], fv, R[0][0]); }
                if constexpr (KH > 5) { fifo_pop(5, fv); ShiftOp<5>::apply(DL[5][0], fv, R[0][0]); }
                if constexpr (KH > 6) { fifo_pop(6, fv); ShiftOp<6>::apply(DL[6][0], fv, R[0][0]); }
            }
            v4f ac[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) ac[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < KH; ++a)
#pragma unroll
                for (int b = 0; b < KW; ++b)
#pragma unroll
                    for (int j = 0; j < NK; ++j)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            ac[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[((a * KW + b) * NK + j) * MT + mt],
                                                                          R[a][b][j], ac[mt], 0, 0, 0);
            float xpk[NKD];
            pack_d<C>(ac, xpk);
#pragma unroll
            for (int j = 0; j < NKD; ++j) xring[(j * C::XSLOTS + xs) * 64 + lane] = xpk[j];
            ++cc; if (cc == W) cc = 0;
            ++zc; if (zc == 12) zc = 0;
            xs = (xs + 1) & 7;
            ++fslot; if (fslot == D) fslot = 0;
        }
        io_store();
    }
}

// -----------------------------------------------------------------------------------------------
// Fragment packing (fp64 math, one workgroup per group).
//   inverse: Linv = L^-1 by forward substitution; z-term fragment = Linv; tap (a,b) fragment
//            = -(Linv * Wc[:,:,KH-1-a,KW-1-b]).
//   forward: tap (a,b) fragment = Wc[:,:,KH-1-a,KW-1-b].
// Lane (q,i) of fragment (tap, j, mt) holds row 16mt+i, column = channel of k-slot q of k-step j.
// -----------------------------------------------------------------------------------------------
__global__ void pack_kernel(const float *__restrict__ wc, float *__restrict__ packed, int Cq, int KH, int KW, int MT,
                            int NKZ, int NKD, int pack_last, int forward, int nfrag)
{
    extern __shared__ __attribute__((aligned(16))) double sm[]; // Linv [Cq][Cq]
    const int g = blockIdx.x;
    const float *wg = wc + (size_t)g * Cq * Cq * KH * KW;
    const int KK = KH * KW;
    double *Linv = sm;
    if (!forward) {
        // column j of Linv: solve L y = e_j  (L unit lower triangular)
        for (int j = threadIdx.x; j < Cq; j += blockDim.x) {
            for (int r = 0; r < Cq; ++r) {
                double s = (r == j) ? 1.0 : 0.0;
                for (int k = j; k < r; ++k)
                    s -= (double)wg[((size_t)r * Cq + k) * KK + (KK - 1)] * Linv[k * Cq + j];
                Linv[r * Cq + j] = (r < j) ? 0.0 : s;
            }
        }
        __syncthreads();
    }
    const int NK = forward ? NKZ : NKD;
    for (int e = threadIdx.x; e < nfrag * 64; e += blockDim.x) {
        const int lane = e & 63, f = e >> 6;
        const int q = lane >> 4, i = lane & 15;
        int tap, j, mt;
        bool zterm = false;
        if (forward) {
            mt = f % MT; j = (f / MT) % NK; tap = f / (MT * NK);
        } else if (f < NKZ * MT) {
            zterm = true; mt = f % MT; j = f / MT; tap = 0;
        } else {
            const int ff = f - NKZ * MT;
            mt = ff % MT; j = (ff / MT) % NK; tap = 1 + ff / (MT * NK);
        }
        const int row = 16 * mt + i;
        const int col = (forward || zterm) ? 4 * j + q : chan_d(MT, pack_last != 0, j, q);
        double v = 0.0;
        if (row < Cq && col < Cq) {
            const int a = tap / KW, b = tap % KW;
            const int widx = (KH - 1 - a) * KW + (KW - 1 - b);
            if (forward) {
                v = (double)wg[((size_t)row * Cq + col) * KK + widx];
            } else if (zterm) {
                v = Linv[row * Cq + col];
            } else {
                double s = 0.0;
                for (int k = 0; k <= row; ++k) s += Linv[row * Cq + k] * (double)wg[((size_t)k * Cq + col) * KK + widx];
                v = -s;
            }
        }
        packed[((size_t)g * nfrag + f) * 64 + lane] = (float)v;
    }
}

// -----------------------------------------------------------------------------------------------
// Instantiation table
// -----------------------------------------------------------------------------------------------
typedef void (*wave_fn)(const float *, const float *, float *, int, int, int, int, int, int, unsigned);

struct Inst {
    int cqp, kh, kw;
    bool fwd;
    wave_fn fn;
    int nkz, nkd, nk, mt, nfrag, pack;
};

template <int CQP, int KH, int KW, bool FWD>
constexpr Inst make_inst()
{
    using C = Cfg<CQP, KH, KW, FWD>;
    return Inst{CQP, KH, KW, FWD, finc_wave_kernel<CQP, KH, KW, FWD>, C::NKZ, C::NKD, C::NK, C::MT, C::NFRAG,
                C::PACK ? 1 : 0};
}

#define FINC_BOTH(cqp, kh, kw) make_inst<cqp, kh, kw, false>(), make_inst<cqp, kh, kw, true>()

const Inst g_insts[] = {
    FINC_BOTH(4, 3, 3),  FINC_BOTH(8, 3, 3),  FINC_BOTH(12, 3, 3), FINC_BOTH(16, 3, 3),
    FINC_BOTH(24, 3, 3), FINC_BOTH(32, 3, 3),
    FINC_BOTH(4, 2, 2),  FINC_BOTH(16, 2, 2),
    FINC_BOTH(4, 5, 5),  FINC_BOTH(16, 5, 5),
    FINC_BOTH(4, 3, 5),
};

const Inst *find_inst(int Cq, int KH, int KW, bool forward)
{
    const int cqp = (Cq + 3) / 4 * 4;
    for (const Inst &i : g_insts)
        if (i.cqp == cqp && i.kh == KH && i.kw == KW && i.fwd == forward) return &i;
    return nullptr;
}

size_t lds_bytes(const Inst &i, int W, int P)
{
    const size_t D = (size_t)(W - P + 1);
    return sizeof(float) * ((size_t)i.nkz * 12 * 64 + (size_t)i.nkd * 8 * 64 + D * i.nk * 4 * (i.kh - 1));
}

} // namespace

bool finc_mfma_supported(int Cq, int H, int W, int KH, int KW, bool forward)
{
    const Inst *i = find_inst(Cq, KH, KW, forward);
    if (!i) return false;
    if (W % 4 != 0 || W < 4 || H < 1) return false;
    const int P = W < 16 ? W : 16;
    if (P < KH - 1) return false;
    if (lds_bytes(*i, W, P) > 160 * 1024) return false;
    return true;
}

size_t finc_mfma_packed_bytes(int G, int Cq, int KH, int KW)
{
    const Inst *a = find_inst(Cq, KH, KW, false);
    const Inst *b = find_inst(Cq, KH, KW, true);
    size_t n = 0;
    if (a) n = (size_t)a->nfrag;
    if (b && (size_t)b->nfrag > n) n = (size_t)b->nfrag;
    return n * 64 * sizeof(float) * (size_t)G;
}

int finc_mfma_pack(const float *wc, void *packed, int G, int Cq, int KH, int KW, bool forward, hipStream_t st)
{
    const Inst *i = find_inst(Cq, KH, KW, forward);
    if (!i) return FINC_ERR_UNSUPPORTED;
    size_t sm = forward ? 16 : sizeof(double) * Cq * Cq;
    hipLaunchKernelGGL(pack_kernel, dim3(G), dim3(256), sm, st, wc, (float *)packed, Cq, KH, KW, i->mt, i->nkz,
                       i->nkd, i->pack, forward ? 1 : 0, i->nfrag);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

int finc_mfma_launch(const float *in, const void *packed, float *out, const FincShape &s, bool forward, hipStream_t st)
{
    const Inst *i = find_inst(s.Cq, s.KH, s.KW, forward);
    if (!i || !finc_mfma_supported(s.Cq, s.H, s.W, s.KH, s.KW, forward)) return FINC_ERR_UNSUPPORTED;
    if (((uintptr_t)in & 15u) || ((uintptr_t)out & 15u)) return FINC_ERR_ALIGNMENT;
    const int P = s.W < 16 ? s.W : 16;
    const int NB = (s.H + P - 1) / P;
    const int Tend = (NB * s.W + P - 1 + 3) / 4 * 4;
    const size_t lds = lds_bytes(*i, s.W, P);
    static thread_local const void *attr_done[64];
    static thread_local int n_attr = 0;
    if (lds > 48 * 1024) {
        bool seen = false;
        for (int k = 0; k < n_attr; ++k) seen |= attr_done[k] == (const void *)i->fn;
        if (!seen) {
            FINC_HIP_TRY(hipFuncSetAttribute((const void *)i->fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (n_attr < 64) attr_done[n_attr++] = (const void *)i->fn;
        }
    }
    hipLaunchKernelGGL(i->fn, dim3(s.B * s.G), dim3(64), lds, st, in, (const float *)packed, out, s.G, s.Cq, s.H, s.W,
                       P, Tend, s.orient);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}
