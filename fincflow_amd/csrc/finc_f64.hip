// Double precision on the matrix cores (gfx950 / CDNA4 only): the reference op dispatches over float AND double
// (AT_DISPATCH_FLOATING_TYPES, cinc_cuda_kernel_level2.cu:117) and its CPU solver computes in fp64
// (solve_parallel_mc.pyx:77-126).  finc_inverse_f64 / finc_forward_f64 (finc_generic.hip) keep the reference's term order and
// are bit-exact with that solver; these kernels are the fast form behind FINC_ALGO_AUTO: the same visitation (anti-diagonal
// order band by band, cinc_cuda_kernel_level2.cu:49-56,98-111), the in-pixel substitution folded into the bank (Linv = L^-1, as
// in finc_mfma.hip), every product on v_mfma_f64_16x16x4_f64.
//
// One wavefront owns one (image, group) problem; no barrier, no inter-wave traffic.  Lanes are 4 k-slots q x 16 rows p of a band;
// lane p trails lane p-1 by one step.  The f64 MFMA's result layout (lane (q, n), register r = row q + 4r of column n) is, for
// natural channel order, exactly the B operand of k-step r -- so a solved pixel goes to a ring of cells in LDS as it leaves the
// accumulators and comes back as operands (lane p - a of the slot of step t - a - b) without any shuffling; the rows above a band
// wait in a FIFO in LDS; a wave's LDS operations execute in order, so nothing needs a fence.  The bank -- 9 * NK * MT fragments of
// one double per lane: 216 registers at Cq = 24 -- stays in registers for the whole kernel (one wave per SIMD: 512 registers).
// A step is 9 * NK * MT MFMAs of 64 cycles (Cq = 24: 108 + 12 = 7,680 cycles), so everything else -- 48 LDS reads, the dword-pair
// loads and stores of a lane's own pixel -- runs in their shadow.
#include "finc_common.h"

#include <type_traits>
#include <utility>

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr unsigned OFF_INVALID = 0x80000000u;     // voffset beyond any slab: buffer loads return 0, stores are dropped
constexpr unsigned OFF_BAD_CHANNEL = 0x40000000u;

constexpr int XS = 6;             // x ring slots: taps reach back KH + KW - 2 <= 4 steps
constexpr int WPW = 4;            // inverse: problems (= waves) per workgroup

template <int CQP, int KH, int KW>
struct DCfg {
    static_assert(CQP % 4 == 0 && KH + KW - 2 <= XS - 2, "bank");
    static constexpr int MT = (CQP + 15) / 16, NK = CQP / 4, NTAP = KH * KW;
    static constexpr int NFRAG = NTAP * NK * MT;                 // z-term first, then the taps (a, b) != (0, 0) row-major
    static constexpr int CELL = NK * 8;                           // bytes of a pixel: NK doubles per lane
    static constexpr int HROWS = KH - 1;                          // rows handed over between bands
    // LDS of a wave: ring [slot][lane] | one zero cell | FIFO [slot][q][row]
    static constexpr int RING_B = 0, ZERO_B = XS * 64 * CELL, FIFO_B = ZERO_B + CELL;
    static constexpr int FSLOT = 4 * (HROWS > 0 ? HROWS : 1) * CELL;
    static constexpr int lds_bytes(int DF) { return FIFO_B + DF * FSLOT; }
};

// -----------------------------------------------------------------------------------------------
// Fragment packing in fp64 (one workgroup per group): fragment (tap, j, mt), lane (q, i) = M_tap[16 mt + i][4 j + q], M_z = Linv,
// M_(a,b) = -(Linv * Wc[:, :, KH-1-a, KW-1-b]); forward: M_(a,b) = Wc[:, :, KH-1-a, KW-1-b] for all nine taps, (0,0) in the z slot.
// -----------------------------------------------------------------------------------------------
__global__ void pack_f64_kernel(const double *__restrict__ wc, double *__restrict__ packed, int Cq, int KH, int KW, int MT, int NK, int forward)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];   // Linv [Cq][Cq]
    const int g = blockIdx.x;
    const double *wg = wc + (size_t)g * Cq * Cq * KH * KW;
    const int KK = KH * KW, nfrag = KK * NK * MT;
    double *Linv = sm;
    if (!forward) {
        for (int j = threadIdx.x; j < Cq; j += blockDim.x)
            for (int r = 0; r < Cq; ++r) {
                double s = (r == j) ? 1.0 : 0.0;
                for (int k = j; k < r; ++k) s -= wg[((size_t)r * Cq + k) * KK + (KK - 1)] * Linv[k * Cq + j];
                Linv[r * Cq + j] = (r < j) ? 0.0 : s;
            }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < nfrag * 64; e += blockDim.x) {
        const int lane = e & 63, f = e >> 6;
        const int q = lane >> 4, i = lane & 15;
        const int mt = f % MT, j = (f / MT) % NK, tap = f / (MT * NK);       // tap 0 = the z-term / the (0,0) tap
        const int row = 16 * mt + i, col = 4 * j + q;
        const int a = tap / KW, b = tap % KW;
        const int widx = (KH - 1 - a) * KW + (KW - 1 - b);
        double v = 0.0;
        if (row < Cq && col < Cq) {
            if (forward) v = wg[((size_t)row * Cq + col) * KK + widx];
            else if (tap == 0) v = Linv[row * Cq + col];
            else {
                double s = 0.0;
                for (int k = 0; k <= row; ++k) s += Linv[row * Cq + k] * wg[((size_t)k * Cq + col) * KK + widx];
                v = -s;
            }
        }
        packed[((size_t)g * nfrag + f) * 64 + lane] = v;
    }
}

// -----------------------------------------------------------------------------------------------
// inverse: grid = B*G one-wave workgroups
// -----------------------------------------------------------------------------------------------
template <int CQP, int KH, int KW>
__global__ __launch_bounds__(64 * WPW, 1) void finc_f64_inverse_kernel(const double *__restrict__ in, const double *__restrict__ packed,
                                                                      double *__restrict__ out, int G, int CQ, int H, int W, int P, int T,
                                                                      unsigned orient, int DF, int nprob, int wpw)
{
    using C = DCfg<CQP, KH, KW>;
    constexpr int MT = C::MT, NK = C::NK, CELL = C::CELL, HROWS = C::HROWS;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    // wpw (<= WPW, as many as the LDS holds) independent problems per workgroup, one wave each: the waves of ONE workgroup go to the four SIMDs in turn, one-wave
    // workgroups do not (two of them on one SIMD halve each other: 2.9 against 2.1 ms at c3)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bg = (int)blockIdx.x * wpw + wave;
    if (bg >= nprob) return;
    double *const ldsd = lds_all + (size_t)wave * (C::lds_bytes(DF) / 8);
    char *const ldsb = reinterpret_cast<char *>(ldsd);
    const int lane = threadIdx.x & 63, q = lane >> 4, p = lane & 15;
    const int g = bg % G;
    const unsigned o = finc_group_orient(orient, g);
    const bool fh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 8u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);

    for (int i = lane; i < C::lds_bytes(DF) / 8; i += 64) ldsd[i] = 0.0;      // (the wave's own part: no barrier anywhere)

    // the bank: registers for the whole kernel
    double fr[C::NTAP][NK][MT];
    const double *pk = packed + (size_t)g * C::NFRAG * 64 + lane;
#pragma unroll
    for (int tp = 0; tp < C::NTAP; ++tp)
#pragma unroll
        for (int j = 0; j < NK; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fr[tp][j][mt] = pk[(size_t)((tp * NK + j) * MT + mt) * 64];

    auto ld = [&](int byte_off) { return *reinterpret_cast<const double *>(ldsb + byte_off); };
    auto st = [&](int byte_off, double v) { *reinterpret_cast<double *>(ldsb + byte_off) = v; };

    // per-lane constants
    unsigned chan_off[NK];                       // byte offset of channel 4j + q in the slab, or the out-of-range mark
#pragma unroll
    for (int j = 0; j < NK; ++j) chan_off[j] = (4 * j + q) < CQ ? (unsigned)((4 * j + q) * HW * 8) : OFF_BAD_CHANNEL;
    int taddr[KH];                               // operand cell of tap row a: lane p - a of the ring slot / the FIFO for the lanes p < a
#pragma unroll
    for (int a = 0; a < KH; ++a) taddr[a] = C::RING_B + (lane - a) * CELL;
    const bool pusher = HROWS > 0 && p >= P - HROWS && p < P;
    const int push_cell = C::FIFO_B + (q * HROWS + (P - 1 - p)) * CELL;      // row a' = P - p above the next band

    // the lane's walk: position n = t - p, column c = n mod W, band n / W
    int c = -p, row = p;                         // column (negative: not started), image row
    auto pix_off = [&](int r, int cc) { return (unsigned)(((fh ? H - 1 - r : r) * W + (fw ? W - 1 - cc : cc)) * 8); };
    // z of the next step is requested one step ahead
    double zn[NK];
    auto request = [&](int cc, int rr) {
        const bool ok = cc >= 0 && rr < H && p < P;
        const unsigned base = ok ? pix_off(rr, cc) : OFF_INVALID;
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            const v2u w2 = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(rin, base + chan_off[j], 0, 0));
            zn[j] = __builtin_bit_cast(double, w2);
        }
    };
    request(c, row);
    const bool W4 = (W & 3) == 0;                // rows of whole 32-byte groups
    int slot = 0;                                // t mod XS
    int fpush = 0;                               // t mod DF
    for (int t = 0; t < T; ++t) {
        double z[NK];
#pragma unroll
        for (int j = 0; j < NK; ++j) z[j] = zn[j];
        const int cn = (c + 1 == W) ? 0 : c + 1, rn = (c + 1 == W) ? row + P : row;
        request(cn, rn);
        // ---- z-term, then the taps by falling age (a + b): the two that need the pixel of the step before come last
        v4d acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < NK; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (16 * mt + 15 < 4 * j) continue;                          // Linv is lower triangular
                acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[0][j][mt], z[j], acc[mt], 0, 0, 0);
            }
#pragma unroll
        for (int age = KH + KW - 2; age >= 1; --age)
#pragma unroll
            for (int a = 0; a < KH; ++a) {
                const int b = age - a;
                if (b < 0 || b >= KW) continue;
                // S_a(t - age): ring slot (t - age) mod XS at lane p - a; the lanes p < a read the band above from the FIFO (the push
                // of step t - age - (W - P)); a missing column (c - b < 0) reads the zero cell
                int sl = slot - age; if (sl < 0) sl += XS;
                int fs = fpush - age - (W - P); fs %= DF; if (fs < 0) fs += DF;
                int addr = taddr[a] + sl * (64 * CELL);
                if (a > 0) addr = p >= a ? addr : C::FIFO_B + fs * C::FSLOT + (q * HROWS + (a - 1 - p)) * CELL;
                if (b > 0) addr = c >= b ? addr : C::ZERO_B;
                double x[NK];
#pragma unroll
                for (int j = 0; j < NK; ++j) x[j] = ld(addr + j * 8);
#pragma unroll
                for (int j = 0; j < NK; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[a * KW + b][j][mt], x[j], acc[mt], 0, 0, 0);
            }
        // ---- the solved pixel: register r of tile mt, lane row q = channel 16 mt + 4 r + q = k-slot q of k-step 4 mt + r
        const bool started = c >= 0 && p < P;
        double xs[NK];
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            const v4d &am = acc[j >> 2];
            const double v = (j & 3) == 0 ? am.x : (j & 3) == 1 ? am.y : (j & 3) == 2 ? am.z : am.w;
            xs[j] = started ? v : 0.0;
        }
        const int cell = C::RING_B + slot * (64 * CELL) + lane * CELL;
#pragma unroll
        for (int j = 0; j < NK; ++j) st(cell + j * 8, xs[j]);
        if (pusher) {
#pragma unroll
            for (int j = 0; j < NK; ++j) st(push_cell + fpush * C::FSLOT + j * 8, xs[j]);
        }
        if (W4) {
            // 32-byte pieces: a lane stores a group of four canonical columns when it has solved the last one -- the three before sit
            // in the ring's last slots at its own lane.  (8-byte stores, 100 M of them at c3, bound the kernel at 2.9 ms: every wave
            // keeps 384 partially written lines open, more than the L2s hold.)
            const bool ok = started && row < H && (c & 3) == 3;
            const unsigned base = ok ? pix_off(row, fw ? c : c - 3) : OFF_INVALID;
            int s1 = slot - 1, s2 = slot - 2, s3 = slot - 3;
            s1 = s1 < 0 ? s1 + XS : s1; s2 = s2 < 0 ? s2 + XS : s2; s3 = s3 < 0 ? s3 + XS : s3;
            const int own = C::RING_B + lane * CELL;
#pragma unroll
            for (int j = 0; j < NK; ++j) {
                const double x3 = ld(own + s3 * (64 * CELL) + j * 8), x2 = ld(own + s2 * (64 * CELL) + j * 8), x1 = ld(own + s1 * (64 * CELL) + j * 8);
                const double lo0 = fw ? xs[j] : x3, lo1 = fw ? x1 : x2, hi0 = fw ? x2 : x1, hi1 = fw ? x3 : xs[j];
                const v2u a0 = __builtin_bit_cast(v2u, lo0), a1 = __builtin_bit_cast(v2u, lo1), b0 = __builtin_bit_cast(v2u, hi0), b1 = __builtin_bit_cast(v2u, hi1);
                // (s_nop 1: a store of more than 8 bytes reads its data registers a few cycles after issue; scripts/check_store_hazard.py
                // wants two wait states before anything is written to them -- here the allocator reuses them for the next LDS read)
                __builtin_amdgcn_raw_buffer_store_b128((v4u){a0.x, a0.y, a1.x, a1.y}, rout, base + chan_off[j], 0, 0);
                asm volatile("s_nop 1");
                __builtin_amdgcn_raw_buffer_store_b128((v4u){b0.x, b0.y, b1.x, b1.y}, rout, base + chan_off[j] + 16, 0, 0);
                asm volatile("s_nop 1");
            }
        } else {
            const bool ok = started && row < H;
            const unsigned base = ok ? pix_off(row, c) : OFF_INVALID;
#pragma unroll
            for (int j = 0; j < NK; ++j)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, xs[j]), rout, base + chan_off[j], 0, 0);
        }
        c = cn; row = rn;
        ++slot; if (slot == XS) slot = 0;
        ++fpush; if (fpush == DF) fpush = 0;
    }
}

// -----------------------------------------------------------------------------------------------
// forward: one wave walks a strip of 16 canonical columns of one slab from the top; grid = (strips, B*G)
// -----------------------------------------------------------------------------------------------
template <int CQP, int KH, int KW>
__global__ __launch_bounds__(64) void finc_f64_forward_kernel(const double *__restrict__ in, const double *__restrict__ packed,
                                                              double *__restrict__ out, int G, int CQ, int H, int W, unsigned orient)
{
    using C = DCfg<CQP, KH, KW>;
    constexpr int MT = C::MT, NK = C::NK;
    const int lane = threadIdx.x, q = lane >> 4, p = lane & 15;
    const int bg = (int)blockIdx.y, g = bg % G;
    const int w0 = (int)blockIdx.x * 16;
    const unsigned o = finc_group_orient(orient, g);
    const bool fh = (o & FINC_FLIP_H) != 0, fw = (o & FINC_FLIP_W) != 0;
    const int HW = H * W;
    const unsigned slab_bytes = (unsigned)CQ * (unsigned)HW * 8u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)bg * CQ * HW), 0, (int)slab_bytes, 0x00020000);
    double fr[C::NTAP][NK][MT];
    const double *pk = packed + (size_t)g * C::NFRAG * 64 + lane;
#pragma unroll
    for (int tp = 0; tp < C::NTAP; ++tp)
#pragma unroll
        for (int j = 0; j < NK; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fr[tp][j][mt] = pk[(size_t)((tp * NK + j) * MT + mt) * 64];
    unsigned chan_off[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) chan_off[j] = (4 * j + q) < CQ ? (unsigned)((4 * j + q) * HW * 8) : OFF_BAD_CHANNEL;
    auto pix_off = [&](int r, int cc) { return (unsigned)(((fh ? H - 1 - r : r) * W + (fw ? W - 1 - cc : cc)) * 8); };
    const int col = w0 + p;
    for (int h = 0; h < H; ++h) {
        v4d acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int a = 0; a < KH; ++a)
#pragma unroll
            for (int b = 0; b < KW; ++b) {
                // x[i, h - a, col - b]: zero outside the image (the padding of layers/conv.py:41-55)
                const bool ok = h - a >= 0 && col - b >= 0 && col - b < W;
                const unsigned base = ok ? pix_off(h - a, col - b) : OFF_INVALID;
                double x[NK];
#pragma unroll
                for (int j = 0; j < NK; ++j) x[j] = __builtin_bit_cast(double, __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(rin, base + chan_off[j], 0, 0)));
#pragma unroll
                for (int j = 0; j < NK; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[a * KW + b][j][mt], x[j], acc[mt], 0, 0, 0);
            }
        const unsigned base = col < W ? pix_off(h, col) : OFF_INVALID;
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            const v4d &am = acc[j >> 2];
            const double v = (j & 3) == 0 ? am.x : (j & 3) == 1 ? am.y : (j & 3) == 2 ? am.z : am.w;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), rout, base + chan_off[j], 0, 0);
        }
    }
}

typedef void (*f64inv_fn)(const double *, const double *, double *, int, int, int, int, int, int, unsigned, int, int, int);
typedef void (*f64fwd_fn)(const double *, const double *, double *, int, int, int, int, unsigned);
struct DInst {
    int cqp, kh, kw, nfrag, lds_fixed, fslot;
    f64inv_fn inv;
    f64fwd_fn fwd;
};
template <int CQP, int KH, int KW>
constexpr DInst make_dinst()
{
    return DInst{CQP, KH, KW, DCfg<CQP, KH, KW>::NFRAG, DCfg<CQP, KH, KW>::lds_bytes(0), DCfg<CQP, KH, KW>::FSLOT,
                 finc_f64_inverse_kernel<CQP, KH, KW>, finc_f64_forward_kernel<CQP, KH, KW>};
}
const DInst g_dinsts[] = {
    make_dinst<4, 3, 3>(), make_dinst<8, 3, 3>(), make_dinst<12, 3, 3>(), make_dinst<16, 3, 3>(), make_dinst<20, 3, 3>(), make_dinst<24, 3, 3>(),
    make_dinst<4, 2, 2>(), make_dinst<8, 2, 2>(), make_dinst<12, 2, 2>(), make_dinst<16, 2, 2>(), make_dinst<24, 2, 2>(), make_dinst<32, 2, 2>(),
};
const DInst *find_dinst(int Cq, int KH, int KW)
{
    const int cqp = (Cq + 3) / 4 * 4;
    for (const DInst &i : g_dinsts)
        if (i.cqp == cqp && i.kh == KH && i.kw == KW) return &i;
    return nullptr;
}
int f64_fifo_depth(int W, int P, int KH, int KW) { return W - P + KH + KW - 2; }

} // namespace

bool finc_f64_supported(const FincShape &s)
{
    const DInst *i = find_dinst(s.Cq, s.KH, s.KW);
    if (!i || s.H < 1 || s.W < 1) return false;
    const int P = s.W < 16 ? s.W : 16;
    if (P < s.KH - 1) return false;
    if ((size_t)s.Cq * s.H * s.W * 8 >= ((size_t)1 << 30)) return false;     // buffer-offset range marks
    return (size_t)i->lds_fixed + (size_t)f64_fifo_depth(s.W, P, s.KH, s.KW) * i->fslot <= 160 * 1024;
}

size_t finc_f64_packed_bytes(int G, int Cq, int KH, int KW)
{
    const DInst *i = find_dinst(Cq, KH, KW);
    return i ? (size_t)G * i->nfrag * 64 * sizeof(double) : 0;
}

int finc_f64_launch(const double *in, const double *wc, double *out, void *packed, const FincShape &s, bool forward, hipStream_t st)
{
    const DInst *i = find_dinst(s.Cq, s.KH, s.KW);
    if (!i) return FINC_ERR_UNSUPPORTED;
    const int MT = (i->cqp + 15) / 16, NK = i->cqp / 4;
    hipLaunchKernelGGL(pack_f64_kernel, dim3(s.G), dim3(256), sizeof(double) * s.Cq * s.Cq, st, wc, (double *)packed, s.Cq, s.KH, s.KW, MT, NK,
                       forward ? 1 : 0);
    FINC_CHECK_LAUNCH();
    if (forward) {
        hipLaunchKernelGGL(i->fwd, dim3((s.W + 15) / 16, s.B * s.G), dim3(64), 0, st, in, (const double *)packed, out, s.G, s.Cq, s.H, s.W, s.orient);
    } else {
        const int P = s.W < 16 ? s.W : 16;
        const int NB = (s.H + P - 1) / P;
        const int DF = f64_fifo_depth(s.W, P, s.KH, s.KW);
        const size_t per_wave = (size_t)i->lds_fixed + (size_t)DF * i->fslot;
        int wpw = WPW;
        while (wpw > 1 && per_wave * wpw > 160 * 1024) wpw >>= 1;
        const size_t lds = per_wave * wpw;
        if (int e = finc_ensure_dynamic_lds((const void *)i->inv, lds)) return e;
        hipLaunchKernelGGL(i->inv, dim3((s.B * s.G + wpw - 1) / wpw), dim3(64 * wpw), lds, st, in, (const double *)packed, out, s.G, s.Cq, s.H, s.W, P,
                           NB * s.W + P - 1, s.orient, DF, s.B * s.G, wpw);
    }
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

unsigned finc_build_flags_f64() { return FINC_BUILD_FLAGS; }
