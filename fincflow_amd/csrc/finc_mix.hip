// Per-pixel channel mixing for gfx950:  out[b, :, p] = M * in[b, :, p] + bias   (fp32, NCHW, M is C x C).
//
// This is the 1x1 convolution that sits next to the FInC unit in the reference's flow step (layers/conv1x1.py:29-43:
// forward F.conv2d(x, W), reverse F.conv2d(z, W^-1); fastflow_cifar_multi_gpu.py:224-256 puts it between ActNorm and
// the coupling), with the per-channel affine neighbour (layers/actnorm.py:39-52) folded into M and bias by the caller
// (SURVEY 8 f3).  As a separate MIOpen launch it costs more than the unit's inverse (c3: 0.70 ms + 0.27 ms for ActNorm,
// profiles/r02/f3_conv1x1_separate_launch.json); here it is one streaming pass at the HBM rate.
//
// Mapping: per image Out[C x HW] = M[C x C] * In[C x HW] on v_mfma_f32_16x16x4_f32 (exact fp32).
//   * one wavefront owns chunks of 16*PX consecutive pixels of one image, all C channels; lane (q,p): k-slot q (input
//     channel 4j+q of k-step j), pixels PX*p .. PX*p+PX-1 of the chunk.  The 16 lanes of a lane row read / write
//     16*PX consecutive floats of one channel row: 64 (PX=1) or 128 (PX=2) contiguous bytes per request.
//   * A operands (M in fragment order, plus one extra k-step that carries the bias against a constant-one B operand) sit in
//     LDS, filled once per workgroup; they are re-read per chunk (one ds_read_b32 per PX MFMAs), so neither registers
//     nor instantiations depend on how M was produced.
//   * D layout (lane (q,p), register r of tile mt = output channel 16mt+4q+r) stores straight back: no LDS, no shuffle.
//   * in == out is allowed: a chunk's pixels are read for all channels before any is written, and no other wave
//     touches them.
// Latency is hidden by occupancy, not by software pipelining: <= 128 registers, 4 workgroups of 4 waves per CU.
#include "finc_common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr unsigned MIX_INVALID = 0x80000000u;

template <int PX>
struct PixVec;
template <>
struct PixVec<1> {
    static __device__ inline void load(float (&d)[1], __amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
    {
        d[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    }
    static __device__ inline void store(const float (&d)[1], __amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
    {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d[0]), r, voff, soff, 0);
    }
};
template <>
struct PixVec<2> {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    static __device__ inline void load(float (&d)[2], __amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
    {
        const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        const unsigned a = v.x, b = v.y;
        d[0] = __builtin_bit_cast(float, a);
        d[1] = __builtin_bit_cast(float, b);
    }
    static __device__ inline void store(const float (&d)[2], __amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
    {
        v2u v;
        v.x = __builtin_bit_cast(unsigned, d[0]);
        v.y = __builtin_bit_cast(unsigned, d[1]);
        __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
    }
};

template <>
struct PixVec<4> {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    static __device__ inline void load(float (&d)[4], __amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
    {
        const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
        const unsigned a = v.x, b = v.y, c = v.z, e = v.w;
        d[0] = __builtin_bit_cast(float, a);
        d[1] = __builtin_bit_cast(float, b);
        d[2] = __builtin_bit_cast(float, c);
        d[3] = __builtin_bit_cast(float, e);
    }
    static __device__ inline void store(const float (&d)[4], __amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
    {
        v4u v;
        v.x = __builtin_bit_cast(unsigned, d[0]);
        v.y = __builtin_bit_cast(unsigned, d[1]);
        v.z = __builtin_bit_cast(unsigned, d[2]);
        v.w = __builtin_bit_cast(unsigned, d[3]);
        __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
    }
};

// MTN = row tiles of 16 output channels, NK = k-steps of 4 input channels (C = 4*NK, MTN = ceil(C/16)).
// Register budget: KB*PX operand + 4*MTN*PX accumulator + 2*MTN fragment registers.  The fragment reads are software
// pipelined one k-step ahead and fenced per k-step: left alone, the scheduler hoists all MTN*NK LDS reads of a chunk
// (350-500 registers at C >= 96, one wave per SIMD and nothing to cover its loads).
// Workgroup size WGW (waves): 16 = one workgroup per CU, one copy of the fragments in LDS filled once (the fill is a
// gather of C*C scattered words per workgroup -- with 4-wave workgroups it was a tenth of the c3 run time); 4 = small
// problems, where more workgroups than CUs matter more.  16 waves per CU need <= 128 registers.
template <int MTN, int NK, int PX, int WGW>
__global__ __launch_bounds__((64 * WGW)) void finc_mix_kernel(const float *__restrict__ in, const float *__restrict__ mat,
                                                           const float *__restrict__ bias, float *out, int C, int HW,
                                                           int chunks_per_image, int total_chunks)
{
    extern __shared__ __attribute__((aligned(16))) float afrag[];   // [(mt*(NK+1) + j)*64 + lane]
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, p = lane & 15;
    // ---- M (and the bias column) -> LDS, fragment order: lane (q,i) of fragment (mt, j) = M[16mt+i][4j+q]
    for (int e = threadIdx.x; e < MTN * (NK + 1) * 64; e += (int)blockDim.x) {
        const int l = e & 63, f = e >> 6;
        const int j = f % (NK + 1), mt = f / (NK + 1);
        const int row = 16 * mt + (l & 15), col = 4 * j + (l >> 4);
        float v = 0.f;
        if (row < C) {
            if (j < NK) v = col < C ? mat[(size_t)row * C + col] : 0.f;
            else v = ((l >> 4) == 0 && bias) ? bias[row] : 0.f;      // bias rides on k-slot 0 of the extra k-step
        }
        afrag[e] = v;
    }
    __syncthreads();
    const float one = q == 0 ? 1.f : 0.f;      // B operand of the bias k-step
    // A DS instruction reaches 64 KB beyond its address register; the fragments of C = 192 span 150 KB.  Three explicit
    // window bases keep every read an immediate offset (left to itself the compiler materialises one address per
    // fragment, hoists them all out of the chunk loop and spills).
    constexpr int WIN = 16128;                 // floats per window (63 KB)
    const float *const ab[3] = {afrag + lane, afrag + WIN + lane, afrag + 2 * WIN + lane};
    auto frag = [&](int mt, int j) {
        const int e = (mt * (NK + 1) + j) * 64;
        return ab[e / WIN][e % WIN];
    };
    // (readfirstlane: the wave index is uniform, but only this tells the compiler -- without it every buffer access of the loop is
    // wrapped in a waterfall loop over its resource descriptor: 49 of them, C = 192: 676 -> 603 us, C = 96: 183 -> 178)
    const int wave = blockIdx.x * WGW + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwaves = gridDim.x * WGW;
    const unsigned img_bytes = (unsigned)C * (unsigned)HW * 4u;
    const unsigned rowpart = (unsigned)q * (unsigned)HW * 4u;        // lane part of an offset: channel q of a group of four
    // k-steps in blocks of at most KB (the operand registers of a block are all in flight together)
    constexpr int KB = PX == 4 ? (NK % 12 == 0 ? 12 : NK % 8 == 0 ? 8 : NK <= 8 ? NK : 4) : NK <= 32 ? NK : 24;
    static_assert(NK % KB == 0, "k-step blocks must tile the k-steps");
    for (int chunk = wave; chunk < total_chunks; chunk += nwaves) {
        const int b = __builtin_amdgcn_readfirstlane(chunk / chunks_per_image), ci = chunk - b * chunks_per_image;   // (the small kernels divide on the VALU: say it again)
        const __amdgpu_buffer_rsrc_t rin =
            __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)b * C * HW), 0, (int)img_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rout =
            __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)b * C * HW), 0, (int)img_bytes, 0x00020000);
        const int pix = ci * 16 * PX + p * PX;
        const bool ok = pix < HW;              // (PX > 1: HW % PX == 0, so a lane's pixels are valid together)
        const unsigned base = ok ? (unsigned)pix * 4u + rowpart : MIX_INVALID;
        v4f acc[MTN][PX];
        float a_cur[MTN], a_nxt[MTN];
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt) a_cur[mt] = frag(mt, NK);   // the bias k-step goes first
#pragma unroll
        for (int k0 = 0; k0 < NK; k0 += KB) {
            float bv[KB][PX];
#pragma unroll
            for (int j = 0; j < KB; ++j) PixVec<PX>::load(bv[j], rin, base, (k0 + j) * 4 * HW * 4);
#pragma unroll
            for (int j = (k0 == 0 ? -1 : 0); j < KB; ++j) {
                if (k0 + j + 1 < NK) {
#pragma unroll
                    for (int mt = 0; mt < MTN; ++mt) a_nxt[mt] = frag(mt, k0 + j + 1);
                }
#pragma unroll
                for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
                    for (int e = 0; e < PX; ++e) {
                        if (j < 0) acc[mt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt], one, (v4f){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        else acc[mt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt], bv[j][e], acc[mt][e], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MTN; ++mt) a_cur[mt] = a_nxt[mt];
            }
        }
        // D: lane (q,p), register r of tile mt = output channel 16mt + 4q + r
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float d[PX];
#pragma unroll
                for (int e = 0; e < PX; ++e) d[e] = r == 0 ? acc[mt][e].x : r == 1 ? acc[mt][e].y : r == 2 ? acc[mt][e].z : acc[mt][e].w;
                // rows >= C exist only in the last tile: their lanes point beyond the image
                const bool rowok = mt < MTN - 1 || 16 * mt + 4 * q + r < C;
                const unsigned vo = (ok && rowok) ? (unsigned)pix * 4u + (unsigned)(4 * q) * (unsigned)HW * 4u : MIX_INVALID;
                PixVec<PX>::store(d, rout, vo, (16 * mt + r) * HW * 4);
            }
        }
    }
}

typedef void (*mix_fn)(const float *, const float *, const float *, float *, int, int, int, int);
struct MixInst {
    int C;
    mix_fn fn[3][2];   // [1, 2 or 4 pixels per lane][4- or 16-wave workgroup]; nullptr = not instantiated
};
template <int C>
constexpr MixInst make_mix()
{
    constexpr int MTN = (C + 15) / 16, NK = C / 4;
    constexpr bool big_lds = MTN * (NK + 1) * 256 > 80 * 1024;            // only one workgroup fits a CU anyway
    constexpr int kb = NK <= 32 ? NK : 24;
    constexpr bool wide16 = 2 * kb + 8 * MTN + 2 * MTN + 24 <= 128;       // two pixels per lane within 128 registers
    constexpr bool quad = MTN <= 6;                                       // four pixels per lane: 204 registers at C = 96, 2 waves per SIMD
    MixInst m{C, {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}}};
    if constexpr (big_lds) m.fn[0][0] = finc_mix_kernel<MTN, NK, 1, 16>; else m.fn[0][0] = finc_mix_kernel<MTN, NK, 1, 4>;
    m.fn[0][1] = finc_mix_kernel<MTN, NK, 1, 16>;
    if constexpr (!big_lds) m.fn[1][0] = finc_mix_kernel<MTN, NK, 2, 4>;
    if constexpr (wide16) m.fn[1][1] = finc_mix_kernel<MTN, NK, 2, 16>;
    if constexpr (quad) { m.fn[2][0] = finc_mix_kernel<MTN, NK, 4, 4>; m.fn[2][1] = finc_mix_kernel<MTN, NK, 4, 8>; }
    return m;
}
// every channel count of the reference's model scripts (4*Cq, Cq in {1,2,3,4,6,12,16,24,48}) and the powers of two between
const MixInst g_mix[] = {make_mix<4>(),  make_mix<8>(),  make_mix<12>(), make_mix<16>(), make_mix<24>(), make_mix<32>(),
                         make_mix<48>(), make_mix<64>(), make_mix<96>(), make_mix<128>(), make_mix<192>()};

const MixInst *find_mix(int C)
{
    for (const MixInst &m : g_mix)
        if (m.C == C) return &m;
    return nullptr;
}

} // namespace

bool finc_mix_supported(int C) { return find_mix(C) != nullptr; }

int finc_mix_launch(const float *in, const float *mat, const float *bias, float *out, int B, int C, int HW, hipStream_t st)
{
    const MixInst *m = find_mix(C);
    if (!m) return FINC_ERR_UNSUPPORTED;
    if ((size_t)C * HW * 4 >= ((size_t)1 << 31)) return FINC_ERR_BAD_DIMS;
    const size_t lds = (size_t)((C + 15) / 16) * (C / 4 + 1) * 64 * sizeof(float);
    // PX pixels per lane need PX*4-byte aligned rows: HW % PX == 0 and aligned bases
    const uintptr_t ptrs = (uintptr_t)in | (uintptr_t)out;
    // big = one large workgroup per CU once there is enough work to give every wave of the chip several chunks
    const long long chunks1 = (long long)B * ((HW + 15) / 16);
    const int big = chunks1 >= 4LL * 256 * 16 ? 1 : 0;
    int pxi = (HW % 4 == 0 && (ptrs & 15u) == 0 && HW >= 64) ? 2 : (HW % 2 == 0 && (ptrs & 7u) == 0 && HW >= 32) ? 1 : 0;
    int wb = big;
    while (pxi > 0 && !m->fn[pxi][wb]) {       // that combination is not instantiated (registers / LDS): other size, then narrower
        if (m->fn[pxi][wb ^ 1]) { wb ^= 1; break; }
        --pxi;
    }
    if (!m->fn[pxi][wb]) wb ^= 1;
    const mix_fn fn = m->fn[pxi][wb];
    if (!fn) return FINC_ERR_UNSUPPORTED;
    const int px = 1 << pxi;
    // waves per workgroup of the chosen instantiation (make_mix)
    const int wgw = lds > 80 * 1024 ? 16 : wb == 0 ? 4 : pxi == 2 ? 8 : 16;
    const int cpi = (HW + 16 * px - 1) / (16 * px);
    const long long total = (long long)B * cpi;
    if (total >= (1LL << 31)) return FINC_ERR_BAD_DIMS;
    if (int e = finc_ensure_dynamic_lds((const void *)fn, lds)) return e;
    // persistent: enough workgroups to fill the chip at this kernel's occupancy, each wave loops over its chunks
    const int waves_cu = pxi == 2 ? 8 : 16;    // four pixels per lane: 2 waves per SIMD
    int wgs_per_cu = (int)((160 * 1024) / lds);
    if (wgs_per_cu > waves_cu / wgw) wgs_per_cu = waves_cu / wgw;
    if (wgs_per_cu < 1) wgs_per_cu = 1;
    long long wgs = (total + wgw - 1) / wgw;
    if (wgs > 256LL * wgs_per_cu) wgs = 256LL * wgs_per_cu;
    hipLaunchKernelGGL(fn, dim3((unsigned)wgs), dim3(64 * wgw), lds, st, in, mat, bias, out, C, HW, cpi, (int)total);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

unsigned finc_build_flags_mix() { return FINC_BUILD_FLAGS; }
