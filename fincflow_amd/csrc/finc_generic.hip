// Reference-order kernels: correct for every shape, the universal fallback and
// the bit-exactness anchor.  Compiled with -ffp-contract=off.
//
//  * inverse_strict: one workgroup per (image, group) walks the H+W-1
//    anti-diagonals with a workgroup barrier where the reference has a kernel
//    launch + cudaDeviceSynchronize (cinc_cuda_kernel_level2.cu:98-130).  One
//    thread owns one pixel of the diagonal and runs the reference's term order
//    kh -> kw -> kc (cinc_cuda_kernel_level2.cu:59-72) with a separate fp32
//    multiply and subtract per term, so the result is bit-identical to the
//    fp32 CPU restatement (utils/solve_mc.py:29-44).
//  * forward_generic / backward_generic: direct evaluation, one output per thread.
#include "finc_common.h"

namespace {

constexpr int STRICT_MAX_BLOCK = 256;

// acc - x*w as two separately rounded operations (the reference's C++ `output -= input*kernel` compiled without
// contraction, and numpy's elementwise ops in utils/solve_mc.py)
__device__ inline float mul_sub(float acc, float x, float w) { return __fsub_rn(acc, __fmul_rn(x, w)); }
__device__ inline double mul_sub(double acc, double x, double w) { return __dsub_rn(acc, __dmul_rn(x, w)); }

// T = float: the fp32 path; T = double: the reference op's other dispatch arm (AT_DISPATCH_FLOATING_TYPES,
// cinc_cuda_kernel_level2.cu:117) and the arithmetic of its Cython CPU solver (solve_parallel_mc.pyx:77-126).
template <typename T>
__global__ void inverse_strict_kernel(const T *__restrict__ z, const T *__restrict__ wc, T *x, int G,
                                      int Cq, int H, int W, int KH, int KW, unsigned orient)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char own_raw[];
    T *own = reinterpret_cast<T *>(own_raw); // [Cq][blockDim.x]: this pixel's solved channels
    const int bg = blockIdx.x;
    const int g = bg % G;
    const unsigned o = finc_group_orient(orient, g);
    const size_t HW = (size_t)H * W;
    const size_t off = (size_t)bg * Cq * HW; // == (b*G*Cq + g*Cq) * HW
    const T *zg = z + off;
    T *xg = x + off;
    const T *wg = wc + (size_t)g * Cq * Cq * KH * KW;
    const int tid = threadIdx.x, nt = blockDim.x;

    for (int d = 0; d < H + W - 1; ++d) {
        const int h_lo = d - (W - 1) > 0 ? d - (W - 1) : 0;
        const int h_hi = d < H - 1 ? d : H - 1;
        const int len = h_hi - h_lo + 1;
        for (int base = 0; base < len; base += nt) {
            const int idx = base + tid;
            if (idx < len) {
                const int h = h_lo + idx, w = d - h;
                const int p = finc_pix(H, W, o, h, w);
                for (int c = 0; c < Cq; ++c) {
                    T acc = zg[(size_t)c * HW + p];
                    for (int kh = 0; kh < KH; ++kh) {
                        if (h - kh < 0) break;
                        for (int kw = 0; kw < KW; ++kw) {
                            if (w - kw < 0) break;
                            const T *wrow = wg + (((size_t)c * Cq) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw);
                            if (kh == 0 && kw == 0) {
                                for (int kc = 0; kc < c; ++kc) { // kc == c skipped, kc > c: zero taps
                                    T xv = own[kc * nt + tid];
                                    T wv = wrow[(size_t)kc * KH * KW];
                                    acc = mul_sub(acc, xv, wv);
                                }
                            } else {
                                const int pn = finc_pix(H, W, o, h - kh, w - kw);
                                for (int kc = 0; kc < Cq; ++kc) {
                                    T xv = xg[(size_t)kc * HW + pn];
                                    T wv = wrow[(size_t)kc * KH * KW];
                                    acc = mul_sub(acc, xv, wv);
                                }
                            }
                        }
                    }
                    own[c * nt + tid] = acc;
                    xg[(size_t)c * HW + p] = acc;
                }
            }
        }
        __syncthreads(); // this diagonal's stores become visible to the workgroup
    }
}

template <typename T>
__global__ void forward_generic_kernel(const T *__restrict__ x, const T *__restrict__ wc,
                                       T *__restrict__ z, int B, int G, int Cq, int H, int W, int KH, int KW,
                                       unsigned orient)
{
    const size_t HW = (size_t)H * W;
    const size_t total = (size_t)B * G * Cq * HW;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int pw = (int)(idx % W);
        const int ph = (int)((idx / W) % H);
        const size_t ch = idx / HW; // b*G*Cq + g*Cq + oc
        const int oc = (int)(ch % Cq);
        const int g = (int)((ch / Cq) % G);
        const unsigned o = finc_group_orient(orient, g);
        const int h = (o & FINC_FLIP_H) ? H - 1 - ph : ph; // canonical coordinates of this output
        const int w = (o & FINC_FLIP_W) ? W - 1 - pw : pw;
        const T *xg = x + (ch - oc) * HW;
        const T *wo = wc + ((size_t)(g * Cq + oc) * Cq) * KH * KW;
        T acc = 0;
        for (int ic = 0; ic < Cq; ++ic)
            for (int a = 0; a < KH && a <= h; ++a)
                for (int b = 0; b < KW && b <= w; ++b)
                    acc = fma(xg[(size_t)ic * HW + finc_pix(H, W, o, h - a, w - b)],
                              wo[((size_t)ic * KH + (KH - 1 - a)) * KW + (KW - 1 - b)], acc);
        z[idx] = acc;
    }
}

// grad_x[b,i,h,w] = sum_{o,a,b} wc[o,i,KH-1-a,KW-1-b] * gz[b,o,h+a,w+b]   (canonical coordinates)
__global__ void backward_input_kernel(const float *__restrict__ gz, const float *__restrict__ wc,
                                      float *__restrict__ gx, int B, int G, int Cq, int H, int W, int KH, int KW,
                                      unsigned orient)
{
    const size_t HW = (size_t)H * W;
    const size_t total = (size_t)B * G * Cq * HW;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int pw = (int)(idx % W);
        const int ph = (int)((idx / W) % H);
        const size_t ch = idx / HW;
        const int ic = (int)(ch % Cq);
        const int g = (int)((ch / Cq) % G);
        const unsigned o = finc_group_orient(orient, g);
        const int h = (o & FINC_FLIP_H) ? H - 1 - ph : ph;
        const int w = (o & FINC_FLIP_W) ? W - 1 - pw : pw;
        const float *gg = gz + (ch - ic) * HW;
        const float *wgp = wc + (size_t)g * Cq * Cq * KH * KW;
        float acc = 0.f;
        for (int oc = 0; oc < Cq; ++oc)
            for (int a = 0; a < KH && h + a < H; ++a)
                for (int b = 0; b < KW && w + b < W; ++b)
                    acc = fmaf(gg[(size_t)oc * HW + finc_pix(H, W, o, h + a, w + b)],
                               wgp[(((size_t)oc * Cq + ic) * KH + (KH - 1 - a)) * KW + (KW - 1 - b)], acc);
        gx[idx] = acc;
    }
}

// One workgroup per (g, o, i): all KH*KW taps of grad_w_canon[g*Cq+o][i], reduced over (b,h,w).
// The corner tap of entries with i >= o is masked to 0 (PaddedConv2d.reset_gradients, layers/conv.py:98-99).
constexpr int BW_BLOCK = 256;
constexpr int BW_MAX_TAPS = 49;
__global__ void backward_weight_kernel(const float *__restrict__ gz, const float *__restrict__ x,
                                       float *__restrict__ gw, int B, int G, int Cq, int H, int W, int KH, int KW,
                                       unsigned orient)
{
    __shared__ float red[BW_BLOCK];
    const int i = blockIdx.x % Cq;
    const int oc = (blockIdx.x / Cq) % Cq;
    const int g = blockIdx.x / (Cq * Cq);
    const unsigned o = finc_group_orient(orient, g);
    const size_t HW = (size_t)H * W;
    const int ntap = KH * KW;
    float acc[BW_MAX_TAPS];
#pragma unroll
    for (int t = 0; t < BW_MAX_TAPS; ++t) acc[t] = 0.f;
    const size_t n = (size_t)B * HW;
    for (size_t e = threadIdx.x; e < n; e += BW_BLOCK) {
        const int b = (int)(e / HW);
        const int r = (int)(e % HW);
        const int h = r / W, w = r % W; // canonical coordinates of the grad_z sample
        const size_t plane = ((size_t)b * G + g) * Cq;
        const float gv = gz[(plane + oc) * HW + finc_pix(H, W, o, h, w)];
        const float *xp = x + (plane + i) * HW;
#pragma unroll 1
        for (int kh = 0; kh < KH; ++kh) {
            const int a = KH - 1 - kh;
            if (h - a < 0) continue;
            for (int kw = 0; kw < KW; ++kw) {
                const int bb = KW - 1 - kw;
                if (w - bb < 0) continue;
                acc[kh * KW + kw] = fmaf(gv, xp[finc_pix(H, W, o, h - a, w - bb)], acc[kh * KW + kw]);
            }
        }
    }
    for (int t = 0; t < ntap; ++t) {
        red[threadIdx.x] = acc[t];
        __syncthreads();
        for (int s = BW_BLOCK / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const bool masked = (t == ntap - 1) && (i >= oc);
            gw[(((size_t)(g * Cq + oc) * Cq + i) * ntap) + t] = masked ? 0.f : red[0];
        }
        __syncthreads();
    }
}

} // namespace

template <typename T>
static int launch_inverse_strict(const T *z, const T *wc, T *x, const FincShape &s, hipStream_t st)
{
    int diag = s.H < s.W ? s.H : s.W;
    int block = ((diag + 63) / 64) * 64;
    if (block > STRICT_MAX_BLOCK) block = STRICT_MAX_BLOCK;
    while (block > 64 && (size_t)block * s.Cq * sizeof(T) > 60 * 1024) block -= 64;
    size_t lds = (size_t)block * s.Cq * sizeof(T);
    if (lds > 64 * 1024) return FINC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(inverse_strict_kernel<T>, dim3(s.B * s.G), dim3(block), lds, st, z, wc, x, s.G, s.Cq, s.H, s.W,
                       s.KH, s.KW, s.orient);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

template <typename T>
static int launch_forward_generic(const T *x, const T *wc, T *z, const FincShape &s, hipStream_t st)
{
    size_t total = (size_t)s.B * s.G * s.Cq * s.H * s.W;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(forward_generic_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, x, wc, z, s.B, s.G, s.Cq,
                       s.H, s.W, s.KH, s.KW, s.orient);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

int finc_launch_inverse_strict(const float *z, const float *wc, float *x, const FincShape &s, hipStream_t st)
{
    return launch_inverse_strict<float>(z, wc, x, s, st);
}
int finc_launch_inverse_strict_f64(const double *z, const double *wc, double *x, const FincShape &s, hipStream_t st)
{
    return launch_inverse_strict<double>(z, wc, x, s, st);
}
int finc_launch_forward_generic(const float *x, const float *wc, float *z, const FincShape &s, hipStream_t st)
{
    return launch_forward_generic<float>(x, wc, z, s, st);
}
int finc_launch_forward_generic_f64(const double *x, const double *wc, double *z, const FincShape &s, hipStream_t st)
{
    return launch_forward_generic<double>(x, wc, z, s, st);
}

int finc_launch_backward_generic(const float *gz, const float *x, const float *wc, float *gx, float *gw,
                                 const FincShape &s, hipStream_t st)
{
    if (gx) {
        size_t total = (size_t)s.B * s.G * s.Cq * s.H * s.W;
        size_t blocks = (total + 255) / 256;
        if (blocks > 256 * 32) blocks = 256 * 32;
        hipLaunchKernelGGL(backward_input_kernel, dim3((unsigned)blocks), dim3(256), 0, st, gz, wc, gx, s.B, s.G,
                           s.Cq, s.H, s.W, s.KH, s.KW, s.orient);
        FINC_CHECK_LAUNCH();
    }
    if (gw) {
        if (s.KH * s.KW > BW_MAX_TAPS) return FINC_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(backward_weight_kernel, dim3(s.G * s.Cq * s.Cq), dim3(BW_BLOCK), 0, st, gz, x, gw, s.B,
                           s.G, s.Cq, s.H, s.W, s.KH, s.KW, s.orient);
        FINC_CHECK_LAUNCH();
    }
    return FINC_OK;
}

// -----------------------------------------------------------------------------------------------
// Width padding for the MFMA inverse (finc_abi.hip, FINC_ALGO_AUTO with W % 4 != 0): rows are copied into a buffer
// whose width is a multiple of 8, zero-filled on the right, and the solved rows are copied back.  Exact: a zero z on
// ghost columns right of the image never feeds a real pixel of an un-flipped group, and stays an exact zero on the
// canonical LEFT of a W-flipped group, where it stands for the out-of-image taps the unpadded problem masks.
// -----------------------------------------------------------------------------------------------
namespace {
__global__ void repitch_kernel(const float *__restrict__ in, float *__restrict__ out, long long rows, int Win, int Wout)
{
    const long long n = rows * Wout;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / Wout;
        const int w = (int)(e - r * Wout);
        out[e] = w < Win ? in[r * Win + w] : 0.f;
    }
}
} // namespace

int finc_launch_repitch(const float *in, float *out, long long rows, int Win, int Wout, hipStream_t st)
{
    const long long n = rows * Wout;
    if (n == 0) return FINC_OK;
    long long blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(repitch_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, out, rows, Win, Wout);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

unsigned finc_build_flags_generic() { return FINC_BUILD_FLAGS; }
