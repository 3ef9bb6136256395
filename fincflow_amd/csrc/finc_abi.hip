// extern "C" surface of libfinc_hip.so -- see include/finc.h for the contract
// and the reference interfaces each entry point replaces.
#include "finc_common.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

namespace {

thread_local char g_hip_error[256] = "no error";

// ---- per-process tables (the only global state of the library; include/finc.h) ----
std::mutex g_table_mutex;
std::vector<std::pair<int, const void *>> g_attr_done;   // (device, kernel) pairs whose LDS attribute is set
constexpr int MAX_DEVICES = 64;
int *g_invariant_flag[MAX_DEVICES];                       // one 4-byte device word per device, allocated on first use
volatile unsigned *g_fault_host[MAX_DEVICES];             // one pinned, mapped host word per device: set by a kernel that gave up a wait
unsigned *g_fault_dev[MAX_DEVICES];                       // ... and the device's pointer to it
std::vector<std::pair<int, size_t>> g_debug_tokens;       // finc_debug_attr_table_insert's own table (never the live one)

std::atomic<bool> g_fault_armed_any{false};
std::vector<const char *> g_env_seen;                      // run-time switches found set (finc_env)

// caller holds g_table_mutex
bool attr_table_has(int device, const void *fn)
{
    for (const auto &e : g_attr_done)
        if (e.first == device && e.second == fn) return true;
    return false;
}

template <typename T>
__global__ void canonicalize_kernel(const T *__restrict__ ws, T *__restrict__ wc, int G, int Cq, int KH,
                                    int KW, unsigned orient)
{
    const int total = G * Cq * Cq * KH * KW;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int kw = idx % KW;
        const int kh = (idx / KW) % KH;
        const int rest = idx / (KW * KH); // (g*Cq + oc)*Cq + ic
        const int g = rest / (Cq * Cq);
        const unsigned o = finc_group_orient(orient, g);
        const int sh = (o & FINC_FLIP_H) ? KH - 1 - kh : kh;
        const int sw = (o & FINC_FLIP_W) ? KW - 1 - kw : kw;
        wc[idx] = ws[(rest * KH + sh) * KW + sw];
    }
}

// flag[0] = 0 if the corner tap is unit lower triangular in every group, else 1 + first bad index
__global__ void invariant_kernel(const float *__restrict__ wc, int G, int Cq, int KH, int KW, int *flag)
{
    const int total = G * Cq * Cq;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int kc = idx % Cq;
        const int c = (idx / Cq) % Cq;
        if (kc < c) continue;
        const float v = wc[((size_t)idx * KH + (KH - 1)) * KW + (KW - 1)];
        const bool ok = (kc == c) ? (v == 1.0f) : (v == 0.0f);
        if (!ok) atomicMax(flag, 1 + idx);
    }
}

int check_shape(int B, int G, int Cq, int H, int W, int KH, int KW)
{
    if (B <= 0 || G <= 0 || Cq <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0) return FINC_ERR_BAD_DIMS;
    if (G > FINC_MAX_GROUPS || Cq > FINC_MAX_CQ || KH > 15 || KW > 15) return FINC_ERR_BAD_DIMS;
    if ((size_t)B * G * Cq * H * W >= ((size_t)1 << 40)) return FINC_ERR_BAD_DIMS;
    if ((size_t)Cq * H * W >= ((size_t)1 << 31)) return FINC_ERR_BAD_DIMS; // per-group offsets are 32-bit in-kernel
    return FINC_OK;
}

inline bool misaligned(const void *p) { return ((uintptr_t)p & 3u) != 0; }

} // namespace

void finc_set_hip_error(hipError_t e)
{
    strncpy(g_hip_error, hipGetErrorString(e), sizeof(g_hip_error) - 1);
    g_hip_error[sizeof(g_hip_error) - 1] = 0;
}

// A kernel's dynamic-LDS ceiling is a per-DEVICE property of the loaded code object: set it once per (device, kernel),
// whichever thread launches first (a per-thread cache would skip the second device of a process that drives two).
int finc_ensure_dynamic_lds(const void *fn, size_t bytes)
{
    if (bytes <= 48 * 1024) return FINC_OK;
    int dev = 0;
    FINC_HIP_TRY(hipGetDevice(&dev));
    // one lock over lookup, hipFuncSetAttribute and insert: a second thread on the device must not see the entry before the
    // attribute is set, and a failed call must leave no entry behind (the next launch tries again)
    std::lock_guard<std::mutex> lk(g_table_mutex);
    if (attr_table_has(dev, fn)) return FINC_OK;
    FINC_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    g_attr_done.emplace_back(dev, fn);
    return FINC_OK;
}

unsigned *finc_fault_device_word()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
    return g_fault_host[dev] ? g_fault_dev[dev] : nullptr;
}

const char *finc_env(const char *name)
{
    const char *v = getenv(name);
    if (v) {
        std::lock_guard<std::mutex> lk(g_table_mutex);
        bool seen = false;
        for (const char *n : g_env_seen) seen = seen || strcmp(n, name) == 0;
        if (!seen) g_env_seen.push_back(name);              // (string literals of the callers: they outlive the table)
    }
    return v;
}

// `arm`: publish the device's fault word if that has not happened yet.  Arming is a hipHostMalloc + a synchronous copy to a
// device symbol: both are illegal while `st` is being captured into a graph, so a capture skips it (the packing calls, which
// run before any capture of the launches, arm too); a failed step frees the word and is retried by the next arming call.
int finc_fault_gate(bool arm, hipStream_t st)
{
    if (!arm && !g_fault_armed_any.load(std::memory_order_acquire)) return FINC_OK;   // nothing armed yet: no HIP call at all
    int dev = 0;
    FINC_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return FINC_ERR_BAD_DIMS;
    volatile unsigned *w = g_fault_host[dev];
    if (!w && arm) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs == hipStreamCaptureStatusNone) {
            std::lock_guard<std::mutex> lk(g_table_mutex);
            if (!g_fault_host[dev]) {
                unsigned *h = nullptr, *d = nullptr;
                FINC_HIP_TRY(hipHostMalloc((void **)&h, sizeof(unsigned), hipHostMallocMapped));
                *h = 0;
                int e = FINC_OK;
                if (hipError_t he = hipHostGetDevicePointer((void **)&d, h, 0); he != hipSuccess) { finc_set_hip_error(he); e = FINC_ERR_LAUNCH; }
                if (!e) e = finc_mfma_arm_fault_word(d);
                if (e) { (void)hipHostFree(h); return e; }
                g_fault_dev[dev] = d;
                g_fault_host[dev] = h;
                g_fault_armed_any.store(true, std::memory_order_release);
            }
            w = g_fault_host[dev];
        }
    }
    if (w && *w != 0) {
        strncpy(g_hip_error, "a helper-wave wait of an earlier launch on this device gave up: its output is not valid (finc_clear_fault() resets)",
                sizeof(g_hip_error) - 1);
        return FINC_ERR_LAUNCH;
    }
    return FINC_OK;
}

template <typename T>
static int canonicalize(const T *w_stored, T *w_canon, int G, int Cq, int KH, int KW, unsigned orient, finc_stream_t stream)
{
    if (!w_stored || !w_canon) return FINC_ERR_NULL_POINTER;
    if (int e = check_shape(1, G, Cq, 1, 1, KH, KW)) return e;
    if (w_stored == w_canon) return FINC_ERR_BAD_DIMS;
    const int total = G * Cq * Cq * KH * KW;
    int blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(canonicalize_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_stored, w_canon, G,
                       Cq, KH, KW, orient);
    FINC_CHECK_LAUNCH();
    return FINC_OK;
}

extern "C" {

int finc_version(void) { return 102; }

unsigned finc_build_flags(void)
{
    return FINC_BUILD_FLAGS | finc_build_flags_mfma() | finc_build_flags_split() | finc_build_flags_chain() | finc_build_flags_f64() | finc_build_flags_conv() | finc_build_flags_gradw() | finc_build_flags_wino4m() |
           finc_build_flags_mix() | finc_build_flags_generic() | finc_build_flags_wino() | finc_build_flags_big() | finc_build_flags_probe() | finc_build_flags_wino5() | finc_build_flags_stream();
}

int finc_fault_pending(void)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return 0;
    volatile unsigned *w = g_fault_host[dev];
    return (w && *w != 0) ? 1 : 0;
}

int finc_runtime_switches(char *h_buf, size_t n)
{
    std::lock_guard<std::mutex> lk(g_table_mutex);
    size_t used = 0;
    int count = 0;
    auto put = [&](const char *t) {
        const size_t l = strlen(t);
        if (h_buf && used + l + 2 < n) {
            if (used) h_buf[used++] = ',';
            memcpy(h_buf + used, t, l);
            used += l;
        }
        ++count;
    };
    for (const char *name : g_env_seen) put(name);
    if (finc_wino_form_override() != 0) put("forward_form_override");
    if (h_buf && n) h_buf[used < n ? used : n - 1] = 0;
    return count;
}

int finc_clear_fault(void)
{
    int dev = 0;
    FINC_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return FINC_ERR_BAD_DIMS;
    if (g_fault_host[dev]) *g_fault_host[dev] = 0;
    return FINC_OK;
}

const char *finc_status_string(int status)
{
    switch (status) {
    case FINC_OK: return "ok";
    case FINC_ERR_NULL_POINTER: return "null pointer argument";
    case FINC_ERR_BAD_DIMS: return "bad dimensions";
    case FINC_ERR_UNSUPPORTED: return "requested algorithm does not support this shape";
    case FINC_ERR_WORKSPACE: return "workspace missing or too small";
    case FINC_ERR_LAUNCH: return "HIP call failed";
    case FINC_ERR_INVARIANT: return "corner tap is not unit lower triangular (layers/conv.py:63-70 invariant)";
    case FINC_ERR_ALIGNMENT: return "pointer not aligned for fp32";
    default: return "unknown status";
    }
}

const char *finc_last_hip_error(void) { return g_hip_error; }

int finc_canonicalize_weights_f32(const float *w_stored, float *w_canon, int G, int Cq, int KH, int KW,
                                  unsigned orient, finc_stream_t stream)
{
    return canonicalize<float>(w_stored, w_canon, G, Cq, KH, KW, orient, stream);
}

int finc_canonicalize_weights_f64(const double *w_stored, double *w_canon, int G, int Cq, int KH, int KW,
                                  unsigned orient, finc_stream_t stream)
{
    return canonicalize<double>(w_stored, w_canon, G, Cq, KH, KW, orient, stream);
}

int finc_check_invariant_f32(const float *w_canon, int G, int Cq, int KH, int KW, finc_stream_t stream)
{
    if (!w_canon) return FINC_ERR_NULL_POINTER;
    if (int e = check_shape(1, G, Cq, 1, 1, KH, KW)) return e;
    hipStream_t st = (hipStream_t)stream;
    int dev = 0;
    FINC_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return FINC_ERR_BAD_DIMS;
    if (int e = finc_fault_gate(false)) return e;          // (a synchronous call is a natural place to report a protocol fault)
    // the call is synchronous anyway: one lock covers the lazily allocated per-device flag word and its use, so
    // concurrent checks (DataParallel replica threads) neither allocate twice nor share the word
    std::lock_guard<std::mutex> lk(g_table_mutex);
    if (!g_invariant_flag[dev]) FINC_HIP_TRY(hipMalloc(&g_invariant_flag[dev], sizeof(int)));
    int *d_flag = g_invariant_flag[dev];
    int h_flag = 0;
    FINC_HIP_TRY(hipMemsetAsync(d_flag, 0, sizeof(int), st));
    const int blocks = (G * Cq * Cq + 255) / 256;
    hipLaunchKernelGGL(invariant_kernel, dim3(blocks), dim3(256), 0, st, w_canon, G, Cq, KH, KW, d_flag);
    FINC_CHECK_LAUNCH();
    FINC_HIP_TRY(hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    FINC_HIP_TRY(hipStreamSynchronize(st));
    return h_flag == 0 ? FINC_OK : FINC_ERR_INVARIANT;
}

int finc_debug_attr_table_insert(int device, size_t kernel_token)
{
    // the key logic of the (device, kernel) table on a table of its OWN: a test token must never land in the live table
    // (a token equal to a real kernel address would make the next launch skip its attribute)
    std::lock_guard<std::mutex> lk(g_table_mutex);
    for (const auto &e : g_debug_tokens)
        if (e.first == device && e.second == kernel_token) return 0;
    g_debug_tokens.emplace_back(device, kernel_token);
    return 1;
}

size_t finc_workspace_bytes(int G, int Cq, int KH, int KW)
{
    if (G <= 0 || Cq <= 0 || KH <= 0 || KW <= 0) return 256;
    size_t n = finc_mfma_packed_bytes(G, Cq, KH, KW);
    const size_t c = finc_conv_packed_bytes(G, Cq, KH, KW);
    if (c > n) n = c;
    return n < 256 ? 256 : n;
}

// The MFMA inverse streams rows in 16-byte pieces (W % 4 == 0).  Any other width runs on a copy whose rows are padded
// with zeros to a multiple of 8 floats -- exact (finc_generic.hip, repitch_kernel) and two cheap copies instead of the
// strict kernel's serial chain (W = 66: 0.4 ms instead of 133 ms at B=64, C=96).
static int padded_width(int Cq, int H, int W, int KH, int KW)
{
    if (W % 4 == 0 || finc_mfma_supported(Cq, H, W, KH, KW)) return 0;
    const int Wp = (W + 7) / 8 * 8;
    return finc_mfma_supported(Cq, H, Wp, KH, KW) ? Wp : 0;
}
static size_t align256(size_t n) { return (n + 255) / 256 * 256; }

size_t finc_inverse_workspace_bytes(int B, int G, int Cq, int H, int W, int KH, int KW)
{
    size_t n = finc_workspace_bytes(G, Cq, KH, KW);
    if (B <= 0 || G <= 0 || Cq <= 0 || H <= 0 || W <= 0) return n;
    if (const int Wp = padded_width(Cq, H, W, KH, KW))
        n = align256(n) + 2 * align256((size_t)B * G * Cq * H * Wp * sizeof(float));
    return n;
}

int finc_inverse_algo_for(int Cq, int H, int W, int KH, int KW)
{
    return finc_mfma_supported(Cq, H, W, KH, KW) ? FINC_ALGO_MFMA : FINC_ALGO_STRICT;
}

int finc_forward_algo_for(int Cq, int H, int W, int KH, int KW)
{
    return finc_conv_supported(Cq, H, W, KH, KW) ? FINC_ALGO_MFMA : FINC_ALGO_STRICT;
}

static int run(const float *in, const float *w_canon, float *out, int B, int G, int Cq, int H, int W, int KH,
               int KW, unsigned orient, int algo, void *workspace, size_t workspace_bytes, finc_stream_t stream,
               bool forward)
{
    if (!in || !w_canon || !out) return FINC_ERR_NULL_POINTER;
    if (int e = check_shape(B, G, Cq, H, W, KH, KW)) return e;
    if (misaligned(in) || misaligned(w_canon) || misaligned(out)) return FINC_ERR_ALIGNMENT;
    if (in == out) return FINC_ERR_BAD_DIMS;
    if (int e = finc_fault_gate(false)) return e;          // an earlier launch on this device gave up a protocol wait
    FincShape s{B, G, Cq, H, W, KH, KW, orient};
    hipStream_t st = (hipStream_t)stream;
    if (algo == FINC_ALGO_AUTO && !forward) {
        // a width the MFMA kernel cannot stream: solve a zero-padded copy when the caller's workspace has room for it
        const int Wp = padded_width(Cq, H, W, KH, KW);
        if (Wp && workspace && workspace_bytes >= finc_inverse_workspace_bytes(B, G, Cq, H, W, KH, KW)) {
            const size_t pk = align256(finc_workspace_bytes(G, Cq, KH, KW));
            const size_t act = align256((size_t)B * G * Cq * H * Wp * sizeof(float));
            float *zp = (float *)((char *)workspace + pk), *xp = (float *)((char *)workspace + pk + act);
            const long long rows = (long long)B * G * Cq * H;
            if (int e = finc_launch_repitch(in, zp, rows, W, Wp, st)) return e;
            if (int e = finc_mfma_pack(w_canon, nullptr, nullptr, workspace, G, Cq, KH, KW, st)) return e;
            FincShape sp{B, G, Cq, H, Wp, KH, KW, orient};
            if (int e = finc_mfma_launch(zp, workspace, xp, sp, st)) return e;
            return finc_launch_repitch(xp, out, rows, Wp, W, st);
        }
    }
    if (algo == FINC_ALGO_AUTO) {
        algo = forward ? finc_forward_algo_for(Cq, H, W, KH, KW) : finc_inverse_algo_for(Cq, H, W, KH, KW);
        // the wavefront kernel moves aligned 16-byte pieces: activations that are only 4-byte aligned take the strict kernel
        // (the role-split kernel of small problem sets has no alignment rule)
        if (!forward && algo == FINC_ALGO_MFMA && ((((uintptr_t)in) | ((uintptr_t)out)) & 15u) && !finc_split_takes(s)) algo = FINC_ALGO_STRICT;
    }
    if (algo == FINC_ALGO_STRICT)
        return forward ? finc_launch_forward_generic(in, w_canon, out, s, st)
                       : finc_launch_inverse_strict(in, w_canon, out, s, st);
    if (algo != FINC_ALGO_MFMA) return FINC_ERR_UNSUPPORTED;
    if (forward) {
        if (!finc_conv_supported(Cq, H, W, KH, KW)) return FINC_ERR_UNSUPPORTED;
        if (!workspace || workspace_bytes < finc_conv_packed_bytes(G, Cq, KH, KW)) return FINC_ERR_WORKSPACE;
        if (int e = finc_conv_pack(w_canon, workspace, G, Cq, KH, KW, false, st)) return e;
        return finc_conv_launch(in, workspace, out, s, st);
    }
    if (!finc_mfma_supported(Cq, H, W, KH, KW)) return FINC_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < finc_mfma_packed_bytes(G, Cq, KH, KW)) return FINC_ERR_WORKSPACE;
    if (int e = finc_mfma_pack(w_canon, nullptr, nullptr, workspace, G, Cq, KH, KW, st)) return e;
    return finc_mfma_launch(in, workspace, out, s, st);
}

int finc_inverse_f32(const float *z, const float *w_canon, float *x, int B, int G, int Cq, int H, int W, int KH,
                     int KW, unsigned orient, int algo, void *workspace, size_t workspace_bytes, finc_stream_t stream)
{
    return run(z, w_canon, x, B, G, Cq, H, W, KH, KW, orient, algo, workspace, workspace_bytes, stream, false);
}

int finc_forward_f32(const float *x, const float *w_canon, float *z, int B, int G, int Cq, int H, int W, int KH,
                     int KW, unsigned orient, int algo, void *workspace, size_t workspace_bytes, finc_stream_t stream)
{
    return run(x, w_canon, z, B, G, Cq, H, W, KH, KW, orient, algo, workspace, workspace_bytes, stream, true);
}

static int pack(const float *w_canon, void *packed, int G, int Cq, int KH, int KW, finc_stream_t stream, bool forward)
{
    if (!w_canon || !packed) return FINC_ERR_NULL_POINTER;
    if (int e = check_shape(1, G, Cq, 1, 1, KH, KW)) return e;
    if (forward) {
        if (finc_conv_packed_bytes(G, Cq, KH, KW) == 0) return FINC_ERR_UNSUPPORTED;
        return finc_conv_pack(w_canon, packed, G, Cq, KH, KW, false, (hipStream_t)stream);
    }
    if (finc_mfma_packed_bytes(G, Cq, KH, KW) == 0) return FINC_ERR_UNSUPPORTED;
    return finc_mfma_pack(w_canon, nullptr, nullptr, packed, G, Cq, KH, KW, (hipStream_t)stream);
}

int finc_pack_inverse_weights_f32(const float *w_canon, void *packed, int G, int Cq, int KH, int KW,
                                  finc_stream_t stream)
{
    return pack(w_canon, packed, G, Cq, KH, KW, stream, false);
}

int finc_pack_inverse_weights_affine_f32(const float *w_canon, const float *scale, const float *shift, void *packed,
                                         int G, int Cq, int KH, int KW, finc_stream_t stream)
{
    if (!w_canon || !packed) return FINC_ERR_NULL_POINTER;
    if (int e = check_shape(1, G, Cq, 1, 1, KH, KW)) return e;
    if (finc_mfma_packed_bytes(G, Cq, KH, KW) == 0) return FINC_ERR_UNSUPPORTED;
    return finc_mfma_pack(w_canon, scale, shift, packed, G, Cq, KH, KW, (hipStream_t)stream);
}

int finc_pack_forward_weights_f32(const float *w_canon, void *packed, int G, int Cq, int KH, int KW,
                                  finc_stream_t stream)
{
    return pack(w_canon, packed, G, Cq, KH, KW, stream, true);
}

int finc_pack_forward_weights_affine_f32(const float *w_canon, const float *scale, const float *shift, void *packed,
                                         int G, int Cq, int KH, int KW, finc_stream_t stream)
{
    if (!w_canon || !packed) return FINC_ERR_NULL_POINTER;
    if (int e = check_shape(1, G, Cq, 1, 1, KH, KW)) return e;
    if (finc_conv_packed_bytes(G, Cq, KH, KW) == 0) return FINC_ERR_UNSUPPORTED;
    return finc_conv_pack(w_canon, packed, G, Cq, KH, KW, false, (hipStream_t)stream, scale, shift);
}

static int run_packed(const float *in, const void *packed, float *out, int B, int G, int Cq, int H, int W, int KH,
                      int KW, unsigned orient, finc_stream_t stream, bool forward)
{
    if (!in || !packed || !out) return FINC_ERR_NULL_POINTER;
    if (int e = check_shape(B, G, Cq, H, W, KH, KW)) return e;
    if (misaligned(in) || misaligned(out)) return FINC_ERR_ALIGNMENT;
    if (in == out) return FINC_ERR_BAD_DIMS;
    if (int e = finc_fault_gate(false)) return e;
    FincShape s{B, G, Cq, H, W, KH, KW, orient};
    if (forward) {
        if (!finc_conv_supported(Cq, H, W, KH, KW)) return FINC_ERR_UNSUPPORTED;
        return finc_conv_launch(in, packed, out, s, (hipStream_t)stream);
    }
    if (!finc_mfma_supported(Cq, H, W, KH, KW)) return FINC_ERR_UNSUPPORTED;
    return finc_mfma_launch(in, packed, out, s, (hipStream_t)stream);
}

int finc_inverse_packed_f32(const float *z, const void *packed, float *x, int B, int G, int Cq, int H, int W, int KH,
                            int KW, unsigned orient, finc_stream_t stream)
{
    return run_packed(z, packed, x, B, G, Cq, H, W, KH, KW, orient, stream, false);
}

int finc_forward_packed_f32(const float *x, const void *packed, float *z, int B, int G, int Cq, int H, int W, int KH,
                            int KW, unsigned orient, finc_stream_t stream)
{
    return run_packed(x, packed, z, B, G, Cq, H, W, KH, KW, orient, stream, true);
}

int finc_inverse_affine_supported(int B, int G, int Cq, int H, int W, int KH, int KW)
{
    if (check_shape(B, G, Cq, H, W, KH, KW)) return 0;
    return finc_mfma_affine_takes(FincShape{B, G, Cq, H, W, KH, KW, 0}) ? 1 : 0;
}

int finc_inverse_premultiplied_supported(int B, int G, int Cq, int H, int W, int KH, int KW)
{
    if (check_shape(B, G, Cq, H, W, KH, KW)) return 0;
    return finc_mfma_zpre_takes(FincShape{B, G, Cq, H, W, KH, KW, 0}) ? 1 : 0;
}

int finc_inverse_packed_premultiplied_f32(const float *zp, const void *packed, float *x, int B, int G, int Cq, int H, int W,
                                          int KH, int KW, unsigned orient, finc_stream_t stream)
{
    if (!zp || !packed || !x) return FINC_ERR_NULL_POINTER;
    if (int e = check_shape(B, G, Cq, H, W, KH, KW)) return e;
    if (misaligned(zp) || misaligned(x)) return FINC_ERR_ALIGNMENT;
    if (zp == x) return FINC_ERR_BAD_DIMS;
    FincShape s{B, G, Cq, H, W, KH, KW, orient};
    return finc_mfma_launch(zp, packed, x, s, (hipStream_t)stream, true);
}

static int run_f64(const double *in, const double *w_canon, double *out, int B, int G, int Cq, int H, int W, int KH,
                   int KW, unsigned orient, finc_stream_t stream, bool forward)
{
    if (!in || !w_canon || !out) return FINC_ERR_NULL_POINTER;
    if (int e = check_shape(B, G, Cq, H, W, KH, KW)) return e;
    if (((uintptr_t)in | (uintptr_t)w_canon | (uintptr_t)out) & 7u) return FINC_ERR_ALIGNMENT;
    if (in == out) return FINC_ERR_BAD_DIMS;
    FincShape s{B, G, Cq, H, W, KH, KW, orient};
    return forward ? finc_launch_forward_generic_f64(in, w_canon, out, s, (hipStream_t)stream)
                   : finc_launch_inverse_strict_f64(in, w_canon, out, s, (hipStream_t)stream);
}

int finc_inverse_f64(const double *z, const double *w_canon, double *x, int B, int G, int Cq, int H, int W, int KH,
                     int KW, unsigned orient, finc_stream_t stream)
{
    return run_f64(z, w_canon, x, B, G, Cq, H, W, KH, KW, orient, stream, false);
}

int finc_forward_f64(const double *x, const double *w_canon, double *z, int B, int G, int Cq, int H, int W, int KH,
                     int KW, unsigned orient, finc_stream_t stream)
{
    return run_f64(x, w_canon, z, B, G, Cq, H, W, KH, KW, orient, stream, true);
}

size_t finc_f64_workspace_bytes(int G, int Cq, int KH, int KW)
{
    if (G <= 0 || Cq <= 0 || KH <= 0 || KW <= 0) return 0;
    return finc_f64_packed_bytes(G, Cq, KH, KW);
}

static int run_f64_algo(const double *in, const double *w_canon, double *out, int B, int G, int Cq, int H, int W, int KH, int KW,
                        unsigned orient, int algo, void *workspace, size_t workspace_bytes, finc_stream_t stream, bool forward)
{
    if (!in || !w_canon || !out) return FINC_ERR_NULL_POINTER;
    if (int e = check_shape(B, G, Cq, H, W, KH, KW)) return e;
    if (((uintptr_t)in | (uintptr_t)w_canon | (uintptr_t)out) & 7u) return FINC_ERR_ALIGNMENT;
    if (in == out) return FINC_ERR_BAD_DIMS;
    if (algo != FINC_ALGO_AUTO && algo != FINC_ALGO_STRICT && algo != FINC_ALGO_MFMA) return FINC_ERR_BAD_DIMS;
    FincShape s{B, G, Cq, H, W, KH, KW, orient};
    const bool can = algo != FINC_ALGO_STRICT && finc_f64_supported(s) && workspace && ((uintptr_t)workspace & 7u) == 0 &&
                     workspace_bytes >= finc_f64_packed_bytes(G, Cq, KH, KW);
    if (!can) {
        if (algo == FINC_ALGO_MFMA) return finc_f64_supported(s) ? FINC_ERR_WORKSPACE : FINC_ERR_UNSUPPORTED;
        return forward ? finc_launch_forward_generic_f64(in, w_canon, out, s, (hipStream_t)stream)
                       : finc_launch_inverse_strict_f64(in, w_canon, out, s, (hipStream_t)stream);
    }
    if (int e = finc_fault_gate(false)) return e;
    return finc_f64_launch(in, w_canon, out, workspace, s, forward, (hipStream_t)stream);
}

int finc_inverse_f64_algo(const double *z, const double *w_canon, double *x, int B, int G, int Cq, int H, int W, int KH, int KW,
                          unsigned orient, int algo, void *workspace, size_t workspace_bytes, finc_stream_t stream)
{
    return run_f64_algo(z, w_canon, x, B, G, Cq, H, W, KH, KW, orient, algo, workspace, workspace_bytes, stream, false);
}

int finc_forward_f64_algo(const double *x, const double *w_canon, double *z, int B, int G, int Cq, int H, int W, int KH, int KW,
                          unsigned orient, int algo, void *workspace, size_t workspace_bytes, finc_stream_t stream)
{
    return run_f64_algo(x, w_canon, z, B, G, Cq, H, W, KH, KW, orient, algo, workspace, workspace_bytes, stream, true);
}

int finc_inverse_kernel_variant(int B, int G, int Cq, int H, int W, int KH, int KW, int *info)
{
    if (!info) return FINC_ERR_NULL_POINTER;
    if (int e = check_shape(B, G, Cq, H, W, KH, KW)) return e;
    return finc_mfma_variant(B, G, Cq, H, W, KH, KW, info);
}

int finc_mix_supported_f32(int C) { return (C > 0 && finc_mix_supported(C)) ? 1 : 0; }

int finc_mix_f32(const float *in, const float *mat, const float *bias, float *out, int B, int C, int HW, finc_stream_t stream)
{
    if (!in || !mat || !out) return FINC_ERR_NULL_POINTER;
    if (B <= 0 || C <= 0 || HW <= 0 || C > FINC_MAX_CQ * FINC_MAX_GROUPS) return FINC_ERR_BAD_DIMS;
    if (misaligned(in) || misaligned(mat) || misaligned(out) || (bias && misaligned(bias))) return FINC_ERR_ALIGNMENT;
    if ((size_t)B * C * HW >= ((size_t)1 << 40)) return FINC_ERR_BAD_DIMS;
    if (int e = finc_fault_gate(false)) return e;
    return finc_mix_launch(in, mat, bias, out, B, C, HW, (hipStream_t)stream);
}

int finc_debug_backward_variant(int B, int G, int Cq, int H, int W, int KH, int KW, int *info)
{
    if (!info) return FINC_ERR_NULL_POINTER;
    if (int e = check_shape(B, G, Cq, H, W, KH, KW)) return e;
    FincShape s{B, G, Cq, H, W, KH, KW, 0};
    info[0] = finc_gradw_variant(s);
    int c[3] = {0, 0, 0};
    const bool mfma = finc_conv_variant(B, G, Cq, H, W, KH, KW, c) == FINC_OK;
    info[1] = mfma ? c[0] : 0;
    info[2] = mfma ? c[1] : 0;
    return FINC_OK;
}

int finc_debug_set_forward_form(int form) { return finc_wino_set_form(form); }

int finc_debug_inverse_remainder_images(int B, int G, int Cq, int H, int W, int KH, int KW)
{
    if (B < 1 || G < 1 || Cq < 1 || H < 1 || W < 1 || KH < 1 || KW < 1) return 0;
    return finc_mfma_remainder_images(FincShape{B, G, Cq, H, W, KH, KW, 0});
}

int finc_debug_row_chunks(long long units, long long slots, int H, int min_rows, int extra, int second_tenant)
{
    if (units < 1 || slots < 1 || H < 1 || min_rows < 1 || extra < 0 || second_tenant < 1 || second_tenant > 16) return 0;
    return finc_row_chunks(units, slots, H, min_rows, extra, second_tenant);
}

int finc_debug_hlp_timeouts(unsigned *h_count)
{
    if (!h_count) return FINC_ERR_NULL_POINTER;
    unsigned a = 0, b = 0;
    if (int e = finc_mfma_hlp_timeouts(&a)) return e;          // helper-wave protocol (finc_mfma.hip)
    if (int e = finc_split_timeouts_count(&b)) return e;       // band split's progress words (finc_split.hip)
    *h_count = a + b;
    return FINC_OK;
}

int finc_debug_inverse_table_row(int row, int *info)
{
    if (!info) return FINC_ERR_NULL_POINTER;
    return finc_mfma_table_row(row, info);
}

size_t finc_backward_workspace_bytes(int B, int G, int Cq, int H, int W, int KH, int KW)
{
    if (B <= 0 || G <= 0 || Cq <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0) return 256;
    FincShape s{B, G, Cq, H, W, KH, KW, 0};
    const size_t n = align256(finc_conv_packed_bytes(G, Cq, KH, KW)) + finc_gradw_workspace_bytes(s);
    return n < 256 ? 256 : n;
}

int finc_backward_f32(const float *grad_z, const float *x, const float *w_canon, float *grad_x, float *grad_w_canon,
                      int B, int G, int Cq, int H, int W, int KH, int KW, unsigned orient, void *workspace,
                      size_t workspace_bytes, finc_stream_t stream)
{
    if (!grad_z) return FINC_ERR_NULL_POINTER;
    if (grad_x && !w_canon) return FINC_ERR_NULL_POINTER;
    if (grad_w_canon && !x) return FINC_ERR_NULL_POINTER;
    if (int e = check_shape(B, G, Cq, H, W, KH, KW)) return e;
    if (grad_x == grad_z) return FINC_ERR_BAD_DIMS;
    if (int e = finc_fault_gate(false)) return e;
    FincShape s{B, G, Cq, H, W, KH, KW, orient};
    hipStream_t st = (hipStream_t)stream;
    const size_t pk = align256(finc_conv_packed_bytes(G, Cq, KH, KW));
    // grad_x: the same conv on the H- and W-flipped image with in/out channels transposed (finc_conv.hip)
    if (grad_x && workspace && finc_conv_supported(Cq, H, W, KH, KW) && pk > 0 && workspace_bytes >= pk) {
        if (int e = finc_conv_pack(w_canon, workspace, G, Cq, KH, KW, true, st)) return e;
        FincShape sb = s;
        sb.orient = orient ^ ((G >= 16) ? 0xFFFFFFFFu : ((1u << (2 * G)) - 1u));
        if (int e = finc_conv_launch(grad_z, workspace, grad_x, sb, st)) return e;
        grad_x = nullptr;
    }
    // grad_w: MFMA strip kernel with the pixels on K + reduce (+ corner-tap mask)
    const size_t gwb = finc_gradw_workspace_bytes(s);
    if (grad_w_canon && workspace && gwb > 0 && workspace_bytes >= pk + gwb) {
        const int e = finc_gradw_launch(grad_z, x, grad_w_canon, (char *)workspace + pk, s, st);
        if (e == FINC_OK) grad_w_canon = nullptr;
        else if (e != FINC_ERR_UNSUPPORTED) return e;       // (unsupported for THIS call, e.g. float-aligned views: the direct kernel below)
    }
    if (!grad_x && !grad_w_canon) return FINC_OK;
    return finc_launch_backward_generic(grad_z, x, w_canon, grad_x, grad_w_canon, s, st);
}

} // extern "C"
