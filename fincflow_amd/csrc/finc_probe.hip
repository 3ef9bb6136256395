// Clock probe (diagnostics; include/finc.h: finc_debug_clock_probe_begin / _end).
//
// The chip lowers its shader clock under load, by an amount that depends on the data (MI355X_MICROARCH.md, "DVFS
// give-back"), so two legs of a benchmark that launch the SAME kernel can differ in wall time with no code difference
// (BENCH_r03: 0.401 ms on z = forward(x), 0.509 ms on z ~ N(0,1)).  The in-kernel clock is
//     d s_memtime / d s_memrealtime x 100 MHz
// (s_memtime ticks with the shader clock, s_memrealtime at a constant 100 MHz).  Rather than stamping the measured
// kernel -- a stamp costs issue slots and a diagnostic build is a different binary -- ONE extra wavefront samples the two
// counters a few thousand times per second from its own stream while the leg under test runs beside it: it needs no LDS
// and 2 of a SIMD's wave slots are all the inverse kernel uses, so it is resident from the first launch on.  The probe
// has three exits, each of which every (= its one) wave reaches: the host's stop word, its sample budget, and a wall-time
// bound counted on the constant-rate counter.
#include "finc_common.h"

#include <mutex>

namespace {

constexpr int MAX_SAMPLES = 16384;

__global__ __launch_bounds__(64) void finc_clock_probe_kernel(unsigned long long *__restrict__ out, unsigned *__restrict__ count,
                                                              int max_samples, unsigned period_rt, unsigned long long max_rt,
                                                              const unsigned *stop_host)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    int n = 0;
    while (n < max_samples) {
        const unsigned long long m = __builtin_amdgcn_s_memtime();
        const unsigned long long r = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) { out[2 * n] = m; out[2 * n + 1] = r; }
        ++n;
        if (r - r0 > max_rt) break;
        if (__hip_atomic_load(stop_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
        while (__builtin_amdgcn_s_memrealtime() - r < period_rt) __builtin_amdgcn_s_sleep(64);
    }
    if (threadIdx.x == 0) *count = (unsigned)n;
}

struct Probe {
    hipStream_t stream = nullptr;
    unsigned long long *d_out = nullptr;
    unsigned *d_count = nullptr;
    unsigned *h_stop = nullptr, *d_stop = nullptr;
    bool running = false;
};
constexpr int MAX_DEV = 64;
Probe g_probe[MAX_DEV];
std::mutex g_probe_mutex;

} // namespace

extern "C" int finc_debug_clock_probe_begin(int period_us, int max_ms)
{
    if (period_us < 20 || period_us > 100000 || max_ms < 1 || max_ms > 10000) return FINC_ERR_BAD_DIMS;
    int dev = 0;
    FINC_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEV) return FINC_ERR_BAD_DIMS;
    std::lock_guard<std::mutex> lk(g_probe_mutex);
    Probe &p = g_probe[dev];
    if (p.running) return FINC_ERR_BAD_DIMS;
    if (!p.stream) {
        // (everything is allocated into locals and committed only when every step has succeeded: a half-built probe would be launched
        // with null pointers by the next call -- ADVICE r4)
        hipStream_t st = nullptr;
        unsigned long long *d_out = nullptr;
        unsigned *d_count = nullptr, *h_stop = nullptr, *d_stop = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void **)&d_out, sizeof(unsigned long long) * 2 * MAX_SAMPLES);
        if (e == hipSuccess) e = hipMalloc((void **)&d_count, sizeof(unsigned));
        if (e == hipSuccess) e = hipHostMalloc((void **)&h_stop, sizeof(unsigned), hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&d_stop, h_stop, 0);
        if (e != hipSuccess) {
            finc_set_hip_error(e);
            if (h_stop) (void)hipHostFree(h_stop);
            if (d_count) (void)hipFree(d_count);
            if (d_out) (void)hipFree(d_out);
            if (st) (void)hipStreamDestroy(st);
            return FINC_ERR_LAUNCH;
        }
        p.d_out = d_out; p.d_count = d_count; p.h_stop = h_stop; p.d_stop = d_stop;
        p.stream = st;
    }
    *(volatile unsigned *)p.h_stop = 0;
    FINC_HIP_TRY(hipMemsetAsync(p.d_count, 0, sizeof(unsigned), p.stream));
    long long want = (long long)max_ms * 1000 / period_us + 2;
    const int max_samples = want > MAX_SAMPLES ? MAX_SAMPLES : (int)want;
    hipLaunchKernelGGL(finc_clock_probe_kernel, dim3(1), dim3(64), 0, p.stream, p.d_out, p.d_count, max_samples, (unsigned)period_us * 100u,
                       (unsigned long long)max_ms * 100000ull, (const unsigned *)p.d_stop);
    FINC_CHECK_LAUNCH();
    p.running = true;
    return FINC_OK;
}

// h_stats[6] = {mean MHz over the whole window (total ticks / total time), min and max over the sampling intervals, number of
// samples, seconds covered, median MHz}
extern "C" int finc_debug_clock_probe_end(double *h_stats)
{
    if (!h_stats) return FINC_ERR_NULL_POINTER;
    int dev = 0;
    FINC_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEV) return FINC_ERR_BAD_DIMS;
    std::lock_guard<std::mutex> lk(g_probe_mutex);
    Probe &p = g_probe[dev];
    if (!p.running) return FINC_ERR_BAD_DIMS;
    *(volatile unsigned *)p.h_stop = 1;
    FINC_HIP_TRY(hipStreamSynchronize(p.stream));
    p.running = false;                                  // (only once the probe kernel is known to have ended)
    unsigned n = 0;
    FINC_HIP_TRY(hipMemcpy(&n, p.d_count, sizeof(unsigned), hipMemcpyDeviceToHost));
    for (int i = 0; i < 6; ++i) h_stats[i] = 0.0;
    h_stats[3] = (double)n;
    if (n < 2 || n > (unsigned)MAX_SAMPLES) return FINC_OK;
    static unsigned long long buf[2 * MAX_SAMPLES];
    FINC_HIP_TRY(hipMemcpy(buf, p.d_out, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost));
    double lo = 1e30, hi = 0.0;
    static double mhz[MAX_SAMPLES];
    int k = 0;
    for (unsigned i = 1; i < n; ++i) {
        const double dm = (double)(buf[2 * i] - buf[2 * i - 2]), dr = (double)(buf[2 * i + 1] - buf[2 * i - 1]);
        if (dr <= 0.0) continue;
        const double f = dm / dr * 100.0;
        mhz[k++] = f;
        lo = f < lo ? f : lo;
        hi = f > hi ? f : hi;
    }
    const double tm = (double)(buf[2 * (n - 1)] - buf[0]), tr = (double)(buf[2 * (n - 1) + 1] - buf[1]);
    h_stats[0] = tr > 0.0 ? tm / tr * 100.0 : 0.0;
    h_stats[1] = k ? lo : 0.0;
    h_stats[2] = hi;
    h_stats[4] = tr / 1e8;
    if (k) {                                   // median by partial selection (k <= 16383)
        for (int i = 0; i <= k / 2; ++i) {
            int m = i;
            for (int j = i + 1; j < k; ++j) m = mhz[j] < mhz[m] ? j : m;
            const double t = mhz[i]; mhz[i] = mhz[m]; mhz[m] = t;
        }
        h_stats[5] = mhz[k / 2];
    }
    return FINC_OK;
}

unsigned finc_build_flags_probe() { return FINC_BUILD_FLAGS; }
