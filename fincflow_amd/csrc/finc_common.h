// Shared host/device helpers for libfinc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/finc.h"
#include "finc_experiment.h"   // the measurement knobs: an error if one is set without -DFINC_EXPERIMENT

#define FINC_FLIP_W 1u
#define FINC_FLIP_H 2u

__host__ __device__ inline unsigned finc_group_orient(unsigned orient, int g) { return (orient >> (2 * g)) & 3u; }

// canonical (h,w) of a group -> offset inside one H*W plane
__device__ inline int finc_pix(int H, int W, unsigned o, int h, int w)
{
    int hh = (o & FINC_FLIP_H) ? H - 1 - h : h;
    int ww = (o & FINC_FLIP_W) ? W - 1 - w : w;
    return hh * W + ww;
}

// Recorded by every failing HIP call; read through finc_last_hip_error().
void finc_set_hip_error(hipError_t e);

#define FINC_HIP_TRY(expr)                     \
    do {                                       \
        hipError_t _e = (expr);                \
        if (_e != hipSuccess) {                \
            finc_set_hip_error(_e);            \
            return FINC_ERR_LAUNCH;            \
        }                                      \
    } while (0)

#define FINC_CHECK_LAUNCH() FINC_HIP_TRY(hipGetLastError())

// hipFuncAttributeMaxDynamicSharedMemorySize = 160 KiB, once per (device, kernel), thread-safe (finc_abi.hip)
int finc_ensure_dynamic_lds(const void *fn, size_t bytes);

// Protocol faults (helper-wave waits that gave up): finc_fault_gate returns FINC_ERR_LAUNCH if the device's fault word is
// set (sticky; no synchronisation -- the word lives in mapped host memory); `arm` allocates and publishes the word on the
// first call per device.  finc_abi.hip owns the per-device table, finc_mfma.hip the device symbol.
int finc_fault_gate(bool arm, hipStream_t st = nullptr);
int finc_mfma_arm_fault_word(unsigned *device_ptr_to_host_word);

// Run-time A/B switches (FINC_NO_HLP, FINC_WINO_FORM, FINC_SPLIT_MAX ...: scripts/ only) are read through finc_env, which
// remembers every switch that was found SET: finc_runtime_switches() reports them, bench.py refuses a judged run with any.
const char *finc_env(const char *name);
int finc_wino_form_override();   // 0 = the library's choice (finc_debug_set_forward_form)

// Row chunks of the forward kernels that run ONE wave per SIMD (F(4,3), its M-split, F(2,5)): a launch of `units` strips (waves or
// workgroups), of which the chip holds `slots` at a time, is cut into row chunks so that every slot has work; a chunk walks its rows
// plus `extra` rows of operands above them.  All units take the same time, so the chip works through units x chunks in ROUNDS of
// `slots`, and the cost of a split is rounds x (rows per chunk + extra).  The count that minimises it is returned (the smallest one
// among equals; chunks of at least `min_rows` rows).  Measured on the F(4,3) forward of c3, 64x64 (profiles/r05/notes/row_chunks.txt):
// B = 96 (384 strips): 3 chunks = 1,152 waves = two rounds of 24 rows 133 us, 5 chunks = two rounds of 15 rows 97 us; B = 160: 2 chunks
// 203 us, 3 chunks 155; B = 384: 1 chunk (two rounds of 66 rows) 395 us, 2 chunks (three rounds of 34) 355.
// `second_tenant`: the kernels that run TWO waves per SIMD (strip kernel, F(2,3)) pass 14 -- once a launch has more waves than SIMDs the
// tenants fill each other's issue gaps, a row costs 14/16 of what it costs a lone wave (F(2,3), c3 shape, B = 24: 5 chunks = 960 lone
// waves 34.5 us, 10 chunks = 1,920 waves in pairs 34.4; 4 against 8 chunks: 39.8 both) -- and count their rounds in SIMDs all the same:
// two tenants share one SIMD's MFMA pipe.  16 = no such effect (one wave per SIMD is all the registers allow).
inline int finc_row_chunks(long long units, long long slots, int H, int min_rows, int extra, int second_tenant = 16)
{
    const int maxc = H / min_rows > 0 ? H / min_rows : 1;
    long long best = -1;
    int nrc = 1;
    for (int c = 1; c <= maxc; ++c) {
        const int rc = (H + c - 1) / c;
        if ((H + rc - 1) / rc != c) continue;                     // (the same split as a smaller count)
        const long long rounds = (units * c + slots - 1) / slots;
        const long long cost = rounds * (rc + extra) * (units * c > slots ? second_tenant : 16);
        if (best < 0 || cost < best) { best = cost; nrc = c; }
    }
    return nrc;
}

// the measurement knobs each kernel translation unit was built with (finc_experiment.h); finc_build_flags() ORs them
unsigned finc_build_flags_mfma();
unsigned finc_build_flags_split();
unsigned finc_build_flags_conv();
unsigned finc_build_flags_gradw();
unsigned finc_build_flags_mix();
unsigned finc_build_flags_generic();
unsigned finc_build_flags_probe();

struct FincShape {
    int B, G, Cq, H, W, KH, KW;
    unsigned orient;
};

// ---- generic (reference-order) kernels: finc_generic.hip ----
int finc_launch_inverse_strict(const float *z, const float *wc, float *x, const FincShape &s, hipStream_t st);
int finc_launch_forward_generic(const float *x, const float *wc, float *z, const FincShape &s, hipStream_t st);
int finc_launch_inverse_strict_f64(const double *z, const double *wc, double *x, const FincShape &s, hipStream_t st);
int finc_launch_forward_generic_f64(const double *x, const double *wc, double *z, const FincShape &s, hipStream_t st);
int finc_launch_backward_generic(const float *gz, const float *x, const float *wc, float *gx, float *gw,
                                 const FincShape &s, hipStream_t st);

// ---- inverse, MFMA wavefront kernel: finc_mfma.hip ----
// rows x Win floats -> rows x Wout floats (Wout > Win: zero fill on the right; Wout < Win: crop)
int finc_launch_repitch(const float *in, float *out, long long rows, int Win, int Wout, hipStream_t st);

bool finc_mfma_supported(int Cq, int H, int W, int KH, int KW);
size_t finc_mfma_packed_bytes(int G, int Cq, int KH, int KW);
// scale / shift [G*Cq] or nullptr: the affine map z = scale*y + shift folded in front of the inverse
int finc_mfma_pack(const float *wc, const float *scale, const float *shift, void *packed, int G, int Cq, int KH, int KW,
                   hipStream_t st);
// zpre: `in` is Linv * z already (the caller's channel mix applied blockdiag(Linv)): the helper-wave form without its z-term
int finc_mfma_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st, bool zpre = false);
bool finc_mfma_zpre_takes(const FincShape &s);
// can an affine map with a SHIFT be folded into the bank this problem set runs on? (not on finc_big.hip's kernels: the big
// banks and the wide-map takeover of the 33..64-channel banks carry a scale only)
bool finc_mfma_affine_takes(const FincShape &s);
// info[0..7] = {Cq padded, waves per problem, problems per workgroup, 32-byte I/O (1) or 16-byte (0), LDS bytes of a
// workgroup, workgroups, index into the instantiation table, rows of the table}; FINC_ERR_UNSUPPORTED if none applies
int finc_mfma_variant(int B, int G, int Cq, int H, int W, int KH, int KW, int *info);
// info[0..5] = {Cq padded, KH, KW, waves per problem, problems per workgroup, max_problems (0 = no limit)}
int finc_mfma_table_row(int row, int *info);
// waits of the helper-wave protocol that ran out of their spin budget since the library was loaded (synchronous copy)
int finc_mfma_hlp_timeouts(unsigned *count);

// Cq padded as the packed bank of (Cq, KH, KW) has it, or 0 when the shape has no MFMA instantiation
int finc_mfma_packed_cqp(int Cq, int KH, int KW);

// ---- inverse for the under-filled chip, role-split kernel: finc_split.hip (same packed bank as the wavefront kernel) ----
// finc_big.hip: 3x3 banks beyond the wavefront kernel's table (64 < Cq <= 96), 8 waves per problem, each owns 12 output channels
bool finc_big_bank(int Cq, int KH, int KW);
bool finc_big_wide_bank(int Cq, int KH, int KW);   // a bank of the wavefront kernel's table that finc_big.hip takes over on maps too wide for it
bool finc_big_supported(int Cq, int H, int W, int KH, int KW);
size_t finc_big_packed_bytes(int G, int Cq, int KH, int KW);
int finc_big_pack(const float *wc, const float *scale, const float *shift, void *packed, int G, int Cq, int KH, int KW, hipStream_t st);
int finc_big_info(const FincShape &s, int *waves, int *lds_bytes, int *cqp);
int finc_big_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st);
size_t finc_bigfwd_packed_bytes(int G, int Cq, int KH, int KW);         // 0: no big-bank forward for this bank
bool finc_bigfwd_takes(const float *in, const float *out, const FincShape &s);
int finc_bigfwd_pack(const float *wc, void *packed, int G, int Cq, int KH, int KW, bool transpose, hipStream_t st, const float *scale,
                     const float *shift);
int finc_bigfwd_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st);
unsigned finc_build_flags_big();
bool finc_split_takes(const FincShape &s);       // this problem set runs on the role-split kernel
int finc_split_info(const FincShape &s, int *waves, int *lds_bytes, int *steps, int *nwg = nullptr);   // nwg: workgroups per problem (band split)
int finc_split_prepare(hipStream_t st);          // allocates the band split's progress words for the current device (not inside a capture)
int finc_split_timeouts_count(unsigned *count);  // progress waits of the band split that gave up (must be 0)
unsigned *finc_fault_device_word();              // device pointer to the current device's (armed) fault word, or nullptr
int finc_split_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st);
// finc_chain.hip: the short-step form of the same idea for the banks of up to 16 channels (output channels, not taps, shared out
// over the preparing waves); finc_split_launch / finc_split_info hand over to it
bool finc_chain_takes(const FincShape &s);        // (can run it; finc_split_uses_chain: does run it)
bool finc_split_uses_chain(const FincShape &s);
int finc_chain_info(const FincShape &s, int *waves, int *lds_bytes, int *steps);
int finc_chain_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st);
unsigned finc_build_flags_chain();

// ---- banks no register-resident kernel holds: the streaming-bank kernel, inverse and forward forms: finc_stream.hip ----
bool finc_stream_bank_ok(int Cq, int KH, int KW);                      // within its limits (Cq <= 256, KH, KW <= 7)
bool finc_stream_supported(int Cq, int H, int W, int KH, int KW, bool inverse);
size_t finc_stream_packed_bytes(int G, int Cq, int KH, int KW, bool inverse);
int finc_stream_pack(const float *wc, const float *scale, const float *shift, void *packed, int G, int Cq, int KH, int KW, bool inverse,
                     bool transpose, hipStream_t st);
int finc_mfma_remainder_images(const FincShape &s);
int finc_stream_info(const FincShape &s, bool inverse, int *cqp, int *lds, int *steps, int *waves = nullptr, int *one_wave_vec = nullptr);
int finc_stream_launch(const float *in, const void *packed, float *out, const FincShape &s, bool inverse, hipStream_t st);
unsigned finc_build_flags_stream();

// ---- forward / grad-input, MFMA strip kernel: finc_conv.hip ----
bool finc_conv_supported(int Cq, int H, int W, int KH, int KW);
size_t finc_conv_packed_bytes(int G, int Cq, int KH, int KW);
// scale / shift [G*Cq] or nullptr: z' = scale * conv(x) + shift folded into the bank (the affine layer BEHIND the conv)
int finc_conv_pack(const float *wc, void *packed, int G, int Cq, int KH, int KW, bool transpose, hipStream_t st,
                   const float *scale = nullptr, const float *shift = nullptr);
int finc_conv_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st);
// ---- 3x3 forward / grad-input with 1.5x fewer multiplies (Winograd F(2,3) along W): finc_wino.hip; its bank sits behind the
// strip kernels' in the packed buffer
size_t finc_wino_packed_bytes(int G, int Cq, int KH, int KW);          // 0: no Winograd kernel for this bank
bool finc_wino_takes(const float *in, const float *out, const FincShape &s);
int finc_wino_pack(const float *wc, void *packed, int G, int Cq, bool transpose, hipStream_t st, const float *scale, const float *shift);
int finc_wino_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st);
bool finc_wino_disabled();                                          // FINC_NO_WINO / finc_debug_set_forward_form(1)
int finc_wino_pack_bank(const float *wc, float *packed, int G, int Cq, int MT, int MTB, int NK, int NF, bool transpose, hipStream_t st,
                        const float *scale, const float *shift);
// 3x3 banks of 28 .. 64 channels: F(4,3) along W, M-split over the waves of a workgroup (finc_wino4m.hip)
size_t finc_wino4m_packed_bytes(int G, int Cq, int KH, int KW);        // 0: no such kernel for this bank
bool finc_wino4m_takes(const float *in, const float *out, const FincShape &s);
int finc_wino4m_pack(const float *wc, void *packed, int G, int Cq, bool transpose, hipStream_t st, const float *scale, const float *shift);
int finc_wino4m_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st);
unsigned finc_build_flags_wino4m();
int finc_wino_form(const FincShape &s);                                // 2: F(2,3), 4: F(4,3) -- which of the two a call runs
int finc_wino_set_form(int form);                                      // 0 library's choice, 1 strip kernel, 2 F(2,3), 4 F(4,3)
unsigned finc_build_flags_wino();
// ---- 5x5 forward / grad-input with 0.6 x the multiplies (Winograd F(2,5) along W): finc_wino5.hip; bank behind the strip kernels'
size_t finc_wino5_packed_bytes(int G, int Cq, int KH, int KW);         // 0: no such kernel for this bank
bool finc_wino5_takes(const float *in, const float *out, const FincShape &s);
int finc_wino5_pack(const float *wc, void *packed, int G, int Cq, bool transpose, hipStream_t st, const float *scale, const float *shift);
int finc_wino5_launch(const float *in, const void *packed, float *out, const FincShape &s, hipStream_t st);
unsigned finc_build_flags_wino5();
size_t finc_gradw_workspace_bytes(const FincShape &s); // 0: no MFMA grad-weight kernel for this shape
int finc_gradw_launch(const float *gz, const float *x, float *gw, void *workspace, const FincShape &s, hipStream_t st);
int finc_gradw_variant(const FincShape &s);   // 0 direct, 1 dword MFMA, 2 staged, 3 tiled, 4 Winograd, 5 Winograd tiled
// info[0..2] = {waves per strip (K-split), staged form (1) or dword form (0), strips per slab}; FINC_ERR_UNSUPPORTED: direct kernel
int finc_conv_variant(int B, int G, int Cq, int H, int W, int KH, int KW, int *info);

// ---- double precision on the matrix cores: finc_f64.hip (the c2 / c3 class of banks: Cq <= 24 at 3x3, <= 32 at 2x2) ----
bool finc_f64_supported(const FincShape &s);
size_t finc_f64_packed_bytes(int G, int Cq, int KH, int KW);          // 0: no such kernel for this bank
// packs the bank into `packed` (finc_f64_packed_bytes) and runs the inverse or the forward on it, all on `st`
int finc_f64_launch(const double *in, const double *wc, double *out, void *packed, const FincShape &s, bool forward, hipStream_t st);
unsigned finc_build_flags_f64();

// ---- per-pixel channel mixing (1x1 conv + folded affine): finc_mix.hip ----
bool finc_mix_supported(int C);
int finc_mix_launch(const float *in, const float *mat, const float *bias, float *out, int B, int C, int HW, hipStream_t st);
