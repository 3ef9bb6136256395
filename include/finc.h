/*
 * finc.h -- C ABI of the MI355X-native FInC Flow hot path (libfinc_hip.so).
 *
 * This is the drop-in boundary.  It replaces the reference's one native entry
 * point
 *
 *     m.def("inverse", &cinc_inverse_level2)            cinc_cuda_level2.cpp:30-32
 *     std::vector<Tensor> cinc_inverse_level2(Tensor input  [B,C,H,W],
 *                                             Tensor kernel [G*Cq,Cq,KH,KW],
 *                                             Tensor output [B,C,H,W])   :19-28
 *
 * and the cuDNN convolution behind PaddedConv2d.forward (layers/conv.py:102-107)
 * with plain-pointer functions: no torch types, no global state besides a
 * per-process table of (device, kernel) attributes and one 4-byte flag word per
 * device (finc_check_invariant_f32).  Every function is stream-ordered
 * and asynchronous unless its comment says otherwise; every pointer is a
 * DEVICE pointer unless its name starts with `h_`.
 *
 * Data layout (the reference's own, fastflow.py:78-100):
 *   activations  fp32 NCHW contiguous, C = G*Cq; group g owns channels
 *                [g*Cq, (g+1)*Cq).
 *   w_canon      fp32 [G*Cq][Cq][KH][KW], TL-canonical -- exactly the `kernel`
 *                tensor of the reference op (fastflow.py:79-84).
 *   w_stored     the same shape in state-dict form: each group's bank flipped
 *                per its order (layers/conv.py:72-79).
 *   orient       2 bits per group g at bits [2g, 2g+1]: bit0 = W-flipped (TR),
 *                bit1 = H-flipped (BL), both = BR.  FastFlowUnit = 0xE4
 *                (TL,TR,BL,BR; fastflow.py:24-27).  The reference flips the
 *                activation chunks in PyTorch around the op (fastflow.py:85-100);
 *                here the kernels index the flipped pixel directly, so the
 *                caller passes the un-flipped tensors.
 *
 * Invariant (layers/conv.py:63-70): w_canon[c][c][KH-1][KW-1] == 1 and
 * w_canon[c][kc>c][KH-1][KW-1] == 0.  finc_check_invariant_f32 verifies it.
 */
#ifndef FINC_H
#define FINC_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *finc_stream_t; /* a hipStream_t; NULL = the default stream */

enum finc_status {
    FINC_OK = 0,
    FINC_ERR_NULL_POINTER = 1,
    FINC_ERR_BAD_DIMS = 2,     /* non-positive dims, G > 16, Cq > FINC_MAX_CQ ... */
    FINC_ERR_UNSUPPORTED = 3,  /* the requested algo has no instantiation for this shape */
    FINC_ERR_WORKSPACE = 4,    /* workspace NULL or smaller than finc_workspace_bytes() */
    FINC_ERR_LAUNCH = 5,       /* a HIP call failed; see finc_last_hip_error() */
    FINC_ERR_INVARIANT = 6,    /* unit-lower-triangular corner tap violated */
    FINC_ERR_ALIGNMENT = 7     /* a pointer is not 4-byte (fp32) aligned */
};

enum finc_algo {
    FINC_ALGO_AUTO = 0,   /* fastest kernel that supports the shape */
    FINC_ALGO_STRICT = 1, /* reference visitation + term order; inverse is bit-exact with the
                             fp32 CPU restatement (cinc_cuda_kernel_level2.cu:59-72)        */
    FINC_ALGO_MFMA = 2    /* wavefront-per-(image,group) MFMA kernel; FINC_ERR_UNSUPPORTED
                             when the shape has no instantiation                             */
};

#define FINC_MAX_GROUPS 16
#define FINC_MAX_CQ 256
#define FINC_ORIENT_FASTFLOW 0xE4u

int finc_version(void);
/* Bit mask of the measurement knobs (csrc/finc_experiment.h: ablations, stamps, reduced tables ...) the library was built
 * with.  0 for a product build; anything else computes wrong results by design and must never be shipped or benchmarked. */
unsigned finc_build_flags(void);
const char *finc_status_string(int status);
/* hipGetErrorString of the last failing HIP call on this thread (never NULL). */
const char *finc_last_hip_error(void);

/* Replaces fastflow.py:79-84 (four torch.flip + cat).  w_stored -> w_canon; may not alias. */
int finc_canonicalize_weights_f32(const float *w_stored, float *w_canon, int G, int Cq, int KH, int KW,
                                  unsigned orient, finc_stream_t stream);

/* SYNCHRONOUS (one 4-byte D2H copy on `stream`; the device flag word is allocated once per device and kept).
 * Returns FINC_OK or FINC_ERR_INVARIANT.  Call once per weight version, not per step. */
int finc_check_invariant_f32(const float *w_canon, int G, int Cq, int KH, int KW, finc_stream_t stream);

/* Scratch the AUTO/MFMA algos need (packed filter fragments); 0 is never returned. */
size_t finc_workspace_bytes(int G, int Cq, int KH, int KW);

/*
 * Workspace finc_inverse_f32 can use for this problem: finc_workspace_bytes(), plus -- when W is not a multiple
 * of 4, which the MFMA inverse cannot stream -- room for a zero-padded copy of z and of x (row pitch rounded up to
 * 8 floats).  With at least this much workspace FINC_ALGO_AUTO solves the padded copy (exact, two extra copies);
 * with less it falls back to FINC_ALGO_STRICT for such widths.  The reference has no counterpart: its kernel
 * takes any width (cinc_cuda_kernel_level2.cu:49-56).
 */
size_t finc_inverse_workspace_bytes(int B, int G, int Cq, int H, int W, int KH, int KW);

/* Which algo FINC_ALGO_AUTO resolves to for this shape (FINC_ALGO_STRICT or FINC_ALGO_MFMA). */
int finc_inverse_algo_for(int Cq, int H, int W, int KH, int KW);
int finc_forward_algo_for(int Cq, int H, int W, int KH, int KW);

/*
 * x = inverse(z): the reference op `inverse(input=z, kernel=w_canon, output=x)`
 * (cinc_cuda_level2.cpp:19-28 -> cinc_cuda_kernel_level2.cu:77-136) including the
 * chunk flips FastFlowUnit.reverse_level2 wraps around it (fastflow.py:85-100).
 * Unlike the reference: `x` need not be zero-filled, B is not limited to 1024,
 * one launch per call instead of (H+W-1)*Cq launches + device syncs.
 * z and x may not alias.
 */
int finc_inverse_f32(const float *z, const float *w_canon, float *x, int B, int G, int Cq, int H, int W,
                     int KH, int KW, unsigned orient, int algo, void *workspace, size_t workspace_bytes,
                     finc_stream_t stream);

/*
 * z = forward(x): FastFlowUnit.forward (fastflow.py:31-50) = per group
 * F.pad on one corner + bias-free cross-correlation (layers/conv.py:102-107),
 * without the pad/chunk/cat copies.  logdet of this layer is identically 0
 * (layers/conv.py:106,220-221), so nothing is written for it.
 * x and z may not alias.
 */
int finc_forward_f32(const float *x, const float *w_canon, float *z, int B, int G, int Cq, int H, int W,
                     int KH, int KW, unsigned orient, int algo, void *workspace, size_t workspace_bytes,
                     finc_stream_t stream);

/*
 * Split form of the two calls above for callers that run many steps on one
 * weight version (sampling): pack once, then launch with the packed buffer.
 * `packed` must hold finc_workspace_bytes() bytes.
 */
int finc_pack_inverse_weights_f32(const float *w_canon, void *packed, int G, int Cq, int KH, int KW,
                                  finc_stream_t stream);
int finc_pack_forward_weights_f32(const float *w_canon, void *packed, int G, int Cq, int KH, int KW,
                                  finc_stream_t stream);
/*
 * SURVEY 8 f3 -- the per-channel affine layer that precedes the unit in the reverse chain folded into the bank:
 * with fragments packed by this call, finc_inverse_packed_f32(y, ...) returns inverse(scale * y + shift), i.e.
 * FastFlowUnit.reverse(ActNorm.reverse(y)) (layers/actnorm.py:39-52: scale = exp(log_scale), shift = translation;
 * called back to back by FlowSequential.sample, layers/flowsequential.py:89-115) in the one launch and at the
 * per-step cost of the plain inverse: Linv*diag(scale) replaces Linv as the z-term and Linv*shift is the
 * accumulators' initial value.  scale / shift: [G*Cq] device floats, either may be NULL (identity).
 * FINC_ERR_UNSUPPORTED when the shape has no MFMA instantiation (there is no strict twin of this call).
 * A SHIFT cannot ride on every kernel: the big banks (64 < Cq <= 96) refuse it here (FINC_ERR_UNSUPPORTED), and the
 * 33..64-channel banks carry it on every map except those too wide for the wavefront kernel's forms (Cq = 50 at 256
 * columns ...), which run on the big-bank kernel.  Packing does not know the map, so ask first:
 * finc_inverse_affine_supported(B, G, Cq, H, W, KH, KW) == 1 iff finc_inverse_packed_f32 on THIS problem set accepts a
 * bank packed with a shift.  A launch that does not (a wide map on a shift-carrying bank) returns FINC_ERR_UNSUPPORTED --
 * it never computes with a partial bank -- and the caller runs the affine layer as its own launch.
 */
int finc_inverse_affine_supported(int B, int G, int Cq, int H, int W, int KH, int KW);
int finc_pack_inverse_weights_affine_f32(const float *w_canon, const float *scale, const float *shift, void *packed,
                                         int G, int Cq, int KH, int KW, finc_stream_t stream);
/*
 * The same fold for the forward direction: with fragments packed by this call, finc_forward_packed_f32(x, ...) returns
 * scale * forward(x) + shift per output channel, i.e. ActNorm.forward(FastFlowUnit.forward(x)) (layers/actnorm.py:39-46:
 * scale = exp(-log_scale), shift = -translation * exp(-log_scale); FlowSequential.forward calls them back to back,
 * layers/flowsequential.py:21-44) in the one launch: the filter rows carry the scale, the accumulators start from the
 * shift.  The layer's log-determinant is the caller's business (it does not depend on the data).
 */
int finc_pack_forward_weights_affine_f32(const float *w_canon, const float *scale, const float *shift, void *packed,
                                         int G, int Cq, int KH, int KW, finc_stream_t stream);
int finc_inverse_packed_f32(const float *z, const void *packed, float *x, int B, int G, int Cq, int H, int W,
                            int KH, int KW, unsigned orient, finc_stream_t stream);
int finc_forward_packed_f32(const float *x, const void *packed, float *z, int B, int G, int Cq, int H, int W,
                            int KH, int KW, unsigned orient, finc_stream_t stream);
/*
 * SURVEY 8 f3, second half -- the inverse of a unit whose input the caller has ALREADY multiplied by blockdiag(Linv_g),
 * Linv_g = inverse(w_canon[g][:, :, KH-1, KW-1]) (the unit lower triangular tap of the pixel itself, layers/conv.py:63-70):
 * in the reverse chain of a flow step (fastflow/fastflow_cifar.py:46-55: ... Conv1x1.reverse -> [ActNorm.reverse] ->
 * FastFlowUnit.reverse) the channel mix in front of the unit is a dense C x C matrix product per pixel anyway, so
 *     zp = finc_mix_f32(u, blockdiag(Linv) * diag(scale) * inverse(W), blockdiag(Linv) * shift)
 * costs what the plain mix costs, and the unit's inverse starts every pixel from what it reads instead of spending 15 of
 * its 162 MFMAs (c3) on Linv * z.  `packed` is the plain bank of finc_pack_inverse_weights_f32 (its z-term and folded
 * shift are not used).  Exists for the problem sets that run the helper-wave form (a full chip; W % 16 == 0):
 * finc_inverse_premultiplied_supported() == 1, else FINC_ERR_UNSUPPORTED and the caller keeps the two plain calls.
 */
int finc_inverse_premultiplied_supported(int B, int G, int Cq, int H, int W, int KH, int KW);
int finc_inverse_packed_premultiplied_f32(const float *zp, const void *packed, float *x, int B, int G, int Cq, int H,
                                          int W, int KH, int KW, unsigned orient, finc_stream_t stream);

/*
 * Backward of the forward conv (replaces autograd through cuDNN,
 * layers/conv.py:105, plus PaddedConv2d.reset_gradients, layers/conv.py:98-99).
 *   grad_x = conv_transpose(grad_z, w)      (same corner geometry, mirrored)
 *   grad_w_canon[o][i][kh][kw] = sum_{b,h,w} grad_z[b,o,h,w] * x[b,i,h-(KH-1-kh),w-(KW-1-kw)]
 * with the corner-tap mask applied in-kernel (masked entries are written as 0).
 * grad_w is OVERWRITTEN (not accumulated).  Either output may be NULL to skip it.
 * `workspace` of finc_backward_workspace_bytes() bytes lets both gradients run on the MFMA strip kernels (grad_x: the
 * forward kernel on the flipped image with transposed fragments; grad_w: pixels on the MFMA K dimension, partial
 * tiles in the workspace, then a reduce).  NULL or a smaller buffer selects the direct kernels.
 */
size_t finc_backward_workspace_bytes(int B, int G, int Cq, int H, int W, int KH, int KW);
int finc_backward_f32(const float *grad_z, const float *x, const float *w_canon, float *grad_x, float *grad_w_canon,
                      int B, int G, int Cq, int H, int W, int KH, int KW, unsigned orient, void *workspace,
                      size_t workspace_bytes, finc_stream_t stream);

/*
 * Double precision: the reference op dispatches over float AND double (AT_DISPATCH_FLOATING_TYPES,
 * cinc_cuda_kernel_level2.cu:117), and its CPU solver computes in fp64 (solve_parallel_mc.pyx:77-126, called from
 * layers/conv.py:113-163).  These entry points run the reference visitation and term order in fp64 (separately
 * rounded multiply and subtract): finc_inverse_f64 is bit-exact with that solver.  Same layout and orientation rules
 * as the f32 calls; no workspace, no packed form, any shape.
 */
int finc_canonicalize_weights_f64(const double *w_stored, double *w_canon, int G, int Cq, int KH, int KW,
                                  unsigned orient, finc_stream_t stream);
int finc_inverse_f64(const double *z, const double *w_canon, double *x, int B, int G, int Cq, int H, int W, int KH,
                     int KW, unsigned orient, finc_stream_t stream);
int finc_forward_f64(const double *x, const double *w_canon, double *z, int B, int G, int Cq, int H, int W, int KH,
                     int KW, unsigned orient, finc_stream_t stream);
/*
 * The same two calls with an algorithm choice (round 5): FINC_ALGO_STRICT = the reference-order kernels above; FINC_ALGO_AUTO /
 * FINC_ALGO_MFMA = the double-precision matrix-core form (finc_f64.hip: v_mfma_f64_16x16x4_f64, one wavefront per (image, group),
 * the reference's anti-diagonal visitation band by band, the in-pixel substitution folded into the bank in fp64) for the banks that
 * have one -- Cq <= 24 at 3x3 (configs[1], configs[2]), Cq <= 32 at 2x2; results within 1e-12 of the reference-order solve, not
 * bit-equal to it (the order in which a pixel's terms are added differs).  AUTO falls back to the reference-order kernel for any
 * other shape or when `workspace` is NULL / smaller than finc_f64_workspace_bytes(); MFMA reports FINC_ERR_UNSUPPORTED /
 * FINC_ERR_WORKSPACE instead.  The workspace (8-byte aligned) holds the packed bank; the call packs and launches on `stream`.
 */
size_t finc_f64_workspace_bytes(int G, int Cq, int KH, int KW);
int finc_inverse_f64_algo(const double *z, const double *w_canon, double *x, int B, int G, int Cq, int H, int W, int KH, int KW,
                          unsigned orient, int algo, void *workspace, size_t workspace_bytes, finc_stream_t stream);
int finc_forward_f64_algo(const double *x, const double *w_canon, double *z, int B, int G, int Cq, int H, int W, int KH, int KW,
                          unsigned orient, int algo, void *workspace, size_t workspace_bytes, finc_stream_t stream);

/*
 * SURVEY 8 f3, second half -- the 1x1 convolution next to the unit (layers/conv1x1.py:29-43: forward
 * F.conv2d(x, W), reverse F.conv2d(z, inverse(W)); one per flow step, fastflow_cifar_multi_gpu.py:224-256) as one
 * streaming pass:      out[b, :, p] = mat * in[b, :, p] + bias      for every pixel p of every image,
 * mat [C][C] row-major (out channel, in channel), bias [C] or NULL, activations fp32 NCHW with HW = H*W.
 * The per-channel affine neighbour folds into the operands on the host: Conv1x1.reverse followed by ActNorm.reverse
 * (layers/actnorm.py:47-52) is mat = diag(exp(log_scale)) * inverse(W), bias = translation.
 * `in == out` is allowed.  FINC_ERR_UNSUPPORTED for channel counts without an instantiation
 * (finc_mix_supported_f32(C) == 0): the caller keeps its own 1x1 convolution for those.
 */
int finc_mix_supported_f32(int C);
int finc_mix_f32(const float *in, const float *mat, const float *bias, float *out, int B, int C, int HW,
                 finc_stream_t stream);

/*
 * Introspection (tests, diagnostics; no reference counterpart).
 * finc_inverse_kernel_variant: which MFMA inverse kernel FINC_ALGO_AUTO / finc_inverse_packed_f32 launches for this
 *   problem.  info[8] = {Cq padded to 4, waves per problem (K-split), problems per workgroup, 3 = sector pairing with
 *   helper waves (8-wave workgroups of 4 problems) / 2 = 64-byte sector pairing / 1 = 32-byte pieces / 0 = 16-byte
 *   groups, LDS bytes per workgroup, workgroups, row of the instantiation table, rows in the table}.
 *   info[3] = 4: the role-split kernel (problem sets that do not outnumber the compute units; row = -1);
 *   info[3] = 6: its short-step form for the banks of up to 16 channels (finc_chain.hip: one wave carries the recurrence on a
 *     16-row tile, one wave per tap with a + b == 2 prepares the rest, one wave owns the HBM side; row = -1);
 *   info[3] = 5: the big-bank kernel (3x3 banks with 64 < Cq <= 96, 8 waves per problem; row = -2).
 *   info[3] = 7: the streaming-bank kernel (finc_stream.hip) of every bank no table holds -- 3x3 above 96 channels per group,
 *     5x5 above 48, 2x2 above 32, 4x4, 6x6, 7x7, non-square filters above 16 channels; Cq <= 256, KH, KW <= 7: the bank
 *     streams from the L2 once per step, the solved pixels of the last KH+KW-2 steps sit in an LDS ring; info[0] = Cq padded
 *     to 16 (one-wave problems, Cq <= 48) or 64 (four waves per problem), info[1] = waves per problem; row = -3.
 *   FINC_ERR_UNSUPPORTED when the shape runs on the strict kernel.
 * finc_debug_attr_table_insert: the (device, kernel) table behind the once-per-device kernel attributes; returns 1 if
 *   the pair was new.  Host-only; exists so the key logic is testable without two GPUs.
 * finc_debug_inverse_table_row: row `row` of the MFMA inverse instantiation table, info[6] = {Cq padded, KH, KW, waves
 *   per problem, problems per workgroup, max_problems (0 = none)}; FINC_ERR_BAD_DIMS past the end.  Lets a test walk
 *   every compiled variant.
 */
int finc_inverse_kernel_variant(int B, int G, int Cq, int H, int W, int KH, int KW, int *info);
/* Which kernels finc_backward_f32 runs for this shape (16-byte aligned activations, full workspace): info[3] =
 * {grad-weight: 0 direct / 1 dword MFMA strip kernel / 2 staged (16-byte pieces through LDS) / 3 tiled (one tile pair per
 * workgroup) / 4 Winograd (3x3 banks of 13..32 channels: F(4,3) transposed, half the multiplies) / 5 Winograd on one tile pair
 * per wave (3x3 above 32 channels: F(4,3) transposed; 5x5 above 12: F(2,5) transposed), grad-input: waves per strip of the MFMA strip kernel (0 = direct kernel; > 1 = K-split), grad-input: staged form
 * (1) or dword form (0); 2 = Winograd F(2,3) along W, 3 = the big banks' M-split, 4 = Winograd F(4,3) along W, 5 = Winograd
 * F(2,5) along W (5x5 banks), 6 = Winograd F(4,3) M-split over a workgroup's waves (3x3 banks of 25 .. 64 channels), 7 = the
 * streaming-bank kernel in its forward form (the banks beyond every table; info[1] = its waves per problem)}.  Lets a parity
 * test assert WHICH kernel its numbers came from. */
int finc_debug_backward_variant(int B, int G, int Cq, int H, int W, int KH, int KW, int *info);
/* Pins the kernel family of the 3x3 forward / grad-input for this process (tests and A/B timing of each form on one shape):
 * 0 = the library's choice (default), 1 = the direct strip kernels (3x3 and 5x5), 2 = Winograd F(2,3), 4 = Winograd F(4,3).  A form a call
 * cannot take (width not a multiple of 4, unaligned activations, bank too large) still falls back to the strip kernel. */
int finc_debug_set_forward_form(int form);
/* How many images of an fp32 inverse call (FINC_ALGO_AUTO / the packed entry points) go to a SECOND launch: a problem set of whole
 * rounds of one-wave problems (1,024: four to a compute unit; 512 for the packed two-wave kernels) plus a remainder of at most 512
 * problems runs the rounds on that kernel and the remainder's images -- the last ones of the batch -- on the kernel the library picks
 * for them alone (`finc_inverse_kernel_variant` of that image count).  0: the call is one launch.  Host-only.  (The reference's
 * counterpart is the host loop of cinc_cuda_level2.cpp:19-32: 3,048 launches whatever the batch.) */
int finc_debug_inverse_remainder_images(int B, int G, int Cq, int H, int W, int KH, int KW);
/* The row-chunk count the one-wave-per-SIMD forward kernels (F(4,3), its M-split, F(2,5)) launch with: `units` strips of which the chip
 * holds `slots` at a time, maps of H rows, chunks of at least `min_rows` rows that each recompute `extra` rows of operands -- the count
 * that minimises rounds x (rows per chunk + extra).  `second_tenant` (1 .. 16): sixteenths a row costs once the launch has more units
 * than slots -- 14 for the kernels that run two waves per SIMD (strip kernel, F(2,3): slots = SIMDs, min_rows 4), 16 otherwise.
 * Host-only; 0 for arguments out of range.  (No reference counterpart: the reference's forward is one cuDNN call,
 * layers/conv.py:98-107.) */
int finc_debug_row_chunks(long long units, long long slots, int H, int min_rows, int extra, int second_tenant);
int finc_debug_inverse_table_row(int row, int *info);
int finc_debug_attr_table_insert(int device, size_t kernel_token);
/* SYNCHRONOUS.  The helper-wave form of the inverse (form 3 of finc_inverse_kernel_variant) pairs each compute wave with a
 * wave that does its HBM traffic; the two meet through progress words in LDS, and every wait is bounded.  *h_count = waits
 * that gave up since the library was loaded on the current device: anything but 0 is a bug. */
int finc_debug_hlp_timeouts(unsigned *h_count);
/* A wait that gives up leaves garbage in that launch's output.  It does not pass silently: the kernel also sets a word in
 * mapped host memory, and every later launching call on that device -- finc_inverse_*, finc_forward_*, finc_mix_f32,
 * finc_backward_f32, finc_check_invariant_f32 -- returns FINC_ERR_LAUNCH (finc_last_hip_error() names the cause) until
 * finc_clear_fault(); no synchronisation is added to the launch path.  The word is armed by the packing calls (and by the
 * first helper-wave launch outside a stream capture).  The launch that faulted has itself returned FINC_OK (it is
 * asynchronous): callers check finc_fault_pending() at their own synchronisation points -- the Python layer does at the
 * end of FlowSequential.sample and in load_reference_checkpoint, bench.py after every timed leg. */
int finc_clear_fault(void);
/* 1 if the current device's fault word is set (host-side read of mapped memory, no synchronisation), else 0. */
int finc_fault_pending(void);
/* Run-time A/B switches in effect in this process: every FINC_* environment switch the library found SET when it looked
 * (FINC_NO_HLP, FINC_NO_S64, FINC_NO_WINO, FINC_WINO_FORM, FINC_SPLIT_MAX ...: scripts/ only) and a forward form pinned by
 * finc_debug_set_forward_form.  Returns their number and writes a comma-separated list into h_buf (n bytes, may be NULL).
 * 0 = the library's own dispatch; bench.py records the list and refuses to report a judged line otherwise. */
int finc_runtime_switches(char *h_buf, size_t n);
/* Clock probe (bench.py's per-leg `sclk_mhz`).  The chip lowers its shader clock under load by an amount that depends on
 * the data, so one kernel can take different wall times on different inputs with no code difference.  _begin starts ONE
 * wavefront on a stream of its own that samples s_memtime (shader clock) against s_memrealtime (100 MHz) every
 * `period_us` until _end, its sample budget or `max_ms` of wall time, whichever comes first -- nothing is stamped in the
 * measured kernels.  _end (SYNCHRONOUS on the probe's stream only) stops it and returns h_stats[6] = {mean MHz of the
 * window, min and max over the sampling intervals, samples, seconds covered, median MHz}.  One probe per device at a
 * time; the caller must not device-synchronise between _begin and _end (stream synchronisation is fine). */
int finc_debug_clock_probe_begin(int period_us, int max_ms);
int finc_debug_clock_probe_end(double *h_stats);

#ifdef __cplusplus
}
#endif
#endif /* FINC_H */
