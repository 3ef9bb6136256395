// Measurement knobs of libfinc_hip.so, in one place.
//
// Every macro listed here changes the timing -- and most of them the RESULTS -- of a kernel.  They exist for diagnostic
// builds (scripts/build_variant.sh, scripts/stamp.sh: ablations, stamps, counters, reduced instantiation tables) and are
// never set for the product.  Three guards keep a stray -D from producing a library that passes finc_version() and
// computes garbage:
//   1. setting any of them without -DFINC_EXPERIMENT is a compile error;
//   2. every translation unit reports the knobs it was built with (FINC_BUILD_FLAGS below) and finc_build_flags() ORs
//      them: 0 for the product, asserted by tests/test_abi.py;
//   3. the knobs keep their own #ifndef defaults next to the code they act on, so the product's value is in the source.
#pragma once

#define FINC_KNOB_LIST(X) X(FINC_ABLATE, 0) X(FINC_ABLATE_IO, 1) X(FINC_STAMP, 2) X(FINC_HLP_COUNT, 3) X(FINC_HLP_NOWAIT, 4) X(FINC_S64_ABLATE, 5) X(FINC_HLP_MODE, 6) X(FINC_ONLY_C3, 7) X(FINC_CONV_ABLATE, 8) X(FINC_CONV_2W_MAX, 9) X(FINC_SPLIT_STAMP, 10) X(FINC_HLP_INJECT_TIMEOUT, 11) X(FINC_HLP_BUDGET_LOG2, 12) X(FINC_WINO_ABLATE, 13) X(FINC_BIG_ABLATE, 14) X(FINC_CHAIN_ABLATE, 15) X(FINC_STREAM_ABLATE, 16)

// (knobs of experiments that DESIGN.md marks closed were removed in round 4 -- FINC_SAMEBUF, FINC_FIFO_EXEC, FINC_HLP_PRIO,
// FINC_S64_MODE, FINC_LD_AUX, FINC_ST_AUX, FINC_ZREP; their code is kept as a patch: profiles/r04/notes/removed_knobs.patch)
#if defined(FINC_ABLATE) || defined(FINC_ABLATE_IO) || defined(FINC_STAMP) || defined(FINC_HLP_COUNT) || defined(FINC_HLP_NOWAIT) || defined(FINC_S64_ABLATE) || defined(FINC_HLP_MODE) || defined(FINC_ONLY_C3) || defined(FINC_CONV_ABLATE) || defined(FINC_CONV_2W_MAX) || defined(FINC_SPLIT_STAMP) || defined(FINC_HLP_INJECT_TIMEOUT) || defined(FINC_HLP_BUDGET_LOG2) || defined(FINC_WINO_ABLATE) || defined(FINC_BIG_ABLATE) || defined(FINC_CHAIN_ABLATE) || defined(FINC_STREAM_ABLATE)
#ifndef FINC_EXPERIMENT
#error "a measurement knob (finc_experiment.h) is set without -DFINC_EXPERIMENT: this would build a library that computes wrong results"
#endif
#define FINC_HAS_KNOBS 1
#else
#define FINC_HAS_KNOBS 0
#endif

// bit i of FINC_BUILD_FLAGS = knob i of FINC_KNOB_LIST is defined in this translation unit; bit 31 = FINC_EXPERIMENT itself
#ifdef FINC_ABLATE
#define FINC_BF_0 1u
#else
#define FINC_BF_0 0u
#endif
#ifdef FINC_ABLATE_IO
#define FINC_BF_1 1u
#else
#define FINC_BF_1 0u
#endif
#ifdef FINC_STAMP
#define FINC_BF_2 1u
#else
#define FINC_BF_2 0u
#endif
#ifdef FINC_HLP_COUNT
#define FINC_BF_3 1u
#else
#define FINC_BF_3 0u
#endif
#ifdef FINC_HLP_NOWAIT
#define FINC_BF_4 1u
#else
#define FINC_BF_4 0u
#endif
#ifdef FINC_S64_ABLATE
#define FINC_BF_5 1u
#else
#define FINC_BF_5 0u
#endif
#ifdef FINC_HLP_MODE
#define FINC_BF_6 1u
#else
#define FINC_BF_6 0u
#endif
#ifdef FINC_ONLY_C3
#define FINC_BF_7 1u
#else
#define FINC_BF_7 0u
#endif
#ifdef FINC_CONV_ABLATE
#define FINC_BF_8 1u
#else
#define FINC_BF_8 0u
#endif
#ifdef FINC_CONV_2W_MAX
#define FINC_BF_9 1u
#else
#define FINC_BF_9 0u
#endif
#ifdef FINC_SPLIT_STAMP
#define FINC_BF_10 1u
#else
#define FINC_BF_10 0u
#endif
#ifdef FINC_HLP_INJECT_TIMEOUT
#define FINC_BF_11 1u
#else
#define FINC_BF_11 0u
#endif
#ifdef FINC_HLP_BUDGET_LOG2
#define FINC_BF_12 1u
#else
#define FINC_BF_12 0u
#endif
#ifdef FINC_WINO_ABLATE
#define FINC_BF_13 1u
#else
#define FINC_BF_13 0u
#endif
#ifdef FINC_BIG_ABLATE
#define FINC_BF_14 1u
#else
#define FINC_BF_14 0u
#endif
#ifdef FINC_CHAIN_ABLATE
#define FINC_BF_15 1u
#else
#define FINC_BF_15 0u
#endif
#ifdef FINC_STREAM_ABLATE
#define FINC_BF_16 1u
#else
#define FINC_BF_16 0u
#endif
#ifdef FINC_EXPERIMENT
#define FINC_BF_31 1u
#else
#define FINC_BF_31 0u
#endif
#define FINC_BUILD_FLAGS ((FINC_BF_0 << 0) | (FINC_BF_1 << 1) | (FINC_BF_2 << 2) | (FINC_BF_3 << 3) | (FINC_BF_4 << 4) | (FINC_BF_5 << 5) | (FINC_BF_6 << 6) | (FINC_BF_7 << 7) | (FINC_BF_8 << 8) | (FINC_BF_9 << 9) | (FINC_BF_10 << 10) | (FINC_BF_11 << 11) | (FINC_BF_12 << 12) | (FINC_BF_13 << 13) | (FINC_BF_14 << 14) | (FINC_BF_15 << 15) | (FINC_BF_16 << 16) | (FINC_BF_31 << 31))
