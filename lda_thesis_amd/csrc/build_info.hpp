// build_info.hpp -- what llda_build_info() returns, from the preprocessor alone (include/llda_gibbs.h: LLDA_BUILD_*).
// Included FIRST by llda_gibbs.hip, before any header gives the switches their defaults; tests/test_abi.py compiles this file on its own
// with and without the switches.
#pragma once
// one bit per compile-time switch that changes what the kernels do or how they are scheduled.  The production library (csrc/Makefile
// without EXTRA) reports 0; bench.py prints no line from a library that reports anything else.
#ifdef LLDA_MARGIN0
#define LLDA_INFO_MARGIN0 LLDA_BUILD_MARGIN0
#else
#define LLDA_INFO_MARGIN0 0
#endif
#ifdef LLDA_WAVES
#define LLDA_INFO_WAVES LLDA_BUILD_WAVES
#else
#define LLDA_INFO_WAVES 0
#endif
#ifdef LLDA_MARGIN0_WIDE
#define LLDA_INFO_MARGIN0_WIDE LLDA_BUILD_MARGIN0_WIDE
#else
#define LLDA_INFO_MARGIN0_WIDE 0
#endif
#ifdef ABL_NOLOAD
#define LLDA_INFO_NOLOAD LLDA_BUILD_ABL_NOLOAD
#else
#define LLDA_INFO_NOLOAD 0
#endif
#ifdef ABL_NOCOMMIT
#define LLDA_INFO_NOCOMMIT LLDA_BUILD_ABL_NOCOMMIT
#else
#define LLDA_INFO_NOCOMMIT 0
#endif
#ifdef ABL_WIDE_NOROW
#define LLDA_INFO_WIDE_NOROW LLDA_BUILD_ABL_WIDE_NOROW
#else
#define LLDA_INFO_WIDE_NOROW 0
#endif
#ifdef ABL_WIDE_NOADDLOAD
#define LLDA_INFO_WIDE_NOADDLOAD LLDA_BUILD_ABL_WIDE_NOADDLOAD
#else
#define LLDA_INFO_WIDE_NOADDLOAD 0
#endif
#ifdef ABL_NOFMA
#define LLDA_INFO_NOFMA LLDA_BUILD_ABL_NOFMA
#else
#define LLDA_INFO_NOFMA 0
#endif
#ifdef ABL_EXTRA_LDS_BYTES
#define LLDA_INFO_EXTRA_LDS LLDA_BUILD_ABL_EXTRA_LDS
#else
#define LLDA_INFO_EXTRA_LDS 0
#endif
#ifdef QUAD_PROFILE
#define LLDA_INFO_QUAD_PROFILE LLDA_BUILD_QUAD_PROFILE
#else
#define LLDA_INFO_QUAD_PROFILE 0
#endif
#ifdef LLDA_BUDGET_MARKS
#define LLDA_INFO_BUDGET_MARKS LLDA_BUILD_BUDGET_MARKS
#else
#define LLDA_INFO_BUDGET_MARKS 0
#endif
#ifdef LLDA_QUAD_PRIO
#define LLDA_INFO_QUAD_PRIO LLDA_BUILD_QUAD_PRIO
#else
#define LLDA_INFO_QUAD_PRIO 0
#endif
#define LLDA_BUILD_INFO_BITS (LLDA_INFO_QUAD_PRIO | LLDA_INFO_QUAD_PROFILE | LLDA_INFO_BUDGET_MARKS | LLDA_INFO_MARGIN0 | LLDA_INFO_WAVES | LLDA_INFO_MARGIN0_WIDE | LLDA_INFO_NOLOAD | LLDA_INFO_NOCOMMIT | \
                              LLDA_INFO_WIDE_NOROW | LLDA_INFO_WIDE_NOADDLOAD | LLDA_INFO_NOFMA | LLDA_INFO_EXTRA_LDS)
