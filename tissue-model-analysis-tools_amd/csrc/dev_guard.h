// Product builds refuse the development switches of the kernel sources.
//
// The kernel files carry compile-time switches for timing experiments: *_ABL_* (ablations: pieces of a kernel removed to see what they
// cost -- THE RESULTS ARE WRONG), *_VAR_* (alternative instruction orders / priorities, same results) and *_DIAG (in-kernel cycle
// stamps).  tools/build.py never defines any of them; tools/build_variant.sh is the only place that does, together with
// -DTMAT_DEV_BUILD, and writes its library to build_variants/ (git- and product-ignored).  Any other way one of them reaches a
// compile of libtmat_hip.so stops here.
#pragma once
#if !defined(TMAT_DEV_BUILD)
#if defined(TMAT_ABL_A9) || defined(TMAT_ABL_NOEPI) || defined(TMAT_ABL_NODMA) || defined(TMAT_ABL_NOBAR)
#error "an *_ABL_* timing ablation (wrong results) is defined in a product build: ablations need -DTMAT_DEV_BUILD (tools/build_variant.sh)"
#endif
#if defined(TMAT_VAR_ORDER) || defined(TMAT_VAR_SETPRIO) || defined(TMAT_VAR_NOPIN) || defined(TMAT_OLD_MASKS) || \
    defined(WS_POOL_SHUFFLE) || defined(TMAT_VAR_BUFSTORE) || defined(WS_VAR_SLEEP) || defined(WS_VAR_FETCHPRIO) || defined(ZH_VAR_NOSKIP) || defined(MA_VAR_NOSKIP)
#error "a *_VAR_* kernel variant is defined in a product build: variants need -DTMAT_DEV_BUILD (tools/build_variant.sh)"
#endif
#if defined(TMAT_DIAG) || defined(WS_DIAG)
#error "a *_DIAG cycle-stamp build is not a product build: diagnostics need -DTMAT_DEV_BUILD (tools/build_variant.sh)"
#endif
#endif
