// instances.h -- every instantiation of the fasty / fastm kernel templates that xrft_hip.cpp launches, as one list.
// build() compiles the library from several translation units in parallel (one hipcc process compiles one unit on one core: ~1500 kernels
// in a single unit took four minutes): xrft_hip.cpp includes this list with XRFT_KW = `extern template __global__` (so it launches the kernels
// without instantiating them), and inst_fasty.cpp / inst_fastm_*.cpp include it with XRFT_KW = `template __global__` and XRFT_KI_GROUP set to
// the group they hold.  A kernel launched but not listed here fails the link (-Wl,-z,defs), not the run.  The emulator build and a plain
// one-unit build (no -DXRFT_SPLIT_TUS) never include this file: the launches instantiate implicitly.
// Must mirror the launch macros of xrft_hip.cpp: fasty_launch_cols / _rows, fastm_launch_cols / _rows, run_fastmy, run_fastmx.
#ifndef XRFT_KW
#error "define XRFT_KW (extern template __global__ | template __global__) before including instances.h"
#endif
#ifndef XRFT_KI_GROUP
#define XRFT_KI_GROUP 0  /* 0: all groups (the declarations of the launching unit) */
#endif
#define XRFT_KI_ON(g) (XRFT_KI_GROUP == 0 || XRFT_KI_GROUP == (g))

#if XRFT_KI_ON(1)  // ---- fasty.h: the y-first float32 kernels
#define XRFT_KI_Y_(NN) \
    XRFT_KW void fasty_cols_kernel<NN, true>(FastY); XRFT_KW void fasty_cols_kernel<NN, false>(FastY); \
    XRFT_KW void fasty_cols_kernel<NN, true, true>(FastY); XRFT_KW void fasty_cols_kernel<NN, false, true>(FastY); \
    XRFT_KW void fasty_rows_kernel<NN, 1, true>(FastY); XRFT_KW void fasty_rows_kernel<NN, 1, false>(FastY); \
    XRFT_KW void fasty_rows_kernel<NN, 2, true>(FastY); XRFT_KW void fasty_rows_kernel<NN, 2, false>(FastY); \
    XRFT_KW void fasty_rows_kernel<NN, 3, false>(FastY); XRFT_KW void fasty_rows_kernel<NN, 0, false>(FastY);
XRFT_KI_Y_(256) XRFT_KI_Y_(512) XRFT_KI_Y_(1024) XRFT_KI_Y_(2048) XRFT_KI_Y_(4096)
#undef XRFT_KI_Y_
#define XRFT_KI_YC_(NN) XRFT_KW void fastyc_cols_kernel<NN>(FastYC); XRFT_KW void fastyc_rows_kernel<NN>(FastYC);
XRFT_KI_YC_(256) XRFT_KI_YC_(512) XRFT_KI_YC_(1024) XRFT_KI_YC_(2048) XRFT_KI_YC_(4096)
#undef XRFT_KI_YC_
#define XRFT_KI_Y2_(NN) XRFT_KW void fastyc_rows_c2r_kernel<NN, false>(FastYC); XRFT_KW void fastyc_rows_c2r_kernel<NN, true>(FastYC);
XRFT_KI_Y2_(256) XRFT_KI_Y2_(512) XRFT_KI_Y2_(1024) XRFT_KI_Y2_(2048)
#undef XRFT_KI_Y2_
#define XRFT_KI_YI_(NN) XRFT_KW void fasty_isorows_kernel<NN, 1, false>(FastY); XRFT_KW void fasty_isorows_kernel<NN, 2, false>(FastY); \
    XRFT_KW void fasty_isorows_kernel<NN, 1, true>(FastY); XRFT_KW void fasty_isorows_kernel<NN, 2, true>(FastY);
XRFT_KI_YI_(1024) XRFT_KI_YI_(2048) XRFT_KI_YI_(4096)
#undef XRFT_KI_YI_
XRFT_KW void fasty_rows_kernel<256, 1, false, true, true>(FastY); XRFT_KW void fasty_rows_kernel<256, 0, false, true, true>(FastY);
XRFT_KW void fasty_rows_kernel<256, 1, false, true>(FastY); XRFT_KW void fasty_rows_kernel<256, 0, false, true>(FastY);
#endif

#if XRFT_KI_ON(2)  // ---- fastm.h, pass 1
#define XRFT_KI_MCD_(NN) XRFT_KW void fastm_cols_kernel<double, NN, true>(FastM); XRFT_KW void fastm_cols_kernel<double, NN, false>(FastM);
#define XRFT_KI_MCF_(NN) XRFT_KW void fastm_cols_kernel<float, NN, true>(FastM); XRFT_KW void fastm_cols_kernel<float, NN, false>(FastM);
#define XRFT_KI_MCW_(NN) XRFT_KW void fastm_cols_kernel<float, NN, true, 4>(FastM); XRFT_KW void fastm_cols_kernel<float, NN, false, 4>(FastM);
XRFT_M_LATLON(XRFT_KI_MCD_) XRFT_M_POW2(XRFT_KI_MCD_) XRFT_M_LATLON(XRFT_KI_MCF_) XRFT_M_F32ONLY(XRFT_KI_MCF_) XRFT_M_WIDE32(XRFT_KI_MCW_)
#undef XRFT_KI_MCD_
#undef XRFT_KI_MCF_
#undef XRFT_KI_MCW_
#endif

#if XRFT_KI_ON(3) || XRFT_KI_ON(5)  // ---- fastm.h, pass 2 (float64: group 3, float32: group 5)
#define XRFT_KI_MR_(TT, NN) \
    XRFT_KW void fastm_rows_kernel<TT, NN, 1, true>(FastM); XRFT_KW void fastm_rows_kernel<TT, NN, 1>(FastM); \
    XRFT_KW void fastm_rows_kernel<TT, NN, 2, true>(FastM); XRFT_KW void fastm_rows_kernel<TT, NN, 2>(FastM); \
    XRFT_KW void fastm_rows_kernel<TT, NN, 3>(FastM); XRFT_KW void fastm_rows_kernel<TT, NN, 0>(FastM);
#define XRFT_KI_MRD_(NN) XRFT_KI_MR_(double, NN)
#define XRFT_KI_MRF_(NN) XRFT_KI_MR_(float, NN)
#if XRFT_KI_ON(3)
XRFT_M_LATLON(XRFT_KI_MRD_) XRFT_M_POW2(XRFT_KI_MRD_)
#endif
#if XRFT_KI_ON(5)
XRFT_M_LATLON(XRFT_KI_MRF_) XRFT_M_F32ONLY(XRFT_KI_MRF_)
#endif
#undef XRFT_KI_MR_
#undef XRFT_KI_MRD_
#undef XRFT_KI_MRF_
#endif

#if XRFT_KI_ON(4)  // ---- fastm.h, the one-axis kernels
#define XRFT_KI_M1_(TT, NN) \
    XRFT_KW void fastm_yonly_kernel<TT, NN, 0>(FastM); XRFT_KW void fastm_yonly_kernel<TT, NN, 1>(FastM); XRFT_KW void fastm_yonly_kernel<TT, NN, 2>(FastM); \
    XRFT_KW void fastm_xonly_kernel<TT, NN, 0>(FastM); XRFT_KW void fastm_xonly_kernel<TT, NN, 1>(FastM); XRFT_KW void fastm_xonly_kernel<TT, NN, 2>(FastM);
#define XRFT_KI_M1D_(NN) XRFT_KI_M1_(double, NN)
#define XRFT_KI_M1F_(NN) XRFT_KI_M1_(float, NN)
XRFT_M_LATLON(XRFT_KI_M1D_) XRFT_M_POW2(XRFT_KI_M1D_) XRFT_M_YONLY(XRFT_KI_M1D_) XRFT_KI_M1D_(2048) XRFT_KI_M1D_(4096)
XRFT_M_LATLON(XRFT_KI_M1F_) XRFT_M_F32ONLY(XRFT_KI_M1F_) XRFT_M_F32_1AX(XRFT_KI_M1F_) XRFT_M_POW2(XRFT_KI_M1F_) XRFT_M_YONLY(XRFT_KI_M1F_) XRFT_KI_M1F_(2048) XRFT_KI_M1F_(4096)
#undef XRFT_KI_M1_
#undef XRFT_KI_M1D_
#undef XRFT_KI_M1F_
#endif
#if XRFT_KI_ON(6) || XRFT_KI_ON(7)  // ---- fastn.h: the y-first pipeline with the lengths as data (float32: group 6, float64: group 7); CAP = the largest radix a variant carries
#define XRFT_KI_N_(TT, CC) \
    XRFT_KW void fastn_cols_kernel<TT, 0, CC>(FastN); \
    XRFT_KW void fastn_rows_kernel<TT, 0, false, CC>(FastN); XRFT_KW void fastn_rows_kernel<TT, 1, false, CC>(FastN); XRFT_KW void fastn_rows_kernel<TT, 1, true, CC>(FastN); \
    XRFT_KW void fastn_rows_kernel<TT, 2, false, CC>(FastN); XRFT_KW void fastn_rows_kernel<TT, 2, true, CC>(FastN); XRFT_KW void fastn_rows_kernel<TT, 3, false, CC>(FastN);
#if XRFT_KI_ON(6)
XRFT_KI_N_(float, 16) XRFT_KI_N_(float, 20) XRFT_KW void fastn_cols_kernel<float, 1, 16>(FastN); XRFT_KW void fastn_cols_kernel<float, 2, 16>(FastN);  /* (the chirp-convolution and the Rader columns: radices up to 16) */
XRFT_KW void fastn_irows_kernel<float, 0, 16>(FastNI); XRFT_KW void fastn_irows_kernel<float, 1, 16>(FastNI); XRFT_KW void fastn_irows_kernel<float, 0, 20>(FastNI); XRFT_KW void fastn_irows_kernel<float, 1, 20>(FastNI);
XRFT_KW void fastn_irows_kernel<float, 2, 16>(FastNI); XRFT_KW void fastn_irows_kernel<float, 2, 20>(FastNI);
XRFT_KW void fastn_fit_inner_kernel<float>(const double*, const float*, C2<float>*, int, int, int, int, int, int);
#endif
#if XRFT_KI_ON(7)
XRFT_KI_N_(double, 16) XRFT_KW void fastn_cols_kernel<double, 1, 16>(FastN); XRFT_KW void fastn_cols_kernel<double, 2, 16>(FastN);
XRFT_KW void fastn_irows_kernel<double, 0, 16>(FastNI); XRFT_KW void fastn_irows_kernel<double, 1, 16>(FastNI); XRFT_KW void fastn_irows_kernel<double, 2, 16>(FastNI);
XRFT_KW void fastn_fit_inner_kernel<double>(const double*, const double*, C2<double>*, int, int, int, int, int, int);
#endif
#undef XRFT_KI_N_
#endif
#undef XRFT_KI_ON
