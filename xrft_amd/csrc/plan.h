#pragma once
// plan.h -- what the host side of libxrft_hip.so shares between its translation units (not part of the C ABI: include/xrft_hip.h is):
// the plan (struct xrfthip_plan), device tables, the small host helpers, and the declarations of the per-family host functions.
//
//   xrft_hip.cpp     the plan builder (xrfthip_plan_create: which kernels serve a descriptor), the generic tile passes (tile_fft.h), workspace
//                    layout, xrfthip_exec's dispatch, describe / profiling, the C ABI of the plan
//   host_fasty.cpp   the y-first float32 two-pass pipeline (fasty.h, fasty_iso.h) and its complex form (fasty_c2c.h): tables, launchers
//   host_fastm.cpp   the mixed-radix pipeline on the lat/lon lengths (fastm.h), its run-time-radix form (fastn.h), the one-axis table kernels
//   host_fastg.cpp   the one-pass lengths-as-data kernels (fastg.h): small slabs, one axis of any smooth length, Rader / Bluestein tables
//   host_rows.cpp    the register-resident one-pass kernels: small float32 slabs (fasts.h), long rows and complex rows (fastr.h)
//   host_inner.cpp   two transform axes that are not the trailing pair (xrfthip_desc.inner / .mid): the fused passes and the composite plan
//   ops.cpp          the stand-alone operations of the ABI (detrend, spectrum tail, gather, convert, reduce, isotropize, ...)
//   inst_g1..7.cpp   the explicit instantiations of the fasty / fastm / fastn kernel templates (instances.h)
//
// A plan is a short list of kernel launches per group of slabs, chosen when the plan is created (xrfthip_plan_describe says which):
//   * generic passes (tile_fft.h), any shape:   [slab_moments -> finalize_coef]  ->  x pass(es)  ->  y pass(es)  [-> radial sums]
//     The x pass reads the user's array (detrend / window / flip / ifftshift fused into its loads) and the last pass writes the
//     user's output (fftshift / phase / scaling / |F|^2 / cross / mirror fused into its stores); the only intermediate is the
//     half spectrum of ONE group of slabs, re-used for every group.
//   * the specialised families above, each with its own launcher (run_fast*).
// Nothing allocates or synchronises in exec.
#include <algorithm>
#include <functional>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/xrft_hip.h"
#ifndef XRFT_EMULATE
#include <hip/hip_ext.h>
#endif
#include "aux_kernels.h"
#include "fastp2.h"
#include "fasty.h"
#include "fasty_iso.h"
#include "fasty_c2c.h"
#include "selftest.h"
#include "fastm.h"
#include "fastn.h"
#include "fastr.h"
#include "fasts.h"
#include "tile_fft.h"
#include "fastg.h"
#ifdef XRFT_SPLIT_TUS  /* the library built from several translation units: the fasty / fastm kernels are instantiated in inst_g*.cpp */
namespace xrft {
#define XRFT_KW extern template __global__
#include "instances.h"
#undef XRFT_KW
}
#endif

using namespace xrft;

namespace xrfth {

extern thread_local int g_last_hip_error;

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess) {                         \
            g_last_hip_error = (int)e_;                 \
            return XRFTHIP_HIP_ERROR;                   \
        }                                               \
    } while (0)

constexpr size_t kLdsMax = 160 * 1024;  // gfx950: 160 KiB per CU, one workgroup may use all of it
constexpr int kCUs = 256;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int upload(const void* host, size_t n) {
        if (p) { (void)hipFree(p); p = nullptr; }
        bytes = n;
        if (hipMalloc(&p, n ? n : 16) != hipSuccess) { p = nullptr; return XRFTHIP_ALLOC_FAILED; }
        HIP_TRY(hipMemcpy(p, host, n, hipMemcpyHostToDevice));
        return XRFTHIP_OK;
    }
    void clear() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
};

struct FftTables {  // per FFT length, in the plan's precision
    int n = 0;
    std::vector<int> radix;
    bool generic = false;
    DevBuf tw, rev;
    // Bluestein (a prime factor above XRFTHIP_MAX_RADIX): the LDS transform has blue_m = 2^a 3^b 5^c >= 2n-1 points;
    // blue_c[k] = exp(+i pi k^2 / n), blue_b = FFT_m(chirp kernel) / m in the order the DIF passes leave it
    int blue_m = 0;
    DevBuf blue_c, blue_b;
};

enum BufKind { B_NONE = 0, B_IN, B_W, B_W2, B_F0, B_OUT };

struct Pass {
    TileGeom g{};
    Prologue pr{};
    Epilogue ep{};
    bool first = false, final_ = false, generic = false;
    int path = 0;  // tile_fft_kernel PATH: 0 = general instantiation, 1..4 = lean single-purpose ones
    int threads = 256;
    size_t lds = 0;
    int in_kind = B_NONE, out_kind = B_NONE;
    long long outer_per_slab = 1;  // n_outer = outer_per_slab * slabs in the group
    std::string label;
};

inline long long env_ll(const char* name, long long dflt) {
    const char* e = getenv(name);
    return e && *e ? atoll(e) : dflt;
}

inline int factorize(long long n, std::vector<int>& out, bool& generic) {
    // prime factors first: anything above XRFTHIP_MAX_RADIX is refused here (Bluestein takes over, see lds_fft_len)
    std::vector<int> big;  // primes other than 2, 3, 5: O(r^2) butterflies of their own
    long long m = n;
    int c2 = 0, c3 = 0, c5 = 0;
    while (m % 2 == 0) { ++c2; m /= 2; }
    while (m % 3 == 0) { ++c3; m /= 3; }
    while (m % 5 == 0) { ++c5; m /= 5; }
    for (long long p = 7; m > 1; p += 2) {
        if (p * p > m) p = m;
        while (m % p == 0) {
            if (p > XRFTHIP_MAX_RADIX) return XRFTHIP_UNSUPPORTED_LENGTH;
            big.push_back((int)p);
            m /= p;
        }
    }
    generic = !big.empty();
    std::sort(big.begin(), big.end(), [](int a, int b) { return a > b; });
    // 2^a 3^b 5^c into as few passes as possible with the in-register butterflies 16, 15, 12, 10, 9, 8, 6, 5, 4, 3, 2
    // (every pass is one trip of every point through LDS): exhaustive search, the exponents are small
    static const int R[] = {16, 15, 12, 10, 9, 8, 6, 5, 4, 3, 2};
    static const int E2[] = {4, 0, 2, 1, 0, 3, 1, 0, 2, 0, 1}, E3[] = {0, 1, 1, 0, 2, 0, 1, 0, 0, 1, 0}, E5[] = {0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    const bool comp = env_ll("XRFTHIP_COMPOSITE", 1) != 0, r16 = env_ll("XRFTHIP_RADIX16", 1) != 0;
    std::vector<int> best, cur;
    std::function<void(int, int, int, int)> dfs = [&](int a2, int a3, int a5, int from) {
        if (a2 == 0 && a3 == 0 && a5 == 0) {
            if (best.empty() || cur.size() < best.size()) best = cur;
            return;
        }
        if (!best.empty() && cur.size() + 1 >= best.size()) return;
        for (int i = from; i < 11; ++i) {  // non-increasing radices: each multiset once, big radices tried first
            if (E2[i] > a2 || E3[i] > a3 || E5[i] > a5) continue;
            if (!comp && (R[i] == 15 || R[i] == 12 || R[i] == 10 || R[i] == 9 || R[i] == 6)) continue;
            if (!r16 && R[i] == 16) continue;
            cur.push_back(R[i]);
            dfs(a2 - E2[i], a3 - E3[i], a5 - E5[i], i);
            cur.pop_back();
        }
    };
    if (c2 || c3 || c5) dfs(c2, c3, c5, 0);
    // DIF order: odd and composite radices first (their passes then run on lane-contiguous LDS), powers of two last
    std::vector<int> odd, two;
    for (int r : best) ((r & (r - 1)) == 0 ? two : odd).push_back(r);
    out = big;
    out.insert(out.end(), odd.begin(), odd.end());
    out.insert(out.end(), two.begin(), two.end());
    if ((int)out.size() > XRFT_MAX_PASSES) return XRFTHIP_UNSUPPORTED_LENGTH;
    return XRFTHIP_OK;
}

// host-side forward FFT (float64) of any length whose prime factors are small: recursive decimation in time over the
// smallest factor (used once per plan for the Bluestein kernel's spectrum)
inline void host_fft_rec(const double* xr, const double* xi, size_t n, size_t stride, double* yr, double* yi) {
    if (n == 1) { yr[0] = xr[0]; yi[0] = xi[0]; return; }
    size_t p = 2;
    while (n % p) ++p;
    const size_t m = n / p;
    std::vector<double> tr(n), ti(n);
    for (size_t r = 0; r < p; ++r) host_fft_rec(xr + r * stride, xi + r * stride, m, stride * p, tr.data() + r * m, ti.data() + r * m);
    const long double w0 = -2.0L * 3.14159265358979323846264338327950288L / (long double)n;
    for (size_t k = 0; k < n; ++k) {
        long double ar = 0.0L, ai = 0.0L;
        const size_t km = k % m;
        for (size_t r = 0; r < p; ++r) {
            const long double a = w0 * (long double)((r * k) % n);
            const long double c = cosl(a), s = sinl(a);
            ar += c * tr[r * m + km] - s * ti[r * m + km];
            ai += c * ti[r * m + km] + s * tr[r * m + km];
        }
        yr[k] = (double)ar; yi[k] = (double)ai;
    }
}
inline void host_fft_smooth(std::vector<double>& re, std::vector<double>& im) {
    std::vector<double> yr(re.size()), yi(re.size());
    host_fft_rec(re.data(), im.data(), re.size(), 1, yr.data(), yi.data());
    re.swap(yr); im.swap(yi);
}

// length of the transform actually run in LDS for an n-point sequence: n, or the Bluestein length m = 2^a 3^b 5^c >= 2n - 1 (a
// power of two can be almost twice as long and then misses the LDS) when n has a prime factor the radix passes do not take
// -- or take badly: a prime above kGenericMax = 6 runs through the O(r^2) butterfly with its operands in scratch memory
// ((64, 721, 1440) float32, 721 = 7 x 103: 22.3 ms in the column pass, 2.9 GFFT/s -> 54.7 through Bluestein; 1001 = 7 x 11 x 13: 13 ->
// 26; a lone factor 7 breaks even), so it goes to Bluestein as long as four sequences of m points fit the LDS
// (scripts/prof_primes.py, prof_primes2.py).
inline long long lds_fft_len(long long n, size_t csize) {
    std::vector<int> r;
    bool g;
    if (n < 2) return n;
    const bool ok = factorize(n, r, g) == XRFTHIP_OK;
    static const long long kGenericMax = env_ll("XRFTHIP_GENERIC_MAX", 6);
    if (ok && (!g || r[0] <= kGenericMax)) return n;  // (g: a prime factor other than 2, 3, 5; those come first in r, largest first)
    long long m = 2 * n - 1;
    for (;; ++m) {
        long long q = m;
        while (q % 2 == 0) q /= 2;
        while (q % 3 == 0) q /= 3;
        while (q % 5 == 0) q /= 5;
        if (q == 1) break;
    }
    if (!ok) return m;
    const size_t per = (size_t)(m + m / 16 + 2) * csize;
    // (a factor above 13 through the O(r^2) butterfly is hopeless -- 1460 = 2 x 2 x 5 x 73 daily samples of four years, float64: 1.0 GFFT/s along the time axis --:
    // Bluestein even when only one or two sequences of m points fit the tile)
    const size_t seqs = r[0] > 13 ? 1 : 4;
    return seqs * per + 4096 <= kLdsMax ? m : n;
}

// host-side radix-2 FFT (float64) for the two 4096-point window spectra the fused detrend needs
inline void host_fft_pow2(std::vector<double>& re, std::vector<double>& im) {
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * 3.14159265358979323846264338327950288 / (double)len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const double wr = cos(ang * (double)k), wi = sin(ang * (double)k);
                const size_t a = i + k, b = i + k + len / 2;
                const double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
    }
}

template <typename T>
int build_tables(FftTables& t, int n_logical) {
    t.n = n_logical;
    const int n = (int)lds_fft_len(n_logical, 2 * sizeof(T));
    t.blue_m = n != n_logical ? n : 0;
    int rc = factorize(n, t.radix, t.generic);
    if (rc) return rc;
    std::vector<C2<T>> tw((size_t)std::max(n, 1));
    for (int k = 0; k < n; ++k) {
        const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)n;
        tw[k].re = (T)cosl(a);
        tw[k].im = (T)sinl(a);
    }
    std::vector<unsigned> rev((size_t)std::max(n, 1));
    for (int pos = 0; pos < n; ++pos) {  // frequency held at LDS position `pos` after the DIF passes
        long long L = n, rem = pos, k = 0, mult = 1;
        for (int r : t.radix) {
            const long long m = L / r;
            k += (rem / m) * mult;
            rem %= m;
            mult *= r;
            L = m;
        }
        rev[(size_t)k] = (unsigned)pos;
    }
    rc = t.tw.upload(tw.data(), tw.size() * sizeof(C2<T>));
    if (rc) return rc;
    if (t.blue_m) {
        const long long N = n_logical;
        const long double pi = 3.14159265358979323846264338327950288L;
        std::vector<C2<T>> c((size_t)N);
        std::vector<double> br((size_t)n, 0.0), bi((size_t)n, 0.0);
        for (long long k = 0; k < N; ++k) {
            const long double a = pi * (long double)((k * k) % (2 * N)) / (long double)N;  // k^2 mod 2N keeps the angle small
            const long double cr = cosl(a), ci = sinl(a);
            c[(size_t)k].re = (T)cr; c[(size_t)k].im = (T)ci;
            br[(size_t)k] = (double)cr; bi[(size_t)k] = (double)ci;
            if (k) { br[(size_t)(n - k)] = (double)cr; bi[(size_t)(n - k)] = (double)ci; }
        }
        host_fft_smooth(br, bi);
        std::vector<C2<T>> bh((size_t)n);
        for (int k = 0; k < n; ++k) {
            bh[(size_t)rev[(size_t)k]].re = (T)(br[(size_t)k] / n);
            bh[(size_t)rev[(size_t)k]].im = (T)(bi[(size_t)k] / n);
        }
        rc = t.blue_c.upload(c.data(), c.size() * sizeof(C2<T>));
        if (!rc) rc = t.blue_b.upload(bh.data(), bh.size() * sizeof(C2<T>));
        if (rc) return rc;
    }
    return t.rev.upload(rev.data(), rev.size() * sizeof(unsigned));
}

template <typename T>
int build_twiddle(DevBuf& buf, long long N, long long count) {  // W_N^k, k < count
    std::vector<C2<T>> tw((size_t)count);
    for (long long k = 0; k < count; ++k) {
        const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)N;
        tw[(size_t)k].re = (T)cosl(a);
        tw[(size_t)k].im = (T)sinl(a);
    }
    return buf.upload(tw.data(), tw.size() * sizeof(C2<T>));
}

}  // namespace xrfth
using namespace xrfth;

struct xrfthip_plan {
    xrfthip_desc d{};
    bool dbl = false, cplx_in = false;
    size_t rsize = 4, csize = 8;
    long long nxh = 0, width = 0, nx_out = 0, w_cols = 0;  // w_cols: columns of the row->column intermediate incl. tile padding
    bool mirror = false;
    int G = 1;
    int mom_chunks = 1;  // blocks per slab of the moments pass (partial sums added in order: deterministic)
    std::map<int, FftTables> tables;
    std::vector<DevBuf*> extra;  // r2c / four-step twiddles
    DevBuf win[2], phase[2], binmap;
    int nbins = 0;
    std::vector<Pass> passes;     // main pipeline (field 1 for CROSS)
    std::vector<Pass> passes_f0;  // CROSS: field 0 -> raw F0 buffer
    // workspace layout (byte offsets)
    size_t off_acc = 0, off_coef = 0, off_w = 0, off_w2 = 0, off_f0 = 0, off_pt = 0, off_rowfit = 0, off_corr = 0, off_isopart = 0, off_isotmp = 0, off_rdv = 0, ws_bytes = 0;
    int iso_chunks = 1;  // workgroups per slab of the generic radial-sum pass (partial sums added in order)
    std::string desc_text;
    // specialised path for real float32 slabs whose two lengths are 256 .. 4096 powers of two (fasty.h)
    bool fast4096 = false;  // (the flag keeps its first name: the headline shape is where the path started)
    DevBuf tw_fx, tw_fy, ones4096, fph[2];
    std::vector<double> host_phase[2];  // complex, as handed to xrfthip_plan_set_phase (empty = none)
    // two-pass "y first" pipeline for full power spectra (fasty.h): columns -> [fit] -> rows, no untile pass
    bool yfirst = false;
    // ... and, as the two steps of a four-step transform, one long real sequence per slab: N = yny * ynx samples viewed as
    // a [yny][ynx] slab (fasty.h, FS).  yny / ynx are d.ny / d.nx for the 2-D plans.
    bool fast1d = false;
    bool fastyc = false;  // ... the same two passes for COMPLEX float32 slabs (fasty_c2c.h): xrft.ifft over two axes, xrft.fft of complex data
    // ... and its mixed-radix float64 form (fastm.h): lengths 360 / 720 / 1440
    bool fastm = false;
    // ... and the same pipeline with the LENGTHS AS DATA (fastn.h): either pass (or both) of a `fastm` plan may be the run-time-radix kernel -- every
    // length that is a product of the butterflies 2 ... 20 (7, 11, 13 included), and for the columns any other length through a chirp convolution
    bool fastn = false;
    struct NSide { bool rt = false; NGeo geo{}; size_t lds = 0; DevBuf twm, geo_dev; };
    NSide n_c, n_r;                 // pass 1 (columns, length ny) and pass 2 (rows, length nx)
    int n_cw = 0, n_rk = 1, n_rpu = 0, n_nxb = 0;  // the intermediate's layout: columns per block, rows per line; rows per pass-2 workgroup; column blocks per row
    long long y_pitch = 0;          // complex elements per row of the intermediate (ynx, or n_nxb * n_cw when the last column block is ragged)
    int n_blue_m = 0;               // pass 1 through a chirp convolution of this length
    DevBuf n_bluec, n_blueb;
    int n_rad_p = 0;                // ... or, ny = q p with ONE prime 17 ... 127 whose p - 1 the butterflies factor: the prime-factor form with Rader's algorithm along p
    std::vector<int> n_rq, n_rp;    // the radices of q and of p - 1
    DevBuf n_rgeo, n_radpin, n_radpout, n_radb;
    // ... and xrfthip_desc.inner > 1 (two adjacent transform axes, the independent elements innermost) as the same two passes (fastn.h: fastn_cols_kernel on the
    // [ny][nx inner] view, fastn_fit_inner_kernel, fastn_irows_kernel).  n_c: the ny-point columns of the view; n_r: GE sequences of nx points per row workgroup
    bool fusedi = false;
    int n_dbg = 0, fi_dbg = 0, fi_vec = 1;  // the measuring scripts' ablation switches (XRFTHIP_FASTN_DBG / _FI_DBG / _FI_VEC), read when the plan is made: xrfthip_exec reads no environment
    DevBuf winx_exp;                // the window along x expanded to the view's columns (never null: ones)
    // ... and pass 1 alone for ONE transform axis that is not the contiguous one (XRFTHIP_AXIS_Y, fastm_yonly_kernel)
    bool fastmy = false;
    // ... and the same transform over short contiguous rows packed in pairs (ndim = 1, fastm_xonly_kernel)
    bool fastmx = false;
    // ... and ONE pass for a small real slab of any smooth shape, either precision, held in LDS with run-time radices (fastg.h)
    bool fastg = false;
    std::vector<int> g_rx, g_ry;
    DevBuf g_twx, g_twy, g_twr, g_revx, g_revy, g_isopos, g_isostart;
    std::vector<unsigned> g_hrevx, g_hrevy;  // (host copies: the radial-sum lists are built from them when the bin map arrives)
    bool g_one_d = false;  // ... the same kernel on groups of g_rows ROWS of a 1-D transform along x (no y passes, a mean / line per row)
    int g_rows = 0, g_lpr = 1, g_nred = 0;
    int g_rs = 0, g_n = 0;  // LDS row stride; length of the x transforms: nx / 2 (rows packed in pairs of samples) or nx (an odd nx)
    bool g_packed = true;
    // ... and ONE pass for one transform axis that is not the contiguous one (XRFTHIP_AXIS_Y), any smooth length, real input (fastg.h: fastgy_kernel)
    bool fastgy = false;
    int gy_G = 0, gy_thr = 0, gy_blue_m = 0;  // gy_blue_m: Bluestein inside the tile on blue_m rows (a prime factor of ny with no butterfly)
    int gy_rad_p = 0;                         // ... or, ny = q p with ONE such prime p <= 127 and p - 1 smooth: the prime-factor form with Rader's algorithm along p
    bool gy_rows = false;                     // ... the same form along the CONTIGUOUS axis of a 1-D plan ([batch rows][nx samples]; fastgy_kernel FORM 3): gy_n = nx
    long long gy_n = 0;                       // the transform length of a fastgy plan
    std::vector<int> gy_rp;                   // the radices of p - 1
    DevBuf gy_twp, gy_radb, gy_permin;
    bool gy_tw_lds = true;
    size_t gy_lds = 0;
    DevBuf gy_bluec, gy_blueb;
    size_t g_lds = 0;
    // ... and ONE pass for a small real float32 slab that fits the registers of a CU: 256 x 256 power spectra (fasts.h)
    bool fasts = false;
    DevBuf tw_sy, tw_sx, s_tfirst;
    long long tune_sstagger = 0;  // XRFTHIP_FASTS_STAGGER: classes << 8 | steps of 3.4 us between the classes of a resident set of slab workgroups
    long long tune_sgrid = -1;    // XRFTHIP_FASTS_GRID: workgroups of the launch (0 = one per slab, the default; else a resident set walking the slabs)
    // ... and ONE pass for a long real float32 row that fits the registers of a CU: 65536 samples per workgroup (fastr.h)
    bool fastr = false;
    bool fastr_rows = false;      // ... complex rows of 256 .. 4096 points: pass 2 of the complex two-pass pipeline on the rows of the input itself (fastyc_rows_kernel, nrows > 0)
    bool fastyc_fs = false;       // fastyc on ONE long complex sequence per batch entry (ndim = 1, 2^16 .. 2^20 points): the [n / 256][256] view, the four-step form of pass 2 (FastYC::fs)
    DevBuf fs_phx;                // ... the x factor of a separable input phase (PHASE_IN): 256 entries (fph[0]: the factor per row of the view)
    bool fastr_cin = false;       // ... its complex-row form: rows of 2048 .. 16384 complex64 points, forward or inverse (fastc_kernel)
    DevBuf tw_rm, tw_rs, tw_rn;   // W_M^p (p < 1024), W_1024^n (n < 32), W_N^p (p < 1024)
    long long tune_rstagger = 0;  // XRFTHIP_FASTR_STAGGER: classes << 8 | units of 3.4 us between the start of consecutive classes of workgroups (FastR::stagger)
    long long tune_rgrid = 0;     // XRFTHIP_FASTR_GRID: workgroups of the launch (0 = one per row; default: a resident set of one per CU walking the rows)
    bool fph_on = false;  // some entry of the combined phase tables (fph) differs from 1
    long long yny = 0, ynx = 0;
    DevBuf tw_big1d;
    int y_nrow_pad = 0;  // rows ky = 0..ny/2 of the intermediate, rounded up to what one row workgroup covers
    DevBuf ywhat0, ywhat1, ytcodes;
    DevBuf ytfirst, ytwin, ytunits;  // (ytwin: the bins each unit of rows reaches; ytunits: the units that reach each bin)
    bool ytfirst_on = false;       // ... and a radial map's: the radial sums are gathered per bin without atomics (fasty_build_tcodes)
    bool ytcodes_compact = false;  // the bin map has a radial map's structure: 4 bytes per 16 samples (fasty_build_tcodes)
    std::vector<double> host_win_y;
    std::vector<double> host_win_x;  // (four-step 1-D: the window of the whole sequence)
    DevBuf win2d;                    // ... as float32, laid out like the slab [yny][ynx]
    bool fast1d_win = false;         // the four-step plan carries a window: slab-shaped window table, per-column window spectra
    // tuning knobs from the environment, read once when the plan is created (never in xrfthip_exec)
    long long tune_group = 0, tune_fast_group = 0, tune_group_bytes = 512LL << 20, tune_cols_grid = 256, tune_max_grid = 8192;
    long long tune_y = 0;  // XRFTHIP_YTUNE: cache policies / start stagger of the y-first float32 kernels (FastY::tune), fixed at plan creation
    long long tune_isorows = 0;  // XRFTHIP_ISOROWS: 1 the persistent radial-sum row kernel (fasty_iso.h) when nothing but the sums leaves pass 2; 0 (default) fasty_rows_kernel<.., ISO>;
                                 // 2 its profiling build (per-phase shader-clock sums printed after every launch: scripts/prof.py iso-phases -- synchronises, never in the product)
    mutable DevBuf iso_tim;      // ... whose counters live here
    // optional per-pass event timing (bench only; a plan with profiling on is not re-entrant)
    bool prof = false;
    struct ProfRec { std::string label; hipEvent_t a, b; };
    std::vector<ProfRec> prof_recs;
    void prof_clear() { for (auto& r : prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); } prof_recs.clear(); }
    // xrfthip_desc.inner > 1: [batch][ny][nx][inner], two adjacent transform axes with the independent elements innermost.  A
    // composite of two in-place one-axis plans (XRFTHIP_AXIS_Y): sub_x transforms x of [batch ny][nx][inner], sub_y transforms y of
    // [batch][ny][nx inner]; a detrend runs first as a pass of its own (plane_inner_* kernels).  No transposed copy anywhere.
    long long inner = 1, mid = 1;
    bool sub_x_1d = false;  // the x stage is a 1-D plan (nothing behind x: inner = 1), its axis is 1
    xrfthip_plan* sub_x = nullptr;
    xrfthip_plan* sub_y = nullptr;
    size_t off_sub = 0, off_det = 0, off_mid = 0, off_dws = 0;
    ~xrfthip_plan() { for (auto* b : extra) delete b; prof_clear(); delete sub_x; delete sub_y; }
};


// geometry of the specialised kernels as the host needs it (describe, launchers, workspace layout)
struct YGeomRt { int thr, gxy, cw, rk, lbs; size_t lds; };
struct MGeomRt { int thr, g; size_t lds_cols, lds_rows; int r0, r1, r2; int thr_r1, g_r1; size_t lds_r1; };  // *_r1: pass 2 of one field
struct SGeomRt { int thr; size_t lds; int per_cu; size_t lds_iso; };

// ---- host functions shared by the translation units of the library (xrft_hip.cpp, host_*.cpp, ops.cpp)
FastM fastm_params(const xrfthip_plan* P, const void* in, void* out, char* ws, long long g0, long long gc, int slot, long long slot_slabs);
FastN fastn_wrap(const xrfthip_plan* P, const FastM& m, bool cols);
FastY fasty_params(const xrfthip_plan* P, const float* in, void* out, double* iso, char* ws, long long g0, long long gc, int slot, long long slot_slabs);
MGeomRt mgeom(long long n, bool dbl);
MGeomRt mgeom_cols(long long ny, long long nx, bool dbl);
MGeomRt mxgeom(long long n, bool dbl);
MGeomRt mygeom(long long n, bool dbl);
SGeomRt sgeom(long long ny, long long nx);
YGeomRt ycols_geom(long long ny);
YGeomRt yrows_geom(long long nx, bool fs = false);
bool fast_on(const xrfthip_plan* P);
bool fastg_factor(long long n, std::vector<int>& out);
bool fastg_try(xrfthip_plan* P);
bool fastgy_try(xrfthip_plan* P, bool rows = false);
bool fastm_iso_fused(const xrfthip_plan* P);
bool fastm_iso_gather(const xrfthip_plan* P);
bool fastm_len(long long n, bool dbl);
bool fastm_wide(long long ny, long long nx, bool dbl);
bool fastmx_len(long long n, bool dbl);
bool fastmy_len(long long n, bool dbl);
bool fastn_factor(long long n, int maxr, std::vector<int>& out, int need_last = 0);
bool fastn_pick(long long n, int g, bool blue, bool dbl, bool cols, int maxr, int thr_force, NGeo& out);
bool fastn_setup(xrfthip_plan* P);
bool fasty_iso_tables_fit(const xrfthip_plan* P, int nbins);
bool fasty_on(const xrfthip_plan* P);
bool phase_nontrivial(const xrfthip_plan* P);
bool plan_two(const xrfthip_plan* P);
bool rader_split(long long n, bool allow17, int& p_out, std::vector<int>& rq, std::vector<int>& rp);
int build_unit_windows(xrfthip_plan* P, const int32_t* bm, int rpu);
int create_inner_plan(xrfthip_plan** plan, const xrfthip_desc& d);
int fast_phase_tables(xrfthip_plan* P);
int fastg_build_iso(xrfthip_plan* P, const int32_t* bm);
int fastg_rev(const std::vector<int>& radix, int n, DevBuf& buf, std::vector<unsigned>& host);
int fastm_build_tfirst(xrfthip_plan* P, const int32_t* bm);
int fastm_cw(long long ny, long long nx, bool dbl);
int fastm_gather_rpu(const xrfthip_plan* P);
int fastm_iso_ncopy(const xrfthip_plan* P);
int fastm_rk(long long ny, long long nx, bool dbl);
int fastm_rk2(long long ny, long long nx, bool two, bool dbl);
int fastm_rows_rpu(const xrfthip_plan* P);
int fastm_rpu(long long nx, bool two, bool dbl);
int fasts_build_tfirst(xrfthip_plan* P, const int32_t* bm);
int fasty_build_tcodes(xrfthip_plan* P, const int32_t* bm);
int fasty_window_spectra(xrfthip_plan* P);
int fasty_window_spectra_1d(xrfthip_plan* P);
int finalize_plan(xrfthip_plan* P);
int fusedi_tables(xrfthip_plan* P);
int ilog2i(int v);
int inner_chunk_cap(long long batch, long long i2);
int inner_chunks(long long ny, long long batch, long long i2);
int iso_bin_window(bool cplx);
int iso_chunk_count(long long total);
int plan_cw(const xrfthip_plan* P);
int plan_nxb(const xrfthip_plan* P);
int plan_rk2(const xrfthip_plan* P);
int rader_maps(int n, int p, const std::vector<int>& rq_, const std::vector<int>& rp_, std::vector<unsigned>& pin, std::vector<unsigned>& pout, std::vector<double>& bre, std::vector<double>& bim);
int run_detrend_inner(int32_t dtype, int32_t ndim, long long batch, long long ny, long long nx, long long inner, int32_t kind, const void* in, void* out, char* ws, hipStream_t st, long long mid = 1);
int run_fastg(const xrfthip_plan* P, const void* in, const void* in_b, void* out, double* iso, hipStream_t st);
int run_fastgy(const xrfthip_plan* P, const void* in, const void* in_b, void* out, hipStream_t st);
int run_fastm(const xrfthip_plan* P, const void* in, const void* in1, void* out, double* iso, char* ws, hipStream_t st);
int run_fastmx(const xrfthip_plan* P, const void* in, const void* in1, void* out, hipStream_t st);
int run_fastmy(const xrfthip_plan* P, const void* in, const void* in1, void* out, hipStream_t st);
int run_fastr(const xrfthip_plan* P, const void* in, void* out, hipStream_t st);
int run_fasts(const xrfthip_plan* P, const void* in, void* out, double* iso, hipStream_t st);
int run_fasty(const xrfthip_plan* P, const float* in, const float* in1, void* out, double* iso, char* ws, hipStream_t st);
int run_fastyc(const xrfthip_plan* P, const void* in, void* out, char* ws, hipStream_t st);
int run_fused_inner(const xrfthip_plan* P, const void* in, const void* in1, void* out, char* ws, hipStream_t st);
int run_inner_plan(const xrfthip_plan* P, const void* in, void* out, char* ws, hipStream_t st);
int run_radial_sums(int32_t dtype, const void* spec, const int32_t* d_binmap, long long bc, long long ny, long long nxo, int sy, int sx, int nbins, int chunks, double* part, double* iso, hipStream_t st);
int upload_real_table(xrfthip_plan* P, DevBuf& buf, const double* h, int64_t n, int cplx);
int ycols_gstr(long long ny);
long long fastg_threads(const xrfthip_plan* P);
long long fasty_rows_gx(const xrfthip_plan* P);
long long resident_workgroups(const void* kernel, int threads, size_t lds);
size_t detrend_inner_ws(bool cplx, long long batch, long long inner);
size_t fastn_lds(const NGeo& g, size_t csize, bool cols);
void fastm_launch_cols(const xrfthip_plan* P, const FastM& p, long long gc, hipStream_t st);
void fastm_launch_rows(const xrfthip_plan* P, const FastM& p, long long gc, hipStream_t st);
void fastn_geom(long long n, const std::vector<int>& rad, int g, int maxthr, bool blue, NGeo& o, int thr_force = 0, int thr_pref = 0);
void fastn_launch_cols(const xrfthip_plan* P, const FastM& m, hipStream_t st);
void fastn_launch_rows(const xrfthip_plan* P, const FastM& m, long long gc, bool fused, hipStream_t st);
void fasty_launch_cols(const xrfthip_plan* P, const FastY& p, long long gc, hipStream_t st, bool prof);
void fasty_launch_rows(const xrfthip_plan* P, const FastY& p, long long gc, hipStream_t st, bool prof);
void layout_workspace(xrfthip_plan* P);
void prof_end(xrfthip_plan::ProfRec* r, hipStream_t st);
xrfthip_plan* create_fused_inner(const xrfthip_desc& d);
xrfthip_plan::ProfRec* prof_begin(const xrfthip_plan* P, const std::string& label, hipStream_t st);

void set_attrs_fasty();
void set_attrs_fastm();
void set_attrs_fastg();
void set_attrs_rows();

// templates whose two precisions are instantiated where they are defined (host_fastm.cpp / host_fastg.cpp)
template <typename T> int fastn_upload_twm(const NGeo& g, DevBuf& buf, bool blue = false);
template <typename T> int fastn_blue_tables(xrfthip_plan* P);
template <typename T> int fastg_setup_t(xrfthip_plan* P);
template <typename T> int fastgy_rader_tables(xrfthip_plan* P);
template <typename T> int fastn_rader_tables(xrfthip_plan* P);
template <typename T> int fastgy_blue_tables(xrfthip_plan* P);
