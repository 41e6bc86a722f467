/*
 * xrft_hip.h -- C ABI of libxrft_hip.so: the MI355X (gfx950) spectral engine behind the xrft API.
 *
 * The reference (xgcm/xrft) is pure Python and has no FFI of its own.  The seam this library replaces is
 * the array-backend module returned by `_fft_module` (reference xrft/xrft.py:32-36) together with the
 * full-array numpy passes `xrft.fft` / `power_spectrum` / `cross_spectrum` / `isotropize` / `detrend`
 * wrap around it.  One plan executes, fused on the device, what the reference does in these steps:
 *
 *   detrend (constant | linear)            xrft/detrend.py:54-55, 64-71, 100-113
 *   window multiply                        xrft/xrft.py:96-103
 *   flip + ifftshift of the input          xrft/xrft.py:436-441   (true_phase)
 *   fftn / rfftn over the last 1-2 axes    xrft/xrft.py:439-444
 *   fftshift                               xrft/xrft.py:446-447
 *   x exp(-i 2 pi k lag), x prod(dx)       xrft/xrft.py:462-472
 *   |F|^2  or  F1 conj(F2), real-dim x2,
 *   / window correction, x prod(dk)        xrft/xrft.py:740-748, 825-833
 *   radial bin-sum                         xrft/xrft.py:895-906, 993-1004
 *
 * Conventions
 *   - every `d_*` pointer is a DEVICE pointer owned by the caller (e.g. torch tensor .data_ptr());
 *     every `h_*` pointer is a HOST pointer, copied during the call;
 *   - arrays are C-contiguous [batch][ny][nx] (ny == 1 for 1-D transforms); the transform runs over the
 *     last `ndim` axes; `batch` slabs are independent;
 *   - a plan is immutable once created and its tables are set: xrfthip_plan_create and the xrfthip_plan_set_* calls build
 *     every device table and the workspace layout (allocations, blocking copies, environment lookups happen THERE);
 *     xrfthip_exec takes the plan as const, never allocates, never copies from the host, never synchronises, reads no
 *     environment variable and enqueues everything on `stream` (a hipStream_t passed as void*; NULL = the default
 *     stream).  It can be captured into a hipGraph from its first call, and one plan can be executed from several
 *     threads at once, each with its own stream and workspace (exception: a plan with profiling switched on records
 *     events into itself and is not re-entrant);
 *   - all functions return 0 on success or a negative xrfthip_status; nothing throws or aborts.
 *
 * Python binding: xrft_amd/_lib.py (ctypes).  A reference-side binding sketch is in INTEGRATION.md.
 */
#ifndef XRFT_HIP_H
#define XRFT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XRFTHIP_VERSION 106 /* 0.1.6: the inner / mid layouts take OUT_CROSS (two real fields) and XRFTHIP_HALF_X / REALDIM_X2 (real_dim along the second axis) on their fused passes; 0.1.5: xrfthip_selftest_floor; 0.1.4: XRFTHIP_AXIS_Y with XRFTHIP_HALF_X / REALDIM_X2 (real_dim along the one transformed axis); 0.1.3: xrfthip_desc.mid (two transform axes anywhere in a C-contiguous array); 0.1.2: xrfthip_plan_uses_bluestein, xrfthip_convert (0.1.1: xrfthip_desc.inner, xrfthip_reduce_axis, xrfthip_detrend_inner) */

typedef enum xrfthip_status {
    XRFTHIP_OK = 0,
    XRFTHIP_BAD_ARG = -1,
    XRFTHIP_UNSUPPORTED_LENGTH = -2, /* a prime factor above XRFTHIP_MAX_RADIX whose Bluestein transform (2^a 3^b 5^c >= 2n-1) does not fit the LDS */
    XRFTHIP_WORKSPACE_TOO_SMALL = -3,
    XRFTHIP_HIP_ERROR = -4, /* see xrfthip_last_hip_error() */
    XRFTHIP_ALLOC_FAILED = -5,
    XRFTHIP_MISSING_TABLE = -6 /* flag needs a table that was not set (window, phase, bin map) */
} xrfthip_status;

#define XRFTHIP_MAX_RADIX 128

typedef enum xrfthip_dtype { /* dtype of the input array; the arithmetic runs in the matching precision */
    XRFTHIP_F32 = 0,
    XRFTHIP_F64 = 1,
    XRFTHIP_C64 = 2,
    XRFTHIP_C128 = 3
} xrfthip_dtype;

typedef enum xrfthip_out_mode {
    XRFTHIP_OUT_COMPLEX = 0, /* F                    -> complex (c64 | c128)          xrft.fft            */
    XRFTHIP_OUT_POWER = 1,   /* |F|^2 * scale        -> real    (f32 | f64)           xrft.power_spectrum */
    XRFTHIP_OUT_CROSS = 2,   /* F0 conj(F1) * scale  -> complex                       xrft.cross_spectrum */
    XRFTHIP_OUT_PHASE = 3    /* arg(F0 conj(F1))     -> real, in [-pi, pi]            xrft.cross_phase (xrft.py:838-874) */
} xrfthip_out_mode;

typedef enum xrfthip_detrend_kind {
    XRFTHIP_DETREND_NONE = 0,
    XRFTHIP_DETREND_CONSTANT = 1, /* subtract the mean over the transform axes, per slab  detrend.py:54-55 */
    XRFTHIP_DETREND_LINEAR = 2    /* subtract the least-squares line (1-D) / plane (2-D)  detrend.py:64-113 */
} xrfthip_detrend_kind;

/* flags */
#define XRFTHIP_HALF_X 0x001u   /* rfftn: keep only kx = 0..nx/2 along the last axis (real input only; no shift) */
#define XRFTHIP_SHIFT_Y 0x002u  /* fftshift the output along y (xrft.py:446-447) */
#define XRFTHIP_SHIFT_X 0x004u  /* fftshift the output along x */
#define XRFTHIP_ISHIFT_Y 0x008u /* ifftshift the input along y (true_phase, xrft.py:440) */
#define XRFTHIP_ISHIFT_X 0x010u /* ifftshift the input along x */
#define XRFTHIP_FLIP_Y 0x020u   /* np.flip the input along y before the ifftshift (descending coordinate); CROSS/PHASE: field 1 (d_in1) */
#define XRFTHIP_FLIP_X 0x040u   /* np.flip the input along x; CROSS/PHASE: field 1 */
#define XRFTHIP_REALDIM_X2 0x080u /* with HALF_X and POWER|CROSS: multiply by [1,2,...,2,(1)] (xrft.py:673-682) */
#define XRFTHIP_ISO 0x100u      /* radial bin-sum of the POWER|CROSS result into d_iso (needs a bin map) */
#define XRFTHIP_NO_SPECTRUM_OUT 0x200u /* with ISO: do not write the full spectrum (d_out may be NULL) */
/* inverse transforms (xrft.ifft, xrft.py:479-646); complex input, out_mode COMPLEX */
#define XRFTHIP_INVERSE 0x400u   /* ifftn: conj(FFT(conj(z))); the caller folds 1/prod(N) into `scale` */
#define XRFTHIP_C2R_X 0x800u     /* irfftn: d_in0 is [batch][ny][nx/2+1] complex, Hermitian-extended on the fly; d_out is REAL [batch][ny][nx] */
#define XRFTHIP_PHASE_IN 0x1000u /* the phase tables multiply the INPUT (indexed by source position) instead of the output (xrft.py:574-576) */
/* Transform along a middle (or the first) axis in place, no transposed copy (the reference transforms any axes of the array
 * where they lie, xrft.py:395-409): with ndim = 2 the array is [batch][ny][nx] and ONLY y is transformed, once per (slab,
 * column); nx is the product of the trailing axes.  Detrending and the window act along y (one line / mean per column);
 * the *_X flags, ISO and C2R_X do not apply (HALF_X / REALDIM_X2: see below).  Output [batch][ny][nx] in the same layout.
 * Inverse transforms along the axis (xrft.ifft of one first / middle axis, xrft.py:479-646): XRFTHIP_INVERSE with complex input; ISHIFT_Y then
 * rotates the fftshifted INPUT rows, SHIFT_Y the output; XRFTHIP_PHASE_IN (the lag's phase on the input, axis 0 table) is accepted where a one-pass
 * kernel takes the plan (any smooth ny that fits a tile, Bluestein lengths included) and XRFTHIP_BAD_ARG otherwise -- the caller then transposes.
 * real_dim along the axis (0.1.4; xrft.py:400-404, 673-682): XRFTHIP_HALF_X with AXIS_Y keeps k = 0 .. ny/2 of the ONE transformed axis -- output
 * [batch][ny/2 + 1][nx], unshifted, real input, no SHIFT_Y / FLIP_Y / INVERSE -- and REALDIM_X2 counts 0 < k < ny/2 twice; accepted where a one-pass
 * kernel takes the plan, XRFTHIP_UNSUPPORTED_LENGTH otherwise (the caller transposes). */
#define XRFTHIP_AXIS_Y 0x2000u
/* CROSS/PHASE: the reference flips each field by its own coordinate (xrft.py:436-441): these flip field 0 (d_in0), FLIP_Y /
 * FLIP_X flip field 1 (d_in1).  The window always multiplies in the source order, before the flip (xrft.py:425-441). */
#define XRFTHIP_FLIP0_Y 0x4000u
#define XRFTHIP_FLIP0_X 0x8000u
/* inner / mid layouts only (ABI 0.1.6): real_dim along the FIRST of the two transform axes -- ky = 0..ny/2 is stored, output [batch][ny/2 + 1][mid][nx][inner], unshifted;
 * with REALDIM_X2 and POWER|CROSS 0 < ky < ny/2 counts twice.  Real input, no SHIFT_*, not together with HALF_X. */
#define XRFTHIP_HALF_Y 0x10000u

typedef struct xrfthip_desc {
    uint32_t struct_size; /* = sizeof(xrfthip_desc) */
    int32_t ndim;         /* 1 or 2: number of trailing axes transformed */
    int64_t batch;        /* independent slabs */
    int64_t ny;           /* 1 when ndim == 1 */
    int64_t nx;
    int32_t dtype;    /* xrfthip_dtype of d_in0 / d_in1 */
    int32_t out_mode; /* xrfthip_out_mode */
    int32_t detrend;  /* xrfthip_detrend_kind */
    uint32_t flags;
    double scale;            /* COMPLEX: multiplies F (prod(dx) for true_amplitude); POWER/CROSS: multiplies the product */
    int32_t slabs_per_group; /* 0 = auto: slabs pushed through all passes together (keeps the intermediate in MALL) */
    int32_t reserved;
    /* Layout with the independent elements INNERMOST: with inner > 1 the arrays are [batch][ny][nx][inner] and the transform runs over
     * (ny, nx) for every (batch, inner) element -- two ADJACENT transform axes anywhere in a C-contiguous array (the reference transforms
     * any axes where they lie, xrft/xrft.py:395-409), e.g. dim = ["y", "x"] of a (y, x, time) array: batch = 1, inner = nt.  No transposed
     * copy is made: x is transformed where it lies ([batch ny][nx][inner], as XRFTHIP_AXIS_Y does for one axis), then y
     * ([batch][ny][nx inner]); a detrend runs first as a pass of its own.  ndim = 2, out_mode COMPLEX | POWER, flags SHIFT_* /
     * ISHIFT_* / FLIP_*; windows, phases and `scale` as usual.  ABI 0.1.6: real input of a smooth shape (the two fused passes) also takes out_mode CROSS
     * (d_in1 = the second field, same layout) and XRFTHIP_HALF_X / REALDIM_X2 (real_dim along the SECOND transform axis: output [batch][ny][mid][nx/2 + 1][inner],
     * unshifted) or XRFTHIP_HALF_Y (along the FIRST: [batch][ny/2 + 1][mid][nx][inner]); XRFTHIP_UNSUPPORTED_LENGTH where only the composite of one-axis plans exists (the caller transposes).  0 or 1 = the trailing-axes layout.  A descriptor with the
     * struct_size of the version without this field is accepted (inner = 1). */
    int64_t inner;
    /* ... and `mid` independent elements BETWEEN the two transform axes (ABI 0.1.3): with inner > 1 or mid > 1 the arrays are
     * [batch][ny][mid][nx][inner] -- batch = the product of the extents in front of the first transform axis, mid of those between the two, inner of those
     * behind the second -- which is every way two transform axes can lie in a C-contiguous array of any rank (xrft/xrft.py:395-409 transforms any axes where
     * they lie): dim = ["t", "x"] of a (t, y, x) array is batch = 1, ny = nt, mid = ny, nx = nx, inner = 1.  Same plan as `inner` alone: x where it lies, then y
     * where it lies, a detrend (one plane over (ny, nx) per (batch, mid, inner) element) first as a pass of its own; no transposed copy.  0 or 1 = none.
     * Descriptors with the struct_size of the versions without `mid` / without `inner` are accepted. */
    int64_t mid;
} xrfthip_desc;

typedef struct xrfthip_plan xrfthip_plan;

int xrfthip_version(void);
const char* xrfthip_strerror(int status);
int xrfthip_last_hip_error(void); /* hipError_t of the most recent XRFTHIP_HIP_ERROR on this thread */

int xrfthip_plan_create(xrfthip_plan** plan, const xrfthip_desc* desc);
int xrfthip_plan_destroy(xrfthip_plan* plan);

/* axis: 0 = y, 1 = x.  h_window: n doubles (scipy.signal.windows.<name>(n, sym=False)).  NULL clears. */
int xrfthip_plan_set_window(xrfthip_plan* plan, int axis, const double* h_window, int64_t n);
/* h_phase: n interleaved (re,im) doubles indexed by UNSHIFTED frequency index: exp(-i 2 pi f_k lag)
 * (for CROSS: the net factor phase0 * conj(phase1)).  With XRFTHIP_PHASE_IN (inverse transforms) the table multiplies
 * the INPUT and is indexed by source position; a C2R_X plan then takes nx/2 + 1 entries on axis 1.  NULL clears. */
int xrfthip_plan_set_phase(xrfthip_plan* plan, int axis, const double* h_phase, int64_t n);
/* h_binmap: [ny][nx_out] int32 bin codes indexed by UNSHIFTED frequency indices (nx_out = nx, or nx/2+1 with
 * HALF_X); negative = not binned.  The host computes it with the reference's float64 pd.cut expression. */
int xrfthip_plan_set_binmap(xrfthip_plan* plan, const int32_t* h_binmap, int64_t ny, int64_t nx_out, int32_t nbins);

/* Per-pass timing with HIP events recorded on the exec stream around every kernel launch (bench.py's roofline
 * figure).  enable != 0 starts a fresh record; xrfthip_plan_profile_read synchronises the recorded events and
 * writes one line per pass: "<label> <launches> <total_ms>\n".  Off by default (no events are recorded). */
int xrfthip_plan_set_profiling(xrfthip_plan* plan, int enable);
int xrfthip_plan_profile_read(xrfthip_plan* plan, char* buf, size_t buflen);

/* 1 if a pass of the plan runs Bluestein's algorithm (a transform length with a prime factor that has no butterfly: numpy.fft takes
 * any length, xrft.py:439-444), else 0.  In float32 the chirp convolution leaves an error of a few 1e-7 of the spectrum's PEAK in every bin;
 * callers that hold small bins to 1e-3 run such plans in float64 (xrft_amd/engine.py does, with xrfthip_convert at both ends). */
int xrfthip_plan_uses_bluestein(const xrfthip_plan* plan);

/* Which kernel family serves the plan (decided when it is created; what xrfthip_plan_describe prints, as numbers -- callers that compose plans decide on
 * these, not on the text): *kind = xrfthip_kernel_kind, *per_workgroup = the independent sequences (columns, rows or slabs) one workgroup transforms
 * (0 where that is not a property of the plan).  ABI 0.1.3. */
typedef enum xrfthip_kernel_kind {
    XRFTHIP_K_GENERIC = 0,    /* tile_fft.h passes */
    XRFTHIP_K_FASTY = 1,      /* two-pass float32 powers of two (and the four-step 1-D form) */
    XRFTHIP_K_FASTM = 2,      /* two-pass table lengths */
    XRFTHIP_K_FASTN = 3,      /* two-pass, lengths as data (either pass may be a table kernel) */
    XRFTHIP_K_FASTM_Y = 4,    /* one axis, not the contiguous one, table lengths */
    XRFTHIP_K_FASTM_X = 5,    /* rows, table lengths */
    XRFTHIP_K_FASTG_Y = 6,    /* one axis, not the contiguous one, lengths as data */
    XRFTHIP_K_FASTG_ROWS = 7, /* groups of rows, lengths as data */
    XRFTHIP_K_FASTG = 8,      /* one pass over a small slab, lengths as data */
    XRFTHIP_K_FASTS = 9,      /* one pass over a small float32 slab in registers */
    XRFTHIP_K_FASTR = 10,     /* one pass over a long float32 row in registers */
    XRFTHIP_K_COMPOSITE = 11  /* xrfthip_desc.inner / .mid: two one-axis plans */
} xrfthip_kernel_kind;
int xrfthip_plan_kernel_info(const xrfthip_plan* plan, int32_t* kind, int32_t* per_workgroup);

size_t xrfthip_workspace_bytes(const xrfthip_plan* plan);
/* human-readable pass list (kernel, tile, grid, LDS) for logs and DESIGN.md; returns bytes written */
int xrfthip_plan_describe(const xrfthip_plan* plan, char* buf, size_t buflen);

/*
 * d_in0 : [batch][ny][nx] input (dtype)
 * d_in1 : second field for CROSS, else NULL
 * d_out : COMPLEX/CROSS: complex [batch][ny][nx_out]; POWER: real [batch][ny][nx_out]; may be NULL with
 *         XRFTHIP_NO_SPECTRUM_OUT
 * d_iso : with XRFTHIP_ISO: float64 [batch][nbins] (POWER) or complex128 [batch][nbins] (CROSS), else NULL
 * d_workspace / ws_bytes : >= xrfthip_workspace_bytes(plan), 256-byte aligned
 */
int xrfthip_exec(const xrfthip_plan* plan, const void* d_in0, const void* d_in1, void* d_out, void* d_iso,
                 void* d_workspace, size_t ws_bytes, void* stream);

/* Stand-alone detrend (xrft.detrend, detrend.py:11-97) over the last `ndim` axes: out = in - trend, same dtype.
 * d_workspace: >= xrfthip_detrend_workspace_bytes(batch) bytes. */
size_t xrfthip_detrend_workspace_bytes(int64_t batch);
int xrfthip_detrend(int32_t dtype, int32_t ndim, int64_t batch, int64_t ny, int64_t nx, int32_t detrend_type,
                    const void* d_in, void* d_out, void* d_workspace, size_t ws_bytes, void* stream);

/* The same with the independent elements innermost: [batch][ny][nx][inner], mean or least-squares plane over (ny, nx) for every
 * (batch, inner) element (ndim = 2), or line over nx of [batch][nx][inner] (ndim = 1, ny = 1) -- xrft.detrend over adjacent axes that
 * are not the trailing ones, without a transposed copy.  d_workspace: >= xrfthip_detrend_inner_workspace_bytes(dtype, batch, inner). */
size_t xrfthip_detrend_inner_workspace_bytes(int32_t dtype, int64_t batch, int64_t inner);
int xrfthip_detrend_inner(int32_t dtype, int32_t ndim, int64_t batch, int64_t ny, int64_t nx, int64_t inner, int32_t detrend_type,
                          const void* d_in, void* d_out, void* d_workspace, size_t ws_bytes, void* stream);

/* The same over the last THREE axes (n0, n1, n2) of [batch][n0][n1][n2]: mean, or the least-squares hyperplane
 * a0 + a1 i + a2 j + a3 k of xrft/detrend.py:116-138 (_detrend_3d_ufunc).  Same workspace size as xrfthip_detrend. */
int xrfthip_detrend3(int32_t dtype, int64_t batch, int64_t n0, int64_t n1, int64_t n2, int32_t detrend_type,
                     const void* d_in, void* d_out, void* d_workspace, size_t ws_bytes, void* stream);

/* Tail of power_spectrum / cross_spectrum (xrft.py:740, 825) on already transformed fields, for transforms over more
 * than two axes that are composed of several plans: d_out[e] = |d_a[e]|^2 * scale (real, d_b NULL) or
 * d_a[e] * conj(d_b[e]) * scale (complex).  dtype = XRFTHIP_C64 | XRFTHIP_C128 (type of d_a / d_b). */
int xrfthip_spectrum_tail(int32_t dtype, int64_t n, const void* d_a, const void* d_b, void* d_out, double scale, void* stream);

/* d_out[e] = arg(d_a[e]) in [-pi, pi], real of the same precision (dtype = XRFTHIP_C64 | XRFTHIP_C128): numpy.angle of a stored cross
 * spectrum, for xrft.cross_phase (xrft.py:838-874) on calls whose cross spectrum is composed of several plans. */
int xrfthip_angle(int32_t dtype, int64_t n, const void* d_a, void* d_out, void* stream);

/* The same over [outer][na][inner] with the real-dim factor [1, 2, ..., 2, (1 if last_is_one)] along axis `na` (xrft.py:673-682):
 * the kept half of a real transform counts twice, except k = 0 and, for an even length, the Nyquist sample. */
int xrfthip_spectrum_tail_axis(int32_t dtype, int64_t outer, int64_t na, int64_t inner, int32_t last_is_one, const void* d_a, const void* d_b,
                               void* d_out, double scale, void* stream);

/* out[o][i][j] = in[o][src(i)][j] over [outer][n_out][inner] (source [outer][n_in][inner]), elem_bytes = 4 | 8 | 16:
 * src(i) = d_index[i] (device int64[n_out]) or, with d_index NULL, (i - roll) mod n_in (numpy.roll: n_out == n_in).
 * The fftshift / ifftshift of the backend module (xrft.py:446-447, 617-621) and the gather of spectra that are not stored
 * in fftshift order (xrft.ifft).  d_out must not alias d_in. */
int xrfthip_gather_axis(int32_t elem_bytes, int64_t outer, int64_t n_out, int64_t inner, int64_t n_in, const int64_t* d_index, int64_t roll,
                        const void* d_in, void* d_out, void* stream);

/* d_out[b][j] = (j < n_in ? d_in[b][j] : 0) * d_table[j], j < n_out: zero padding or truncation with a pointwise complex factor.
 * dtype of d_in F32|F64|C64|C128; d_table (min(n_in, n_out) entries) and d_out complex of that precision.  The pointwise steps of Bluestein's algorithm
 * through global memory, for lengths with a prime factor > XRFTHIP_MAX_RADIX that exceed the in-tile bound
 * (numpy.fft takes any length: xrft.py:398-447). */
int xrfthip_table_mul(int32_t dtype, int64_t batch, int64_t n_in, int64_t n_out, const void* d_in, const void* d_table, void* d_out, void* stream);

/* d_out[e] = (precision of dtype_out) d_in[e], e < n: F32 <-> F64 or C64 <-> C128 (n counts real or complex elements of dtype_in's kind).
 * The precision change around float32 plans that run in float64 (Bluestein lengths, see xrfthip_plan_uses_bluestein). */
int xrfthip_convert(int32_t dtype_in, int32_t dtype_out, int64_t n, const void* d_in, void* d_out, void* stream);

/* d_out[o][i] = scale * sum_k d_in[o][k][i] over [outer][n][inner] -> [outer][inner], same dtype (F32|F64|C64|C128), accumulated in float64
 * in the order k = 0, 1, ...: the sum / mean over a batch dimension (scale = 1 / n), bit-reproducible.  What the reference's users do with
 * batches of isotropic spectra (`iso_ps.mean("time")`, xrft/tests/test_xrft.py:1011-1013) and, with scale = 1 / (slabs of ALL ranks), the
 * local term of the multi-GPU batch mean (xrft_amd/dist.py: one all_reduce(SUM) of nbins values finishes it). */
int xrfthip_reduce_axis(int32_t dtype, int64_t outer, int64_t n, int64_t inner, const void* d_in, void* d_out, double scale, void* stream);

/* The memory floor of the headline path (BASELINE.json configs[2]: xrft.power_spectrum of 4096 x 4096 float32 slabs, xrft/xrft.py:685-750),
 * measured on the spot: the access patterns of the two passes with the transforms removed (csrc/selftest.h), and a plain copy.
 * `reps` timed rounds after one warm-up; us[0..2] = microseconds per slab of the copy (nslab x 64 MB read + written), the column-pass
 * skeleton and the row-pass skeleton.  d_in: [nslab][4096][4096] float32; d_w2: scratch, nslab * 2052 * 4096 * 8 bytes; d_out:
 * [nslab][4096][4096] float32, overwritten.  Synchronises the stream: a measurement for bench.py, not part of the hot path. */
int xrfthip_selftest_floor(const void* d_in, void* d_w2, void* d_out, int64_t nslab, int32_t reps, double* us, void* stream);

/* Stand-alone radial bin-sum of an existing spectrum (xrft.isotropize, xrft.py:948-1010):
 * d_in [batch][ny][nx] (dtype F32|F64|C64|C128), d_binmap device int32 [ny][nx] (bin of each sample, < 0 = none),
 * d_iso float64|complex128 [batch][nbins] (every entry written).  The sums are bit-reproducible (per-workgroup integer
 * fixed-point sums, combined in a fixed order); d_workspace holds the per-workgroup tables,
 * xrfthip_isotropize_workspace_bytes(...) bytes (0 = bad arguments).  Any nbins >= 1. */
size_t xrfthip_isotropize_workspace_bytes(int32_t dtype, int64_t batch, int64_t ny, int64_t nx, int32_t nbins);
int xrfthip_isotropize(int32_t dtype, int64_t batch, int64_t ny, int64_t nx, const void* d_in,
                       const int32_t* d_binmap, int32_t nbins, void* d_iso, void* d_workspace, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XRFT_HIP_H */
