#include "plan.h"

extern "C" {

// per-chunk partial sums of at most 32768 slabs at a time (slabs x chunks <= 32768, 8 sums each) + 8 coefficients per slab
static constexpr long long kDetrendPart = 32768;
size_t xrfthip_detrend_workspace_bytes(int64_t batch) { return ((size_t)(kDetrendPart + std::max<int64_t>(batch, 1)) * 8 * sizeof(double) + 255) & ~(size_t)255; }

int xrfthip_detrend(int32_t dtype, int32_t ndim, int64_t batch, int64_t ny, int64_t nx, int32_t detrend_type,
                    const void* d_in, void* d_out, void* d_workspace, size_t ws_bytes, void* stream) {
    if (!d_in || !d_out || dtype < XRFTHIP_F32 || dtype > XRFTHIP_C128 || batch < 0 || ny < 1 || nx < 1) return XRFTHIP_BAD_ARG;
    if ((ndim != 1 && ndim != 2) || (ndim == 1 && ny != 1)) return XRFTHIP_BAD_ARG;
    if (detrend_type != XRFTHIP_DETREND_CONSTANT && detrend_type != XRFTHIP_DETREND_LINEAR) return XRFTHIP_BAD_ARG;
    if (ws_bytes < xrfthip_detrend_workspace_bytes(batch) || !d_workspace) return XRFTHIP_WORKSPACE_TOO_SMALL;
    if (batch == 0) return XRFTHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    double* acc = (double*)d_workspace;
    double* coef = acc + kDetrendPart * 8;
    const bool dbl = dtype == XRFTHIP_F64 || dtype == XRFTHIP_C128, cplx = dtype >= XRFTHIP_C64;
    const long long total = ny * nx;
    const size_t esz = (dbl ? 8 : 4) * (cplx ? 2 : 1);
    for (long long b0 = 0; b0 < batch; b0 += 32768) {  // grid.y limit
        const long long bc = std::min<long long>(32768, batch - b0);
        const long long chunks = std::max<long long>(1, std::min<long long>(std::min<long long>(ny, 64), kDetrendPart / bc));
        const dim3 grid((unsigned)chunks, (unsigned)bc), block(256);
        const void* src = (const char*)d_in + (size_t)b0 * total * esz;
        void* dst = (char*)d_out + (size_t)b0 * total * esz;
        const size_t lds = 6 * 256 * sizeof(double);
#define MOM(TT, CC) do { auto k = &slab_moments_kernel<TT, CC>; XRFT_LAUNCH(k, grid, block, lds, st, src, (long long)ny, (long long)nx, total, (long long)nx, acc); } while (0)
        if (dbl) { if (cplx) MOM(double, true); else MOM(double, false); } else { if (cplx) MOM(float, true); else MOM(float, false); }
#undef MOM
        auto kf = &finalize_coef_kernel;
        XRFT_LAUNCH(kf, dim3((unsigned)bc), dim3(64), 0, st, (const double*)acc, coef + b0 * 6, bc, (long long)ny, (long long)nx, (int)detrend_type, (int)chunks);
        const long long gx = std::max<long long>(1, std::min<long long>(2048, (total + 255) / 256));
        const dim3 grid2((unsigned)gx, (unsigned)bc);
#define APP(TT, CC) do { auto k = &detrend_apply_kernel<TT, CC>; XRFT_LAUNCH(k, grid2, block, 0, st, src, dst, (long long)ny, (long long)nx, (const double*)(coef + b0 * 6)); } while (0)
        if (dbl) { if (cplx) APP(double, true); else APP(double, false); } else { if (cplx) APP(float, true); else APP(float, false); }
#undef APP
        HIP_TRY(hipGetLastError());
    }
    return XRFTHIP_OK;
}

int xrfthip_detrend3(int32_t dtype, int64_t batch, int64_t n0, int64_t n1, int64_t n2, int32_t detrend_type,
                     const void* d_in, void* d_out, void* d_workspace, size_t ws_bytes, void* stream) {
    if (!d_in || !d_out || dtype < XRFTHIP_F32 || dtype > XRFTHIP_C128 || batch < 0 || n0 < 1 || n1 < 1 || n2 < 1) return XRFTHIP_BAD_ARG;
    if (detrend_type != XRFTHIP_DETREND_CONSTANT && detrend_type != XRFTHIP_DETREND_LINEAR) return XRFTHIP_BAD_ARG;
    if (ws_bytes < xrfthip_detrend_workspace_bytes(batch) || !d_workspace) return XRFTHIP_WORKSPACE_TOO_SMALL;
    if (batch == 0) return XRFTHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    double* acc = (double*)d_workspace;
    double* coef = acc + kDetrendPart * 8;
    const bool dbl = dtype == XRFTHIP_F64 || dtype == XRFTHIP_C128, cplx = dtype >= XRFTHIP_C64;
    const long long rows = n0 * n1, total = rows * n2;
    const size_t esz = (dbl ? 8 : 4) * (cplx ? 2 : 1);
    for (long long b0 = 0; b0 < batch; b0 += 32768) {  // grid.y limit
        const long long bc = std::min<long long>(32768, batch - b0);
        const long long gx = std::max<long long>(1, std::min<long long>(rows, std::min<long long>(4096, kDetrendPart / bc)));
        const dim3 grid((unsigned)gx, (unsigned)bc), block(256);
        const void* src = (const char*)d_in + (size_t)b0 * total * esz;
        void* dst = (char*)d_out + (size_t)b0 * total * esz;
        const size_t lds = 8 * 256 * sizeof(double);
#define MOM(TT, CC) do { auto k = &block3_moments_kernel<TT, CC>; XRFT_LAUNCH(k, grid, block, lds, st, src, (long long)n0, (long long)n1, (long long)n2, acc); } while (0)
        if (dbl) { if (cplx) MOM(double, true); else MOM(double, false); } else { if (cplx) MOM(float, true); else MOM(float, false); }
#undef MOM
        auto kf = &finalize_coef3_kernel;
        XRFT_LAUNCH(kf, dim3((unsigned)((bc + 63) / 64)), dim3(64), 0, st, (const double*)acc, coef + b0 * 8, bc, (long long)n0, (long long)n1, (long long)n2, (int)detrend_type, (int)gx);
        const dim3 grid2((unsigned)std::max<long long>(1, std::min<long long>(rows, 4096)), (unsigned)bc);
#define APP(TT, CC) do { auto k = &detrend3_apply_kernel<TT, CC>; XRFT_LAUNCH(k, grid2, block, 0, st, src, dst, (long long)n0, (long long)n1, (long long)n2, (const double*)(coef + b0 * 8)); } while (0)
        if (dbl) { if (cplx) APP(double, true); else APP(double, false); } else { if (cplx) APP(float, true); else APP(float, false); }
#undef APP
        HIP_TRY(hipGetLastError());
    }
    return XRFTHIP_OK;
}

int xrfthip_spectrum_tail(int32_t dtype, int64_t n, const void* d_a, const void* d_b, void* d_out, double scale, void* stream) {
    if (!d_a || !d_out || n < 0 || (dtype != XRFTHIP_C64 && dtype != XRFTHIP_C128)) return XRFTHIP_BAD_ARG;
    if (n == 0) return XRFTHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)std::max<long long>(1, std::min<long long>(16384, (n + 255) / 256))), block(256);
#define TAIL(TT, CC) do { auto k = &spectrum_tail_kernel<TT, CC>; XRFT_LAUNCH(k, grid, block, 0, st, (const C2<TT>*)d_a, (const C2<TT>*)d_b, d_out, (long long)n, scale); } while (0)
    if (dtype == XRFTHIP_C128) { if (d_b) TAIL(double, true); else TAIL(double, false); }
    else { if (d_b) TAIL(float, true); else TAIL(float, false); }
#undef TAIL
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

int xrfthip_angle(int32_t dtype, int64_t n, const void* d_a, void* d_out, void* stream) {
    if (!d_a || !d_out || n < 0 || (dtype != XRFTHIP_C64 && dtype != XRFTHIP_C128)) return XRFTHIP_BAD_ARG;
    if (n == 0) return XRFTHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)std::max<long long>(1, std::min<long long>(16384, (n + 255) / 256))), block(256);
    if (dtype == XRFTHIP_C128) { auto k = &angle_kernel<double>; XRFT_LAUNCH(k, grid, block, 0, st, (const C2<double>*)d_a, (double*)d_out, (long long)n); }
    else { auto k = &angle_kernel<float>; XRFT_LAUNCH(k, grid, block, 0, st, (const C2<float>*)d_a, (float*)d_out, (long long)n); }
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

int xrfthip_spectrum_tail_axis(int32_t dtype, int64_t outer, int64_t na, int64_t inner, int32_t last_is_one, const void* d_a, const void* d_b,
                               void* d_out, double scale, void* stream) {
    if (!d_a || !d_out || outer < 0 || na < 1 || inner < 1 || (dtype != XRFTHIP_C64 && dtype != XRFTHIP_C128)) return XRFTHIP_BAD_ARG;
    const long long n = (long long)outer * na * inner;
    if (n == 0) return XRFTHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)std::max<long long>(1, std::min<long long>(16384, (n + 255) / 256))), block(256);
#define TAIL(TT, CC) do { auto k = &spectrum_tail_axis_kernel<TT, CC>; XRFT_LAUNCH(k, grid, block, 0, st, (const C2<TT>*)d_a, (const C2<TT>*)d_b, d_out, n, scale, (long long)na, (long long)inner, (int)last_is_one); } while (0)
    if (dtype == XRFTHIP_C128) { if (d_b) TAIL(double, true); else TAIL(double, false); }
    else { if (d_b) TAIL(float, true); else TAIL(float, false); }
#undef TAIL
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

int xrfthip_gather_axis(int32_t elem_bytes, int64_t outer, int64_t n_out, int64_t inner, int64_t n_in, const int64_t* d_index, int64_t roll,
                        const void* d_in, void* d_out, void* stream) {
    if (!d_in || !d_out || d_in == d_out || outer < 0 || n_out < 0 || inner < 0 || n_in < 1) return XRFTHIP_BAD_ARG;
    if (elem_bytes != 4 && elem_bytes != 8 && elem_bytes != 16) return XRFTHIP_BAD_ARG;
    const long long n = (long long)outer * n_out * inner;
    if (n == 0) return XRFTHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)std::max<long long>(1, std::min<long long>(16384, (n + 255) / 256))), block(256);
    struct alignas(16) E16 { double a, b; };
#define GA(EE) do { auto k = &gather_axis_kernel<EE>; XRFT_LAUNCH(k, grid, block, 0, st, (const EE*)d_in, (EE*)d_out, (long long)outer, (long long)n_out, (long long)inner, (long long)n_in, (const long long*)d_index, (long long)roll); } while (0)
    if (elem_bytes == 4) GA(float); else if (elem_bytes == 8) GA(double); else GA(E16);
#undef GA
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

int xrfthip_table_mul(int32_t dtype, int64_t batch, int64_t n_in, int64_t n_out, const void* d_in, const void* d_table, void* d_out, void* stream) {
    if (!d_in || !d_table || !d_out || dtype < XRFTHIP_F32 || dtype > XRFTHIP_C128 || batch < 0 || n_in < 1 || n_out < 1) return XRFTHIP_BAD_ARG;
    if (batch == 0) return XRFTHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long total = batch * n_out;
    const dim3 grid((unsigned)std::min<long long>((total + 255) / 256, 8LL * kCUs * 8)), block(256);
#define TM_(TT, CC) do { auto k = &table_mul_kernel<TT, CC>; XRFT_LAUNCH(k, grid, block, 0, st, d_in, (const C2<TT>*)d_table, (C2<TT>*)d_out, (long long)batch, (long long)n_in, (long long)n_out); } while (0)
    if (dtype == XRFTHIP_F32) TM_(float, false); else if (dtype == XRFTHIP_F64) TM_(double, false); else if (dtype == XRFTHIP_C64) TM_(float, true); else TM_(double, true);
#undef TM_
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

}  // extern "C"

static constexpr int kInnerMaxChunks = 256;
// Row chunks per (batch, inner tile): enough workgroups to fill the chip (a (y, x, t) array is ONE slab), never more than the rows.  The cap
// depends only on (batch, inner) -- what the workspace query knows -- and falls to 1 as soon as the tiles alone fill the chip, so the partial
// sums stay a few MB whatever the inner extent (they were 257 chunks' worth always: 6168 bytes per inner element, 103 GB for a 4096^2 grid).
int inner_chunk_cap(long long batch, long long i2) {
    const long long tiles = std::max<long long>(1, batch * ((i2 + kInnerThreads - 1) / kInnerThreads));
    return (int)std::max<long long>(1, std::min<long long>(kInnerMaxChunks, (2048 + tiles - 1) / tiles));
}
int inner_chunks(long long ny, long long batch, long long i2) { return (int)std::max<long long>(1, std::min<long long>(inner_chunk_cap(batch, i2), ny)); }
size_t detrend_inner_ws(bool cplx, long long batch, long long inner) {
    const size_t i2 = (size_t)inner * (cplx ? 2 : 1);
    return (((size_t)std::max<long long>(batch, 1) * i2 * 3 * sizeof(double) * ((size_t)inner_chunk_cap(batch, (long long)i2) + 1)) + 255) & ~(size_t)255;  // partial sums of <= cap chunks + the coefficients
}
int run_detrend_inner(int32_t dtype, int32_t ndim, long long batch, long long ny, long long nx, long long inner, int32_t kind, const void* in, void* out,
                             char* ws, hipStream_t st, long long mid) {  // (batch counts (outer, mid) pairs: [batch / mid][ny][mid][nx][inner])
    (void)ndim;
    const bool dbl = dtype == XRFTHIP_F64 || dtype == XRFTHIP_C128, cplx = dtype >= XRFTHIP_C64;
    const long long i2 = inner * (cplx ? 2 : 1);
    const int nch = inner_chunks(ny, batch, i2);
    double* part = reinterpret_cast<double*>(ws);
    double* coef = part + (size_t)batch * inner_chunk_cap(batch, i2) * i2 * 3;
    for (long long b0 = 0; b0 < batch; b0 += 65535) {  // grid.z limit
        const long long bc = std::min<long long>(65535, batch - b0);
        const int ib = (int)std::min<long long>(i2, kInnerThreads), xsn = kInnerThreads / ib;  // lanes across the inner index x column slots
        const dim3 grid((unsigned)((i2 + ib - 1) / ib), (unsigned)nch, (unsigned)bc), block(kInnerThreads);  // (tiles of the inner index on grid.x: no 65535 limit)
        // (the kernel takes the first (outer, mid) pair of the launch and addresses from the array's base: with mid > 1 a block of pairs is not a contiguous piece)
        const size_t lds = (size_t)xsn * 3 * ib * sizeof(double);
        if (dbl) { auto k = &plane_inner_moments_kernel<double>; XRFT_LAUNCH(k, grid, block, lds, st, (const double*)in, (long long)ny, (long long)nx, i2, part, ib, xsn, mid, b0); }
        else { auto k = &plane_inner_moments_kernel<float>; XRFT_LAUNCH(k, grid, block, lds, st, (const float*)in, (long long)ny, (long long)nx, i2, part, ib, xsn, mid, b0); }
    }
    {
        auto k = &plane_inner_finalize_kernel;
        XRFT_LAUNCH(k, dim3((unsigned)((batch * i2 + 255) / 256)), dim3(256), 0, st, (const double*)part, coef, (long long)batch, (long long)ny, (long long)nx, i2, nch, (int)kind);
    }
    const dim3 grid((unsigned)std::min<long long>(batch * ny, 8LL * kCUs * 4)), block(256);
    const size_t clds = (size_t)i2 * 3 * sizeof(double);  // the coefficients of one batch element in LDS (four workgroups per CU at 40 KB)
    const int lds_coef = clds <= 40 * 1024 ? 1 : 0;
    if (dbl) { auto k = &plane_inner_apply_kernel<double>; XRFT_LAUNCH(k, grid, block, lds_coef ? clds : 0, st, (const double*)in, (double*)out, (const double*)coef, (long long)batch, (long long)ny, (long long)nx, i2, lds_coef, mid); }
    else { auto k = &plane_inner_apply_kernel<float>; XRFT_LAUNCH(k, grid, block, lds_coef ? clds : 0, st, (const float*)in, (float*)out, (const double*)coef, (long long)batch, (long long)ny, (long long)nx, i2, lds_coef, mid); }
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

extern "C" {

size_t xrfthip_detrend_inner_workspace_bytes(int32_t dtype, int64_t batch, int64_t inner) {
    if (dtype < XRFTHIP_F32 || dtype > XRFTHIP_C128 || batch < 0 || inner < 1) return 0;
    return detrend_inner_ws(dtype >= XRFTHIP_C64, batch, inner);
}

int xrfthip_detrend_inner(int32_t dtype, int32_t ndim, int64_t batch, int64_t ny, int64_t nx, int64_t inner, int32_t detrend_type,
                          const void* d_in, void* d_out, void* d_workspace, size_t ws_bytes, void* stream) {
    if (!d_in || !d_out || dtype < XRFTHIP_F32 || dtype > XRFTHIP_C128 || batch < 0 || ny < 1 || nx < 1 || inner < 1) return XRFTHIP_BAD_ARG;
    if ((ndim != 1 && ndim != 2) || (ndim == 1 && ny != 1)) return XRFTHIP_BAD_ARG;
    if (detrend_type != XRFTHIP_DETREND_CONSTANT && detrend_type != XRFTHIP_DETREND_LINEAR) return XRFTHIP_BAD_ARG;
    if (inner > (1LL << 30) || nx > (1LL << 31) - 1 || ny > (1LL << 31) - 1) return XRFTHIP_BAD_ARG;  // (a row's length nx * inner is carried in 64 bits, the positions within it in 32)
    if (!d_workspace || ws_bytes < xrfthip_detrend_inner_workspace_bytes(dtype, batch, inner)) return XRFTHIP_WORKSPACE_TOO_SMALL;
    if (batch == 0) return XRFTHIP_OK;
    return run_detrend_inner(dtype, ndim, batch, ny, nx, inner, detrend_type, d_in, d_out, (char*)d_workspace, (hipStream_t)stream);
}

int xrfthip_convert(int32_t dtype_in, int32_t dtype_out, int64_t n, const void* d_in, void* d_out, void* stream) {
    if (!d_in || !d_out || n < 0) return XRFTHIP_BAD_ARG;
    const bool up = (dtype_in == XRFTHIP_F32 && dtype_out == XRFTHIP_F64) || (dtype_in == XRFTHIP_C64 && dtype_out == XRFTHIP_C128);
    const bool down = (dtype_in == XRFTHIP_F64 && dtype_out == XRFTHIP_F32) || (dtype_in == XRFTHIP_C128 && dtype_out == XRFTHIP_C64);
    if (!up && !down) return XRFTHIP_BAD_ARG;
    const long long cnt = n * (dtype_in >= XRFTHIP_C64 ? 2 : 1);
    if (cnt == 0) return XRFTHIP_OK;
    const dim3 grid((unsigned)std::min<long long>((cnt + 255) / 256, 8LL * kCUs * 4)), block(256);
    if (up) { auto k = &convert_kernel<float, double>; XRFT_LAUNCH(k, grid, block, 0, (hipStream_t)stream, (const float*)d_in, (double*)d_out, cnt); }
    else { auto k = &convert_kernel<double, float>; XRFT_LAUNCH(k, grid, block, 0, (hipStream_t)stream, (const double*)d_in, (float*)d_out, cnt); }
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

int xrfthip_reduce_axis(int32_t dtype, int64_t outer, int64_t n, int64_t inner, const void* d_in, void* d_out, double scale, void* stream) {
    if (!d_in || !d_out || dtype < XRFTHIP_F32 || dtype > XRFTHIP_C128 || outer < 0 || n < 1 || inner < 0) return XRFTHIP_BAD_ARG;
    if (outer == 0 || inner == 0) return XRFTHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long in2 = inner * (dtype >= XRFTHIP_C64 ? 2 : 1), total = outer * in2;  // complex data: two real components per sample
    const dim3 grid((unsigned)std::min<long long>((total + 255) / 256, 8LL * kCUs * 8)), block(256);
    if (dtype == XRFTHIP_F32 || dtype == XRFTHIP_C64) { auto k = &reduce_axis_kernel<float>; XRFT_LAUNCH(k, grid, block, 0, st, (const float*)d_in, (float*)d_out, (long long)outer, (long long)n, in2, scale); }
    else { auto k = &reduce_axis_kernel<double>; XRFT_LAUNCH(k, grid, block, 0, st, (const double*)d_in, (double*)d_out, (long long)outer, (long long)n, in2, scale); }
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

size_t xrfthip_isotropize_workspace_bytes(int32_t dtype, int64_t batch, int64_t ny, int64_t nx, int32_t nbins) {
    if (dtype < XRFTHIP_F32 || dtype > XRFTHIP_C128 || batch < 0 || ny < 1 || nx < 1 || nbins < 1) return 0;
    return (size_t)batch * iso_chunk_count(ny * nx) * nbins * (dtype >= XRFTHIP_C64 ? 2 : 1) * sizeof(double);
}

int xrfthip_isotropize(int32_t dtype, int64_t batch, int64_t ny, int64_t nx, const void* d_in,
                       const int32_t* d_binmap, int32_t nbins, void* d_iso, void* d_workspace, size_t ws_bytes, void* stream) {
    if (!d_in || !d_binmap || !d_iso || dtype < XRFTHIP_F32 || dtype > XRFTHIP_C128 || batch < 0 || ny < 1 || nx < 1 || nbins < 1) return XRFTHIP_BAD_ARG;
    if (batch == 0) return XRFTHIP_OK;
    if (!d_workspace || ws_bytes < xrfthip_isotropize_workspace_bytes(dtype, batch, ny, nx, nbins)) return XRFTHIP_WORKSPACE_TOO_SMALL;
    return run_radial_sums(dtype, d_in, d_binmap, batch, ny, nx, 0, 0, nbins, iso_chunk_count(ny * nx), (double*)d_workspace, (double*)d_iso, (hipStream_t)stream);
}

}  // extern "C"

