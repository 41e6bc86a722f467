// fused.hip -- skeleton of the PERSISTENT FUSED y-first pipeline (DESIGN.md 9.1 -> 3.2): column units of slab n+LAG and row
// units of slab n run side by side in ONE launch, the intermediate W2 lives in a ring of RING slabs that is meant to stay in
// the 256-MB Infinity Cache.  Memory access patterns, LDS footprint and workgroup shape of the real 4096^2 kernels
// (fasty.h), optional dummy arithmetic, and a CHECK of every 16-byte piece handed from a column unit to a row unit
// (the hand-off protocol of cdna_hip_programming.md Guideline 16, under real load).
//   column unit (slab, xb < 512): 8 adjacent real columns (32-byte row segments, 16 rows per thread), writes 2 x 2049 pieces
//                                 of 16 bytes into W2[ky/4][x/8][set][ky%4][4 complex]
//   row unit    (slab, j < 513):  rows ky = 4j..4j+3 of W2 = one contiguous 128-KB block; writes 8 output rows (direct + mirror)
// Work is handed out by one queue per XCD (a returning atomicAdd on a head word); a row unit waits for its slab's 512
// column units (counter), a column unit for the 513 row units that read the ring slot before (counter).
// hipcc --offload-arch=gfx950 -O3 fused.hip -o fused
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <unistd.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int NY = 4096, NX = 4096, NROWP = 2052;           // rows of W2 per slab (ny/2 + 1 padded to 4)
constexpr size_t W2_BYTES = (size_t)NROWP * NX * 8;         // 67.2 MB
constexpr int CU_PER_SLAB = 512, RU_PER_SLAB = 513, RU_PER_XCD = 65, MAXS = 256;

struct Ctl {
    unsigned head[8 * 32];   // per-XCD queue heads, 128 bytes apart
    unsigned cdone[MAXS];    // column units finished, per slab
    unsigned rdone[MAXS];    // row units that have READ their block, per slab
    unsigned timeout, errors, checked, xcdmis;
};

#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ unsigned expect_word(unsigned tag, unsigned piece) { return tag * 2654435761u + piece * 40503u + 12345u; }

__device__ __forceinline__ bool wait_ge(unsigned* p, unsigned want, unsigned* tmo, int slp = 1) {
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(p, RLX, AGENT) >= want) return true;
        if ((spins & 63u) == 63u && __hip_atomic_load(tmo, RLX, AGENT)) return false;  // someone gave up: do not hang
        if (spins > (1u << 15)) { __hip_atomic_store(tmo, 1u, RLX, AGENT); return false; }
        for (int z = 0; z < slp; ++z) __builtin_amdgcn_s_sleep(8);
    }
}

// WP: 0 plain stores + release fence, 1 sc1 (write-through) stores, 2 non-temporal stores + release fence
template <int WP>
__device__ __forceinline__ void w2_store(char* base, unsigned off, v4u v) {
    if (WP == 1 || WP == 3) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, WP == 1 ? 16 : 17);
    } else if (WP == 2) {
        __builtin_nontemporal_store(v, reinterpret_cast<v4u*>(base + off));
    } else {
        *reinterpret_cast<v4u*>(base + off) = v;
    }
}

template <int WP, int INP>  // INP: 0 plain loads of the input, 1 non-temporal
__device__ __forceinline__ void col_unit(const float* in_slab, char* w2_slab, int xb, unsigned tag, int work, unsigned* done, float* sink) {
    const int tid = threadIdx.x, g = tid & 1, u = tid >> 1;
    const char* src = reinterpret_cast<const char*>(in_slab) + ((size_t)xb * 8 + 4 * g) * 4;
    v4f raw[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const v4f* ap = reinterpret_cast<const v4f*>(src + (size_t)(u + 256 * q) * NX * 4);
        raw[q] = INP ? __builtin_nontemporal_load(ap) : *ap;
    }
    for (int w = 0; w < work; ++w) {
#pragma unroll
        for (int q = 0; q < 16; ++q) raw[q] = raw[q] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += raw[q].x + raw[q].y + raw[q].z + raw[q].w;
    if (s == 1.2345f) *sink = s;
#pragma unroll
    for (int set = 0; set < 2; ++set)
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int k = u + 256 * q;
            if (q < 8 || u == 0) {
                const unsigned piece = ((((unsigned)(k >> 2) * 512u + (unsigned)xb) * 2u + set) * 8u) + (k & 3) * 2 + g;  // 16-byte pieces
                v4u v; v.x = expect_word(tag, piece); v.y = piece; v.z = tag; v.w = __float_as_uint(s);
                w2_store<WP>(w2_slab, piece * 16u, v);
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains
    __syncthreads();
    if (tid == 0) {
        if (WP != 1 && WP != 3) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __hip_atomic_fetch_add(done, 1u, RLX, AGENT);
    }
}

template <int W2L>  // W2L: 0 plain loads of the intermediate, 1 non-temporal, 2 sc1
__device__ __forceinline__ void row_unit(const char* w2_slab, float* out_slab, int j, unsigned tag, int work, unsigned* rdone, Ctl* ctl, bool check, int outp = 0) {
    const int tid = threadIdx.x;
    const char* src = w2_slab + (size_t)j * 131072;
    v4u v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const v4u* ap = reinterpret_cast<const v4u*>(src) + tid + 512 * r;
        if (W2L == 1) v[r] = __builtin_nontemporal_load(ap);
        else if (W2L == 2) { __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0x7fffffff, 0x00020000); v[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + 512 * r) * 16, 0, 16); }
        else v[r] = *ap;
    }
    unsigned bad = 0, acc = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned piece = (unsigned)j * 8192u + tid + 512 * r;
        const bool valid = j < 512 || ((piece & 7u) < 2u);  // the Nyquist unit holds one live row in four
        if (check && valid && (v[r].x != expect_word(tag, piece) || v[r].y != piece || v[r].z != tag)) ++bad;
        acc += v[r].x ^ v[r].w;
    }
    __syncthreads();  // every wave holds its block in registers: the ring slot may be overwritten as far as this unit goes
    if (tid == 0) __hip_atomic_fetch_add(rdone, 1u, RLX, AGENT);
    if (check && bad) __hip_atomic_fetch_add(&ctl->errors, bad, RLX, AGENT);
    float f = __uint_as_float((acc & 0x007fffffu) | 0x3f800000u);
    for (int w = 0; w < work; ++w) {
#pragma unroll
        for (int r = 0; r < 16; ++r) f = f * 1.0001f + __uint_as_float((v[r].x & 0x007fffffu) | 0x3f800000u);
    }
    v4f o = {f, f + 1.f, f + 2.f, f + 3.f};
    const int nvalid = j < 512 ? 4 : 1;
    for (int r = 0; r < nvalid; ++r) {
        const int ky = 4 * j + r;
        const int rd = (ky + NY / 2) & (NY - 1), rm = ((NY - ky) + NY / 2) & (NY - 1);
        v4f* d0 = reinterpret_cast<v4f*>(out_slab + (size_t)rd * NX);
        if (outp) { d0[tid] = o; d0[tid + 512] = o; } else { __builtin_nontemporal_store(o, d0 + tid); __builtin_nontemporal_store(o, d0 + tid + 512); }
        if (ky != 0 && ky != NY / 2) {
            v4f* d1 = reinterpret_cast<v4f*>(out_slab + (size_t)rm * NX);
            if (outp) { d1[tid] = o; d1[tid + 512] = o; } else { __builtin_nontemporal_store(o, d1 + tid); __builtin_nontemporal_store(o, d1 + tid + 512); }
        }
    }
}

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}

// ------------------------------------------------------------------------------------------------
// the fused persistent kernel.  Queue position t of XCD x: pair index m = t >> 1 on a grid of 65 pairs per slab;
//   t even: column unit (slab m / 65, xb = x * 64 + m % 65), void when m % 65 == 64
//   t odd : row unit of pair index m - LAGU: (slab r / 65, j = (r % 65) * 8 + x), void when j >= 513
// ------------------------------------------------------------------------------------------------
template <int WP, int INP, int W2L>
__global__ void __launch_bounds__(512, 2) k_fused(const float* in, char* w2, float* out, Ctl* ctl, int ns, int ring, int lagu, int work, int check, int in_slabs, float* sink, int slp, int outp) {
    extern __shared__ float lds_fp[];
    if (ns < 0) lds_fp[threadIdx.x] = 0.f;
    const int tid = threadIdx.x;
    const unsigned xcd = blockIdx.x & 7u;   // the queue a workgroup serves is a software role: results never depend on placement
    if (tid == 0 && xcc_id() != xcd) __hip_atomic_fetch_add(&ctl->xcdmis, 1u, RLX, AGENT);   // (observed: block b runs on XCD b % 8)
    const unsigned tend = 2u * ((unsigned)ns * 65u + (unsigned)lagu);
    unsigned iters = 0;
    for (;;) {
        unsigned* s_t = reinterpret_cast<unsigned*>(lds_fp);
        if (tid == 0) *s_t = __hip_atomic_fetch_add(&ctl->head[xcd * 32], 1u, RLX, AGENT);
        __syncthreads();
        const unsigned t = __builtin_amdgcn_readfirstlane(*s_t);   // provably wave-uniform: scalar branches around the barriers below
        __syncthreads();
        if (t >= tend || ++iters > 8192u) break;
        const unsigned m = t >> 1;
        if ((t & 1u) == 0) {
            const unsigned slab = m / 65u, jj = m % 65u;
            if (jj == 64u || slab >= (unsigned)ns) continue;
            if (slab >= (unsigned)ring) {  // the slot's previous contents have been read
                if (tid == 0) wait_ge(&ctl->rdone[slab - ring], RU_PER_SLAB, &ctl->timeout, slp);
                __syncthreads();
            }
            col_unit<WP, INP>(in + (size_t)(slab % in_slabs) * NY * NX, w2 + (size_t)(slab % ring) * W2_BYTES, (int)(xcd * 64u + jj), slab + 1u, work, &ctl->cdone[slab], sink);
        } else {
            if (m < (unsigned)lagu) continue;
            const unsigned r = m - (unsigned)lagu, slab = r / 65u, j = (r % 65u) * 8u + xcd;
            if (j >= (unsigned)RU_PER_SLAB || slab >= (unsigned)ns) continue;
            if (tid == 0) {
                wait_ge(&ctl->cdone[slab], CU_PER_SLAB, &ctl->timeout, slp);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            row_unit<W2L>(w2 + (size_t)(slab % ring) * W2_BYTES, out + (size_t)(slab % in_slabs) * NY * NX, (int)j, slab + 1u, work, &ctl->rdone[slab], ctl, check != 0, outp);
        }
    }
}

// the same units as two plain launches (the round-2 structure): XCD-aware unit order in the column pass
template <int WP, int INP>
__global__ void __launch_bounds__(512, 2) k_cols(const float* in, char* w2, Ctl* ctl, int ns, int work, float* sink) {
    extern __shared__ float lds_fp[];
    if (ns < 0) lds_fp[threadIdx.x] = 0.f;
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3, slab = jq / 64, xb = xcd * 64 + jq % 64;
    col_unit<WP, INP>(in + (size_t)slab * NY * NX, w2 + (size_t)slab * W2_BYTES, xb, slab + 1u, work, &ctl->cdone[slab], sink);
}
template <int W2L>
__global__ void __launch_bounds__(512, 2) k_rows(const char* w2, float* out, Ctl* ctl, int ns, int work, int check) {
    extern __shared__ float lds_fp[];
    if (ns < 0) lds_fp[threadIdx.x] = 0.f;
    const int slab = blockIdx.x / RU_PER_SLAB, j = blockIdx.x % RU_PER_SLAB;
    row_unit<W2L>(w2 + (size_t)slab * W2_BYTES, out + (size_t)slab * NY * NX, j, slab + 1u, work, &ctl->rdone[slab], ctl, check != 0);
}

__global__ void k_xcc(unsigned* hist) { if (threadIdx.x == 0) atomicAdd(&hist[(blockIdx.x & 7) * 16 + (xcc_id() & 15)], 1u); }

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    {
        unsigned* dh; CK(hipMalloc(&dh, 128 * 4)); CK(hipMemset(dh, 0, 128 * 4));
        asm volatile("" ::: "memory");
        unsigned hh[128];
        hipLaunchKernelGGL(k_xcc, dim3(4096), dim3(64), 0, 0, dh); CK(hipMemcpy(hh, dh, sizeof(hh), hipMemcpyDeviceToHost));
        printf("XCC_ID by blockIdx %% 8 (rows) x hardware id (columns 0..15):\n");
        for (int r = 0; r < 8; ++r) { for (int c = 0; c < 16; ++c) printf(" %4u", hh[r * 16 + c]); printf("\n"); }
    }
    const int NS = 64, NSEP = 32;   // slabs per fused launch; slabs of the two-launch baseline (its W2 is NSEP slabs long)
    const size_t LDSB = 71680;
    float* in; float* out; char* w2; Ctl* ctl; float* sink;
    CK(hipMalloc(&in, (size_t)NS * NY * NX * 4)); CK(hipMemset(in, 0, (size_t)NS * NY * NX * 4));
    CK(hipMalloc(&out, (size_t)NS * NY * NX * 4)); CK(hipMemset(out, 0, (size_t)NS * NY * NX * 4));
    CK(hipMalloc(&w2, (size_t)NSEP * W2_BYTES)); CK(hipMemset(w2, 0xff, (size_t)NSEP * W2_BYTES));
    CK(hipMalloc(&ctl, sizeof(Ctl))); CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Ctl h; int nfail = 0;
    hipStream_t side; CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    printf("slab: in 67.1 MB, W2 %.1f MB, out 67.1 MB; %d slabs per fused launch\n", W2_BYTES / 1e6, NS);

    // ---- baseline: two launches per group of NSEP slabs, the same unit code
#define BASE(WP, INP, W2L, WORK) do { \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cols<WP, INP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSB)); \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rows<W2L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSB)); \
        float best = 1e9f, bc = 0, br = 0; unsigned err = 0; \
        for (int rep = 0; rep < 4; ++rep) { \
            CK(hipMemsetAsync(ctl, 0, sizeof(Ctl))); \
            hipEvent_t em; CK(hipEventCreate(&em)); \
            CK(hipEventRecord(e0)); \
            k_cols<WP, INP><<<NSEP * 512, 512, LDSB>>>(in, w2, ctl, NSEP, WORK, sink); \
            CK(hipEventRecord(em)); \
            k_rows<W2L><<<NSEP * RU_PER_SLAB, 512, LDSB>>>(w2, out, ctl, NSEP, WORK, 1); \
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
            float ms, m1, m2; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&m1, e0, em)); CK(hipEventElapsedTime(&m2, em, e1)); \
            if (ms < best) { best = ms; bc = m1; br = m2; } \
            CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost)); err += h.errors; \
        } \
        printf("two launches  wp=%d inp=%d w2l=%d work=%2d: %6.1f us / slab (cols %5.1f rows %5.1f)  errors %u\n", WP, INP, W2L, WORK, best * 1e3 / NSEP, bc * 1e3 / NSEP, br * 1e3 / NSEP, err); \
    } while (0)
    BASE(2, 0, 0, 0); BASE(0, 0, 0, 0); BASE(1, 0, 0, 0); BASE(2, 1, 0, 0); BASE(2, 0, 0, 32);

    // ---- fused
#define FUSED(WP, INP, W2L, RING, LAGU, WORK, GRID) FUSEDX(WP, INP, W2L, RING, LAGU, WORK, GRID, 1, 0)
#define FUSEDX(WP, INP, W2L, RING, LAGU, WORK, GRID, SLP, OUTP) do { \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused<WP, INP, W2L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSB)); \
        float best = 1e9f; unsigned err = 0, tmo = 0, mis = 0; \
        for (int rep = 0; rep < 3; ++rep) { \
            CK(hipMemsetAsync(ctl, 0, sizeof(Ctl))); \
            CK(hipEventRecord(e0)); \
            k_fused<WP, INP, W2L><<<GRID, 512, LDSB>>>(in, w2, out, ctl, NS, RING, LAGU, WORK, 1, NS, sink, SLP, OUTP); \
            CK(hipEventRecord(e1)); \
            for (int w = 0; hipEventQuery(e1) == hipErrorNotReady; ++w) { \
                usleep(1000); \
                if (w > 5000) { \
                    CK(hipMemcpyAsync(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost, side)); CK(hipStreamSynchronize(side)); \
                    printf("HUNG: wp=%d ring=%d lag=%d  timeout %u errors %u\n   heads:", WP, RING, LAGU, h.timeout, h.errors); for (int q = 0; q < 8; ++q) printf(" %u", h.head[q * 32]); \
                    printf("\n   cdone:"); for (int q = 0; q < 8; ++q) printf(" %u", h.cdone[q]); printf("\n   rdone:"); for (int q = 0; q < 8; ++q) printf(" %u", h.rdone[q]); printf("\n"); \
                    _exit(3); \
                } \
            } \
            CK(hipEventSynchronize(e1)); \
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; \
            CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost)); err += h.errors; tmo += h.timeout; mis += h.xcdmis; \
        } \
        printf("fused wp=%d inp=%d w2l=%d ring=%d lag=%3d work=%2d grid=%4d sleep=%2d outp=%d: %6.1f us / slab  errors %u timeouts %u\n", WP, INP, W2L, RING, LAGU, WORK, GRID, SLP, OUTP, best * 1e3 / NS, err, tmo); \
        if (tmo) { printf("   heads:"); for (int q = 0; q < 8; ++q) printf(" %u", h.head[q * 32]); printf("  cdone[0..3] %u %u %u %u  rdone[0..3] %u %u %u %u\n", h.cdone[0], h.cdone[1], h.cdone[2], h.cdone[3], h.rdone[0], h.rdone[1], h.rdone[2], h.rdone[3]); if (++nfail >= 3) { printf("giving up\n"); return 1; } } \
        fflush(stdout); \
    } while (0)
    if (argc > 1) {   // second exploration (r03d)
        FUSED(1, 0, 0, 3, 72, 0, 256); FUSED(1, 0, 0, 3, 81, 0, 256); FUSED(1, 0, 0, 3, 98, 0, 256); FUSED(1, 0, 0, 3, 114, 0, 256); FUSED(1, 0, 0, 2, 65, 0, 256);
        FUSED(1, 0, 0, 4, 130, 0, 256); FUSED(1, 0, 0, 8, 260, 0, 256);
        FUSED(3, 0, 0, 3, 98, 0, 256); FUSED(1, 0, 2, 3, 98, 0, 256); FUSED(1, 1, 0, 3, 98, 0, 256); FUSED(1, 0, 1, 3, 98, 0, 256);
        FUSEDX(1, 0, 0, 3, 98, 0, 256, 8, 0); FUSEDX(1, 0, 0, 3, 98, 0, 256, 1, 1); FUSEDX(1, 0, 0, 3, 98, 0, 512, 8, 0); FUSEDX(1, 0, 0, 3, 110, 0, 512, 8, 0);
        FUSED(1, 0, 0, 3, 98, 32, 256); FUSED(1, 0, 0, 3, 98, 64, 256); FUSED(1, 0, 0, 3, 98, 32, 384); FUSED(1, 0, 0, 3, 98, 0, 384); FUSED(1, 0, 0, 3, 98, 0, 320);
        return 0;
    }
    // store policy of the intermediate
    FUSED(1, 0, 0, 3, 98, 0, 512); FUSED(0, 0, 0, 3, 98, 0, 512); FUSED(2, 0, 0, 3, 98, 0, 512);
    // lag (in pairs of units per XCD: 65 = one slab) and ring
    FUSED(1, 0, 0, 2, 33, 0, 512); FUSED(1, 0, 0, 2, 65, 0, 512); FUSED(1, 0, 0, 3, 65, 0, 512); FUSED(1, 0, 0, 3, 130, 0, 512);
    FUSED(1, 0, 0, 4, 130, 0, 512); FUSED(1, 0, 0, 4, 195, 0, 512); FUSED(1, 0, 0, 8, 260, 0, 512); FUSED(1, 0, 0, 32, 1300, 0, 512);
    // load policies
    FUSED(1, 1, 0, 3, 98, 0, 512); FUSED(1, 0, 1, 3, 98, 0, 512); FUSED(1, 0, 2, 3, 98, 0, 512); FUSED(1, 1, 2, 3, 98, 0, 512);
    // with dummy arithmetic (about the VALU issue time of the real kernels)
    FUSED(1, 0, 0, 3, 98, 32, 512); FUSED(1, 0, 0, 3, 130, 32, 512); FUSED(1, 0, 0, 2, 65, 32, 512);
    // grid size (1 / 2 workgroups per CU; more than fit stay queued and cost nothing)
    FUSED(1, 0, 0, 3, 98, 0, 256); FUSED(1, 0, 0, 3, 98, 0, 768);
    return 0;
}
