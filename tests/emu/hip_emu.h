// hip_emu.h -- TEST INFRASTRUCTURE ONLY.  Not part of the product, never shipped, never a fallback.
//
// A minimal single-process emulation of the handful of HIP constructs used by xrft_amd/csrc, so that the
// *index arithmetic* of the kernels (tile decode, digit reversal, twiddle indices, shift/mirror remaps,
// four-step addressing) can be exercised on the GPU-less build container through the very same C ABI.
// The GPU box has a 90-minute budget per round; this catches addressing bugs before spending it.
//
// Model: one OS thread runs one workgroup at a time; every GPU thread of the workgroup is a ucontext fiber;
// __syncthreads() yields to a round-robin scheduler that releases the barrier once every live fiber has
// arrived.  Workgroups of a launch are distributed over a small pool of OS threads.  Launches are synchronous.
// Built only by tests/emu/build_emu.py with -DXRFT_EMULATE (see xrft_amd/csrc/gpu_rt.h).
#pragma once
#include <sys/mman.h>
#include <ucontext.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

namespace emu {

struct Fiber {
    ucontext_t ctx;
    dim3 tid;
    int state;  // 0 runnable, 1 waiting at barrier, 2 done
    void* stack;
};

struct BlockCtx {
    dim3 blockIdx, blockDim, gridDim;
    unsigned char* smem;
    double* shfl;  // one slot per thread: cross-lane exchange of the emulated __shfl_xor
    ucontext_t sched;
    Fiber* cur;
    const std::function<void()>* body;
};

inline BlockCtx*& tls() {
    static thread_local BlockCtx* p = nullptr;
    return p;
}

static constexpr size_t kStack = 128 * 1024;

inline void trampoline() {
    BlockCtx* b = tls();
    Fiber* f = b->cur;
    (*b->body)();
    f->state = 2;
    swapcontext(&f->ctx, &b->sched);
}

inline void barrier() {
    BlockCtx* b = tls();
    Fiber* f = b->cur;
    f->state = 1;
    swapcontext(&f->ctx, &b->sched);
}

struct Worker {
    std::vector<Fiber> fibers;
    std::vector<unsigned char> smem;
    std::vector<double> shfl;
    void ensure(size_t n) {
        if (shfl.size() < n) shfl.resize(n);
        while (fibers.size() < n) {
            Fiber f;
            f.stack = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (f.stack == MAP_FAILED) { perror("emu mmap"); abort(); }
            f.state = 2;
            fibers.push_back(f);
        }
    }
    ~Worker() { for (auto& f : fibers) munmap(f.stack, kStack); }
    void run_block(const std::function<void()>& body, dim3 grid, dim3 block, dim3 bidx, size_t smem_bytes) {
        const size_t nt = (size_t)block.x * block.y * block.z;
        ensure(nt);
        BlockCtx ctx;
        ctx.blockIdx = bidx; ctx.blockDim = block; ctx.gridDim = grid;
#ifdef XRFT_EMU_ASAN
        // the AddressSanitizer build (scripts/run_emu_asan.sh): the workgroup's LDS is an allocation of exactly the launch's dynamic size, so that an access
        // behind it is a heap-buffer-overflow report instead of a read of a previous launch's bytes
        void* exact = nullptr;
        if (posix_memalign(&exact, 64, smem_bytes ? smem_bytes : 64) != 0) abort();
        ctx.smem = (unsigned char*)exact;
#else
        if (smem.size() < smem_bytes + 64) smem.resize(smem_bytes + 64);
        ctx.smem = (unsigned char*)(((uintptr_t)smem.data() + 63) & ~(uintptr_t)63);
#endif
        ctx.body = &body;
        ctx.shfl = shfl.data();
        tls() = &ctx;
        size_t t = 0;
        for (unsigned z = 0; z < block.z; ++z)
            for (unsigned y = 0; y < block.y; ++y)
                for (unsigned x = 0; x < block.x; ++x, ++t) {
                    Fiber& f = fibers[t];
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = nullptr;
                    f.tid = dim3(x, y, z);
                    f.state = 0;
                    makecontext(&f.ctx, (void (*)())trampoline, 0);
                }
        for (;;) {
            size_t live = 0;
            for (size_t i = 0; i < nt; ++i) {
                Fiber& f = fibers[i];
                if (f.state == 2) continue;
                f.state = 0;
                ctx.cur = &f;
                swapcontext(&ctx.sched, &f.ctx);
                if (f.state != 2) ++live;
            }
            if (live == 0) break;
        }
        tls() = nullptr;
#ifdef XRFT_EMU_ASAN
        free(exact);
#endif
    }
};

inline int n_workers() {
    static int n = [] {
        const char* e = getenv("XRFT_EMU_THREADS");
        int v = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        return v < 1 ? 1 : (v > 16 ? 16 : v);
    }();
    return n;
}

template <typename F>
inline void launch_body(const F& body, dim3 grid, dim3 block, size_t smem_bytes) {
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    std::function<void()> fn = body;
    std::atomic<size_t> next{0};
    auto work = [&]() {
        Worker w;
        for (;;) {
            size_t b = next.fetch_add(1);
            if (b >= nblocks) break;
            dim3 bidx((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y)));
            w.run_block(fn, grid, block, bidx, smem_bytes);
        }
    };
    int nw = n_workers();
    if ((size_t)nw > nblocks) nw = (int)nblocks;
    if (nw <= 1) { work(); return; }
    std::vector<std::thread> th;
    for (int i = 0; i < nw; ++i) th.emplace_back(work);
    for (auto& t : th) t.join();
}

template <typename K, typename... A>
inline void launch(K kernel, dim3 grid, dim3 block, size_t smem, hipStream_t, A... args) {
    launch_body([=]() { kernel(args...); }, grid, block, smem);
}

template <typename T>
inline T atomic_add(T* p, T v) {
    T old = *reinterpret_cast<volatile T*>(p), desired;
    do { desired = old + v; } while (!__atomic_compare_exchange(p, &old, &desired, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return old;
}
}  // namespace emu

#define threadIdx (emu::tls()->cur->tid)
#define blockIdx (emu::tls()->blockIdx)
#define blockDim (emu::tls()->blockDim)
#define gridDim (emu::tls()->gridDim)
#define __syncthreads() emu::barrier()

// wave-level exchange (64 lanes): EVERY thread of the workgroup must call it (it contains two barriers)
inline double __shfl_xor(double v, int mask) {
    emu::BlockCtx* b = emu::tls();
    const unsigned t = b->cur->tid.x;
    b->shfl[t] = v;
    emu::barrier();
    const double r = b->shfl[t ^ (unsigned)mask];
    emu::barrier();
    return r;
}
inline double __shfl_down(double v, unsigned delta, int width = 64) {
    emu::BlockCtx* b = emu::tls();
    const unsigned t = b->cur->tid.x;
    b->shfl[t] = v;
    emu::barrier();
    const double r = ((t % (unsigned)width) + delta < (unsigned)width) ? b->shfl[t + delta] : v;
    emu::barrier();
    return r;
}
inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
inline float __uint_as_float(unsigned v) { float f; memcpy(&f, &v, 4); return f; }
struct alignas(16) uint4 { unsigned x, y, z, w; };
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
inline unsigned __float_as_uint(float f) { unsigned v; memcpy(&v, &f, 4); return v; }
using std::max;
using std::min;

inline double atomicAdd(double* p, double v) { return emu::atomic_add(p, v); }
inline float atomicAdd(float* p, float v) { return emu::atomic_add(p, v); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

// ---- runtime API shim: "device memory" is host memory
inline hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) & ~(size_t)255); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulated hip error"; }
#include <chrono>
typedef std::chrono::steady_clock::time_point* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new std::chrono::steady_clock::time_point(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { *e = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(*b - *a).count(); return hipSuccess; }
template <typename F>
inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 3; return hipSuccess; }  // (a small odd "chip": persistent grids take several units per workgroup)
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 1; return hipSuccess; }
