// fastp2.h -- the power-of-two transform core of the specialised float32 kernels (fasty.h): lengths 256, 512, 1024, 2048,
// 4096.  (Round 1's three-pass x-first pipeline -- rows, columns, untile + mirror -- lived here; the two-pass y-first
// pipeline of fasty.h replaced it for every mode: DESIGN.md 3.2, 4.)
// Core: an N-point complex FFT (N = 256 R3, R3 = 1, 2, 4, 8, 16; R3 = 1 keeps the exchange and skips the butterfly) by N/16 threads, 16 points per thread held in registers,
// radix 16 x 16 x R3 (decimation in frequency) with two padded LDS exchanges; twiddles W^(u k), k = 1..15, are
// generated in registers from one table load W^u by a depth-4 product tree (no strided table gathers).
#pragma once
#include <type_traits>
#include "aux_kernels.h"

namespace xrft {

typedef C2<float> cf;
struct alignas(16) F4 { float x, y, z, w; };

// a[k] *= w1^k for k = 1..15, powers built by a product tree of depth <= 4 (error ~ 4 ulp)
template <typename T> __device__ __forceinline__ void twiddle16(C2<T>* a, C2<T> w1) {
    C2<T> w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
    C2<T> w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3), w8 = cmul(w4, w4);
    a[1] = cmul(a[1], w1); a[2] = cmul(a[2], w2); a[3] = cmul(a[3], w3); a[4] = cmul(a[4], w4);
    a[5] = cmul(a[5], w5); a[6] = cmul(a[6], w6); a[7] = cmul(a[7], w7); a[8] = cmul(a[8], w8);
    a[9] = cmul(a[9], cmul(w8, w1)); a[10] = cmul(a[10], cmul(w8, w2)); a[11] = cmul(a[11], cmul(w8, w3));
    a[12] = cmul(a[12], cmul(w8, w4)); a[13] = cmul(a[13], cmul(w8, w5)); a[14] = cmul(a[14], cmul(w8, w6));
    a[15] = cmul(a[15], cmul(w8, w7));
}

template <int N> struct P2 {
    static_assert(N == 256 || N == 512 || N == 1024 || N == 2048 || N == 4096, "radix 16 x 16 x {1, 2, 4, 8, 16}");
    static constexpr int NT = N / 16;    // threads per sequence
    static constexpr int R3 = N / 256;   // last radix
    static constexpr int NB = 16 / R3;   // last-pass butterflies per thread
    static constexpr int S1 = NT + R3;   // exchange 1: 16 blocks of NT, padded so that the strided reads spread over the banks
    static constexpr int RP = R3 + 1;    // exchange 2: runs of R3, padded to an odd length
    static constexpr int S2 = 16 * RP;   // = 16 (mod 32): two half-waves read disjoint bank sets
    static constexpr int LDS = N + 256;  // elements one sequence needs (>= 16 S1, = 16 S2, > natural-order slots)
};
__device__ __forceinline__ int nat16(int k) { return k + (k >> 4); }  // natural-order slot of frequency k (1 pad per 16)

// fill the stage-2 table (16 * R3 entries) from the W_N^k table; visible after the next barrier
template <int N> __device__ __forceinline__ void fill_tw2(cf* tw2, const cf* __restrict__ tw, int tid, int nthreads) {
    typedef P2<N> G;
    for (int e = tid; e < 16 * G::R3; e += nthreads) {
        const int k = e / G::R3, v = e % G::R3;
        tw2[e] = tw[16 * v * k];  // W_N^(16 v k), 16 v k < N
    }
}

}  // namespace xrft
