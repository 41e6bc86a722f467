// gpu_rt.h -- the one place the runtime is chosen.
// Product builds (hipcc, gfx950) get the HIP runtime.  -DXRFT_EMULATE is defined ONLY by
// tests/emu/build_emu.py, which compiles the same sources with g++ against tests/emu/hip_emu.h to
// exercise kernel index arithmetic on the GPU-less build container.  The product never selects it.
#pragma once
#ifdef XRFT_EMULATE
#include "hip_emu.h"
#define XRFT_LAUNCH(kernel, grid, block, smem, stream, ...) emu::launch(kernel, grid, block, smem, stream, __VA_ARGS__)
#define XRFT_DYN_SMEM(name) unsigned char* name = emu::tls()->smem
#define XRFT_OPAQUE(x) asm volatile("" : "+r"(x))
#else
#include <hip/hip_runtime.h>
#define XRFT_LAUNCH(kernel, grid, block, smem, stream, ...) hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__)
#define XRFT_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
// make a per-lane value opaque to the optimiser (stops LICM from hoisting everything derived from it out of a
// persistent loop and then spilling it: measured 500 B/lane of scratch in fast4096_rows_kernel without this)
#define XRFT_OPAQUE(x) asm volatile("" : "+v"(x))
#endif
