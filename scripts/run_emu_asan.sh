#!/bin/bash
# The kernels under AddressSanitizer, on the CPU (GPU ASan is not available on this pool): the emulated library (tests/emu: the real kernel sources, g++ -DXRFT_EMULATE,
# one fiber per GPU thread) built with -fsanitize=address, every workgroup's LDS an exact-size allocation, and the emulated test suite run against it.
#   scripts/run_emu_asan.sh [pytest arguments, default: tests/test_emulated_api.py -x -q]
cd "$(dirname "$0")/.." || exit 1
export XRFT_EMU_ASAN=1
python -c "import sys; sys.path.insert(0, 'tests/emu'); import build_emu; print(build_emu.build())" || exit 1
export LD_PRELOAD=$(gcc -print-file-name=libasan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:detect_stack_use_after_return=0:allocator_may_return_null=1
if [ $# -eq 0 ]; then set -- tests/test_emulated_api.py -x -q; fi
python -m pytest "$@"
