"""Build the EMULATED copy of the C ABI (g++ -DXRFT_EMULATE) used by the CPU-side tests.  Test infrastructure only:
the product never loads this library (see xrft_amd/_lib.py::_load_for_testing)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build", "libxrft_emu.so")
SRC = os.path.join(REPO, "xrft_amd", "csrc")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(SRC, f) for f in os.listdir(SRC)] + [os.path.join(HERE, "hip_emu.h"),
                                                               os.path.join(REPO, "include", "xrft_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-DXRFT_EMULATE", f"-I{HERE}", f"-I{SRC}",
           os.path.join(SRC, "xrft_hip.cpp"), "-o", OUT, "-lpthread"]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
