"""Build the EMULATED copy of the C ABI (g++ -DXRFT_EMULATE) used by the CPU-side tests.  Test infrastructure only:
the product never loads this library (see xrft_amd/_lib.py::_load_for_testing)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
ASAN = os.environ.get("XRFT_EMU_ASAN", "0") not in ("", "0")  # scripts/run_emu_asan.sh: the same library under AddressSanitizer, in a directory of its own
OUT = os.path.join(HERE, "_build_asan" if ASAN else "_build", "libxrft_emu.so")
SRC = os.path.join(REPO, "xrft_amd", "csrc")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(SRC, f) for f in os.listdir(SRC)] + [os.path.join(HERE, "hip_emu.h"),
                                                               os.path.join(REPO, "include", "xrft_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


# the same translation units as the gfx950 build (__graft_entry__.py): xrft_hip.cpp with the fasty / fastm kernels `extern template`, and the
# instantiation groups, one g++ process each
HOST_UNITS = ["xrft_hip.cpp", "host_fastm.cpp", "host_fastg.cpp", "host_fasty.cpp", "host_rows.cpp", "host_inner.cpp", "ops.cpp"]  # plan builder + C ABI, and one host unit per kernel family
UNITS = [(u, ["-DXRFT_SPLIT_TUS"]) for u in HOST_UNITS] + [(f"inst_g{g}.cpp", []) for g in (6, 7, 4, 5, 3, 1, 2)]


def build(force=False):
    if not force and not needs_build():
        return OUT
    obj_dir = os.path.join(os.path.dirname(OUT), "obj")
    os.makedirs(obj_dir, exist_ok=True)
    import fcntl

    with open(os.path.join(os.path.dirname(OUT), ".lock"), "w") as lock:  # xdist workers and spawned ranks: one builds, the others wait and find it built
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or needs_build():
            _compile(obj_dir)
    return OUT


def _compile(obj_dir):
    procs = []
    for src, extra in UNITS:
        obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        san = ["-fsanitize=address", "-fno-omit-frame-pointer", "-g1", "-DXRFT_EMU_ASAN"] if ASAN else []
        cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-DXRFT_EMULATE", f"-I{HERE}", f"-I{SRC}"] + san + extra + ["-c", os.path.join(SRC, src), "-o", obj]
        procs.append((src, obj, subprocess.Popen(cmd)))
    failed = [src for src, _obj, pr in procs if pr.wait() != 0]
    if failed:
        raise RuntimeError(f"g++ failed on {failed}")
    tmp = OUT + f".{os.getpid()}.tmp"  # (the library appears whole or not at all)
    # (-z defs cannot hold under the sanitizer: its runtime is resolved from the preloaded libasan)
    subprocess.run(["g++", "-shared", "-fPIC"] + (["-fsanitize=address"] if ASAN else []) + [obj for _src, obj, _pr in procs] + ([] if ASAN else ["-Wl,-z,defs"]) + ["-o", tmp, "-lpthread"], check=True)
    os.replace(tmp, OUT)


if __name__ == "__main__":
    print(build(force=True))
