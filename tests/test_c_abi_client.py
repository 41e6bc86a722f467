"""The drop-in boundary from plain C/C++: tests/c_abi/ps_example.cpp links libxrft_hip.so and drives it with raw device
pointers (no Python, no torch).  CPU: the client must compile and link against every symbol it uses.  GPU: it must run
and pass its own closed-form checks (plane removed, Parseval)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(HERE, "c_abi", "ps_example.cpp")
EXE = os.path.join(HERE, "c_abi", "ps_example")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _build():
    lib = os.path.join(REPO, "xrft_amd", "libxrft_hip.so")
    if not os.path.exists(lib):
        pytest.skip("libxrft_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    if os.path.exists(EXE) and os.path.getmtime(EXE) >= max(os.path.getmtime(SRC), os.path.getmtime(lib)):
        return
    cmd = [HIPCC, "--offload-arch=gfx950", "-w", SRC, "-I" + os.path.join(REPO, "include"), "-L" + os.path.join(REPO, "xrft_amd"),
           "-lxrft_hip", "-Wl,-rpath," + os.path.join(REPO, "xrft_amd"), "-o", EXE]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]


def test_c_client_builds_and_links():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_c_client_runs_on_the_gpu():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
