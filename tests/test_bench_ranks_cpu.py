"""bench.py's rank logic (sharding for weak / strong scaling, barriers, max-over-ranks timing, the c4 workload's all_gather)
run as two gloo processes on the CPU with the emulated library and tiny shapes (tests/bench_ranks_harness.py hands bench.run()
a CPU environment; bench.py has no such switch) -- so that the first real multi-GPU run is not the first run of this code.
Nothing here is a measurement."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, extra, launcher=True, environ=None, expect_status=0):
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import build_emu

    build_emu.build()  # once, before the ranks race for it
    cmd = [sys.executable]
    if launcher:  # the driver's form: torch.distributed.run around the script
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += [os.path.join(HERE, "bench_ranks_harness.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
            "--ny", "256", "--nx", "256", "--cpu-slabs", "0"] + extra
    env = dict(os.environ, OMP_NUM_THREADS="1", XRFT_EMU_THREADS="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(environ or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    if expect_status:
        assert r.returncode == expect_status, (r.returncode, r.stderr[-3000:])
        return r
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_power_spectrum(scaling):
    out = _run(2, ["--nt", "3", "--scaling", scaling])
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["unit"] == "GFFT/s" and out["value"] > 0
    assert out["config"]["ranks_in_process_group"] == 2 and "EMULATED" in out["data"]
    cfg = out["config"]
    if scaling == "weak":
        assert cfg["nt_per_gpu"] == 3 and cfg["nt_total"] == 6
    else:
        assert cfg["nt_total"] == 3 and cfg["nt_per_gpu"] == 2  # rank 0 owns slabs [0, 2)
    assert out["roofline"]["achieved"] >= 0 and "fasts_slab" in out["roofline"]["kernels_ms_per_step"]  # (256 x 256 slabs: the one-pass kernel)


def test_bench_two_ranks_c4_all_gather():
    out = _run(2, ["--nt", "2", "--workload", "c4"])
    assert out["n_gpus"] == 2 and "cross_spectrum" in out["metric"]
    col = out["config"]["collective"]
    assert col["op"] == "all_gather" and col["bytes_per_rank"] == 2 * 64 * 16  # (nt/world, nbins = 256/4) complex128 per rank


def test_bench_two_ranks_c5_float64():
    """BASELINE.json configs[4] through the harness (float64, the mixed-radix kernels), two ranks, strong scaling."""
    out = _run(2, ["--nt", "3", "--workload", "c5", "--ny", "360", "--nx", "360", "--scaling", "strong"])
    assert out["n_gpus"] == 2 and out["dtype"] == "f64" and "fp64" in out["metric"] and out["config"]["nt_total"] == 3
    assert "fastm_cols" in out["roofline"]["kernels_ms_per_step"] and out["roofline"]["bytes_per_point"] == 16.0


def test_bench_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver starts the 1-GPU bench): the script starts its two
    ranks itself, the process group reports two, ONE JSON line comes out."""
    out = _run(2, ["--nt", "3"], launcher=False)
    assert out["n_gpus"] == 2 and out["config"]["ranks_in_process_group"] == 2 and out["config"]["nt_total"] == 6
    assert out["config"]["process_group_backend"] == "gloo" and out["value"] > 0


def test_bench_plain_c4_command_gathers_identical_isotropic_blocks():
    """`python bench.py --gpus 2 --workload c4` as the driver would type it (no launcher): two ranks in the process group, the isotropic blocks all-gathered
    over the group are the same bytes on both ranks and hold every rank's slabs."""
    out = _run(2, ["--nt", "2", "--workload", "c4"], launcher=False)
    assert out["n_gpus"] == 2 and out["config"]["ranks_in_process_group"] == 2 and out["scaling"] == "weak"
    col = out["config"]["collective"]
    assert col["op"] == "all_gather" and col["identical_on_all_ranks"] is True
    assert col["gathered_shape"] == [4, 64] and out["config"]["shard_sizes"] == [2, 2]  # (nt_total, nbins = 256 / 4)


def test_bench_plain_c5_command_strong_scaling_shards():
    """`python bench.py --gpus 2 --workload c5 --scaling strong` (no launcher): contiguous blocks of the total, never a split slab."""
    out = _run(2, ["--nt", "5", "--workload", "c5", "--ny", "360", "--nx", "360", "--scaling", "strong"], launcher=False)
    assert out["n_gpus"] == 2 and out["config"]["ranks_in_process_group"] == 2 and out["scaling"] == "strong" and out["dtype"] == "f64"
    assert out["config"]["shard_sizes"] == [3, 2] and out["config"]["nt_total"] == 5 and out["config"]["nt_per_gpu"] == 3


def test_bench_plain_command_refuses_more_ranks_than_devices():
    """Fewer visible devices than --gpus: a clear message and a non-zero status before any rank is started."""
    r = _run(4, ["--nt", "2"], launcher=False, environ={"XRFT_EMU_DEVICES": "2"}, expect_status=3)
    assert "only 2 device(s) visible" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_has_no_emulator_switch():
    """bench.py measures on the GPU or not at all: no flag, environment variable or import reaches the emulated test build."""
    src = open(os.path.join(REPO, "bench.py")).read()
    assert "--emulate" not in src and "_load_for_testing" not in src and "build_emu" not in src
