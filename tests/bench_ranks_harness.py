"""Test harness (run under torch.distributed.run by tests/test_bench_ranks_cpu.py): bench.py's rank logic -- sharding for weak /
strong scaling, barriers, max-over-ranks timing, the c4 all_gather, the JSON line -- on CPU tensors over gloo with the EMULATED
test build of the C ABI.  bench.py itself knows only the GPU environment; this file supplies another one to bench.run().
Nothing printed from here is a measurement (the `data` field says so)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(HERE, "emu"))

import bench  # noqa: E402


class EmulatedEnv:
    backend = "gloo"
    backend_label = "gloo"
    data_label = "synthetic (EMULATED library on CPU: rank-logic test, not a measurement)"
    measures = False
    script = os.path.abspath(__file__)

    def visible_devices(self):
        return int(os.environ.get("XRFT_EMU_DEVICES", "8"))  # the harness pretends to a node's worth of devices

    def device(self, local):
        import torch

        return torch.device("cpu")

    def init_process_group(self, dist, local):
        dist.init_process_group("gloo")

    def sync(self, dev):
        pass

    def load_library(self):
        import build_emu
        from xrft_amd import _lib

        _lib._load_for_testing(build_emu.build())


if __name__ == "__main__":
    out = bench.run(bench.parse_args(sys.argv[1:]), EmulatedEnv())
    if isinstance(out, int):
        sys.exit(out)
