import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver's GPU tier)")
    config.addinivalue_line("markers", "slow: longer CPU test")


@pytest.fixture(scope="session")
def emu_lib_path():
    """g++ build of the UNMODIFIED kernel sources against the SIMT emulator (tests/emu), and the compiled PyTorch binding
    (host code only; a no-op when __graft_entry__.build() has already made it) that the operators go through by default."""
    if os.environ.get("MI355GS_EMU_LIB"):
        # a sanitizer run (tools/asan_tests_emu.sh): instrumented builds of the emulated kernels and of the compiled binding
        from instantsplat_amd import _lib
        if os.environ.get("MI355GS_TORCH_EXT"):
            _lib.EXT_PATH = os.environ["MI355GS_TORCH_EXT"]
        return os.environ["MI355GS_EMU_LIB"]
    subprocess.check_call(["bash", os.path.join(ROOT, "tests", "emu", "build_emu.sh")])
    if torch.version.hip is None:
        # a CPU-only PyTorch wheel has no ATen/hip headers or libc10_hip to build the compiled binding against: the emulated
        # tier then runs through the ctypes binding (same C-ABI calls), and the tests that compare the two bindings skip
        from instantsplat_amd import _lib
        os.environ["MI355GS_BINDING"] = _lib.BINDING = "ctypes"
    else:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "instantsplat_amd", "csrc_torch", "build.py")])
    return os.path.join(ROOT, "tests", "emu", "libmi355gs_emu.so")


@pytest.fixture()
def emu(emu_lib_path):
    """Route instantsplat_amd's operators to the emulated library for one CPU test."""
    from instantsplat_amd import _lib
    _lib._use_library_for_testing(emu_lib_path)
    yield torch.device("cpu")
    _lib._use_library_for_testing(None)


@pytest.fixture()
def gpu():
    """The product path: hipcc-built libmi355gs.so on cuda:0. Fails loudly if either is missing."""
    from instantsplat_amd import _lib
    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    _lib._use_library_for_testing(None)
    _lib.lib()
    return torch.device("cuda:0")


def pytest_terminal_summary(terminalreporter):
    """GS_CALIBRATE=1: print every measured bound of the session (tests/ops_util.py::bound) instead of enforcing it."""
    if os.environ.get("GS_CALIBRATE") != "1":
        return
    from tests import ops_util
    worst = {}
    for label, value, limit in ops_util.MEASURED:
        v0, _ = worst.get(label, (0.0, limit))
        worst[label] = (max(v0, value), limit)
    terminalreporter.write_line("---- measured bounds (max over the session) ----")
    for label, (value, limit) in sorted(worst.items()):
        terminalreporter.write_line("CAL %-70s measured %.3e  limit %.3e  %s" % (label, value, limit, "OVER" if value > limit else ""))
