import ctypes
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_count() -> int:
    lib = ROOT / "lynx_amd" / "_lib" / "liblynxhip.so"
    if not lib.exists():
        return 0
    try:
        from lynx_amd import _ffi

        n = ctypes.c_int(0)
        _ffi.load().lynx_device_count(ctypes.byref(n))
        return n.value
    except Exception:
        return 0


@pytest.fixture(scope="session")
def built_library():
    """liblynxhip.so, built on demand (hipcc cross-compiles gfx950 without a GPU)."""
    lib = ROOT / "lynx_amd" / "_lib" / "liblynxhip.so"
    if not lib.exists():
        subprocess.run(["bash", str(ROOT / "lynx_amd" / "csrc" / "build.sh")], check=True)
    return lib


@pytest.fixture(scope="session")
def host_harness():
    """tests/harness/libhostcheck.so: the kernels' map-builder source compiled for the host."""
    so = ROOT / "tests" / "harness" / "libhostcheck.so"
    src = ROOT / "tests" / "harness" / "host_check.cpp"
    hdr = ROOT / "lynx_amd" / "csrc" / "lynx_maps.hpp"
    if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["bash", str(ROOT / "tests" / "harness" / "build.sh")], check=True)
    return ctypes.CDLL(str(so))


@pytest.fixture(autouse=True)
def _knobs_follow_the_environment(request, monkeypatch):
    """
    liblynxhip reads its launch-plan switches (LYNX_XPOSE, LYNX_TRACK_UNITS, ...) once, when the context is created
    (include/lynx_hip.h: lynx_ctx_reload_knobs).  GPU tests flip them with `monkeypatch.setenv` on the live context:
    every such change -- and its undoing at the end of the test -- is followed by a reload here.
    """
    if "gpu" not in request.keywords:
        yield
        return
    from lynx_amd import device

    def reload():
        rt = device._runtime
        if rt is not None and not rt.closed:
            rt.reload_knobs()

    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def setenv_and_reload(name, value, *args, **kwargs):
        setenv(name, value, *args, **kwargs)
        reload()

    def delenv_and_reload(name, *args, **kwargs):
        delenv(name, *args, **kwargs)
        reload()

    monkeypatch.setenv, monkeypatch.delenv = setenv_and_reload, delenv_and_reload  # this test's instance only
    reload()
    yield
    monkeypatch.undo()
    reload()


def pytest_collection_modifyitems(config, items):
    """`-m gpu` on a box without a GPU must fail loudly, not pass by skipping."""
    if any("gpu" in item.keywords for item in items) and config.getoption("-m") == "gpu":
        if _gpu_count() == 0:
            raise pytest.UsageError("-m gpu requested but liblynxhip sees no GPU (or is not built)")


# hypothesis: the same examples on every run (no flaky discoveries in CI), no per-example deadline
try:
    from hypothesis import settings as _hyp_settings

    _hyp_settings.register_profile("lynx", derandomize=True, deadline=None)
    _hyp_settings.load_profile("lynx")
except ImportError:  # pragma: no cover
    pass
