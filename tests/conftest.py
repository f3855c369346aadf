import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


_CRASHNOTE = None
_TRACE = None


def _crashnote_lib():
    """tests/native/crashnote.c -> _crashnote.so (gcc, built on demand; test infrastructure only)."""
    import ctypes
    import subprocess
    src = os.path.join(ROOT, "tests", "native", "crashnote.c")
    so = os.path.join(ROOT, "tests", "native", "_crashnote.so")
    try:
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            tmp = so + f".{os.getpid()}.tmp"
            subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-o", tmp, src], check=True, capture_output=True)
            os.replace(tmp, so)
        lib = ctypes.CDLL(so)
        lib.osp_crashnote_set.argtypes = [ctypes.c_char_p]
        lib.osp_crashnote_install.argtypes = [ctypes.c_int]
        return lib
    except Exception:
        return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A process-killing fault (SIGABRT from the HIP runtime, SIGSEGV in a kernel launch path) must NAME its test:
    #  * every test start is written straight to the real stderr (fd 2 is not captured between test phases) and to
    #    OSP_TEST_TRACE (default gpurun_out/test_trace.txt when that directory exists);
    #  * faulthandler dumps ALL threads; underneath it sits tests/native/crashnote.c, which faulthandler chains to, so
    #    the node id is the last line of the log even when only the tail survives.
    global _CRASHNOTE, _TRACE
    import faulthandler
    try:
        fd = os.dup(2)
        _CRASHNOTE = _crashnote_lib()
        faulthandler.disable()
        if _CRASHNOTE is not None and _CRASHNOTE.osp_crashnote_install(fd) != 0:
            _CRASHNOTE = None
        config._osp_fault_file = os.fdopen(os.dup(fd), "w")
        faulthandler.enable(file=config._osp_fault_file, all_threads=True)
    except Exception:
        pass
    trace = os.environ.get("OSP_TEST_TRACE")
    if trace is None and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        trace = os.path.join(ROOT, "gpurun_out", "test_trace.txt")
    if trace:
        try:
            _TRACE = open(trace, "a", buffering=1)
        except OSError:
            _TRACE = None


def pytest_runtest_logstart(nodeid, location):
    if _CRASHNOTE is not None:
        _CRASHNOTE.osp_crashnote_set(nodeid.encode())
    if os.environ.get("OSP_TEST_NAMES", "1") != "0":
        try:
            os.write(2, f"\n[osp-test] {nodeid}\n".encode())
        except OSError:
            pass
    if _TRACE is not None:
        _TRACE.write(f"{time.time():.2f} {os.getpid()} START {nodeid}\n")


def pytest_runtest_logfinish(nodeid, location):
    if _CRASHNOTE is not None:
        _CRASHNOTE.osp_crashnote_set(f"(between tests, after {nodeid})".encode())
    if _TRACE is not None:
        _TRACE.write(f"{time.time():.2f} {os.getpid()} END   {nodeid}\n")


def pytest_collection_modifyitems(config, items):
    """GPU tests must fail loudly on a GPU box without the HIP library, and are skipped only when
    there is no GPU at all and they were not explicitly selected with -m gpu."""
    import torch
    if torch.cuda.is_available():
        return
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load
