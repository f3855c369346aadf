"""Run a crash-prone GPU test in a child interpreter, so that a fault which kills the process (a HIP runtime abort at the end of
a hipGraph capture, a GPU memory fault, RCCL tearing down) is a NAMED FAILURE of that test instead of the death of the whole
suite (round 2's driver run died with SIGABRT and zero recorded passes).

    @isolated
    def test_x(...): ...

In the parent the decorated test re-invokes pytest on its own node id with OSP_ISOLATED_CHILD=1 and asserts on the exit status
(the child's output tail -- including the faulthandler dump and the crash note of tests/conftest.py -- goes into the failure
message); in the child the decorator is transparent.
"""
import functools
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def isolated(fn=None, *, timeout=900):
    def deco(f):
        @functools.wraps(f)
        def wrapper(*args, **kwargs):
            if os.environ.get("OSP_ISOLATED_CHILD") == "1":
                return f(*args, **kwargs)
            nodeid = os.environ["PYTEST_CURRENT_TEST"].rsplit(" (", 1)[0]
            env = dict(os.environ, OSP_ISOLATED_CHILD="1", OSP_TEST_NAMES="0")
            env.pop("PYTEST_CURRENT_TEST", None)
            try:
                r = subprocess.run([sys.executable, "-m", "pytest", nodeid, "-x", "-q", "-p", "no:cacheprovider", "-m", "gpu"],
                                   cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
            except subprocess.TimeoutExpired as e:
                out = (e.stdout or b"")
                out = out.decode(errors="replace") if isinstance(out, bytes) else out
                raise AssertionError(f"isolated child of {nodeid} did not finish in {timeout} s (killed)\n{out[-3000:]}")
            if r.returncode != 0:
                how = f"died on signal {-r.returncode}" if r.returncode < 0 else f"exit status {r.returncode}"
                raise AssertionError(f"isolated child of {nodeid}: {how}\n--- child stdout (tail)\n{r.stdout[-6000:]}\n"
                                     f"--- child stderr (tail)\n{r.stderr[-3000:]}")
            assert " passed" in r.stdout, r.stdout[-2000:]
        return wrapper
    return deco(fn) if fn is not None else deco
