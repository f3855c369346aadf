"""Build libosp_hip.so (gfx950 only) in-tree with hipcc.  `python -m optispeech_amd.build [--force]`.

The shared object has no torch dependency: it is the C-ABI boundary declared in include/osp.h.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libosp_hip.so")
OBJDIR = os.path.join(HERE, "csrc", "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-unused-result"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _stale(obj, src, hdrs):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(p) > t for p in [src] + hdrs)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    # a library newer than every source needs nothing (the object directory does not travel to the GPU box; the .so does)
    if not force and os.path.exists(LIB):
        t = os.path.getmtime(LIB)
        if all(os.path.getmtime(os.path.join(CSRC, f)) <= t for f in _sources()) and all(os.path.getmtime(h) <= t for h in hdrs):
            from . import fastcall
            fastcall.build(verbose=verbose)
            return LIB
    jobs = []
    for f in _sources():
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJDIR, f.rsplit(".", 1)[0] + ".o")
        if force or _stale(obj, src, hdrs):
            cmd = [HIPCC] + FLAGS + (["-x", "hip"] if f.endswith(".cpp") else []) + ["-c", src, "-o", obj]
            jobs.append((f, cmd))
    def run(job):
        f, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        return f, r.returncode, r.stdout + r.stderr
    failed = False
    with cf.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for f, rc, out in ex.map(run, jobs):
            if verbose:
                print(f"[osp build] {f}: {'ok' if rc == 0 else 'FAILED'}")
            if rc != 0:
                failed = True
                sys.stderr.write(out)
    if failed:
        raise RuntimeError("hipcc failed")
    objs = [os.path.join(OBJDIR, f.rsplit(".", 1)[0] + ".o") for f in _sources()]
    if force or jobs or not os.path.exists(LIB):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        if verbose:
            print(f"[osp build] linked {LIB}")
    from . import fastcall
    fastcall.build(verbose=verbose)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
