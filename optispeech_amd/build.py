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


def source_hash():
    """sha1 over the names and contents of csrc/*.{hip,cpp,h} and the compile flags: what the library was built FROM.
    It is compiled into the library (``osp_source_hash()``, csrc/api.cpp) so that a shipped, git-ignored .so can be checked
    against the sources next to it by content -- mtimes do not survive a checkout or the copy to the GPU box."""
    import hashlib
    h = hashlib.sha1(" ".join(FLAGS).encode())
    for f in sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h"))):
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()


def library_hash(path=None):
    """The source hash embedded in a built library (read from the file: no dlopen, so a rebuild in this process is safe)."""
    path = path or LIB
    if not os.path.exists(path):
        return None
    data = open(path, "rb").read()
    i = data.find(b"OSP_SOURCE_HASH=")
    return data[i + 16:i + 56].decode() if i >= 0 else None


def _object_key(src, hdrs):
    """Content key of one translation unit: flags + its source + every header of csrc/ (any header may be included)."""
    import hashlib
    h = hashlib.sha1(" ".join(FLAGS).encode())
    for p in [src] + sorted(hdrs):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()


def _file_flags(src):
    """Extra compile flags a translation unit asks for in a line `// osp-flags: ...` (part of its source, hence of its content key)."""
    for line in open(src, encoding="utf-8", errors="replace"):
        if line.startswith("// osp-flags:"):
            return line.split(":", 1)[1].split()
    return []


def _stale(obj, key):
    """An object is reused only when the key file next to it names exactly the contents it was compiled from (mtimes play no
    role: VERDICT r03 found that a stale .o could be linked under a fresh source hash)."""
    kf = obj + ".key"
    return not (os.path.exists(obj) and os.path.exists(kf) and open(kf).read().strip() == key)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    # a library built from exactly these sources needs nothing (the object directory does not travel to the GPU box; the
    # .so does): compared by CONTENT hash, not by mtime
    want = source_hash()
    if not force and library_hash() == want:
        from . import fastcall
        fastcall.build(verbose=verbose)
        return LIB
    jobs = []
    for f in _sources():
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJDIR, f.rsplit(".", 1)[0] + ".o")
        key = _object_key(src, hdrs) + (want if f == "api.cpp" else "")   # api.cpp carries the library hash: recompiled with it
        if force or _stale(obj, key):
            cmd = [HIPCC] + FLAGS + _file_flags(src) + (["-x", "hip"] if f.endswith(".cpp") else []) + ["-c", src, "-o", obj]
            if f == "api.cpp":
                cmd.insert(-4, f'-DOSP_SOURCE_HASH="{want}"')
            jobs.append((f, cmd, obj, key))
    def run(job):
        f, cmd, obj, key = job
        if os.path.exists(obj + ".key"):
            os.remove(obj + ".key")
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0:
            with open(obj + ".key", "w") as fh:
                fh.write(key)
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
    if True:
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        if verbose:
            print(f"[osp build] linked {LIB}")
        assert library_hash() == want, "the linked library does not carry the source hash"
    from . import fastcall
    fastcall.build(verbose=verbose)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
