"""CPU-only: the C-ABI library loads and exports every symbol include/osp.h declares (no compute calls)."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "osp.h")).read()
    return sorted(set(re.findall(r"\b(osp_\w+)\s*\(", txt)))


def test_header_is_valid_c():
    subprocess.check_call(["gcc", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "osp.h")])


def test_library_exports_every_declared_symbol():
    from optispeech_amd.build import build
    lib = ctypes.CDLL(build(verbose=False))
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/osp.h but not exported"
    lib.osp_abi_version.restype = ctypes.c_int
    assert lib.osp_abi_version() >= 1


def test_header_matches_sources():
    """include/osp.h is generated; regenerate and compare so it cannot drift from the kernels."""
    before = open(os.path.join(ROOT, "include", "osp.h")).read()
    subprocess.check_call(["python", os.path.join(ROOT, "tools", "gen_header.py")], stdout=subprocess.DEVNULL)
    assert open(os.path.join(ROOT, "include", "osp.h")).read() == before


def test_product_never_imports_the_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "optispeech_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dp, f)


def test_native_marshalling_module_covers_the_header():
    """_ospfast (generated from include/osp.h) has one typed call per stream-taking entry point and rejects malformed calls
    before anything is launched (no GPU needed: every call below fails in argument conversion)."""
    import pytest
    from optispeech_amd import _lib, fastcall
    from optispeech_amd.build import build
    build(verbose=False)
    fast = _lib._load_fast()
    assert fast is not None, "optispeech_amd/lib/_ospfast*.so not built"
    ents = fastcall.entries()
    assert fast.N_ENTRIES == len(ents) >= 40
    for i, (name, _) in enumerate(ents):
        assert fast.index(name) == i
    assert fast.index("osp_no_such_entry") is None
    i = fast.index("osp_sumsq")
    with pytest.raises(TypeError, match="arguments given"):
        fast.call(i, 0, 1, 2)
    nargs = len(dict(ents)["osp_sumsq"])
    with pytest.raises(TypeError, match="must match include/osp.h"):
        fast.call(i, 0, *(["not a tensor"] * nargs))
    import torch
    with pytest.raises(RuntimeError, match="not on the GPU"):
        fast.call(i, 0, *([torch.zeros(4)] * nargs))


def _build_comm_example(out):
    """hipcc: the C++ host program of tests/native/comm_example.cpp against include/osp.h + libosp_hip.so (nothing from torch)."""
    from optispeech_amd.build import build
    lib = build(verbose=False)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", f"-I{os.path.join(ROOT, 'include')}",
                           os.path.join(ROOT, "tests", "native", "comm_example.cpp"), "-o", out, f"-L{os.path.dirname(lib)}", "-losp_hip",
                           f"-Wl,-rpath,{os.path.dirname(lib)}"])
    return out


def test_native_comm_example_compiles_and_links(tmp_path):
    """The non-torch gradient-exchange example INTEGRATION.md shows is real code: it compiles against the generated header and
    links against the library's osp_comm_* / osp_allreduce_bucket symbols (it is RUN by tests/test_gpu_dp.py on the GPU box)."""
    exe = _build_comm_example(os.path.join(tmp_path, "comm_example"))
    assert os.path.getsize(exe) > 0
    syms = subprocess.check_output(["nm", "-D", "--undefined-only", exe], text=True)
    for n in ("osp_comm_unique_id", "osp_comm_init", "osp_allreduce_bucket", "osp_comm_destroy"):
        assert n in syms, n
