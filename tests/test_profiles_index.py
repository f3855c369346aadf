"""Repo hygiene: every file under profiles/ is named in profiles/README.md, DESIGN.md, DESIGN_LOG.md, KERNELS.md or
README.md -- a summary nobody cites is a summary nobody can interpret -- and tools/ stays small with one README line per script."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_profile_file_is_cited():
    docs = ""
    for f in ("profiles/README.md", "DESIGN.md", "DESIGN_LOG.md", "KERNELS.md", "README.md"):
        p = os.path.join(ROOT, f)
        if os.path.exists(p):
            docs += open(p, encoding="utf-8").read()
    missing = []
    for f in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if f == "README.md":
            continue
        stem = f.rsplit(".", 1)[0]
        if f not in docs and stem not in docs:
            missing.append(f)
    assert not missing, missing


def test_tools_is_small_and_indexed():
    files = []
    for root, _, names in os.walk(os.path.join(ROOT, "tools")):
        if "__pycache__" in root:
            continue
        files += [os.path.relpath(os.path.join(root, n), os.path.join(ROOT, "tools")) for n in names]
    assert len(files) <= 40, len(files)
    readme = open(os.path.join(ROOT, "tools", "README.md"), encoding="utf-8").read()
    unlisted = [f for f in files if f != "README.md" and os.path.basename(f) not in readme
                and os.path.basename(f).replace("make_golden", "") not in readme]
    assert not unlisted, unlisted
