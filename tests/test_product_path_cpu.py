"""The contract on the product path: nothing under moss-ttsd_amd/ imports or executes the oracle, and the package fails
loudly (no CPU fallback) when the HIP library is missing.  Static checks, CPU only."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "moss-ttsd_amd")


def _sources():
    for d, _, files in os.walk(PKG):
        if os.sep + "build" in d or "__pycache__" in d:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                yield os.path.join(d, f)


def test_product_path_never_touches_the_oracle():
    bad = []
    for p in _sources():
        text = open(p, encoding="utf-8", errors="replace").read()
        for m in re.finditer(r"^\s*(from\s+oracle\b|import\s+oracle\b)|oracle[/\\]|asteroid_oracle|codec_oracle|kv_seal_oracle", text, re.M):
            line = text[:m.start()].count("\n") + 1
            src = text.splitlines()[line - 1]
            if p.endswith(".py") and src.lstrip().startswith("#"):
                continue
            if src.lstrip().startswith(("//", "*", "\"\"\"")) or "oracle/" in src and ("//" in src.split("oracle/")[0] or "#" in src.split("oracle/")[0] or "`" in src):
                continue                                   # a comment / docstring pointing at the test infrastructure
            bad.append((os.path.relpath(p, ROOT), line, src.strip()))
    assert not bad, bad


def test_only_the_checkers_import_the_oracle():
    """bench.py (cpu_baseline leg) and __graft_entry__.smoke() may; nothing else at the repo root does."""
    for name in os.listdir(ROOT):
        if name.endswith(".py") and name not in ("bench.py", "__graft_entry__.py"):
            text = open(os.path.join(ROOT, name)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), name


def test_binding_refuses_to_run_without_the_library(monkeypatch, tmp_path):
    import importlib
    import sys
    sys.path.insert(0, PKG)
    from mtts import capi
    importlib.reload(capi)
    monkeypatch.setattr(capi, "LIB_PATH", str(tmp_path / "libmtts.so"))
    monkeypatch.setattr(capi, "_lib", None, raising=False)
    try:
        capi.lib()
    except Exception as ex:              # noqa: BLE001
        assert "libmtts" in str(ex) or "build" in str(ex).lower()
    else:
        raise AssertionError("capi.lib() returned without a library")
    importlib.reload(capi)
