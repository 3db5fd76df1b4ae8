"""The C-ABI library loads and exports every symbol include/mtts.h declares (no GPU calls)."""
import os
import re

from mtts import capi
from mtts import codec  # noqa: F401  (registers the codec signatures)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "mtts.h")).read()
    declared = set(re.findall(r"\b(mtts_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = capi.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in mtts.h but not exported by libmtts.so"
    assert declared == set(capi.exported_symbols()), declared ^ set(capi.exported_symbols())
    assert lib.mtts_version() >= 100


def test_struct_layouts_match_header():
    import ctypes as C
    assert C.sizeof(capi.MttsConfig) == 19 * 4
    assert C.sizeof(capi.MttsSamplerCfg) == 6 * 4
    assert C.sizeof(codec.MttsCodecConfig) == 32 * 4
