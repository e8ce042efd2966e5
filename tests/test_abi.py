"""CPU-only: the C-ABI library loads and exports every symbol include/m17hip.h declares (no compute calls)."""
import os
import re

import m17hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "m17hip.h")).read()
    declared = sorted(set(re.findall(r"\b(m17hip_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    lib = m17hip.load_library()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in m17hip.h but not exported"
    assert sorted(m17hip.EXPORTS) == declared
    assert lib.m17hip_version() >= 100
    assert lib.m17hip_strerror(-2) == b"HIP runtime error"


def test_library_exports_nothing_but_the_c_abi():
    """`nm -D --defined-only`: only m17hip_* (no kernel stubs, no helpers, no C++ symbols) — csrc/m17hip.map."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", m17hip.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = [line.split()[-1] for line in out.splitlines() if line.strip()]
    assert names, "no dynamic symbols listed"
    stray = [n for n in names if not n.startswith("m17hip_")]
    assert not stray, stray
    assert sorted(names) == sorted(m17hip.EXPORTS)


def test_record_layouts_match_header():
    import oracle_lib as ol
    assert m17hip.FRAME_REC == ol.FRAME_REC and m17hip.DIAG == ol.DIAG
    hdr = open(os.path.join(ROOT, "include", "m17hip.h")).read()
    assert "/* 64 bytes */" in hdr


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "m17-cxx-demod_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for line in text.splitlines():
                    s = line.strip()
                    if s.startswith(("#include", "import ", "from ")) or "CDLL" in s or "dlopen" in s:
                        assert "oracle" not in s, (f, s)
