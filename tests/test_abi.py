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


def test_ctx_create_refuses_a_run_length_the_kernels_cannot_address():
    """the limit-filter replay stores sixteen channel rows through one descriptor with 32-bit byte offsets; a max_samples beyond
    M17HIP_MAX_SAMPLES_PER_RUN is an argument error (checked before any HIP call: no GPU needed here), not silently dropped stores."""
    import ctypes as C
    hdr = open(os.path.join(ROOT, "include", "m17hip.h")).read()
    lim = int(re.search(r"#define M17HIP_MAX_SAMPLES_PER_RUN (\d+)u", hdr).group(1))
    assert 16 * 4 * (lim + 104) <= 0x7FFF0000 < 16 * 4 * (lim + 104 + 256)
    lib = m17hip.load_library()
    h = C.c_void_p()
    assert lib.m17hip_ctx_create(C.c_int(0), C.c_uint32(16), C.c_uint32(lim + 1), C.byref(h)) == -1 and not h.value
    assert lib.m17hip_ctx_create(C.c_int(0), C.c_uint32(0), C.c_uint32(100), C.byref(h)) == -1


def test_fake_rccl_exports_what_the_product_binds():
    """The test double of librccl (tests/fake_rccl, used by tests/test_gpu_gather_ranks.py only) defines every symbol m17_gather.hpp binds."""
    import subprocess
    so = os.path.join(ROOT, "tests", "fake_rccl", "librccl.so.1")
    if not os.path.exists(so):
        subprocess.run(["make", "-s", "-C", os.path.dirname(so)], check=True)
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    gather = open(os.path.join(ROOT, "m17-cxx-demod_amd", "csrc", "m17_gather.hpp")).read()
    bound = re.findall(r"M17_RCCL_SYM\(\w+, (nccl\w+)\)", gather)
    assert len(bound) >= 9
    for name in bound:
        assert re.search(rf"\bT {name}\b", out), name
    # ... and nothing of the product names it
    for dirpath, _, files in os.walk(os.path.join(ROOT, "m17-cxx-demod_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", "Makefile")):
                assert "fake_rccl" not in open(os.path.join(dirpath, f), errors="ignore").read(), f


def test_ctx_create_refuses_the_slow_state_unless_the_host_accepts_it():
    """VERDICT r5 #7a: with the HIP runtime's default of four hardware queues a context's five streams share queues and a step takes 1.4-1.6 x as
    long.  That state is not entered silently: m17hip_ctx_create returns M17HIP_ECONFIG (before any HIP call: no GPU needed here) unless
    GPU_MAX_HW_QUEUES >= 8 or the host says M17HIP_FEW_HW_QUEUES_OK=1.  (A child process: the variable is this process's too.)"""
    import subprocess, sys
    code = r"""
import ctypes as C, os, sys
lib = C.CDLL(sys.argv[1])
lib.m17hip_strerror.restype = C.c_char_p
h = C.c_void_p()
os.environ.pop("GPU_MAX_HW_QUEUES", None); os.environ.pop("M17HIP_FEW_HW_QUEUES_OK", None)
r1 = lib.m17hip_ctx_create(C.c_int(0), C.c_uint32(4), C.c_uint32(1000), C.byref(h))
os.environ["GPU_MAX_HW_QUEUES"] = "4"
r2 = lib.m17hip_ctx_create(C.c_int(0), C.c_uint32(4), C.c_uint32(1000), C.byref(h))
print(r1, r2, b"GPU_MAX_HW_QUEUES" in lib.m17hip_strerror(C.c_int(-8)))
"""
    import m17hip as m
    r = subprocess.run([sys.executable, "-c", code, m.LIB_PATH], capture_output=True, text=True, timeout=120)
    assert r.stdout.split() == ["-8", "-8", "True"], r.stdout + r.stderr


def test_get_stream_refuses_null_arguments():
    """m17hip_get_stream (round 6: the library owns a context's main stream): argument errors need no GPU."""
    import ctypes as C
    lib = m17hip.load_library()
    h = C.c_void_p()
    assert lib.m17hip_get_stream(C.c_void_p(), C.byref(h)) == -1
    assert lib.m17hip_set_stream(C.c_void_p(), C.c_void_p()) == -1
