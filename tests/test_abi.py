"""CPU tests of the C-ABI boundary: libartmips.so loads without a GPU and exports exactly the entry
points include/artmips.h declares (no compute calls here)."""
import ctypes
import os
import re

from art_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    with open(os.path.join(ROOT, "include", "artmips.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(artmips_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert getattr(lib, name) is not None


def test_version_and_error_string_without_gpu():
    assert _lib.lib.artmips_version() == 2
    assert isinstance(_lib.lib.artmips_last_error(), bytes)
    # argument errors are reported through the status code + thread-local string, no exceptions
    rc = _lib.lib.artmips_create(None, 0, 768, 0, None)
    assert rc == -1
    assert b"out is NULL" in _lib.lib.artmips_last_error()


def test_info_struct_layout_matches_header():
    # artmips_info_t: 4 x int32, int64, 16 x int64, 2 x int64, 2 x float, 2 x int32, int64, 2 x float
    assert ctypes.sizeof(_lib.ArtmipsInfo) == 16 + 8 + 16 * 8 + 16 + 8 + 8 + 8 + 8


def test_no_oracle_or_cpu_fallback_in_product():
    # the product package must not import the oracle, nor carry a torch matmul/topk search path
    pkg = os.path.join(ROOT, "art_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn
            code = re.sub(r'""".*?"""', "", src, flags=re.S)
            code = re.sub(r"#.*", "", code)
            assert "torch.topk(" not in code and "torch.matmul(" not in code, fn


def test_only_tests_smoke_and_bench_import_the_oracle():
    """oracle/ is test infrastructure: nothing under art_amd/ or tools/ may import it (the fixture
    generator and the stress runner live under tests/ for that reason)."""
    for top in ("art_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".h", ".sh")):
                    src = open(os.path.join(dirpath, fn)).read()
                    assert "import oracle" not in src and "from oracle" not in src, os.path.join(dirpath, fn)
