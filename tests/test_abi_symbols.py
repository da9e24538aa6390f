"""The C-ABI library loads and exports every symbol include/*.h declares.
No compute call is made here (no GPU in the CPU tier)."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols():
    names = set()
    for path in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(path).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names.update(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


@pytest.fixture(scope="module")
def built():
    from mosaicmagnifique_amd import _build
    return _build.build()


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("mm_create", "mm_destroy", "mm_set_library", "mm_set_masks", "mm_set_target",
                 "mm_generate", "mm_score_cells", "mm_resolve", "mm_colour_diff_batch", "mm_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built)
    for name in declared_symbols():
        assert hasattr(lib, name), "missing export %s" % name
    lib.mm_abi_version.restype = ctypes.c_int
    assert lib.mm_abi_version() == 4


def test_the_collective_is_linked_behind_the_c_abi(built):
    """north_star: a single RCCL all-gather assembles the best-fit grid's candidate lists.  The library imports
    ncclAllGather (and the communicator calls) itself; no Python sits between the C++ host and the collective."""
    import subprocess
    syms = subprocess.run(["nm", "-D", "--undefined-only", built], capture_output=True, text=True, check=True).stdout
    for name in ("ncclAllGather", "ncclCommInitAll", "ncclGroupStart", "ncclGroupEnd", "ncclCommDestroy"):
        assert name in syms, name
    needed = subprocess.run(["readelf", "-d", built], capture_output=True, text=True, check=True).stdout
    assert "librccl.so" in needed


def test_binding_covers_every_declared_symbol():
    from mosaicmagnifique_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_struct_layouts_match_the_header():
    from mosaicmagnifique_amd import _lib
    from mosaicmagnifique_amd.grid import CELL_DTYPE
    assert ctypes.sizeof(_lib.MMCell) == 36 == CELL_DTYPE.itemsize
    assert [n for n, _ in _lib.MMCell._fields_] == list(CELL_DTYPE.names)
    assert ctypes.sizeof(_lib.MMTiming) == 232
    assert ctypes.sizeof(_lib.MMStageTiming) == 72          # six clocks (f64) + three counters (i64)


def test_product_never_touches_the_oracle():
    """The package must not import, link or call anything under oracle/."""
    for path in glob.glob(os.path.join(ROOT, "mosaicmagnifique_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
            text = open(path).read()
            hit = re.search(r"import\s+oracle|from\s+oracle|oracle_ffi|liboracle|oracle[/\\.]|orc_[a-z]", text)
            assert hit is None, (path, hit.group(0))
