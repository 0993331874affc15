"""The C-ABI library loads and exports every symbol include/goldibear_gpu.h declares (no GPU needed)."""
import os
import re

from plonky2_goldibear_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "goldibear_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(gb_[a-z0-9_]+)\s*\(", src))


def test_header_symbols_are_exported_and_bound():
    lib = native.load()
    names = _declared()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert names == set(native.SIGNATURES), names ^ set(native.SIGNATURES)


def test_null_arguments_fail_cleanly_without_gpu():
    lib = native.load()
    assert lib.gb_commit_values(None, 0, None, 1, 4, 3, 4, None, 0, None) == native.GB_ERR_INVALID
    assert lib.gb_batch_free(None) == native.GB_OK
    assert lib.gb_ctx_destroy(None) == native.GB_OK
    assert b"null" in lib.gb_last_error(None)
