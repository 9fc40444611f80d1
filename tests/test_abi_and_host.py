"""CPU-only checks: the C-ABI library loads and exports every symbol include/dme_hip.h declares,
the ctypes table mirrors the header, argument validation that needs no GPU, host-side logic."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dme_hip.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|int64_t|const char\s*\*)\s+(dme_\w+)\s*\(([^;]*?)\)\s*;", text, re.S):
        args = m.group(3).strip()
        n = 0 if args in ("void", "") else len([a for a in args.split(",") if a.strip()])
        out[m.group(2)] = n
    return out


def test_header_declares_the_expected_entry_points():
    fns = _declared_functions()
    for name in ("dme_stats_dense_u16", "dme_stats_dense_i32", "dme_stats_csr_i32",
                 "dme_suffix_scan", "dme_cli_init_alphas", "dme_eval", "dme_predicted_step",
                 "dme_predicted_step_alt", "dme_solve", "dme_last_error", "dme_abi_version"):
        assert name in fns


def test_library_exports_every_declared_symbol():
    from bayespy_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with `make -C bayespy_amd/csrc`"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(handle, name), f"{name} declared in dme_hip.h but not exported"


def test_ctypes_table_matches_header_arity():
    from bayespy_amd import _lib
    fns = _declared_functions()
    assert set(fns) == set(_lib.SIGNATURES)
    for name, nargs in fns.items():
        assert len(_lib.SIGNATURES[name][1]) == nargs, name


def test_abi_version_and_sizes_without_gpu():
    from bayespy_amd import _lib
    L = _lib.lib()
    assert L.dme_abi_version() == _lib.ABI_VERSION == 2
    assert L.dme_solver_workspace_bytes(1000, 500, 500) > 0
    assert L.dme_solver_workspace_bytes(0, 1, 1) == -1
    assert L.dme_stats_workspace_bytes(10_000_000, 1000) > 256 * 1000 * 8
    assert L.dme_synth_scratch_bytes(500) > 0
    # CSR: no partition workspace for small inputs or beyond the 32-bit entry offsets; otherwise at
    # least one 8-byte record per stored entry
    assert L.dme_stats_csr_workspace_bytes(1000, 1000, 50_000) == 0
    assert L.dme_stats_csr_workspace_bytes(10**6, 1 << 32, 50_000) == 0
    need = L.dme_stats_csr_workspace_bytes(10**6, 5 * 10**7, 50_000)
    assert need >= 8 * 5 * 10**7 and need % 256 == 0


def test_argument_validation_needs_no_gpu():
    """Bad shapes are rejected before any HIP call."""
    from bayespy_amd import _lib
    L = _lib.lib()
    rc = L.dme_stats_dense_u16(0, 10, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0)
    assert rc == _lib.ERR_INVALID_ARG
    assert b"bad shape" in L.dme_last_error()
    rc = L.dme_stats_csr_i32(0, 0, 0, 5, 10, 0, 0, 2, 0, 2, 0, 0, 0, 0, 0, 0)
    assert rc == _lib.ERR_INVALID_ARG and b"bad shape" in L.dme_last_error()
    rc = L.dme_stats_csr_i32(0, 0, 0, 5, 10, 7, 0, 2, 0, 2, 0, 0, 0, 0, 0, 0)
    assert rc == _lib.ERR_INVALID_ARG and b"null output" in L.dme_last_error()
    rc = L.dme_suffix_scan(0, 0, 1, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0)
    assert rc == _lib.ERR_INVALID_ARG
    with pytest.raises(ValueError):
        _lib.check(rc)
    with pytest.raises(ZeroDivisionError):
        _lib.check(_lib.ERR_ZERO_DIVISION)
    with pytest.raises(ValueError, match="math domain"):
        _lib.check(_lib.ERR_MATH_DOMAIN)
    with pytest.raises(NotImplementedError):
        _lib.check(_lib.ERR_UNSUPPORTED)


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bayespy_amd import device
    from bayespy_amd import dirichletMultinomialEstimation as DME
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        device.DeviceStats(2)
    obj = DME.CompressedRowData(2)
    obj.appendRow([1, 2], 1)  # buffering is host-side
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _ = obj.U


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under bayespy_amd/ may mention it."""
    pkg = os.path.join(ROOT, "bayespy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, re.M), f
                assert "dme_oracle" not in text, f


def test_shard_rows_partitions_exactly():
    from bayespy_amd.parallel import shard_rows
    for N in (0, 1, 7, 10_000_000, 10_000_001):
        for world in (1, 2, 3, 8):
            spans = [shard_rows(N, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == N
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_row_buffering_semantics_on_host():
    from bayespy_amd import dirichletMultinomialEstimation as DME
    obj = DME.CompressedRowData(3)
    with pytest.raises(IndexError):
        obj.appendRow([1, 2], 1)            # row[k] out of range, DE/dme.py:117
    obj.appendRow([1, 2, 3], 0.5)           # fractional weights are kept apart (fp64 histograms)
    assert obj._wpending == [[1, 2, 3]] and obj._wweights == [0.5]
    obj.appendRow([1, 0, 2, 4], 1)          # over-long row keeps its excess for the total
    assert obj._pending == [[1, 0, 2]] and obj._excess == [4]


def test_draw_category_matches_reference_contract():
    import random
    from bayespy_amd.samplingTools import drawCategory
    random.seed(3)
    counts = [0] * 3
    for _ in range(3000):
        counts[drawCategory([1, 0, 3])] += 1
    assert counts[1] == 0 and 600 < counts[0] < 900


def _parse(text, K, delim="\t", threads=4):
    from bayespy_amd import _lib
    L = _lib.lib()
    raw = text.encode()
    n = L.dme_parse_count_rows(raw, len(raw))
    out = np.full((max(n, 1), K), -7, dtype=np.int32)
    got = ctypes.c_int64(0)
    rc = L.dme_parse_counts(raw, len(raw), ord(delim), K, out.ctypes.data, n,
                            ctypes.addressof(got), threads)
    return rc, out[:got.value]


def test_fast_parser_matches_the_reference_row_loop():
    """dme_parse_counts vs csv.reader + int() (DE/fdp.py:67-79) on the reference's sample shapes."""
    import csv
    import io
    from conftest import load_golden
    fx = load_golden("test_csv")
    for delim in ("\t", ","):
        text = "".join(delim.join(str(v) for v in r) + "\n" for r in fx["rows"])
        rc, X = _parse(text, 2, delim)
        assert rc == 0
        ref = [list(map(int, row)) for row in csv.reader(io.StringIO(text), delimiter=delim)]
        assert X.tolist() == ref
    # no trailing newline, CRLF, padding blanks, explicit signs
    rc, X = _parse("1\t2\r\n +3 \t-4\n5\t6", 2)
    assert rc == 0 and X.tolist() == [[1, 2], [3, -4], [5, 6]]
    # a large text exercises the multi-threaded split (> 1 MiB)
    rng = np.random.default_rng(0)
    big = rng.integers(0, 100000, size=(60000, 7))
    text = "\n".join("\t".join(map(str, r)) for r in big) + "\n"
    assert len(text) > (1 << 20)
    rc, X = _parse(text, 7, threads=8)
    assert rc == 0 and np.array_equal(X, big)


def test_fast_parser_gives_up_on_anything_else():
    from bayespy_amd import _lib
    for bad in ("1\t2\n\n3\t4\n",        # blank line
                "1\t2\t3\n",             # too many fields
                "1\n",                   # too few fields
                "1\t2.0\n",              # not an integer
                "1\t\"2\"\n",            # quoted field
                "1\t1_0\n",              # python int() accepts this; the fast grammar does not
                "1\t99999999999\n"):     # beyond int32
        rc, X = _parse(bad, 2)
        assert rc == _lib.PARSE_FALLBACK, bad
