"""CPU: the C-ABI library loads and exports every symbol include/gpzoo_hip.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "gpzoo_hip.h")).read()
    return sorted(set(re.findall(r"\b(gpz_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from gpzoo_amd import _lib, build
    build.build(force=False, verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gpzoo_hip.h but not exported"
    assert sorted(_lib.exported_symbols()) == names  # the ctypes table binds exactly the header


def test_version_and_error_string():
    from gpzoo_amd import _lib
    lib = _lib.load()
    assert lib.gpz_version() == 212          # 210: unknown gpz_svgp_problem.flags bits rejected, gpz_svgp_forward_path; 211: GPZ_SVGP_PANEL_PRODUCTS; 212: gpz_vnngp_state_bytes
    assert isinstance(lib.gpz_last_error(), bytes)


def test_argument_errors_do_not_touch_the_gpu():
    """Bad arguments are rejected on the host (negative return + message) before any launch."""
    from gpzoo_amd import _lib
    lib = _lib.load()
    rc = lib.gpz_kfill(None, None, 4, None, 4, 2, None, None, None, 4, 16, 0.0, 0, None)
    assert rc < 0 and b"null" in lib.gpz_last_error()
    p = _lib.SvgpProblem()
    p.dtype = 7
    assert lib.gpz_svgp_workspace_bytes(ctypes.byref(p), 0) == 0
    assert b"dtype" in lib.gpz_last_error()


def test_collective_entry_rejects_bad_arguments_on_the_host():
    """gpz_comm_* / gpz_allreduce_sum_f64 validate their arguments before RCCL is even bound."""
    from gpzoo_amd import _lib
    lib = _lib.load()
    assert lib.gpz_allreduce_sum_f64(None, None, 1, None) < 0 and b"gpz_allreduce_sum_f64" in lib.gpz_last_error()
    assert lib.gpz_comm_init(None, 1, 0, None) < 0
    comm = ctypes.c_void_p()
    ident = (ctypes.c_char * 128)()
    assert lib.gpz_comm_init(ctypes.byref(comm), 2, 5, ident) < 0 and b"bad rank" in lib.gpz_last_error()
    assert lib.gpz_comm_destroy(None) == 0


def test_struct_layout_matches_header():
    from gpzoo_amd import _lib
    assert ctypes.sizeof(_lib.KernelDesc) == 56
    assert ctypes.sizeof(_lib.SvgpProblem) == 56 + 16 + 16 + 8 * 6 + 16 + 16 + 8 * 8 + 16 + 16
    assert ctypes.sizeof(_lib.SvgpGrads) == 80      # 212: + point_order


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No silent fallback: without the built .so every entry into the product path raises."""
    import pytest
    from gpzoo_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libgpzoo_hip.so"))
    with pytest.raises(RuntimeError, match="no CPU/torch fallback"):
        _lib.load()


def test_library_carries_the_hash_of_its_sources(monkeypatch):
    """A stale binary is recognised by CONTENT: the library embeds the sha256 of the sources it was built from, the
    build script compares it (not file times) to decide what is stale, and the loader refuses a mismatch."""
    import pytest
    from gpzoo_amd import _lib, build
    build.build(force=False, verbose=False)
    want = build.source_hash()
    assert want is not None and len(want) == 32
    assert build.embedded_hash() == want                      # read from the file, no dlopen
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.gpz_source_hash.restype = ctypes.c_char_p
    assert lib.gpz_source_hash().decode() == want             # the exported entry says the same
    assert not build._stale()
    # sources that differ from what the binary was built from: stale for the build script, refused by the loader --
    # file times play no part (nothing on disk changes here)
    monkeypatch.setattr(build, "source_hash", lambda: "0" * 32)
    assert build._stale()
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(RuntimeError, match="built from other sources"):
        _lib.load()
    # a binary-only deployment (no sources next to the library) has nothing to compare
    monkeypatch.setattr(build, "source_hash", lambda: None)
    assert _lib.load() is not None
    monkeypatch.setattr(_lib, "_lib", None)


def test_flag_constants_match_the_header():
    """The Python mirror of gpz_svgp_problem.flags carries the header's values."""
    from gpzoo_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "gpzoo_hip.h")).read()
    vals = {n: int(v) for n, v in re.findall(r"#define\s+(GPZ_SVGP_[A-Z_]+)\s+(\d+)", hdr)}
    assert vals == {"GPZ_SVGP_MATERIALIZE_KZX": _lib.SVGP_MATERIALIZE_KZX, "GPZ_SVGP_NARROW_TILES": _lib.SVGP_NARROW_TILES,
                    "GPZ_SVGP_GENERATE_KZX": _lib.SVGP_GENERATE_KZX, "GPZ_SVGP_PANEL_PRODUCTS": _lib.SVGP_PANEL_PRODUCTS,
                    "GPZ_SVGP_BACKWARD_ALGEBRA": _lib.SVGP_BACKWARD_ALGEBRA, "GPZ_SVGP_BACKWARD_CLASSIC": _lib.SVGP_BACKWARD_CLASSIC}
    assert len(set(vals.values())) == 6 and all(v & (v - 1) == 0 for v in vals.values())     # distinct single bits
    fields = dict(_lib.SvgpProblem._fields_)
    assert "flags" in fields and ctypes.sizeof(fields["flags"]) == 4


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("inverse", [1, 0])
def test_factor_claim_order_is_a_linear_extension_of_the_tile_dag(inverse, fused):
    """The one-launch factorisation (csrc/coop.hip) makes progress with ANY number of resident workgroups because its
    claim list is a linear extension of the tile DAG: whatever a task waits for is produced by a task claimed earlier.
    Checked here, on the host, for every order the library can be asked for up to 26 block columns (M = 3328)."""
    from gpzoo_amd import _lib
    lib = _lib.load()
    lib.gpz_debug_coop_order.restype = C.c_int
    lib.gpz_debug_coop_order.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint16), C.c_int]
    buf = (C.c_uint16 * 1536)()
    for nblk in list(range(1, 27)):
        for G in (1, 2, 6, 8, 32):
            n = lib.gpz_debug_coop_order(nblk, G, inverse, fused, buf, 1536)
            assert n > 0, (nblk, G)
            have = set()                          # ("C", i, j), ("X", i, j), ("INV", j), ("PRE", j)
            for k in range(n):
                code = buf[k]
                kind, i, j = code >> 12, (code >> 6) & 63, code & 63
                need = []
                if kind == 2:                      # C(j, j-1) then C(j, j)
                    assert i == j and j >= 1 and fused
                    need += [("C", j, k2) for k2 in range(j - 1)] + [("C", j - 1, k2) for k2 in range(j - 1)] + [("INV", j - 1)]
                    need += [("PRE", j)] if j >= 2 else []
                    made = [("C", j, j - 1), ("C", j, j), ("INV", j)]
                elif kind == 3:                    # the sums of C(j, j) over k < j - 1
                    assert i == j and j >= 2 and fused
                    need += [("C", j, k2) for k2 in range(j - 1)]
                    made = [("PRE", j)]
                elif kind == 0 and i == j:
                    assert not (fused and j >= 1)
                    need += [("C", j, k2) for k2 in range(j)]
                    made = [("C", j, j), ("INV", j)]
                elif kind == 0:
                    assert not (fused and i == j + 1)
                    need += [("C", i, k2) for k2 in range(j)] + [("C", j, k2) for k2 in range(j)] + [("INV", j)]
                    made = [("C", i, j)]
                else:
                    assert kind == 1 and inverse and i > j
                    need += [("C", i, k2) for k2 in range(j, i)] + [("C", j, j)] + [("X", k2, j) for k2 in range(j + 1, i)] + [("INV", i)]
                    made = [("X", i, j)]
                missing = [t for t in need if t not in have]
                assert not missing, (nblk, G, k, kind, i, j, missing[:3])
                for t in made:
                    assert t not in have, (nblk, G, t)
                    have.add(t)
            want = {("C", i, j) for j in range(nblk) for i in range(j, nblk)} | {("INV", j) for j in range(nblk)}
            if fused:
                want |= {("PRE", j) for j in range(2, nblk)}
            if inverse:
                want |= {("X", i, j) for j in range(nblk) for i in range(j + 1, nblk)}
            assert have == want, (nblk, G)
