"""C-ABI boundary: libymk.so loads and exports every symbol include/ymk.h declares (no compute calls)."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def header_functions(name="ymk.h"):
    txt = (ROOT / "include" / name).read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ymk_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_expected_surface():
    fns = header_functions()
    for must in ("ymk_conv2d", "ymk_dwconv2d", "ymk_esmoe_route", "ymk_esmoe_dw", "ymk_esmoe_pw", "ymk_area_attn",
                 "ymk_detect_decode", "ymk_nms_batched", "ymk_cw_refine"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    from yolo_master_amd import _lib

    if not _lib.LIB_PATH.exists():
        pytest.fail(f"{_lib.LIB_PATH} missing: run `python -m yolo_master_amd.build` (build() in __graft_entry__)")
    h = _lib.load()
    fns = header_functions()
    assert sorted(_lib.SYMBOLS) == fns, "ctypes table and include/ymk.h disagree"
    for f in fns:
        assert hasattr(h, f), f"libymk.so does not export {f}"
    # config-5 rows: second header, second table (first implementation, see include/ymk_mixture.h)
    mix = header_functions("ymk_mixture.h")
    assert sorted(_lib.SYMBOLS_MIXTURE) == mix, "ctypes table and include/ymk_mixture.h disagree"
    assert not set(mix) & set(fns)
    for f in mix:
        assert hasattr(h, f), f"libymk.so does not export {f}"
    assert sorted(_lib.SYMBOLS_NEXT) == header_functions("ymk_next.h") and all(hasattr(h, f) for f in _lib.SYMBOLS_NEXT)
    assert h.ymk_abi_version() == _lib.ABI_VERSION == 4
    assert b"gfx950" in h.ymk_build_info()
    # pure host-side queries work without a GPU
    assert h.ymk_nms_workspace_bytes(2, 80, 8400, 0, 30000) > 0
    # partial-GAP chunks: 64 pixels below 4096 pixels per image, 256 above
    assert h.ymk_esmoe_route_workspace_bytes(2, 128, 40, 40) == 2 * 25 * 128 * 4
    assert h.ymk_esmoe_route_workspace_bytes(2, 128, 80, 80) == 2 * 25 * 128 * 4


def test_missing_library_fails_loudly(tmp_path):
    from yolo_master_amd import _lib

    with pytest.raises(_lib.YmkLibraryError):
        _lib.load(tmp_path / "libymk_missing.so")


def header_prototypes(name):
    """name -> list of parameter declarations, parsed from the header text."""
    txt = (ROOT / "include" / name).read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(ymk_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        args = [a.strip() for a in m.group(2).replace("\n", " ").split(",")]
        out[m.group(1)] = [] if args == ["void"] else args
    return out


def test_ctypes_tables_match_the_prototypes():
    """Argument count and scalar/pointer kind of every ctypes binding against the C prototype it binds."""
    import ctypes as C

    from yolo_master_amd import _lib

    for header, table in (("ymk.h", _lib.SYMBOLS), ("ymk_mixture.h", _lib.SYMBOLS_MIXTURE), ("ymk_next.h", _lib.SYMBOLS_NEXT)):
        protos = header_prototypes(header)
        assert set(protos) == set(table)
        for name, (_, argtypes) in table.items():
            decl = protos[name]
            assert len(decl) == len(argtypes), f"{name}: {len(decl)} parameters in {header}, {len(argtypes)} in the ctypes table"
            for d, t in zip(decl, argtypes):
                is_ptr = "*" in d
                t_ptr = t is C.c_void_p or t is C.c_char_p or hasattr(t, "_type_") and isinstance(getattr(t, "_type_"), type)
                assert is_ptr == bool(t_ptr), f"{name}: parameter `{d}` vs ctypes {t}"
                if not is_ptr:
                    want = {"float": C.c_float, "int32_t": C.c_int32, "int64_t": C.c_int64, "size_t": C.c_size_t, "int": C.c_int32}
                    base = d.replace("const", "").split()[0]
                    assert t is want[base] or (base == "int" and t is C.c_int), f"{name}: parameter `{d}` vs ctypes {t}"
