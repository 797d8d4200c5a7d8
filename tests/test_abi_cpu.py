"""The ctypes binding (`wild_deep_mvs_amd/_lib.py`) against the prototypes of `include/pscv.h`: same exports, same arity, same C
type per argument and the same return type.  A swapped / missing `int` in either file fails here, on the CPU, instead of
corrupting a launch on the GPU (round-3 review: the hand-mirrored argtypes lists were checked by name only)."""
import ctypes as C
import os
import re
import types

import pytest

from wild_deep_mvs_amd import _lib as L

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _prototypes():
    src = open(os.path.join(REPO, "include", "pscv.h")).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = "\n".join(ln for ln in src.splitlines() if not ln.lstrip().startswith("#"))
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(pscv_\w+)\s*\(([^()]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        args = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        protos[name] = (ret, args)
    return protos


def _ctype_of(decl: str, is_return=False):
    """C declaration ('const float* cams', 'int B') -> the category the binding must use."""
    d = re.sub(r"\s+", " ", decl.replace("*", " * ")).strip()
    toks = d.split(" ")
    if not is_return and toks[-1] != "*" and len(toks) > 1:
        toks = toks[:-1]                                   # drop the parameter name
    stars = toks.count("*")
    base = " ".join(t for t in toks if t not in ("*", "const", "signed"))      # ("unsigned char*" = a byte buffer, not a string)
    if stars == 0:
        return {"int": "int", "long": "long", "float": "float", "size_t": "size_t", "double": "double"}[base]
    if stars >= 2:
        return "ptr_to_ptr"
    return {"char": "char_ptr", "int": "int_ptr"}.get(base, "ptr")


def _category(ct):
    if ct is C.c_int:
        return "int"
    if ct is C.c_long:
        return "long"
    if ct is C.c_float:
        return "float"
    if ct is C.c_double:
        return "double"
    if ct is C.c_size_t:
        return "size_t"
    if ct is C.c_char_p:
        return "char_ptr"
    if ct is C.c_void_p:
        return "ptr"
    if isinstance(ct, type) and issubclass(ct, C._Pointer):
        return "ptr_to_ptr" if ct._type_ is C.c_void_p else "int_ptr" if ct._type_ is C.c_int else "ptr"
    raise AssertionError(f"unexpected ctypes type {ct!r}")


class _Recorder:
    def __init__(self):
        object.__setattr__(self, "fns", {})

    def __getattr__(self, name):
        return self.fns.setdefault(name, types.SimpleNamespace(restype="unset", argtypes="unset"))


def test_every_prototype_of_the_header_is_bound_with_the_same_signature():
    protos = _prototypes()
    rec = _Recorder()
    L._declare(rec)
    assert set(protos) == set(L.EXPORTS), f"header vs EXPORTS: {sorted(set(protos) ^ set(L.EXPORTS))}"
    assert set(rec.fns) == set(protos), f"header vs _declare: {sorted(set(rec.fns) ^ set(protos))}"
    assert len(protos) >= 57
    problems = []
    for name, (ret, args) in sorted(protos.items()):
        fn = rec.fns[name]
        if fn.argtypes == "unset" or fn.restype == "unset":
            problems.append(f"{name}: restype / argtypes not declared")
            continue
        want_ret = _ctype_of(ret, is_return=True)
        if _category(fn.restype) != want_ret:
            problems.append(f"{name}: returns {want_ret} in pscv.h, {_category(fn.restype)} in _lib.py")
        if len(args) != len(fn.argtypes):
            problems.append(f"{name}: {len(args)} parameters in pscv.h, {len(fn.argtypes)} in _lib.py")
            continue
        for k, (decl, ct) in enumerate(zip(args, fn.argtypes)):
            want, got = _ctype_of(decl), _category(ct)
            ok = want == got or (want == "int_ptr" and got == "ptr")      # (int arrays may be passed as raw device / host pointers)
            if not ok:
                problems.append(f"{name}: parameter {k} `{decl}` is {want} in pscv.h, {got} in _lib.py")
    assert not problems, "\n".join(problems)


def test_the_parser_sees_what_it_should():
    protos = _prototypes()
    ret, args = protos["pscv_warp_cost"]
    assert ret == "int" and len(args) == 21
    assert [_ctype_of(a) for a in args[:6]] == ["ptr", "ptr_to_ptr", "int", "ptr", "ptr", "long"]
    assert _ctype_of(args[9]) == "float" and _ctype_of(args[-1]) == "ptr"
    assert _ctype_of(protos["pscv_last_error"][0], is_return=True) == "char_ptr"
    assert _ctype_of("const char* key") == "char_ptr" and _ctype_of("int* value") == "int_ptr"


def test_abi_version_constant_matches_the_header():
    src = open(os.path.join(REPO, "include", "pscv.h")).read()
    assert int(re.search(r"#define PSCV_ABI_VERSION (\d+)", src).group(1)) == L.ABI_VERSION
