"""ctypes binding of libwfdgan_hip.so.  The signatures are parsed from include/wfdgan.h (single source of truth).

There is no fallback: if the shared library is missing or a symbol is absent, importing the product fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Callable, Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
# WFD_LIB_PATH: another build of the same library (same-session A/B of kernel variants on one GPU box); no fallback either way
LIB_PATH = os.environ.get("WFD_LIB_PATH") or os.path.join(_HERE, "libwfdgan_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "wfdgan.h")

_SCALARS = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
    "unsigned long long": ctypes.c_ulonglong,
}


class WfdError(RuntimeError):
    pass


def parse_header(path: str = HEADER_PATH) -> Dict[str, List[Tuple[str, str]]]:
    """{function name: [(ctype string, arg name), ...]} for every ``int wfd_*(...)`` declaration."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    out: Dict[str, List[Tuple[str, str]]] = {}
    for m in re.finditer(r"\bint\s+(wfd_\w+)\s*\(([^)]*)\)\s*;", src):
        name, args = m.group(1), m.group(2).strip()
        lst: List[Tuple[str, str]] = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                typ, an = mm.group(1).strip(), mm.group(2)
                lst.append((typ, an))
        out[name] = lst
    return out


def _ctype(typ: str):
    if "*" in typ:
        return ctypes.c_void_p
    t = typ.replace("const", "").strip()
    if t not in _SCALARS:
        raise WfdError(f"wfdgan.h: unsupported C type '{typ}'")
    return _SCALARS[t]


class _Lib:
    def __init__(self) -> None:
        if not os.path.exists(LIB_PATH):
            raise WfdError(
                f"{LIB_PATH} not found: build it with `make -C wfd_gan_amd/csrc` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the wfd-gan hot path.")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.decls = parse_header()
        for name, args in self.decls.items():
            try:
                fn = getattr(self.cdll, name)
            except AttributeError as e:
                raise WfdError(f"libwfdgan_hip.so does not export {name} declared in include/wfdgan.h") from e
            fn.restype = ctypes.c_int
            fn.argtypes = [_ctype(t) for t, _ in args]
            setattr(self, "_raw_" + name, fn)
            if name in ("wfd_version", "wfd_gemm_pick_splits", "wfd_gemm_pick_splits_mode", "wfd_conv8_direct", "wfd_conv8_direct_mode", "wfd_get_gemm_mode",
                        "wfd_static_scratch_floats", "wfd_gemm_wk_pick", "wfd_mesh_flag_bytes"):      # return a value, not a status
                setattr(self, name[4:], fn)
            else:
                setattr(self, name[4:], self._checked(name, fn))

    @staticmethod
    def _checked(name: str, fn) -> Callable:
        def call(*a):
            rc = fn(*a)
            if rc != 0:
                raise WfdError(f"{name} failed with code {rc} "
                               f"({ {1: 'bad argument', 2: 'kernel launch failure', 3: 'workspace'}.get(rc, '?')})")
        call.__name__ = name
        return call


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
