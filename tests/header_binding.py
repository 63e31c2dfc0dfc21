"""A ctypes binding of libgnnome_hip.so built from NOTHING but the text of include/gnnome_hip.h - what INTEGRATION.md
promises a maintainer of the reference can do.  Used by tests/test_host.py (every prototype parses and agrees with the
package's own table) and tests/test_hip_parity.py (a kernel is called through it)."""
import ctypes
import re

_SCALARS = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t, "float": ctypes.c_float}


def _ctype(decl):
    decl = decl.strip()
    if decl == "void":
        return None
    if "*" in decl:
        base = decl.replace("const", "").split("*")[0].strip()
        if base == "size_t":
            return ctypes.POINTER(ctypes.c_size_t)
        if base == "int" and decl.rstrip().endswith(("rows_host", "_host")):
            return ctypes.POINTER(ctypes.c_int)
        return ctypes.c_void_p
    base = decl.replace("const", "").split()
    return _SCALARS[base[0]]


def parse_header(path):
    """-> {name: (restype, [argtypes], [argument names])} for every `int|const char* gnnome_*(...)` prototype."""
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    out = {}
    for ret, name, args in re.findall(r"\b(int|const char\s*\*)\s+(gnnome_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        parts = [a.strip() for a in args.split(",")] if args.strip() and args.strip() != "void" else []
        types, names = [], []
        for a in parts:
            m = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a)
            types.append(_ctype(m.group(1) + (m.group(2) if "host" in m.group(2) else "")))
            names.append(m.group(2))
        out[name] = (ctypes.c_char_p if "char" in ret else ctypes.c_int, types, names)
    return out


def bind(lib_path, header_path):
    lib = ctypes.CDLL(lib_path)
    for name, (res, argtypes, _) in parse_header(header_path).items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, argtypes
    return lib
