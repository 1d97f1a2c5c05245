"""ctypes binding of libclimate_hip.so, generated from include/climate_hip.h.

The product path has no CPU fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
import re

# torch bundles its own HIP runtime (torch/lib/libamdhip64.so).  It must be in the process BEFORE libclimate_hip.so
# is dlopen'ed so that our NEEDED libamdhip64.so.7 resolves to that same runtime instance; loaded the other way round
# the process ends up with two runtimes and every launch fails with hipErrorNoDevice.
import torch  # noqa: F401  (device memory / streams come from torch anyway)

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, "include", "climate_hip.h")
# (CM_LIB_TAG: a second build of the same sources made with CM_BUILD_TAG, for A/B runs of a compiler option)
LIB_PATH = os.path.join(_HERE, f"libclimate_hip{os.environ.get('CM_LIB_TAG', '')}.so")

_CTYPES = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "long long": ctypes.c_longlong,
    "size_t": ctypes.c_size_t,
    "cm_stream": ctypes.c_void_p,
    "void*": ctypes.c_void_p,
    "const void*": ctypes.c_void_p,
    "float*": ctypes.c_void_p,
    "const float*": ctypes.c_void_p,
    "const char*": ctypes.c_char_p,
    "const int*": ctypes.c_void_p,
    "int*": ctypes.c_void_p,
    "const long long*": ctypes.c_void_p,
    "unsigned": ctypes.c_uint,
    "unsigned*": ctypes.c_void_p,
    "const unsigned*": ctypes.c_void_p,
    "double": ctypes.c_double,
    "double*": ctypes.c_void_p,
    "const double*": ctypes.c_void_p,
    "const float* const*": ctypes.c_void_p,
    "float* const*": ctypes.c_void_p,
    "cm_engine*": ctypes.c_void_p,
    "const cm_engine*": ctypes.c_void_p,
    "cm_engine**": ctypes.c_void_p,
    "const cm_config*": ctypes.c_void_p,
    "char*": ctypes.c_char_p,
    "void": None,
}


def parse_header(path=HEADER):
    """Return {name: (restype_str, [(type_str, arg_name), ...])} for every `cm_*` prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(cm_\w+)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef") or "(" in ret:
            continue
        arglist = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                typ = mm.group(1).strip().replace(" *", "*").replace("* ", "*")
                arglist.append((typ, mm.group(2)))
        protos[name] = (ret.replace(" *", "*"), arglist)
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self.protos = parse_header()

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is missing: build it with `python physics-based-climate-model_amd/build.py` "
                    "(the HIP path has no fallback)")
            dll = ctypes.CDLL(LIB_PATH)
            for name, (ret, args) in self.protos.items():
                fn = getattr(dll, name)      # AttributeError if the header declares a symbol the library lacks
                fn.restype = _CTYPES[ret]
                fn.argtypes = [_CTYPES[t] for t, _ in args]
            self._dll = dll
        return self._dll

    def __getattr__(self, name):
        if name.startswith("cm_"):
            fn = getattr(self.load(), name)
            hook = self.__dict__.get("_hook")
            if hook is not None:
                return hook.wrap(name, fn)
            return fn
        raise AttributeError(name)

    def set_hook(self, hook):
        """Install (or clear, with None) a launch hook: an object with wrap(name, fn) -> callable (see profiler.py)."""
        self.__dict__["_hook"] = hook


lib = _Lib()


def check(rc, what=""):
    if rc != 0:
        kind = "argument error" if rc < 0 else "hipError"
        raise RuntimeError(f"climate_hip {what} failed: {kind} {rc}")
