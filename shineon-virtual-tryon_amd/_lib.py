"""ctypes binding of libshineon_hip.so (the C ABI declared in include/shineon_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a kernel launch fails, the
call raises.  `prototypes()` is also what the CPU test-suite uses to check that every symbol the
header declares is exported.
"""
import ctypes
import os
import re
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# SHINEON_LIB: another build of the library (measurement variants of tools/ablate_igemm.sh); the product path is the default
LIB_PATH = os.environ.get("SHINEON_LIB") or os.path.join(_HERE, "libshineon_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "shineon_hip.h")

# Measured igemm plans (tile, waves, split-K per layer shape) committed with the package: loaded at library load so
# that the parity tests and bench.py launch the SAME kernel instantiations, run after run (split-K factor => summation
# order => bit pattern).  SHINEON_PLANS=<file> overrides the path, SHINEON_PLANS=none skips loading (shapes missing
# from the file are still measured once per process unless SHINEON_AUTOTUNE=0).
PLANS_PATH = os.path.join(_HERE, "plans", "gfx950.txt")
PLANS_LOADED = None  # (path, number of plans) once lib() has loaded a file

_lock = threading.Lock()
_lib = None

_CTYPES = {
    "const float*": ctypes.c_void_p,
    "const double*": ctypes.c_void_p,
    "const void*": ctypes.c_void_p,
    "float*": ctypes.c_void_p,
    "double*": ctypes.c_void_p,
    "int*": ctypes.c_void_p,
    "void*": ctypes.c_void_p,
    "const char*": ctypes.c_char_p,
    "int": ctypes.c_int,
    "long long": ctypes.c_longlong,
    "float": ctypes.c_float,
    "void": None,
}


def prototypes(header_path=HEADER_PATH):
    """Parse the C header -> {name: (restype, [argtypes])}."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|long long|void)\s+(so_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes = []
        for a in [s.strip() for s in args.split(",") if s.strip()]:
            a = re.sub(r"\s+", " ", a)
            if a == "void":
                continue
            mm = re.match(r"(const float\*|const double\*|const void\*|const char\*|float\*|double\*|int\*|void\*|long long|int|float)\s*\w*$", a)
            if not mm:
                raise RuntimeError(f"cannot parse argument '{a}' of {name}")
            argtypes.append(_CTYPES[mm.group(1)])
        protos[name] = (_CTYPES[ret], argtypes)
    return protos


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library with argtypes set.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C shineon-virtual-tryon_amd/csrc`). There is no CPU fallback."
            )
        cdll = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in prototypes().items():
            fn = getattr(cdll, name)  # AttributeError if the header declares something not exported
            fn.restype = restype
            fn.argtypes = argtypes
        # measured (tile, split-K) plans per layer shape; SHINEON_AUTOTUNE=0 falls back to the cost model
        # SHINEON_AUTOTUNE=2: thorough measurement (chip warmed up, 6 timed launches per candidate) for tools/make_plans.py
        cdll.so_igemm_autotune(int(os.environ.get("SHINEON_AUTOTUNE", "1") or 1))
        global PLANS_LOADED
        plans = os.environ.get("SHINEON_PLANS", PLANS_PATH)
        if plans and plans.lower() != "none" and os.path.exists(plans):
            n = cdll.so_igemm_plans_load(plans.encode())
            if n < 0:
                raise RuntimeError(f"cannot read the igemm plans file {plans}")
            PLANS_LOADED = (plans, n)
        save = os.environ.get("SHINEON_PLANS_SAVE")
        if save:  # tools: write every plan known at exit (loaded + measured in this process), to refresh the committed file
            import atexit

            atexit.register(lambda: cdll.so_igemm_plans_save(save.encode()))
        _lib = cdll
        return _lib


class HipKernelError(RuntimeError):
    pass


_ERRS = {-1: "SO_ERR_ALIGN (pointer/stride/channel alignment)", -2: "SO_ERR_SHAPE (unsupported shape)",
         -3: "SO_NOT_APPLICABLE (specialised entry point declined; nothing was launched)"}


def check(err, what=""):
    if err != 0:
        raise HipKernelError(f"{what}: libshineon_hip returned {err} {_ERRS.get(err, '(hipError_t)')}")
