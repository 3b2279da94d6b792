"""ctypes binding of libuncr_hip.so (the C ABI declared in include/uncr_hip.h).

Prototypes are parsed from the header, so the loader, the ABI test and INTEGRATION.md share one source
of truth.  There is NO fallback: if the library is missing or a symbol does not resolve, importing the
product path raises -- the GPU path is never silently replaced by PyTorch ops."""
import ctypes
import os
import re
from typing import Dict, List, Tuple

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HEADER = os.path.join(ROOT, "include", "uncr_hip.h")
LIB_PATH = os.environ.get("UNCR_HIP_LIB") or os.path.join(HERE, "lib", "libuncr_hip.so")   # env: development builds only

_CTYPES = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "unsigned long long": ctypes.c_ulonglong,
    "long long": ctypes.c_longlong,
    "double": ctypes.c_double,
    "hipStream_t": ctypes.c_void_p,
}


def parse_header(path: str = HEADER) -> Dict[str, List[Tuple[str, str]]]:
    """-> {function name: [(ctype string, arg name), ...]} for every `int uncr_*(...)` declaration."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(uncr_\w+)\s*\(([^)]*)\)\s*;", text):
        name, args = m.group(1), m.group(2).strip()
        lst = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.+?)\s*(\w+)$", a)
                typ, an = mm.group(1).strip(), mm.group(2)
                lst.append((typ, an))
        protos[name] = lst
    return protos


def _ctype_of(typ: str):
    if "*" in typ:
        return ctypes.c_void_p
    return _CTYPES[typ.replace("const ", "").strip()]


DEV_HEADER = os.path.join(ROOT, "include", "uncr_dev.h")
DEV_LIB_PATH = os.path.join(HERE, "lib", "libuncr_dev.so")


class HipLib:
    def __init__(self, path: str = LIB_PATH, header: str = HEADER):
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found: build it with `python -m uncrtaints_amd.build` (hipcc --offload-arch=gfx950). "
                "uncrtaints_amd has no CPU / PyTorch fallback for its kernels.")
        self.path = path
        self.cdll = ctypes.CDLL(path)
        self.protos = parse_header(header)
        self.fn = {}
        for name, args in self.protos.items():
            f = getattr(self.cdll, name)   # AttributeError if the header and the library disagree
            f.restype = ctypes.c_int
            f.argtypes = [_ctype_of(t) for t, _ in args]
            self.fn[name] = f


_LIB = None
_DEV = None


def dev_lib() -> HipLib:
    """The development-probe library (include/uncr_dev.h); tools and accuracy tests only, never the product path."""
    global _DEV
    if _DEV is None:
        _DEV = HipLib(DEV_LIB_PATH, DEV_HEADER)
    return _DEV


def lib() -> HipLib:
    global _LIB
    if _LIB is None:
        _LIB = HipLib()
    return _LIB


def _ptr(x):
    """torch tensor -> device pointer (checked), None -> NULL, int -> as is."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    import torch
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"expected tensor/None, got {type(x)}")
    if not x.is_cuda:
        raise RuntimeError("uncrtaints_amd kernels need tensors on the GPU (cuda device); there is no CPU path")
    if not x.is_contiguous():
        raise RuntimeError("uncrtaints_amd kernels need contiguous tensors")
    if x.dtype not in (torch.float32, torch.bfloat16, torch.float64, torch.int32, torch.int64):
        raise RuntimeError(f"unsupported dtype {x.dtype}")
    return x.data_ptr()


class EventProfiler:
    """Optional per-launch timing with HIP events on torch's current stream (the stream every kernel of this
    library is enqueued on).  Used by bench.py for the live roofline numbers; off by default."""

    def __init__(self, names):
        self.names = set(names)
        self.records = []          # (entry point, int-arg tuple, start event, end event)
        self.scope = None          # while set (a tag), EVERY launch is timed and also listed in scope_records
        self.scope_records = []    # (tag, entry point, start event, end event)
        self.scope_exclude = set() # entry points left out of scope_ms()
        self.tag = None            # optional callable (entry point, args) -> record name (variants one entry point's ints cannot tell apart)

    def scope_ms(self):
        """-> {tag: summed kernel milliseconds} of the launches made while a scope tag was set."""
        import torch
        torch.cuda.synchronize()
        out = {}
        for tag, name, e0, e1 in self.scope_records:
            if name in self.scope_exclude:
                continue
            out[tag] = out.get(tag, 0.0) + e0.elapsed_time(e1)
        return out

    def scope_detail(self):
        """-> {tag: {entry point: [launches, summed ms]}} of the scoped launches (scope_exclude entries marked by a leading '-')."""
        import torch
        torch.cuda.synchronize()
        out = {}
        for tag, name, e0, e1 in self.scope_records:
            d = out.setdefault(tag, {}).setdefault(("-" if name in self.scope_exclude else "") + name, [0, 0.0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
        return out

    def summarize(self):
        import torch
        torch.cuda.synchronize()
        out = {}
        for name, key, e0, e1 in self.records:
            d = out.setdefault((name, key), [0, 0.0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
        return {k: (n, ms / n) for k, (n, ms) in out.items()}   # launches, mean ms


_PROF = None


def set_profiler(p):
    global _PROF
    _PROF = p


def call(name: str, *args):
    """Call an entry point; tensors are converted to pointers; raises RuntimeError on a non-zero code."""
    L = dev_lib() if name.startswith("uncr_debug_") else lib()
    f = L.fn[name]
    proto = L.protos[name]
    if len(args) != len(proto):
        raise TypeError(f"{name}: expected {len(proto)} args, got {len(args)}")
    conv = []
    for a, (typ, _) in zip(args, proto):
        if "*" in typ or typ == "hipStream_t":
            conv.append(_ptr(a) if typ != "hipStream_t" else a)
        else:
            conv.append(a)
    if _PROF is not None and (name in _PROF.names or _PROF.scope is not None):
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = f(*conv)
        e1.record()
        if name in _PROF.names:
            key = tuple(a for a, (typ, _) in zip(args, proto) if typ == "int")
            _PROF.records.append((_PROF.tag(name, args) if _PROF.tag else name, key, e0, e1))
        if _PROF.scope is not None:
            _PROF.scope_records.append((_PROF.scope, name, e0, e1))
    else:
        rc = f(*conv)
    if rc != 0:
        kind = "argument/shape error" if rc < 0 else "hipError_t"
        raise RuntimeError(f"{name} failed with code {rc} ({kind})")
    return rc


def query(name: str, *args) -> int:
    """Call a pure size-query entry point (returns its int result, no error convention)."""
    return lib().fn[name](*args)
