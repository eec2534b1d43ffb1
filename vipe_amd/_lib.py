"""ctypes binding of libvipe_amd.so (the C ABI declared in include/vipe_amd.h).

The prototypes are parsed from the header itself, so the Python side cannot drift from the ABI.
There is NO fallback: if the library is missing, `lib()` raises with the build command.
"""

import ctypes
import os
import re
import threading

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "vipe_amd.h")
LIB_PATH = os.path.join(HERE, "lib", "libvipe_amd.so")

F16, F32, F64 = 0, 1, 2
DTYPE_CODE = {torch.float16: F16, torch.float32: F32, torch.float64: F64}
CAMERA_CODE = {"pinhole": 0, "mei": 1}


def parse_struct(name, path=HEADER):
    """ctypes.Structure mirroring `typedef struct { ... } name;` of the header (int / float scalars, pointers of any type
    as addresses, fixed-size pointer arrays) - field order and types come from the header itself."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    m = re.search(r"typedef\s+struct\s*\{([^{}]*)\}\s*" + name + r"\s*;", src)
    if m is None:
        raise KeyError(name)
    fields = []
    for decl in m.group(1).split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        first, *rest = decl.split(",")
        toks = first.replace("*", " * ").split()
        base = [t for t in toks[:-1] if t != "*"]
        names = [(first.count("*") > 0, toks[-1])] + [(r.count("*") > 0, r.replace("*", "").strip()) for r in rest]
        scalar = {"int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "int64_t": ctypes.c_int64}
        for is_ptr, nm in names:
            arr = re.match(r"(\w+)\[(\d+)\]", nm)
            # anything that is not a scalar of the table (function-pointer typedefs) travels as an address
            ctype = ctypes.c_void_p if is_ptr else scalar.get([b for b in base if b != "const"][0], ctypes.c_void_p)
            if arr:
                nm, ctype = arr.group(1), ctype * int(arr.group(2))
            fields.append((nm, ctype))
    return type(name, (ctypes.Structure,), {"_fields_": fields})


BAParams = parse_struct("vipe_ba_params")  # include/vipe_amd.h: field order / types come from the header itself


_SCALARS = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "double": ctypes.c_double}


def _ctype_of(decl):
    decl = decl.strip()
    if "*" in decl:
        if "vipe_ba_params" in decl:
            return ctypes.POINTER(BAParams)
        return ctypes.c_void_p  # every other pointer (device or host array) is passed as an address
    base = decl.replace("const", "").split()[0]
    return _SCALARS[base]


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int64_t|int|const char\*)\s+(vipe_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.groups()
        args = args.strip()
        argtypes = [] if args in ("", "void") else [_ctype_of(a) for a in args.split(",")]
        restype = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "const char*": ctypes.c_char_p}[ret]
        protos[name] = (restype, argtypes)
    return protos


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("VIPE_AMD_LIB", LIB_PATH)  # override: diagnostic builds of the same library
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing - the HIP backend is mandatory (there is no CPU fallback). "
                "Build it with `python -m vipe_amd.build` (hipcc --offload-arch=gfx950).")
        L = ctypes.CDLL(path)
        for name, (restype, argtypes) in parse_header().items():
            fn = getattr(L, name)  # AttributeError here means header and library disagree
            fn.restype = restype
            fn.argtypes = argtypes
        _LIB = L
    return _LIB


class VipeError(RuntimeError):
    pass


_TLS = threading.local()  # .restore: set by stream_ptr() when it had to switch devices for one call, undone by check()


def check(code, what):
    restore = getattr(_TLS, "restore", None)
    if restore is not None:
        torch.cuda.set_device(restore)
        _TLS.restore = None
    if code == 0:
        return
    names = {-1: "VIPE_EINVAL (bad argument)", -2: "VIPE_ENOSPACE (workspace too small)",
             -3: "VIPE_EUNSUPPORTED (not implemented in this build)"}
    if code == -3:
        raise NotImplementedError(f"{what}: {names[code]}")
    raise VipeError(f"{what}: {names.get(code, f'hipError_t {code}')}")


def stream_ptr(t=None):
    """hipStream_t of torch's current stream on the tensor's device.  Every entry point is called as
    `check(lib().vipe_x(..., stream_ptr(t)), "x")`: if `t` lives on another device than the current one, that device is
    made current for the duration of the call (kernel launches and function attributes go to the CURRENT device) and
    `check` switches back - the device guard of the reference's bindings (correlation_sampler.cpp:44-58)."""
    dev = t.device if t is not None else None
    if dev is not None and dev.type == "cuda" and dev.index is not None:
        cur = torch.cuda.current_device()
        if cur != dev.index:
            if getattr(_TLS, "restore", None) is None:
                _TLS.restore = cur
            torch.cuda.set_device(dev)
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def upload(data, device, dtype=torch.int64):
    """Small host array (list / numpy / CPU tensor) -> device tensor WITHOUT draining the stream: the data is staged in
    pinned host memory (torch's caching pinned allocator: reuse is event-guarded) and copied asynchronously on the
    current stream.  `torch.tensor(x, device=...)` / `.to(device)` from pageable memory synchronise the stream - a
    pipeline bubble per call; the edge bookkeeping of the keyframe frontend makes a dozen of them per keyframe."""
    import numpy as np
    if torch.is_tensor(data):
        if data.is_cuda:
            return data.to(device=device, dtype=dtype)
        data = data.detach().numpy()
    arr = np.ascontiguousarray(np.asarray(data), dtype=torch.empty(0, dtype=dtype).numpy().dtype)
    if torch.device(device).type != "cuda":
        return torch.from_numpy(arr.copy()).to(device)
    stage = torch.empty(arr.shape, dtype=dtype, pin_memory=True)
    if arr.size:
        stage.numpy()[...] = arr
    return stage.to(device, non_blocking=True)


def upload_many(arrays, device, dtype=torch.int64):
    """Several small host arrays of one dtype -> device tensors through ONE staged asynchronous copy (views of a single
    device buffer).  Every `upload` is a copy command on the stream (4 us of a GPU-bound keyframe each; the frontend made
    43 per frame): index vectors that are produced together travel together."""
    import numpy as np
    npdt = torch.empty(0, dtype=dtype).numpy().dtype
    arrs = [np.ascontiguousarray(np.asarray(a.detach().numpy() if torch.is_tensor(a) else a), dtype=npdt) for a in arrays]
    if torch.device(device).type != "cuda":
        return [torch.from_numpy(a.copy()).to(device) for a in arrs]
    sizes = [a.size for a in arrs]
    stage = torch.empty(max(1, sum(sizes)), dtype=dtype, pin_memory=True)
    off = 0
    for a in arrs:
        if a.size:
            stage.numpy()[off:off + a.size] = a.reshape(-1)
        off += a.size
    dev = stage.to(device, non_blocking=True)
    out, off = [], 0
    for a in arrs:
        out.append(dev[off:off + a.size].view(a.shape))
        off += a.size
    return out


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def require(cond, msg):
    """TORCH_CHECK equivalent (the reference raises RuntimeError, droid.cpp:10-11)."""
    if not cond:
        raise RuntimeError(msg)


def check_gpu_contig(*tensors):
    for t in tensors:
        require(t.is_cuda, "tensor must be a CUDA (HIP) tensor")
        require(t.is_contiguous(), "tensor must be contiguous")
