"""
ctypes binding of libnmrfit_amd.so (include/nmrfit_amd.h).  This is the whole Python <-> GPU
boundary: plain pointers and sizes, no torch types.  There is no CPU fallback: if the
library is missing or no gfx950 device is visible the calls raise ``NmrfitError``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libnmrfit_amd.so")
# the A/B library: the product + the A/B kernel variants (BASELINE, NOSKIP, SINGLE, QUAD, STAGED); what the reference
# kernels of the parity tests, tools/ab.py and bench.py's `variants` entry load.  NMRFIT_LIB=<path> makes it (or any
# other build) the library of this process.
AB_LIB_PATH = os.path.join(_HERE, "lib", "libnmrfit_amd_ab.so")
BUILD_SCRIPT = os.path.join(_HERE, "csrc", "build.sh")

OK = 0
E_INVALID, E_NO_DEVICE, E_HIP, E_UNSUPPORTED, E_STATE, E_COMM = -1, -2, -3, -4, -5, -6
UNIQUE_ID_BYTES = 128
FIT_IM_OFF, FIT_IM_REFERENCE, FIT_IM_SUM = 0, 1, 2
VARIANT_DEFAULT, VARIANT_BASELINE, VARIANT_NOSKIP, VARIANT_SINGLE, VARIANT_QUAD, VARIANT_STAGED, VARIANT_FARFIELD = 0, 1, 2, 3, 4, 5, 6
VARIANT_NOREC = 7
VARIANT_FARFIELD32 = 8     # opt-in mixed precision: the far-field kernel's shared polynomial in packed fp32
HANDOVER_FAST, HANDOVER_FENCED, HANDOVER_TWO_LAUNCH = 0, 1, 2
ABI_VERSION = 6
_VARIANT_NAMES = {"default": 0, "baseline": 1, "noskip": 2, "single": 3, "quad": 4, "staged": 5, "farfield": 6,
                  "norec": 7, "farfield32": 8}


def variant_id(v):
    """NMRFIT_VARIANT_* number from a name ("farfield") or a number; ValueError otherwise."""
    if isinstance(v, str):
        if v.lower() not in _VARIANT_NAMES:
            raise ValueError("unknown kernel variant %r (one of %s)" % (v, ", ".join(sorted(_VARIANT_NAMES))))
        return _VARIANT_NAMES[v.lower()]
    v = int(v)
    if v not in _VARIANT_NAMES.values():
        raise ValueError("unknown kernel variant %d" % v)
    return v

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_void_pp = ctypes.POINTER(ctypes.c_void_p)


class NmrfitError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libnmrfit_amd error %d: %s" % (code, msg))
        self.code = code


class PsoParams(ctypes.Structure):
    _fields_ = [("omega", ctypes.c_double), ("phip", ctypes.c_double), ("phig", ctypes.c_double),
                ("minstep", ctypes.c_double), ("minfunc", ctypes.c_double), ("seed", ctypes.c_uint64)]


# name -> (argtypes) ; every entry returns int except nmrfit_last_error.  SIGNATURES: include/nmrfit_amd.h (the product
# interface); DIAG_SIGNATURES: include/nmrfit_amd_diag.h (diagnostics, A/B knobs, the phases of a generation one by one)
_I64, _I32, _INT = ctypes.c_int64, ctypes.c_int32, ctypes.c_int
_VP = ctypes.c_void_p
ALL_SIGNATURES = {
    "nmrfit_abi_version": [],
    "nmrfit_device_count": [ctypes.POINTER(_INT)],
    "nmrfit_device_info": [_INT, ctypes.c_char_p, _INT, ctypes.POINTER(_INT), ctypes.c_char_p, _INT],
    "nmrfit_ctx_create": [_INT, _I64, _VP, _VP, _VP, _VP, _c_void_pp],
    "nmrfit_ctx_destroy": [_VP],
    "nmrfit_ctx_set_weights": [_VP, _VP],
    "nmrfit_ctx_synchronize": [_VP],
    "nmrfit_ctx_set_variant": [_VP, _INT],
    "nmrfit_ctx_set_fit_im": [_VP, _INT],
    "nmrfit_ctx_set_stream": [_VP, _VP],
    "nmrfit_ctx_n": [_VP, ctypes.POINTER(_I64)],
    "nmrfit_objective_batch": [_VP, _I64, _I32, _VP, _INT, _VP],
    "nmrfit_residual_batch": [_VP, _I64, _I32, _VP, _VP, _VP],
    "nmrfit_contributions": [_VP, _I32, _VP, _I64, _VP, _VP, _VP],
    "nmrfit_generate_result": [_VP, _I32, _VP, _I64, _VP, _VP, _VP, _VP, _VP],
    "nmrfit_objective_batch_dev": [_VP, _I64, _I32, _VP, _VP],
    "nmrfit_residual_batch_dev": [_VP, _I64, _I32, _VP, _VP, _VP],
    "nmrfit_dev_alloc": [_VP, _I64, _c_void_pp],
    "nmrfit_dev_free": [_VP, _VP],
    "nmrfit_memcpy_h2d": [_VP, _VP, _VP, _I64],
    "nmrfit_memcpy_d2h": [_VP, _VP, _VP, _I64],
    "nmrfit_timer_begin": [_VP],
    "nmrfit_timer_end": [_VP, _c_double_p],
    "nmrfit_last_launch": [_VP, ctypes.POINTER(_I64), ctypes.POINTER(_I32), ctypes.POINTER(_I64)],
    "nmrfit_last_launch_workgroup": [_VP, ctypes.POINTER(_I32)],
    "nmrfit_pso_create": [_VP, _I64, _I64, _I64, _I32, _VP, _VP, ctypes.POINTER(PsoParams), _c_void_pp],
    "nmrfit_pso_destroy": [_VP],
    "nmrfit_pso_init": [_VP],
    "nmrfit_pso_step_local": [_VP],
    "nmrfit_pso_candidate_dev": [_VP, _c_void_pp],
    "nmrfit_pso_set_candidate_dev": [_VP, _VP],
    "nmrfit_pso_apply_global_dev": [_VP, _VP, _I32],
    "nmrfit_pso_status": [_VP, ctypes.POINTER(_I64), ctypes.POINTER(_I32), _c_double_p],
    "nmrfit_pso_best": [_VP, _VP, _c_double_p],
    "nmrfit_pso_run": [_VP, _I64, _I32],
    "nmrfit_pso_set_handover": [_VP, _INT],
    "nmrfit_pso_set_fused_pbest": [_VP, _INT],
    "nmrfit_pso_set_fused_tail": [_VP, _INT],
    "nmrfit_pso_last_launches": [_VP, ctypes.POINTER(_I32)],
    "nmrfit_pso_get_state": [_VP, _VP, _VP, _VP, _VP, _VP],
    "nmrfit_device_pci_bus_id": [_INT, ctypes.c_char_p, _INT],
    "nmrfit_comm_available": [],
    "nmrfit_comm_describe": [_VP, ctypes.c_char_p, _INT],
    "nmrfit_comm_unique_id": [_VP],
    "nmrfit_comm_create": [_VP, _I32, _I32, _VP, _c_void_pp],
    "nmrfit_comm_destroy": [_VP],
    "nmrfit_comm_info": [_VP, ctypes.POINTER(_I32), ctypes.POINTER(_I32), ctypes.POINTER(_I32)],
    "nmrfit_comm_all_gather_dev": [_VP, _VP, _VP, _I64],
    "nmrfit_comm_all_reduce_host": [_VP, _c_double_p, _I32, _I32],
    "nmrfit_comm_broadcast_host": [_VP, _VP, _I64, _I32],
    "nmrfit_comm_barrier": [_VP],
    "nmrfit_pso_set_comm": [_VP, _VP],
    "nmrfit_pso_step": [_VP],
    "nmrfit_prof_enable": [_VP, _I64],
    "nmrfit_prof_mark": [_VP],
    "nmrfit_prof_read": [_VP, _VP, _I64, ctypes.POINTER(_I64), _VP, _I64, ctypes.POINTER(_I64), _c_double_p],
    "nmrfit_diag_ab_build": [],
    "nmrfit_batch_create": [_INT, _I32, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I64, _VP, _INT, _INT, _c_void_pp],
    "nmrfit_batch_create_ragged": [_INT, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _INT, _INT, _c_void_pp],
    "nmrfit_batch_destroy": [_VP],
    "nmrfit_batch_run": [_VP, _I64, _I32],
    "nmrfit_batch_status": [_VP, _VP, _VP, _VP],
    "nmrfit_batch_best": [_VP, _VP, _VP],
    "nmrfit_batch_contributions": [_VP, _VP, _VP, _VP, _VP, _VP, _VP],
    "nmrfit_batch_step": [_VP],
    "nmrfit_batch_synchronize": [_VP],
    "nmrfit_batch_set_geometry": [_VP, _INT],
    "nmrfit_batch_geometry": [_VP, ctypes.POINTER(_I32), ctypes.POINTER(_I32), ctypes.POINTER(_I32), ctypes.POINTER(_I64)],
    "nmrfit_batch_get_state": [_VP, _I32, _VP, _VP, _VP, _VP, _VP],
}

_DIAG_NAMES = (
    "nmrfit_device_pci_bus_id", "nmrfit_ctx_set_stream", "nmrfit_timer_begin", "nmrfit_timer_end", "nmrfit_last_launch",
    "nmrfit_last_launch_workgroup", "nmrfit_prof_enable", "nmrfit_prof_mark", "nmrfit_prof_read", "nmrfit_pso_step_local",
    "nmrfit_pso_candidate_dev", "nmrfit_pso_set_candidate_dev", "nmrfit_pso_apply_global_dev", "nmrfit_pso_set_handover",
    "nmrfit_pso_set_fused_pbest", "nmrfit_pso_set_fused_tail", "nmrfit_pso_last_launches", "nmrfit_pso_get_state",
    "nmrfit_comm_describe", "nmrfit_comm_all_gather_dev", "nmrfit_comm_all_reduce_host", "nmrfit_comm_barrier",
    "nmrfit_diag_ab_build", "nmrfit_batch_step", "nmrfit_batch_synchronize", "nmrfit_batch_set_geometry",
    "nmrfit_batch_geometry", "nmrfit_batch_get_state")
DIAG_SIGNATURES = {k: ALL_SIGNATURES[k] for k in _DIAG_NAMES}
SIGNATURES = {k: v for k, v in ALL_SIGNATURES.items() if k not in DIAG_SIGNATURES}

_LIB = None


def build(verbose=False, ab=False):
    """Compile libnmrfit_amd.so (``ab``: libnmrfit_amd_ab.so, the A/B library) for gfx950 with hipcc (cross-compiles
    without a GPU)."""
    out = subprocess.run(["bash", BUILD_SCRIPT] + (["--ab"] if ab else []), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
    if out.returncode != 0:
        raise RuntimeError("building %s failed" % ("libnmrfit_amd_ab.so" if ab else "libnmrfit_amd.so"))
    return AB_LIB_PATH if ab else LIB_PATH


STAMP_PATH = os.path.join(_HERE, "lib", "BUILD_STAMP.json")


def source_digest():
    """sha256 over the sources the library is built from (nmrfit_amd/csrc/*, include/*.h), names and contents, sorted."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(_HERE)
    files = []
    for d in (os.path.join(_HERE, "csrc"), os.path.join(root, "include")):
        files += [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hip", ".h", ".sh"))]
    for f in sorted(files):
        h.update(os.path.relpath(f, root).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()


def file_sha256(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for block in iter(lambda: fh.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()


def write_stamp():
    """What build() leaves next to the libraries it has just made: the digest of the sources and of each library file
    (the build is deterministic -- fixed compilation-unit ids, csrc/build.sh -- so the same sources give the same
    bytes).  __graft_entry__.smoke() checks the library it LOADED against this on the GPU box."""
    import json
    import time
    stamp = {"source_sha256": source_digest(), "built_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
             "libraries": {os.path.basename(p): {"sha256": file_sha256(p), "bytes": os.path.getsize(p)}
                           for p in (LIB_PATH, AB_LIB_PATH) if os.path.exists(p)}}
    with open(STAMP_PATH, "w") as fh:
        json.dump(stamp, fh, indent=1)
    return stamp


def read_stamp():
    import json
    if not os.path.exists(STAMP_PATH):
        return None
    with open(STAMP_PATH) as fh:
        return json.load(fh)


def lib_path():
    """The library this process loads: NMRFIT_LIB if set (e.g. the A/B library), else the product library."""
    return os.environ.get("NMRFIT_LIB") or LIB_PATH


def has_ab_variants():
    """True when the loaded library is an A/B build (the BASELINE / NOSKIP / SINGLE / QUAD / STAGED kernels exist)."""
    return bool(lib().nmrfit_diag_ab_build())


PRODUCT_VARIANTS = (VARIANT_DEFAULT, VARIANT_FARFIELD, VARIANT_NOREC, VARIANT_FARFIELD32)


def available_variants():
    """Kernel variants of the loaded library: the three product kernels, plus the A/B forms in an A/B build."""
    if has_ab_variants():
        return [VARIANT_DEFAULT, VARIANT_BASELINE, VARIANT_NOSKIP, VARIANT_SINGLE, VARIANT_QUAD, VARIANT_STAGED,
                VARIANT_FARFIELD, VARIANT_NOREC, VARIANT_FARFIELD32]
    return list(PRODUCT_VARIANTS)


def lib():
    """The loaded library.  Raises (never falls back) when it is not built."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise NmrfitError(E_NO_DEVICE, "%s not found: build it with nmrfit_amd/csrc/build.sh "
                              "(or __graft_entry__.build()); there is no CPU fallback" % path)
        # One process per GPU, several per node: the ROCm driver of the machines this runs on offers
        # dmabuf IPC only, and RCCL's intra-node transport (hipIpcGetMemHandle) fails with "invalid
        # argument" unless the HSA runtime is told so BEFORE it initialises -- i.e. before the first
        # HIP call this library makes.  A value the user exported wins.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        L = ctypes.CDLL(path)
        for name, argtypes in ALL_SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        L.nmrfit_last_error.argtypes = []
        L.nmrfit_last_error.restype = ctypes.c_char_p
        _LIB = L
    return _LIB


def check(rc):
    if rc != OK:
        raise NmrfitError(rc, lib().nmrfit_last_error().decode("utf-8", "replace"))


def f64(a):
    """Contiguous float64 copy/view: the reference hands out negative-stride views
    (nmrfit/core.py:60) and possibly float32 data; the ABI takes contiguous float64."""
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def device_count():
    n = ctypes.c_int(0)
    rc = lib().nmrfit_device_count(ctypes.byref(n))
    if rc == E_NO_DEVICE:
        return 0
    check(rc)
    return n.value


def device_pci_bus_id(device=0):
    buf = ctypes.create_string_buffer(64)
    check(lib().nmrfit_device_pci_bus_id(device, buf, 64))
    return buf.value.decode()


def device_info(device=0):
    name = ctypes.create_string_buffer(256)
    arch = ctypes.create_string_buffer(256)
    cu = ctypes.c_int(0)
    check(lib().nmrfit_device_info(device, name, 256, ctypes.byref(cu), arch, 256))
    return dict(name=name.value.decode(), compute_units=cu.value, arch=arch.value.decode())
