"""
CPU tier: the C-ABI library loads and exports every symbol include/nmrfit_amd.h declares,
the ctypes table covers all of them, and -- with no GPU -- the product path fails loudly
instead of falling back to anything.  No compute calls here.
"""
import ctypes
import os
import re

import numpy as np
import pytest

from nmrfit_amd import _cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="nmrfit_amd.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nmrfit_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_and_exports_every_declared_symbol():
    if not os.path.exists(_cabi.LIB_PATH):
        _cabi.build()
    L = ctypes.CDLL(_cabi.LIB_PATH)
    names = _declared_symbols() + _declared_symbols("nmrfit_amd_diag.h")
    assert len(names) >= 60
    for n in names:
        assert hasattr(L, n), "libnmrfit_amd.so does not export " + n


def test_ctypes_table_covers_header():
    """The product interface (include/nmrfit_amd.h: what a binding for the hot path binds) and the diagnostic one
    (include/nmrfit_amd_diag.h) against the two ctypes tables; the product header stays a thin interface."""
    names = set(_declared_symbols())
    bound = set(_cabi.SIGNATURES) | {"nmrfit_last_error"}
    assert names == bound, (names - bound, bound - names)
    assert len(names) <= 46, len(names)
    diag = set(_declared_symbols("nmrfit_amd_diag.h"))
    assert diag == set(_cabi.DIAG_SIGNATURES), (diag - set(_cabi.DIAG_SIGNATURES), set(_cabi.DIAG_SIGNATURES) - diag)
    assert not (names & diag)


def test_ab_variants_exist_in_the_ab_library_only():
    """The A/B kernel forms are not part of the product: the product library refuses them by name (no GPU needed:
    the check precedes any device work), the A/B library says what it is."""
    L = _cabi.lib()
    if os.environ.get("NMRFIT_LIB"):
        pytest.skip("another library is loaded through NMRFIT_LIB")
    assert L.nmrfit_diag_ab_build() == 0
    assert not _cabi.has_ab_variants() and _cabi.available_variants() == [0, 6, 7, 8]
    if os.path.exists(_cabi.AB_LIB_PATH):
        A = ctypes.CDLL(_cabi.AB_LIB_PATH)
        assert A.nmrfit_diag_ab_build() == 1


def test_abi_version():
    assert _cabi.lib().nmrfit_abi_version() == _cabi.ABI_VERSION


def test_no_silent_fallback_without_gpu():
    """On a box without a GPU every evaluation must raise; on a GPU box this test checks the
    invalid-device path instead."""
    from nmrfit_amd import equations
    w = np.linspace(0, 1, 64)
    if _cabi.device_count() == 0:
        with pytest.raises(equations.NmrfitError) as ei:
            equations.Evaluator(w, w, w, w)
        assert ei.value.code == _cabi.E_NO_DEVICE
        with pytest.raises(equations.NmrfitError):
            equations.objective(np.zeros(7), w, w, w, w)
    else:
        with pytest.raises(equations.NmrfitError) as ei:
            equations.Evaluator(w, w, w, w, device=1 << 20)
        assert ei.value.code == _cabi.E_NO_DEVICE


def test_argument_validation_needs_no_gpu():
    L = _cabi.lib()
    out = ctypes.c_void_p()
    rc = L.nmrfit_ctx_create(0, 0, None, None, None, None, ctypes.byref(out))
    assert rc == _cabi.E_INVALID
    assert b"N must be" in L.nmrfit_last_error()
    assert L.nmrfit_ctx_destroy(None) == _cabi.OK
    assert L.nmrfit_pso_destroy(None) == _cabi.OK


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "nmrfit_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".sh")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in text.lower(), \
                    "%s mentions the oracle: product code must not depend on it" % fn


def test_product_package_never_imports_torch():
    """Nothing under nmrfit_amd/ imports torch (the gloo rehearsal of the CPU tests, TorchExchange, lives in
    tests/swarm_support.py since round 5)."""
    pkg = os.path.join(ROOT, "nmrfit_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                text = open(os.path.join(dirpath, fn)).read()
                for line in text.splitlines():
                    if re.match(r"^(import|from)\s+torch\b", line):
                        raise AssertionError("%s imports torch at module level" % fn)
                assert not re.search(r"^\s+(import|from)\s+torch\b", text, flags=re.M), fn
    import subprocess
    import sys
    code = "import sys; sys.path.insert(0, %r); import nmrfit_amd, nmrfit_amd.rendezvous; assert 'torch' not in sys.modules" % ROOT
    subprocess.run([sys.executable, "-c", code], check=True)


def test_comm_argument_validation_needs_no_gpu():
    L = _cabi.lib()
    out = ctypes.c_void_p()
    assert L.nmrfit_comm_create(None, 0, 1, None, ctypes.byref(out)) == _cabi.E_INVALID
    assert L.nmrfit_comm_destroy(None) == _cabi.OK
    assert L.nmrfit_comm_barrier(None) == _cabi.E_INVALID
    assert L.nmrfit_pso_step(None) == _cabi.E_INVALID
    assert L.nmrfit_prof_enable(None, 4) == _cabi.E_INVALID


def test_batch_argument_validation_needs_no_gpu():
    """nmrfit_batch_*: argument errors are reported before anything touches a device; without a GPU the creation itself
    fails loudly (NMRFIT_E_NO_DEVICE) -- there is no CPU path behind nmrfit_amd.fit_many either."""
    L = _cabi.lib()
    out = ctypes.c_void_p()
    assert L.nmrfit_batch_create(0, 0, 64, None, None, None, None, None, None, None, 8, None, 0, 0, ctypes.byref(out)) == _cabi.E_INVALID
    assert b"K, N, swarmsize" in L.nmrfit_last_error()
    assert L.nmrfit_batch_destroy(None) == _cabi.OK
    assert L.nmrfit_batch_run(None, 1, 1) == _cabi.E_INVALID
    assert L.nmrfit_batch_status(None, None, None, None) == _cabi.E_INVALID
    assert L.nmrfit_batch_step(None) == _cabi.E_INVALID
    if _cabi.device_count() == 0:
        from nmrfit_amd import synth
        from nmrfit_amd.batch import FitBatch
        sp = synth.make_spectrum(1024, 2, seed=1)
        with pytest.raises(_cabi.NmrfitError) as ei:
            FitBatch([(sp["w"], sp["u"], sp["v"], sp["weights"])] * 2, [sp["lower"]] * 2, [sp["upper"]] * 2, swarmsize=8, seeds=[1, 2])
        assert ei.value.code == _cabi.E_NO_DEVICE


def test_variant_names():
    import pytest
    assert _cabi.variant_id("farfield") == _cabi.VARIANT_FARFIELD == 6
    assert _cabi.variant_id("Default") == 0 and _cabi.variant_id(4) == _cabi.VARIANT_QUAD
    with pytest.raises(ValueError):
        _cabi.variant_id("fastest")
    with pytest.raises(ValueError):
        _cabi.variant_id(9)


def test_build_stamp_names_the_sources_and_the_library(tmp_path, monkeypatch):
    """build() leaves the digests of the sources and of the libraries next to them; smoke() checks the
    library it loads against them on the GPU box (and rebuilds: the build is deterministic).  Here: the digest follows
    the sources, and the stamp of the library on disk matches it."""
    d0 = _cabi.source_digest()
    assert len(d0) == 64 and d0 == _cabi.source_digest()
    stamp = _cabi.read_stamp()
    if stamp is not None and stamp["source_sha256"] == d0:      # (a library built by __graft_entry__.build() from these sources)
        rec = stamp["libraries"]["libnmrfit_amd.so"]
        assert rec["sha256"] == _cabi.file_sha256(_cabi.LIB_PATH) and rec["bytes"] == os.path.getsize(_cabi.LIB_PATH)
    # the build script pins what would make two builds of the same sources differ
    script = open(_cabi.BUILD_SCRIPT).read()
    assert "-cuid=" in script and "-ffile-prefix-map=" in script
