"""
The C-ABI from a plain-C caller (examples/c_abi_client.c, built with gcc against
include/nmrfit_amd.h -- no Python, no HIP headers on the caller's side).

not gpu: the client compiles warning-free against the header, links and loads the library.
gpu:     the client's own checks (objective and residual rows against the formulas written out
         in C, the device-resident swarm, error codes) pass on cuda:0.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "nmrfit_amd", "lib")


@pytest.fixture(scope="module")
def client(tmp_path_factory):
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    from nmrfit_amd import _cabi
    _cabi.lib()      # builds the library if the .so is missing; fails loudly if it cannot
    exe = str(tmp_path_factory.mktemp("cabi") / "c_abi_client")
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_abi_client.c"), "-L", LIBDIR, "-lnmrfit_amd",
           "-Wl,-rpath," + LIBDIR, "-lm", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_client_builds_and_loads(client):
    out = subprocess.run([client, "--abi"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ABI version 6 (header 6)" in out.stdout


@pytest.mark.gpu
def test_c_client_end_to_end(client):
    out = subprocess.run([client], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("ok"), out.stdout
