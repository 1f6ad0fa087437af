"""The C-ABI shared library loads without a GPU and exports every entry point include/transport_se_hip.h declares
(no compute calls here); the product path refuses to run without a HIP device instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import pytest

from transport_se_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "transport_se_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(tse_[a-z0-9_]+)\s*\(", text))
    names.discard("tse_exchange_fn")
    return names


def test_every_declared_symbol_is_exported():
    L = _lib.lib()
    declared = _declared()
    assert len(declared) >= 20
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)


def test_init_fails_loudly_without_gpu_or_with_bad_arguments():
    """no CPU fallback: tse_init reports an error (no device here / bad limiter) and leaves the handle null"""
    L = _lib.lib()
    a = _lib.InitArgs()
    a.nelemd, a.qsize, a.device, a.limiter_option = 1, 1, -1, 4
    h = C.c_void_p()
    assert L.tse_init(C.byref(h), C.byref(a)) != 0
    assert b"limiter_option=8" in L.tse_last_error()
    assert not h.value


def test_fortran_seam_declares_the_reference_hooks():
    """cuda_mod_hip.F90 provides the public names of the reference's seam (cuda_mod.F90:57-64) that the hooks call"""
    src = open(os.path.join(ROOT, "transport_se_amd", "fortran", "cuda_mod_hip.F90")).read().lower()
    assert "module cuda_mod" in src
    for name in ("cuda_mod_init", "euler_step_cuda", "qdp_time_avg_cuda", "vertical_remap_cuda", "copy_qdp_d2h", "copy_qdp_h2d"):
        assert re.search(r"subroutine\s+%s\s*\(" % name, src), name
    for c_name in re.findall(r"name='(tse_[a-z0-9_]+)'", src):
        assert c_name in _declared(), c_name
