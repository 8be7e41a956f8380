"""The C-ABI library builds for gfx950, loads, and exports every symbol that
include/rt_segmentize.h declares.  No compute calls here (no GPU in the CPU suite)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from raytracing_jl_amd import _capi

    _capi.build()
    return _capi


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "rt_segmentize.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_]+)\s*\(", hdr)))


def test_header_symbols_are_exported(capi):
    L = ctypes.CDLL(capi.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 15
    for sym in declared:
        assert hasattr(L, sym), f"{sym} declared in rt_segmentize.h but not exported"
    assert sorted(capi.SYMBOLS) == declared  # the ctypes binding covers the whole header


def test_abi_version_and_messages(capi):
    L = capi.lib()
    assert L.rt_abi_version() == 1
    assert "Try increasing `k`" in capi.status_message(capi.RT_TRACK_LOCATE_FAILED)
    msg = capi.status_message(capi.RT_TRACK_LENGTH_MISMATCH, 17)
    assert msg.startswith("Track with `uid` 17 has a length that do not match")
    assert capi.status_message(capi.RT_TRACK_OK) == ""


def test_fails_loudly_without_gpu(capi, rt, traced):
    if capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    tg = traced(8, 0.02)
    with pytest.raises(capi.RtError, match="no HIP device"):
        rt.segmentize(tg)


def test_code_object_targets_gfx950(capi):
    blob = open(capi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
