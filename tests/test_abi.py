"""CPU: librcfm.so loads and exports exactly what include/rcfm.h + include/rcfm_tools.h declare.

No compute calls (this container has no GPU); the parity tests proper are the
`-m gpu` ones, which go through this same ABI.
"""

import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "rcfm.h")                # what a host binds
TOOLS_HEADER = os.path.join(ROOT, "include", "rcfm_tools.h")    # arenas, explicit FFT plans, kernel-form switches, profiling
LIB = os.path.join(ROOT, "radio-core_amd", "radiocore", "_lib", "librcfm.so")


def declared_in(header):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rcfm_[a-z0-9_]+)\s*\(", text)))


def declared_functions():
    return sorted(set(declared_in(HEADER)) | set(declared_in(TOOLS_HEADER)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(LIB)


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("rcfm_tuner_create", "rcfm_tuner_load", "rcfm_tuner_run", "rcfm_demod_create",
                 "rcfm_demod_run", "rcfm_pipeline_run", "rcfm_resampler_run", "rcfm_filtfilt",
                 "rcfm_lfilter_fir", "rcfm_hilbert", "rcfm_pll_phase", "rcfm_discriminator"):
        assert must in names
    assert len(names) >= 28


def test_the_host_header_is_free_of_tooling():
    """rcfm.h = the surface a host needs; what exists for this repo's tools and tests lives in rcfm_tools.h.  The two do
    not overlap, the host header stands alone, and the tools header builds on it."""
    host, tools = set(declared_in(HEADER)), set(declared_in(TOOLS_HEADER))
    assert not host & tools
    for name in host:
        assert not name.startswith(("rcfm_arena_", "rcfm_profile_")), name
    assert {"rcfm_fft_describe", "rcfm_fft_describe_plan", "rcfm_fft_c2c_plan", "rcfm_fft_c2c_rocfft",
            "rcfm_tuner_set_option", "rcfm_demod_get_option", "rcfm_arena_create", "rcfm_profile_read"} <= tools
    assert "rcfm_demod_set_option" in host          # RCFM_OPT_STATE_FENCE / _NARROW_TILES / _GRAPH are host options
    assert '#include "rcfm.h"' in open(TOOLS_HEADER).read()
    assert "rcfm_tools.h" not in re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    # the C host example binds the host header only
    example = open(os.path.join(ROOT, "examples", "c_host.c")).read()
    assert "rcfm_tools.h" not in example
    example = re.sub(r"/\*.*?\*/", "", example, flags=re.S)
    used = set(re.findall(r"\b(rcfm_[a-z0-9_]+)\s*\(", example))
    assert used <= host, sorted(used - host)


def test_every_symbol_the_library_exports_is_declared(lib):
    """nm -D: the exported rcfm_* symbols are exactly the two headers' declarations (no undeclared back doors)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r"\bT (rcfm_[a-z0-9_]+)$", out, flags=re.M)))
    assert exported == declared_functions()


def test_library_exports_every_declared_symbol(lib):
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_matches_header():
    from radiocore._internal import hip
    bound = set(hip.SIGNATURES) | {"rcfm_last_error", "rcfm_profile_stage_name"}
    assert bound == set(declared_functions())


def test_version_and_error_string(lib):
    lib.rcfm_version.restype = ctypes.c_int
    assert lib.rcfm_version() == 102
    lib.rcfm_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.rcfm_last_error(), bytes)


def test_argument_errors_need_no_device(lib):
    """NULL / bad-size arguments are rejected before any HIP call."""
    lib.rcfm_tuner_create.restype = ctypes.c_int
    assert lib.rcfm_tuner_create(ctypes.c_int64(0), 0, None, None, None) == -4
    out = ctypes.c_void_p()
    assert lib.rcfm_demod_create(7, 1, 100, 10, ctypes.c_double(75e-6), 0, ctypes.byref(out)) == -4
    lib.rcfm_last_error.restype = ctypes.c_char_p
    assert b"kind" in lib.rcfm_last_error()
    # ingest and gather entry points (round 2)
    assert lib.rcfm_feeder_create(ctypes.c_size_t(0), 2, None, ctypes.byref(out)) == -4
    assert lib.rcfm_feeder_create(ctypes.c_size_t(1024), 0, None, ctypes.byref(out)) == -4
    assert lib.rcfm_feeder_submit(None, None) == -4
    assert lib.rcfm_host_register(None, ctypes.c_size_t(16)) == -4
    assert lib.rcfm_comm_init_rank(2, 2, None, ctypes.byref(out)) == -4
    assert lib.rcfm_gather_audio(None, 0, None, ctypes.c_size_t(0), None, None) == -4
    assert lib.rcfm_tuner_shard(None, 0, 0) == -4


def test_package_fails_loudly_without_a_device():
    """The product has no CPU fallback: constructing a class without a HIP device raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    import radiocore
    assert radiocore.HasCuda() is False
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        radiocore.MFM(240000, 48000)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        radiocore.Tuner()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "radio-core_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(d, f)).read()
                assert "radiocore_oracle" not in text, f
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert not re.search(r"#include\s+\"[^\"]*oracle", text), f
