"""CPU: roofline.traffic is only quoted for the device code it was measured on (provenance.py).

tools/traffic_summary.py stamps profiles/hbm_traffic.json with a digest of radio-core_amd/csrc + include/rcfm.h;
bench.py asks provenance.stage_traffic() and reports `"traffic": null, "traffic_stale": true` on a mismatch."""

import json
import os
import shutil

import provenance
from conftest import ROOT


def _table(path, sha):
    json.dump({"tuner_fft_N": {"hbm_bytes_per_launch": 11.2e9, "source": "profiles/x.md"},
               "_meta": {"kernel_source_sha": sha, "commit": "abc", "kernels": ["k_fft_tile<600>"]}}, open(path, "w"))


def test_matching_stamp_gives_the_number(tmp_path):
    p = str(tmp_path / "t.json")
    _table(p, provenance.kernel_source_sha())
    assert provenance.stage_traffic("tuner_fft_N", p) == (11.2e9, "profiles/x.md", False)
    assert provenance.stage_traffic("no_such_stage", p) == (None, None, False)


def test_changed_kernels_or_missing_stamp_are_stale(tmp_path):
    p = str(tmp_path / "t.json")
    _table(p, "0" * 16)
    assert provenance.stage_traffic("tuner_fft_N", p) == (None, "profiles/x.md", True)
    json.dump({"tuner_fft_N": {"hbm_bytes_per_launch": 1.0, "source": "old"}}, open(p, "w"))     # pre-stamp table
    assert provenance.stage_traffic("tuner_fft_N", p) == (None, "old", True)
    assert provenance.stage_traffic("tuner_fft_N", str(tmp_path / "absent.json")) == (None, None, False)


def test_digest_follows_the_kernel_sources(tmp_path):
    root = tmp_path / "copy"
    shutil.copytree(os.path.join(ROOT, "radio-core_amd", "csrc"), root / "radio-core_amd" / "csrc")
    os.makedirs(root / "include")
    shutil.copy(os.path.join(ROOT, "include", "rcfm.h"), root / "include" / "rcfm.h")
    shutil.copy(os.path.join(ROOT, "include", "rcfm_tools.h"), root / "include" / "rcfm_tools.h")
    assert provenance.kernel_source_sha(str(root)) == provenance.kernel_source_sha()
    with open(root / "radio-core_amd" / "csrc" / "kernels.hip", "a") as fh:
        fh.write("// touched\n")
    assert provenance.kernel_source_sha(str(root)) != provenance.kernel_source_sha()
    # documentation and Python do not enter the digest
    assert all(f.endswith((".h", ".hip")) for f in provenance.kernel_source_files())


def test_committed_table_is_either_fresh_or_reported_stale():
    value, source, stale = provenance.stage_traffic("tuner_fft_N")
    assert source is not None
    assert (value is None) == stale
