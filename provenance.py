"""Ties measured evidence to the code it describes.

`kernel_source_sha()` is a digest of everything that defines the device code of librcfm.so (the HIP sources and
headers under radio-core_amd/csrc plus include/rcfm.h and rcfm_tools.h).  tools/traffic_summary.py stamps it (and the commit) into
profiles/hbm_traffic.json when the PMC passes are summarised; bench.py recomputes it at run time and reports
`"traffic": null, "traffic_stale": true` when the kernels have changed since the counters were collected.
"""

import hashlib
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.abspath(__file__))


def kernel_source_files(root=ROOT):
    csrc = os.path.join(root, "radio-core_amd", "csrc")
    files = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".h", ".hip"))]
    files.append(os.path.join(root, "include", "rcfm.h"))
    files.append(os.path.join(root, "include", "rcfm_tools.h"))
    return files


def kernel_source_sha(root=ROOT):
    h = hashlib.sha256()
    for path in kernel_source_files(root):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def git_head(root=ROOT):
    """Short commit id, with '+dirty' when tracked files differ from it; None outside a git checkout (GPU boxes)."""
    try:
        head = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True,
                              timeout=10, check=True).stdout.strip()
        dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--untracked-files=no"],
                               capture_output=True, text=True, timeout=10, check=True).stdout.strip()
        return head + ("+dirty" if dirty else "")
    except (OSError, subprocess.SubprocessError):
        return None


def stage_traffic(stage, path=None, current_sha=None):
    """(bytes per launch | None, source | None, stale: bool) for one profiled stage.

    stale = the committed counters were collected on other device code than what is being run (or carry no stamp
    at all): the number is withheld rather than quoted against kernels it does not describe."""
    path = path or os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as fh:
            table = json.load(fh)
    except (OSError, ValueError):
        return None, None, False
    entry = table.get(stage)
    if not entry:
        return None, None, False
    meta = table.get("_meta") or {}
    sha = current_sha if current_sha is not None else kernel_source_sha()
    if meta.get("kernel_source_sha") != sha:
        return None, entry.get("source"), True
    try:
        return float(entry["hbm_bytes_per_launch"]), entry.get("source"), False
    except (KeyError, TypeError, ValueError):
        return None, entry.get("source"), False
