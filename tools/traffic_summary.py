#!/usr/bin/env python3
"""Per-kernel HBM-side bytes from the raw counter table of tools/profile_traffic.sh.

    tools/traffic_summary.py gpurun_out/<tag>_traffic_raw.txt [--json profiles/hbm_traffic.json --source NAME]

Read side: the L2's memory-side requests come in three sizes and gfx950 exposes one counter per size
(TCC_EA0_RDREQ_32B/_64B/_128B; their sum equals TCC_EA0_RDREQ), so
    bytes_read  = 32 n32 + 64 n64 + 128 n128
is exact, and the printed `fetch_factor` = bytes_read / (FETCH_SIZE KiB * 1024) is the correction the
MI355X guide describes for that kernel's access shape (FETCH_SIZE tallies every request at 64 bytes: 2.0 for
128-byte requests, 1.0 for 64-byte ones, 0.5 for 32-byte ones).  Write side: TCC_EA0_WRREQ_64B counts the 64-byte
writes, the rest are 32-byte: bytes_written = 64 n64 + 32 (n - n64); WRITE_SIZE (KiB) is printed beside it.

--json: also writes the per-stage totals bench.py reports as roofline.traffic (kernel -> stage map below, cfg4),
stamped with the commit, a digest of the kernel sources (provenance.py) and the names of the profiled kernels.
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import provenance  # noqa: E402

STAGES = [   # (stage of rcfm_profile_*, regex on the kernel name); first match wins
    ("tuner_fft_N", r"k_fft_tile<(600|625|640),.*LoadPlainT<false>, (StorePlainT<false>|StoreRowWindow)"),
    ("tuner_ifft_B", r"LoadTunerGather|StorePhase"),
    ("pilot_stage", r"k_pilot_stage"),
    ("rfft_B", r"LoadRealPair|LoadPhaseStepPair"),
    ("ifft_B", r"k_fft_tile2<.*MidHilbertMask|k_fft_tile2_pair"),
    ("fft_B", r"k_fft_tile2_decim|k_fft_tile<\d+, [\d, ]+true, \d+, LoadPlainT<false>, StorePlainT<true>"),
    ("ifft_A", r"StoreRealImagSplit"),
    ("deemphasis", r"k_fir51|k_fir<|k_fir\b"),
    ("deemph_state", r"k_fir_state"),
]


def parse(path):
    data = {}
    key = None
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith("=="):
            continue
        if line.startswith("("):
            m = re.match(r"\('(.*)', '(\d+)'\)", line)
            key = (m.group(1), m.group(2)) if m else None
            continue
        m = re.match(r"\s+(\S+)\s+([\d.]+)$", line)
        if m and key:
            data.setdefault(key, {})[m.group(1)] = float(m.group(2))
    return data


def main():
    args = sys.argv[1:]
    raw = args[0]
    out_json = args[args.index("--json") + 1] if "--json" in args else None
    source = args[args.index("--source") + 1] if "--source" in args else raw
    data = parse(raw)
    print("| kernel (grid) | read bytes | = n32/n64/n128 requests | FETCH_SIZE KiB | fetch_factor | written bytes | WRITE_SIZE KiB |")
    print("|---|---:|---|---:|---:|---:|---:|")
    stages = {}
    for (name, grid), c in data.items():
        if "RDREQ" not in "".join(c) or not name.startswith(("k_", "cal_")):
            continue          # (input synthesis and self-check kernels of torch / rocFFT are not the product)
        n32, n64, n128 = c.get("TCC_EA0_RDREQ_32B_sum", 0), c.get("TCC_EA0_RDREQ_64B_sum", 0), c.get("TCC_EA0_RDREQ_128B_sum", 0)
        tot = c.get("TCC_EA0_RDREQ_sum", 0)
        other = tot - n32 - n64 - n128
        rd = 32 * n32 + 64 * n64 + 128 * n128 + 64 * max(other, 0)
        w = c.get("TCC_EA0_WRREQ_sum", 0)
        w64 = c.get("TCC_EA0_WRREQ_64B_sum", 0)
        wr = 64 * w64 + 32 * (w - w64)
        fs = c.get("FETCH_SIZE", 0)
        ws = c.get("WRITE_SIZE", 0)
        fac = rd / (fs * 1024) if fs else float("nan")
        if rd + wr < 1e6:
            continue
        print("| `%s` (%s) | %.4g | %.3g / %.3g / %.3g | %.0f | %.2f | %.4g | %.0f |" %
              (name[:95], grid, rd, n32, n64, n128, fs, fac, wr, ws))
        for st, rx in STAGES:
            if re.search(rx, name):
                e = stages.setdefault(st, {"kernels": [], "read_bytes": 0.0, "written_bytes": 0.0})
                e["kernels"].append({"kernel": name[:120], "grid": grid, "read_bytes": rd, "written_bytes": wr,
                                     "fetch_size_kib": fs, "fetch_factor": round(fac, 3), "write_size_kib": ws})
                e["read_bytes"] += rd
                e["written_bytes"] += wr
                break
    # the input synthesis of bench.py runs the wideband FFT once with the plain last-pass store; the bench itself
    # (rcfm_tuner_shard) uses the row-window store: keep the latter
    fft = stages.get("tuner_fft_N")
    if fft and any("StoreRowWindow" in k["kernel"] for k in fft["kernels"]):
        windowed = {k["kernel"].split(", Store")[0] for k in fft["kernels"] if "StoreRowWindow" in k["kernel"]}
        keep = [k for k in fft["kernels"] if "StoreRowWindow" in k["kernel"] or k["kernel"].split(", Store")[0] not in windowed]
        fft["kernels"] = keep
        fft["read_bytes"] = sum(k["read_bytes"] for k in keep)
        fft["written_bytes"] = sum(k["written_bytes"] for k in keep)
    if out_json:
        for st, e in stages.items():
            e["hbm_bytes_per_launch"] = e["read_bytes"] + e["written_bytes"]
            e["source"] = source
            e["note"] = "sum over the stage's kernels of one launch each (one chunk of channels; the wideband FFT once per buffer)"
        # the stamp bench.py checks: these counters describe THIS device code (run the summary in the checkout the
        # profiled library was built from, before touching radio-core_amd/csrc again)
        names = sorted({name for (name, _), c in data.items() if name.startswith("k_")})
        stages["_meta"] = {"kernel_source_sha": provenance.kernel_source_sha(), "commit": provenance.git_head(),
                           "source": source, "kernels": [n[:140] for n in names]}
        json.dump(stages, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
