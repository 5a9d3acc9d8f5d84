#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one translation unit (compiler view):
    tools/kernel_resources.py radio-core_amd/csrc/fft_engine.hip [filter-substring] [-DFLAG ...]
Runs hipcc -Rpass-analysis=kernel-resource-usage for gfx950 and demangles the names."""
import re
import subprocess
import sys


def main():
    src = sys.argv[1]
    flt = [a for a in sys.argv[2:] if not a.startswith("-")]
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Iinclude",
           "-Iradio-core_amd/csrc", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    names = [b.split("\n")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    print("%5s %5s %5s %7s %4s %7s  kernel" % ("VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS"))
    for b, n in zip(blocks, dem):
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        n = n.replace("rcfm::fftk::", "").replace("rcfm::", "").replace("void ", "").replace("(anonymous namespace)::", "")
        if flt and not all(f in n for f in flt):
            continue
        print("%5d %5d %5d %7d %4d %7d  %s" % (g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                                               g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"), n[:150]))


if __name__ == "__main__":
    main()
