#!/usr/bin/env python3
"""Instruction mix of one kernel in a device assembly listing (hipcc --cuda-device-only -S):
    tools/isa_mix.py file.s <substring of the mangled kernel name>
Counts instructions by class between the kernel's label and its s_endpgm, plus the s_waitcnt flavours."""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next((i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and (l.rstrip().endswith((":", ")")) or ": " in l)),
                 None)
    if start is None:
        names = sorted({l.split(":")[0][:100] for l in lines if l.startswith("_Z") and ":" in l})
        sys.exit("no kernel label contains %r; %d labels, e.g.\n  %s" % (key, len(names), "\n  ".join(names[:8])))
    mix = collections.Counter()
    waits = collections.Counter()
    ops = collections.Counter()
    n = 0
    for l in lines[start + 1:]:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        n += 1
        ops[op] += 1
        if op.startswith("v_pk_"): mix["valu_pk"] += 1
        elif op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): mix["valu_trans"] += 1
        elif op.startswith("v_"): mix["valu"] += 1
        elif op.startswith("ds_"): mix["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): mix["vmem"] += 1
        elif op.startswith("s_waitcnt"):
            mix["s_waitcnt"] += 1
            waits[" ".join(t.split()[1:])] += 1
        elif op.startswith("s_barrier"): mix["s_barrier"] += 1
        elif op.startswith("s_"): mix["salu"] += 1
        else: mix["other"] += 1
        if op == "s_endpgm":
            break
    print("instructions:", n)
    for k, v in mix.most_common():
        print("  %-12s %6d" % (k, v))
    print("top opcodes:", ", ".join("%s %d" % kv for kv in ops.most_common(22)))
    print("waits:", ", ".join("%s x%d" % kv for kv in waits.most_common(8)))


if __name__ == "__main__":
    main()
