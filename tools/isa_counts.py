#!/usr/bin/env python3
"""Static instruction counts per kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only): VALU / DS / SALU /
memory instructions, DPP moves, FLAT and scratch accesses.  usage: isa_counts.py file.s [substring of the kernel name]"""
import collections
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if want not in name:
            continue
        lines = [l.strip() for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        ops = [l.split()[0] for l in lines if l]
        c = collections.Counter(ops)
        cls = lambda p: sum(v for k, v in c.items() if k.startswith(p))
        print(name)
        print(f"  total {len(ops)}  valu {cls('v_')}  ds {cls('ds_')}  salu {cls('s_')}  global {cls('global_')}  flat {cls('flat_')}"
              f"  scratch {cls('scratch_')}  dpp {sum(1 for l in lines if 'dpp' in l or 'quad_perm' in l or 'row_' in l)}"
              f"  waitcnt {c.get('s_waitcnt', 0)}  branches {cls('s_cbranch')}")
        print("  top:", ", ".join(f"{k} {v}" for k, v in c.most_common(24)))


if __name__ == "__main__":
    main()
