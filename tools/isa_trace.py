#!/usr/bin/env python3
"""Condensed instruction-class trace of one kernel (loads / LDS / MFMA / waits / barriers in
program order) from hipcc -S output: shows where the compiler put the weight loads and the
waits relative to the MFMA blocks.  usage: isa_trace.py file.s kernel-substring"""
import sys

s = open(sys.argv[1]).read()
i = s.index(sys.argv[2])
i = s.index(":\n", s.index(sys.argv[2], i))
j = s.index(".Lfunc_end", i)


def kind(l):
    l = l.strip()
    for pre, k in (("v_mfma", "MFMA"), ("global_load", "GLOAD"), ("global_store", "GSTORE"),
                   ("ds_read", "DSR"), ("ds_write", "DSW"), ("ds_bpermute", "BPERM"),
                   ("scratch_", "SCRATCH"), ("s_barrier", "BARRIER"), ("buffer_", "BUF")):
        if l.startswith(pre):
            return k
    if l.startswith("s_waitcnt"):
        return l.split(";")[0].strip().replace("s_waitcnt ", "wait ")
    if l.startswith("s_cbranch") or l.startswith("s_branch"):
        return "br"
    if l.startswith(".LBB"):
        return l.split(":")[0]
    if l.startswith("v_") or l.startswith("s_"):
        return "alu"
    return None


out, last, cnt = [], None, 0
for l in s[i:j].split("\n"):
    k = kind(l)
    if k is None:
        continue
    if k == last:
        cnt += 1
    else:
        if last:
            out.append(f"{last}x{cnt}" if cnt > 1 else last)
        last, cnt = k, 1
out.append(f"{last}x{cnt}")
print(" ".join(out))
