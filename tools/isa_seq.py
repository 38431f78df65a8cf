"""Memory-operation / wait / branch skeleton of every barrier interval of a kernel in an AMDGPU .s file (experiments):
L global load, S global store, r / w LDS read / write, [vN lM] s_waitcnt, B conditional branch, . fp64 VALU op.
    python tools/isa_seq.py file.s [first_interval] [last_interval]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 10 ** 9
b, seq = 0, {}
for ln in lines:
    if not ln.startswith("\t") or ln.startswith("\t."):
        continue
    t = ln.split()
    if not t:
        continue
    op = t[0]
    if op == "s_barrier":
        b += 1
        continue
    s = seq.setdefault(b, [])
    if op.startswith("global_load") or op.startswith("scratch_load"):
        s.append("L")
    elif op.startswith("global_store") or op.startswith("scratch_store"):
        s.append("S")
    elif op.startswith("ds_read"):
        s.append("r")
    elif op.startswith("ds_write"):
        s.append("w")
    elif op == "s_waitcnt":
        a = " ".join(t[1:])
        m = re.search(r"vmcnt\((\d+)\)", a)
        l = re.search(r"lgkmcnt\((\d+)\)", a)
        s.append("[" + ("v%s" % m.group(1) if m else "") + ("l%s" % l.group(1) if l else "") + "]")
    elif op.startswith("s_cbranch"):
        s.append("B")
    elif op.startswith("v_") and "f64" in op:
        if not s or s[-1] != ".":
            s.append(".")
for k in sorted(seq):
    if lo <= k <= hi:
        print(f"{k:3d}: {''.join(seq[k])}")
