"""Per barrier interval of every kernel in an AMDGPU .s file: instruction, scratch (spill) and v_readlane / v_writelane
(SGPR spill) counts.   python tools/isa_stats.py file.s [kernel-substring]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
cur, data = None, {}
for ln in lines:
    m = re.match(r"^(_Z\w+):\s*; @", ln)
    if m:
        cur = m.group(1)
        data[cur] = {"b": 0, "scr": {}, "lane": {}, "n": {}, "vmem": {}, "lds": {}}
    if cur is None or not ln.startswith("\t") or ln.startswith("\t."):
        continue
    d = data[cur]
    op = ln.split()[0] if ln.split() else ""
    if op == "s_barrier":
        d["b"] += 1
    b = d["b"]
    d["n"][b] = d["n"].get(b, 0) + 1
    if op.startswith("scratch_"):
        d["scr"][b] = d["scr"].get(b, 0) + 1
    if op in ("v_readlane_b32", "v_writelane_b32"):
        d["lane"][b] = d["lane"].get(b, 0) + 1
    if op.startswith("global_") or op.startswith("buffer_"):
        d["vmem"][b] = d["vmem"].get(b, 0) + 1
    if op.startswith("ds_"):
        d["lds"][b] = d["lds"].get(b, 0) + 1
for k, d in data.items():
    if want not in k:
        continue
    print(k[:60], "barriers", d["b"])
    print("  interval: instrs / vmem / lds / scratch / sgpr-spill-moves")
    for b in range(d["b"] + 1):
        print(f"   {b:3d}: {d['n'].get(b, 0):5d} {d['vmem'].get(b, 0):4d} {d['lds'].get(b, 0):4d} {d['scr'].get(b, 0):4d} {d['lane'].get(b, 0):4d}")
