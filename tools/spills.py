"""Registers, spills, scratch and LDS of every kernel in an AMDGPU .s file (the metadata block at its end):
   python tools/spills.py build/isa/fvt_variants.s [kernel-substring]"""
import re, sys

def main(path, want=""):
    cur = {}
    rows = []
    for ln in open(path):
        m = re.match(r"\s+\.(name|vgpr_count|agpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size|max_flat_workgroup_size):\s+(\S+)", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "name" and not v.startswith("_Z"):
            continue
        cur[k] = v
        if k == "vgpr_spill_count":
            rows.append(dict(cur))
    print(f"{'kernel':70s} {'threads':>7s} {'vgpr':>5s} {'agpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s}")
    for r in rows:
        if want in r.get("name", ""):
            print(f"{r.get('name','')[:70]:70s} {r.get('max_flat_workgroup_size','?'):>7s} {r.get('vgpr_count','?'):>5s} {r.get('agpr_count','?'):>5s} "
                  f"{r.get('vgpr_spill_count','?'):>6s} {r.get('sgpr_spill_count','?'):>6s} {r.get('private_segment_fixed_size','?'):>7s} {r.get('group_segment_fixed_size','?'):>6s}")

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
