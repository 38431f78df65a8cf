"""Static instruction census of the kernels in an AMDGPU .s file (or of a .hip source compiled here):
   python tools/isa_census.py <file.s | stem> [kernel-substring]
Classes: fp64 arithmetic, other VALU, v_cmp, v_cndmask, lane moves (SGPR spills), SALU, branches, waits, LDS, VMEM, scratch."""
import collections, re, subprocess, sys, os

def census(path, want=""):
    cur = None
    data = collections.OrderedDict()
    meta = {}
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            data[cur] = collections.Counter()
            continue
        m = re.match(r"^\s*\.amdhsa_next_free_vgpr\s+(\d+)", ln)
        if m and cur:
            meta.setdefault(cur, {})["vgpr"] = int(m.group(1))
        m = re.match(r"^\s*\.amdhsa_group_segment_fixed_size\s+(\d+)", ln)
        if m and cur:
            meta.setdefault(cur, {})["lds"] = int(m.group(1))
        m = re.match(r"^\s*;\s*ScratchSize:\s*(\d+)", ln)
        if m and cur:
            meta.setdefault(cur, {})["scratch"] = int(m.group(1))
        if cur and ln.startswith("\t") and not ln.startswith("\t.") and not ln.startswith("\t;"):
            op = ln.split()[0]
            c = data[cur]
            if op.startswith("v_cmp"): c["v_cmp"] += 1
            elif op.startswith("v_cndmask"): c["cndmask"] += 1
            elif op in ("v_readlane_b32", "v_writelane_b32"): c["lane"] += 1
            elif op.startswith("v_") and op.endswith("_f64") or op.startswith("v_fmac_f64") or op.startswith("v_rcp_f64"): c["f64"] += 1
            elif op.startswith("v_mov") or op.startswith("v_accvgpr"): c["vmov"] += 1
            elif op.startswith("v_"): c["v_int"] += 1
            elif op.startswith("s_waitcnt") or op.startswith("s_nop"): c["wait"] += 1
            elif op.startswith("s_cbranch") or op.startswith("s_branch"): c["branch"] += 1
            elif op.startswith("s_barrier"): c["barrier"] += 1
            elif op.startswith("s_load") or op.startswith("s_buffer"): c["smem"] += 1
            elif op.startswith("s_"): c["salu"] += 1
            elif op.startswith("ds_"): c["lds"] += 1
            elif op.startswith("scratch_"): c["scratch"] += 1
            elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): c["vmem"] += 1
            else: c["other"] += 1
    keys = ["f64", "v_int", "vmov", "v_cmp", "cndmask", "lane", "salu", "branch", "wait", "smem", "lds", "vmem", "scratch", "barrier", "other"]
    print(f"{'kernel':58s} {'total':>6s} " + " ".join(f"{k:>7s}" for k in keys) + "   vgpr  lds")
    for k, c in data.items():
        if want not in k or not c:
            continue
        mt = meta.get(k, {})
        print(f"{k[:58]:58s} {sum(c.values()):6d} " + " ".join(f"{c[x]:7d}" for x in keys) + f"   {mt.get('vgpr', 0):4d} {mt.get('lds', 0):6d}")

if __name__ == "__main__":
    src = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    if not src.endswith(".s"):
        d = "build/isa"
        os.makedirs(d, exist_ok=True)
        extra = os.environ.get("ISA_FLAGS", "").split()
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                               "-Wno-unused-function", "-Wno-unused-const-variable", "--cuda-device-only", "-S", *extra,
                               f"pace_amd/csrc/{src}.hip", "-o", f"{d}/{os.path.basename(src)}.s"])
        src = f"{d}/{os.path.basename(src)}.s"
    census(src, want)
