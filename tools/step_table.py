"""The per-kernel table of one acoustic substep (d_sw + riem_solver3): launches, microseconds alone, counted HBM-side traffic
(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected as tools/pmc_summary.py does), the kernel's own algorithmic bytes (distinct
3-D fields it reads / writes, once each) and the rates both imply.

    python tools/step_table.py gpurun_out/r04/kernel_times_single_stream.txt gpurun_out/r04/pmc_kernels.json [N nz] > profiles/r04_step_table.json
"""
import json
import re
import sys

# distinct 3-D fields per launch (in + out), by kernel-name fragment; the sponge-level / edge-strip kernels touch a few levels or rows only
FIELDS = [
    ("k_fxadv_main", 4, "uc, vc -> ut, vt"),
    ("k_fxadv_edges", 0.2, "edge strips of ut, vt"),
    ("k_fxadv_fluxes", 10, "ut, vt, cx, cy -> crx, cry, xfx, yfx, cx, cy"),
    ("k_fvt_scalars", 18, "delp, w, q_con, pt, crx, cry, xfx, yfx, mfx, mfy -> delp, w, q_con, pt, mfx, mfy, diss_est, heat term"),
    ("k_copy_scalars", 8, "(in-place API only) four fields copied back"),
    ("k_kinetic_energy", 5, "uc, vc, u, v -> ke"),
    ("k_vorticity", 3, "u, v -> rel. vorticity"),
    ("k_divdamp_low_and_copy", 0.3, "sponge levels"),
    ("k_divdamp_fused", 12, "u, v, ua, va, divgd, ke, vorticity -> divgd, delpc, uc, vc, ke, damped vorticity"),
    ("k_fvtILi6ELi0ELi0E", 12, "vorticity, crx, cry, xfx, yfx, u, v, ke -> u*, v*, vorticity damping fluxes (2)"),
    ("k_fvt<6, 0, 0>", 12, "vorticity, crx, cry, xfx, yfx, u, v, ke -> u*, v*, vorticity damping fluxes (2)"),
    ("k_fvtILi6ELi2ELi1E", 9, "q, crx, cry, xfx, yfx, mass fluxes (2), delp -> q'"),
    ("k_fvt<6, 2, 1>", 9, "q, crx, cry, xfx, yfx, mass fluxes (2), delp -> q'"),
    ("k_heat_source", 11, "u*, v*, damped vorticity, damping fluxes (2), delp, heat term -> heat_source, diss_est, u, v"),
    ("k_riem_column", 13, "cappa, q_con, delp, pt, zh, pe, w -> delz, zh, pe, ppe, pk3, w"),
]


def fields_of(name):
    for frag, n, what in FIELDS:
        if frag in name:
            return n, what
    return None, ""


def main(times_path, pmc_path, n=192, nz=79):
    field_mb = n * n * nz * 8 / 1e6
    pmc = json.load(open(pmc_path))
    rows = []
    for ln in open(times_path):
        m = re.match(r"\s+(\S+)\s+calls\s+(\d+)\s+avg\s+([\d.]+)\s+min\s+([\d.]+)", ln)
        if not m:
            continue
        mangled, calls, avg, mn = m.group(1), int(m.group(2)), float(m.group(3)), float(m.group(4))
        nf, what = fields_of(mangled)
        counted = None
        for k, v in pmc.items():
            key = re.sub(r"[^A-Za-z0-9_]", "", k.split("<")[0].split("::")[-1])
            if key and key in mangled:
                digits_k = re.findall(r"-?\d+", k.split("<", 1)[1]) if "<" in k else []
                digits_m = re.findall(r"Li(n?\d+)E", mangled)
                digits_m = [d.replace("n", "-") for d in digits_m]
                if digits_k == digits_m[:len(digits_k)]:
                    counted = v.get("hbm_bytes_per_launch")
        row = {"kernel": mangled, "us_alone_avg": avg, "us_alone_min": mn, "distinct_fields": nf, "fields": what,
               "algorithmic_MB": None if nf is None else round(nf * field_mb, 1),
               "counted_MB": None if counted is None else round(counted / 1e6, 1)}
        if counted:
            row["counted_TBps"] = round(counted / (avg * 1e-6) / 1e12, 2)
            if nf:
                row["counted_over_algorithmic"] = round(counted / 1e6 / (nf * field_mb), 2)
        if nf:
            row["algorithmic_TBps"] = round(nf * field_mb * 1e6 / (avg * 1e-6) / 1e12, 2)
        rows.append(row)
    step = [r for r in rows if "copy_scalars" not in r["kernel"] and "Li2ELi1E" not in r["kernel"]]
    out = {"size": f"C{n} x {nz}, fp64", "field_MB": round(field_mb, 2), "kernels": rows,
           "step_sum_us_alone": round(sum(r["us_alone_avg"] for r in step), 1),
           "step_sum_counted_MB": round(sum(r["counted_MB"] or 0 for r in step), 1),
           "note": "one launch of each kernel per substep except k_copy_scalars (in-place C API only) and k_fvt<6,2,1> (measured by the "
                   "harness, not launched by the step); algorithmic = the kernel's own distinct fields, intermediates between kernels included "
                   "(the operator as a whole: 45 fields = 360 B per cell, SURVEY.md section 8d)"}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0], a[1], *(int(x) for x in a[2:4]))
