"""The per-kernel table of one acoustic substep (d_sw + riem_solver3) as bench.py runs it: launches, microseconds (rocprofv3
--kernel-trace of bench.py: the kernels IN the step), counted HBM-side traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of the same
command, corrected as tools/pmc_summary.py does), the kernel's own algorithmic bytes (distinct 3-D fields it reads / writes, once
each) and the rates both imply.

    python tools/step_table.py gpurun_out/r05/kernel_stats.csv gpurun_out/r05/pmc_traffic.json [N nz] > profiles/r05_step_table.json
"""
import csv
import json
import re
import sys

# distinct 3-D fields per launch (in + out), by kernel-name fragment; the edge-strip kernel touches a few rows only
FIELDS = [
    ("k_fxadv_fused", 10.65, "uc, vc, cx, cy -> crx, cry, xfx, yfx, cx, cy (ut, vt on the frame of the plane, 11 % of it): the frame's stages and the interior's stream in one launch"),
    ("k_fxadv_frame", 0.45, "uc, vc -> ut, vt on the frame of the plane (11 % of it)"),
    ("k_fxadv_edges", 0.2, "edge strips of ut, vt"),
    ("k_fxadv_fluxes", 10, "uc, vc (ut, vt on the frame), cx, cy -> crx, cry, xfx, yfx, cx, cy"),
    ("k_fvt_scalars", 26, "delp, w, q_con, pt, crx, cry, xfx, yfx, mfx, mfy, vorticity, u, v, ke, damped vorticity, heat_source -> delp, w, "
                          "q_con, pt, mfx, mfy, diss_est, u, v, heat_source"),
    ("k_ke_vorticity", 6, "uc, vc, u, v -> ke, rel. vorticity (two kinds of workgroups of one launch)"),
    ("k_kinetic_energy", 5, "uc, vc, u, v -> ke"),
    ("k_vorticity", 3, "u, v -> rel. vorticity"),
    ("k_divdamp_fused", 3, "divgd, vorticity -> damped vorticity (+ the sponge levels: u, v, ua, va, uc, vc on two or three levels)"),
    ("k_riem_column", 13, "cappa, q_con, delp, pt, zh, pe, w -> delz, zh, pe, ppe, pk3, w"),
]


def fields_of(name):
    for frag, n, what in FIELDS:
        if frag in name:
            return n, what
    return None, ""


def main(stats_path, pmc_path, n=192, nz=79):
    field_mb = n * n * nz * 8 / 1e6
    pmc = json.load(open(pmc_path))
    rows = []
    for r in csv.DictReader(open(stats_path)):
        mangled = r["kernel"]
        nf, what = fields_of(mangled)
        if nf is None:
            continue
        avg, mn = float(r["avg_us"]), float(r["min_us"])
        counted = None
        for k, v in pmc.items():
            key = re.sub(r"[^A-Za-z0-9_]", "", k.split("<")[0].split("::")[-1])
            if key and key in mangled:
                counted = v.get("hbm_bytes_per_launch")
        row = {"kernel": mangled[:60], "launches_in_trace": int(r["calls"]), "us_avg": avg, "us_min": mn, "distinct_fields": nf, "fields": what,
               "algorithmic_MB": round(nf * field_mb, 1), "counted_MB": None if counted is None else round(counted / 1e6, 1),
               "vgpr": r.get("vgpr"), "lds_bytes": r.get("lds_bytes")}
        if counted:
            row["counted_TBps"] = round(counted / (avg * 1e-6) / 1e12, 2)
            row["counted_over_algorithmic"] = round(counted / 1e6 / (nf * field_mb), 2)
        row["algorithmic_TBps"] = round(nf * field_mb * 1e6 / (avg * 1e-6) / 1e12, 2)
        rows.append(row)
    out = {"size": f"C{n} x {nz}, fp64", "field_MB": round(field_mb, 2), "kernels": rows,
           "step_sum_us": round(sum(r["us_avg"] for r in rows), 1),
           "step_sum_counted_MB": round(sum(r["counted_MB"] or 0 for r in rows), 1),
           "step_sum_algorithmic_MB": round(sum(r["algorithmic_MB"] for r in rows), 1),
           "note": "one launch of each kernel per substep (bench.py's configuration: the divergence damping's dead work fields skipped, the "
                   "scalars' and the winds' outputs in buffers of their own); algorithmic = the kernel's own distinct fields, intermediates "
                   "between kernels included (the operator as a whole: 45 fields = 360 B per cell, SURVEY.md section 8d)"}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0], a[1], *(int(x) for x in a[2:4]))
