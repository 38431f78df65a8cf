"""Does any stencil of the path assign, under `with horizontal(region[...])`, to an API field at points OUTSIDE its launch
window?  tools/gtinterp.py restricts every write to an API field to origin .. origin + domain (GT4Py's rule for API fields:
only temporaries get extended compute extents); a region whose bounds reach past the window would then be silently clipped
-- and if GT4Py did NOT clip it, fixtures generated through the interpreter would be wrong there.
This script runs one whole DynamicalCore.step_dynamics of the reference (six ranks on threads, n_split = 2) with the
interpreter's region audit on and lists, per (stencil, field), the number of region-masked assignments and how many points of
their masks lay outside the launch window.  Dev container only.

    python tools/region_write_audit.py  -> tools/region_write_audit.json
"""
import datetime
import json
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")


def main():
    import capture
    import gtinterp
    import pace.fv3core as fv3core
    import refenv
    from threadcomm import run_ranks

    gtinterp.REGION_AUDIT = {}
    config = capture.dycore_config(n_split=2, k_split=1, npx=13, npz=79, do_sat_adj=False)

    def rank(comm):
        env = refenv.build_rank(comm, 12, 79)
        dycore = fv3core.DynamicalCore(
            comm=env.cube, grid_data=env.grid_data, stencil_factory=env.stencil_factory, quantity_factory=env.qf,
            damping_coefficients=env.damping, config=config, timestep=datetime.timedelta(seconds=config.dt_atmos),
            phis=env.state.phis, state=env.state)
        dycore.step_dynamics(env.state)
        return 0

    run_ranks(6, rank)
    rows = sorted(gtinterp.REGION_AUDIT.items())
    clipped = {f"{k[0]}:{k[1]}": {"region_assignments": v[0], "mask_points_outside_launch_window": v[1],
                                  "of_which_from_explicit_bounds": v[2]} for k, v in rows if v[1]}
    out = {"stencil_field_pairs_with_region_assignments_to_api_fields": len(rows),
           "region_assignments": sum(v[0] for _, v in rows),
           "pairs_whose_region_mask_reached_outside_the_launch_window": clipped,
           "pairs_where_an_EXPLICIT_region_bound_lies_outside_the_window": sorted(k for k, v in clipped.items() if v["of_which_from_explicit_bounds"]),
           "all_pairs": {f"{k[0]}:{k[1]}": v[0] for k, v in rows}}
    with open(os.path.join(HERE, "region_write_audit.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "all_pairs"}, indent=1))


if __name__ == "__main__":
    main()
