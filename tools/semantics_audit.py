"""Which stencils of the reference rely on the one execution-model rule that no reference-held test pins?

tools/gtinterp.py writes API (argument) fields only inside origin .. origin + domain; a later statement of the SAME stencil
that reads such a field at a horizontal offset therefore sees the new value inside the domain and the caller's old value
outside it.  (GT4Py's extent analysis could instead widen the write.)  This script runs one whole DynamicalCore.step_dynamics
of the reference (six ranks on threads, n_split = 2) with the interpreter's audit switched on and lists every
(stencil, offset) where an API field is read at a non-zero horizontal offset after having been written by the same call.
Dev container only.

    python tools/semantics_audit.py  -> tools/semantics_audit.json
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

    gtinterp.AUDIT = []
    gtinterp.DIFFERENTIAL = {}
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
    hits = {}
    launch = {}
    for name, off, origin, domain, shape in gtinterp.AUDIT:
        hits.setdefault(name, set()).add(tuple(off))
        launch.setdefault(name, set()).add((origin, domain, shape))
    out = {"offset_reads_of_api_fields_written_by_the_same_call": {k: sorted(v) for k, v in sorted(hits.items())},
           "their_launch_windows_origin_domain_storage": {k: sorted(v) for k, v in sorted(launch.items())},
           "differential_entry_vs_sequential": {k: {"calls": v[0], "calls_that_differ": v[1]} for k, v in sorted(gtinterp.DIFFERENTIAL.items())}}
    with open(os.path.join(HERE, "semantics_audit.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out['differential_entry_vs_sequential'].items() if v['calls_that_differ']}, indent=1))
    print(json.dumps(out['their_launch_windows_origin_domain_storage']))
    print(len(out['differential_entry_vs_sequential']), 'stencils,', sum(v['calls'] for v in out['differential_entry_vs_sequential'].values()), 'calls')


if __name__ == "__main__":
    main()
