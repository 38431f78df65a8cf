"""Hunt for the rare abort of the six-tiles-in-one-process GPU runs.

Runs the child of tests/helpers.run_in_child N times with the fault handler on, and records for every run the return code,
whether the result file had been written completely BEFORE the process died (i.e. whether the death happened during
interpreter / runtime teardown, after all work was done), and the tail of stderr.  Writes gpurun_out/abort_hunt.json.

    python tools/abort_hunt.py [--what acoustic] [--runs 60] [--exit hard|normal]
"""
import argparse
import json
import os
import pickle
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="acoustic")
    ap.add_argument("--runs", type=int, default=60)
    ap.add_argument("--exit", default="normal", choices=("normal", "hard"))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "abort_hunt.json"))
    args = ap.parse_args()
    tmp = tempfile.mkdtemp()
    out = os.path.join(tmp, "result.pkl")
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r}); "
            f"import helpers; helpers._child_main({args.what!r}, {out!r}, hard_exit={args.exit == 'hard'!r})")
    env = dict(os.environ, PYTHONFAULTHANDLER="1")
    records = []
    for r in range(args.runs):
        if os.path.exists(out):
            os.remove(out)
        t0 = time.time()
        p = subprocess.run([sys.executable, "-X", "faulthandler", "-c", code], capture_output=True, text=True, env=env, timeout=900)
        complete = False
        if os.path.exists(out):
            try:
                with open(out, "rb") as f:
                    pickle.load(f)
                complete = True
            except Exception:  # noqa: BLE001
                complete = False
        rec = {"run": r, "rc": p.returncode, "seconds": round(time.time() - t0, 2), "result_complete": complete}
        if p.returncode != 0:
            rec["stderr_tail"] = p.stderr[-6000:]
            rec["stdout_tail"] = p.stdout[-2000:]
        records.append(rec)
        print(json.dumps({k: v for k, v in rec.items() if k not in ("stderr_tail", "stdout_tail")}), flush=True)
    bad = [r for r in records if r["rc"] != 0]
    summary = {"what": args.what, "runs": args.runs, "exit": args.exit, "failures": len(bad),
               "failures_after_result_was_written": sum(1 for r in bad if r["result_complete"]), "records": records}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "records"}))


if __name__ == "__main__":
    main()
