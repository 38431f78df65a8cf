"""Halo exchange: the product path (HIP pack/unpack kernel sources under emulation + ThreadComm / gloo transports)
against oracle/halo.py, which is bit-exact against the reference's CubedSphereCommunicator
(tools/crosscheck_oracle.py halo) and its own halo tests.  CPU only."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT, build_emu

N, NZ = 12, 5
DIMS = {"c": ["x", "y", "z"], "xi": ["x_interface", "y", "z"], "yi": ["x", "y_interface", "z"],
        "b": ["x_interface", "y_interface", "z"], "zi": ["x", "y", "z_interface"]}



def _wait_all(procs, timeout):
    """Wait for the rank processes of one run; as soon as one of them fails (or the time is up) the others -- which would
    otherwise sit in the rendezvous until their own timeouts -- are killed, and the test fails with the return codes."""
    import time

    deadline = time.time() + timeout
    codes = [None] * len(procs)
    try:
        while any(c is None for c in codes):
            for n, p in enumerate(procs):
                if codes[n] is None:
                    codes[n] = p.poll()
            if any(c not in (None, 0) for c in codes) or time.time() > deadline:
                break
            time.sleep(0.05)
    finally:
        for n, p in enumerate(procs):
            if p.poll() is None:
                p.kill()
                p.wait()
    assert codes == [0] * len(procs), f"rank processes ended with {codes} (None = killed: timeout or a sibling failed)"

def _base(seed=5):
    rng = np.random.default_rng(seed)
    return {k: [rng.random((N + 7, N + 7, NZ + 1)) for _ in range(6)] for k in DIMS}


def _expected(base):
    from oracle import halo as oh

    out = {}
    f = [a.copy() for a in base["c"]]; oh.halo_update(f, N, nk=NZ); out["c"] = f
    f = [a.copy() for a in base["b"]]; oh.halo_update(f, N, xi=1, yi=1, nk=NZ); out["b"] = f
    f = [a.copy() for a in base["zi"]]; oh.halo_update(f, N, n_pts=2); out["zi"] = f
    u, v = [a.copy() for a in base["yi"]], [a.copy() for a in base["xi"]]; oh.vector_halo_update(u, v, N, grid="d", nk=NZ)
    out["du"], out["dv"] = u, v
    u, v = [a.copy() for a in base["xi"]], [a.copy() for a in base["yi"]]; oh.vector_halo_update(u, v, N, grid="c", nk=NZ)
    out["cu"], out["cv"] = u, v
    u, v = [a.copy() for a in base["yi"]], [a.copy() for a in base["xi"]]; oh.synchronize_vector_interfaces(u, v, N, nk=NZ)
    out["su"], out["sv"] = u, v
    return out


def tile_program(comm, lib, base, device="cpu"):
    """What every tile runs (shared by the thread and the gloo tests)."""
    from pace_amd.util import CubedSphereCommunicator, QuantityFactory, SubtileGridSizer

    sizer = SubtileGridSizer.from_tile_params(nx_tile=N, ny_tile=N, nz=NZ, n_halo=3, extra_dim_lengths={}, layout=(1, 1))
    qf = QuantityFactory(sizer, device=device)
    cube = CubedSphereCommunicator(comm, device=device, lib=lib)
    r = cube.rank

    def q(key):
        x = qf.zeros(DIMS[key], "")
        x.set(base[key][r])
        return x

    out = {}
    s = q("c"); cube.halo_update(s, n_points=3); out["c"] = s.numpy()
    s = q("b"); cube.halo_update(s, n_points=3); out["b"] = s.numpy()
    s = q("zi"); cube.halo_update(s, n_points=2); out["zi"] = s.numpy()
    u, v = q("yi"), q("xi"); cube.vector_halo_update(u, v, n_points=3); out["du"], out["dv"] = u.numpy(), v.numpy()
    u, v = q("xi"), q("yi"); cube.vector_halo_update(u, v, n_points=3); out["cu"], out["cv"] = u.numpy(), v.numpy()
    u, v = q("yi"), q("xi"); cube.synchronize_vector_interfaces(u, v); out["su"], out["sv"] = u.numpy(), v.numpy()
    # a reusable multi-field updater with two exchanges in flight, waited out of order (dyn_core.py:686-699)
    a, b, cc = q("c"), q("c"), q("c")
    b.data[:] = b.data * 2.0
    cc.data[:] = cc.data + 1.0
    up3 = cube.get_scalar_halo_updater([qf.get_quantity_halo_spec(DIMS["c"])] * 2)
    up1 = cube.get_scalar_halo_updater([qf.get_quantity_halo_spec(DIMS["c"])])
    up3.start([a, b])
    up1.start([cc])
    up1.wait()
    up3.wait()
    out["m_a"], out["m_b"], out["m_c"] = a.numpy(), b.numpy(), cc.numpy()
    return out


def _check(results, base):
    exp = _expected(base)
    for t in range(6):
        for k, e in exp.items():
            assert np.array_equal(results[t][k], e[t]), (t, k)
        assert np.array_equal(results[t]["m_a"], exp["c"][t])
        # compute domain + edge halos (corners are never exchanged): b = 2 * field, c = field + 1 elementwise
        m = np.zeros((N + 7, N + 7), dtype=bool)
        m[3 : 3 + N, 0 : N + 6] = True
        m[0 : N + 6, 3 : 3 + N] = True
        assert np.array_equal(results[t]["m_b"][m][:, :NZ], 2.0 * exp["c"][t][m][:, :NZ])
        assert np.array_equal(results[t]["m_c"][m][:, :NZ], exp["c"][t][m][:, :NZ] + 1.0)


# ---- the reference's own pace.util, run natively (tools/make_golden_halo.py; no interpreter involved) ----
def native_fixture():
    from helpers import golden

    d = golden("halo_native_c12.npz")
    base = {k[3:]: [d[k][t] for t in range(6)] for k in d if k.startswith("in_")}
    exp = {k[4:]: d[k] for k in d if k.startswith("out_")}
    return int(d["n"]), int(d["nz"]), base, exp


def native_tile_program(comm, lib, base, n, nz, device="cpu"):
    from pace_amd.util import CubedSphereCommunicator, QuantityFactory, SubtileGridSizer

    sizer = SubtileGridSizer.from_tile_params(nx_tile=n, ny_tile=n, nz=nz, n_halo=3, extra_dim_lengths={}, layout=(1, 1))
    qf = QuantityFactory(sizer, device=device)
    cube = CubedSphereCommunicator(comm, device=device, lib=lib)
    r = cube.rank

    def q(key):
        x = qf.zeros(DIMS[key], "")
        x.set(base[key][r])
        return x

    out = {}
    s = q("c"); cube.halo_update(s, n_points=3); out["c"] = s.numpy()
    s = q("b"); cube.halo_update(s, n_points=3); out["b"] = s.numpy()
    s = q("zi"); cube.halo_update(s, n_points=2); out["zi"] = s.numpy()
    u, v = q("yi"), q("xi"); cube.vector_halo_update(u, v, n_points=3); out["du"], out["dv"] = u.numpy(), v.numpy()
    u, v = q("xi"), q("yi"); cube.vector_halo_update(u, v, n_points=3); out["cu"], out["cv"] = u.numpy(), v.numpy()
    u, v = q("yi"), q("xi"); cube.synchronize_vector_interfaces(u, v); out["su"], out["sv"] = u.numpy(), v.numpy()
    return out


def check_native(results, exp):
    for t in range(6):
        for k, e in exp.items():
            assert np.array_equal(results[t][k], e[t]), (t, k)


def test_oracle_halo_equals_the_reference_run():
    """oracle/halo.py against what the reference's CubedSphereCommunicator left in the same arrays (scalar, B-grid, z-interface,
    D- and C-grid vector updates, interface synchronisation): exactly."""
    from oracle import halo as oh

    n, nz, base, exp = native_fixture()
    cp = lambda k: [a.copy() for a in base[k]]  # noqa: E731
    got = {}
    f = cp("c"); oh.halo_update(f, n, nk=nz); got["c"] = f
    f = cp("b"); oh.halo_update(f, n, xi=1, yi=1, nk=nz); got["b"] = f
    f = cp("zi"); oh.halo_update(f, n, n_pts=2); got["zi"] = f
    u, v = cp("yi"), cp("xi"); oh.vector_halo_update(u, v, n, grid="d", nk=nz); got["du"], got["dv"] = u, v
    u, v = cp("xi"), cp("yi"); oh.vector_halo_update(u, v, n, grid="c", nk=nz); got["cu"], got["cv"] = u, v
    u, v = cp("yi"), cp("xi"); oh.synchronize_vector_interfaces(u, v, n, nk=nz); got["su"], got["sv"] = u, v
    for k, e in exp.items():
        for t in range(6):
            assert np.array_equal(got[k][t], e[t]), (k, t)


def test_halo_updates_six_tiles_equal_the_reference_run():
    """The product's pack / exchange / unpack (HIP kernel sources under emulation, ThreadComm) against the reference run."""
    from pace_amd import _lib
    from pace_amd.util import run_tiles

    lib = _lib.Library(build_emu())
    n, nz, base, exp = native_fixture()
    results = run_tiles(6, lambda comm: native_tile_program(comm, lib, base, n, nz))
    check_native(results, exp)


def test_halo_updates_six_tiles_on_threads():
    from pace_amd import _lib
    from pace_amd.util import run_tiles

    lib = _lib.Library(build_emu())
    base = _base()
    results = run_tiles(6, lambda comm: tile_program(comm, lib, base))
    _check(results, base)


@pytest.mark.gpu
def test_halo_updates_six_tiles_on_one_gpu(tmp_path):
    """The gfx950 pack/unpack kernels: six tiles resident on one device, one host thread per tile (in a child process, like
    helpers.run_in_child)."""
    import pickle

    script = tmp_path / "halo_gpu.py"
    out = tmp_path / "halo_gpu.pkl"
    script.write_text(
        f"import sys, pickle\nsys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "import test_halo\nfrom pace_amd import _lib\nfrom pace_amd.util import run_tiles\n"
        "lib = _lib.load(); base = test_halo._base()\n"
        "res = run_tiles(6, lambda comm: test_halo.tile_program(comm, lib, base, device='cuda'))\n"
        f"pickle.dump(res, open({str(out)!r}, 'wb'))\n")
    p = subprocess.run([sys.executable, "-X", "faulthandler", str(script)], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.returncode, p.stderr[-4000:])  # no retry (240 clean runs: profiles/r02_abort_hunt.json)
    _check(pickle.load(open(out, "rb")), _base())


_WORKER = r"""
import os, sys, pickle
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=6)
from pace_amd import _lib
from pace_amd.util import TorchDistComm
import test_halo
lib = _lib.Library(os.path.join({root!r}, "tests", "emu", "libpace_emu.so"))
out = test_halo.tile_program(TorchDistComm(), lib, test_halo._base())
pickle.dump(out, open(sys.argv[2], "wb"))
dist.barrier(); dist.destroy_process_group()
"""


def test_halo_updates_six_processes_gloo(tmp_path):
    """One process per tile over torch.distributed (gloo here; the same code path runs RCCL on the GPUs)."""
    import pickle

    build_emu()
    port = 29500 + os.getpid() % 2000
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(tmp_path / f"out{r}.pkl")]) for r in range(6)]
    _wait_all(procs, 300)
    results = [pickle.load(open(tmp_path / f"out{r}.pkl", "rb")) for r in range(6)]
    _check(results, _base())


def test_halo_updater_error_behaviour():
    """The misuse errors of the reference's HaloUpdater / communicator (halo_updater.py:225-232,277-278;
    communicator.py:521-553,700-709)."""
    from pace_amd import _lib
    from pace_amd.util import CubedSphereCommunicator, QuantityFactory, SubtileGridSizer, run_tiles
    from pace_amd.util.comm import ThreadComm, _World

    lib = _lib.Library(build_emu())
    with pytest.raises(ValueError, match="ranks"):
        CubedSphereCommunicator(ThreadComm(_World(4), 0), device="cpu", lib=lib)

    def program(comm):
        sizer = SubtileGridSizer.from_tile_params(nx_tile=N, ny_tile=N, nz=NZ, n_halo=3, extra_dim_lengths={}, layout=(1, 1))
        qf = QuantityFactory(sizer, device="cpu")
        cube = CubedSphereCommunicator(comm, device="cpu", lib=lib)
        spec = qf.get_quantity_halo_spec(DIMS["c"])
        with pytest.raises(RuntimeError):
            cube.get_scalar_halo_updater([])
        with pytest.raises(ValueError, match="zero halo points"):
            cube.get_scalar_halo_updater([qf.get_quantity_halo_spec(DIMS["c"], n_halo=0)])
        up = cube.get_scalar_halo_updater([spec])
        with pytest.raises(RuntimeError, match="before"):
            up.wait()
        q = qf.zeros(DIMS["c"], "")
        with pytest.raises(ValueError):
            up.start([q, q])
        up.start([q])
        with pytest.raises(RuntimeError, match="finished"):
            up.start([q])
        up.wait()
        return True

    assert all(run_tiles(6, program))


# ---- the ring stand-in topology (bench.py at 2 / 4 / 8 ranks; world-size-2 coverage of the multi-process path) ----
def ring_program(comm, lib, n_ranks, device="cpu"):
    from pace_amd.util import CubedSphereCommunicator, QuantityFactory, RingPartitioner, SubtileGridSizer

    sizer = SubtileGridSizer.from_tile_params(nx_tile=N, ny_tile=N, nz=NZ, n_halo=3, extra_dim_lengths={}, layout=(1, 1))
    qf = QuantityFactory(sizer, device=device)
    ring = CubedSphereCommunicator(comm, RingPartitioner(n_ranks), device=device, lib=lib)
    base = _base(seed=11)
    out = {}
    for key in ("c", "b"):
        qs = []
        for scale in (1.0, 3.0):
            x = qf.zeros(DIMS[key], "")
            x.set(base[key][ring.rank] * scale)
            qs.append(x)
        up = ring.get_scalar_halo_updater([qf.get_quantity_halo_spec(DIMS[key])] * 2)
        up.update(qs)
        out[key] = [x.numpy() for x in qs]
    return out


def _check_ring(results, n_ranks):
    """West halo = the previous tile's easternmost compute columns, etc.; corners and the compute domain untouched."""
    base = _base(seed=11)
    h = 3
    for key, st in (("c", 0), ("b", 1)):
        e = h + N + st  # end of the compute domain (staggered fields own one more point)
        for r in range(n_ranks):
            prev, nxt = base[key][(r - 1) % n_ranks], base[key][(r + 1) % n_ranks]
            for f, scale in enumerate((1.0, 3.0)):
                got = results[r][key][f]
                exp = base[key][r] * scale
                exp[0:h, h:e, :NZ] = prev[e - st - h:e - st, h:e, :NZ] * scale      # west  <- previous tile's east side
                exp[e:e + h, h:e, :NZ] = nxt[h + st:h + st + h, h:e, :NZ] * scale    # east  <- next tile's west side
                exp[h:e, 0:h, :NZ] = prev[h:e, e - st - h:e - st, :NZ] * scale      # south <- previous tile's north side
                exp[h:e, e:e + h, :NZ] = nxt[h:e, h + st:h + st + h, :NZ] * scale    # north <- next tile's south side
                assert np.array_equal(got, exp), (key, r, f)


@pytest.mark.parametrize("n_ranks", [2, 4])
def test_ring_exchange_on_threads(n_ranks):
    from pace_amd import _lib
    from pace_amd.util import run_tiles

    lib = _lib.Library(build_emu())
    _check_ring(run_tiles(n_ranks, lambda comm: ring_program(comm, lib, n_ranks)), n_ranks)


_RING_WORKER = r"""
import os, sys, pickle
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
from pace_amd import _lib
from pace_amd.util import TorchDistComm
import test_halo
lib = _lib.Library(os.path.join({root!r}, "tests", "emu", "libpace_emu.so"))
out = test_halo.ring_program(TorchDistComm(), lib, 2)
pickle.dump(out, open(sys.argv[2], "wb"))
dist.barrier(); dist.destroy_process_group()
"""


def test_ring_exchange_two_processes_gloo(tmp_path):
    """world_size 2 over torch.distributed: both ranks exchange all four edges with the same peer in one message."""
    import pickle

    build_emu()
    port = 31500 + os.getpid() % 2000
    script = tmp_path / "ring_worker.py"
    script.write_text(_RING_WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(tmp_path / f"ring{r}.pkl")]) for r in range(2)]
    _wait_all(procs, 300)
    _check_ring([pickle.load(open(tmp_path / f"ring{r}.pkl", "rb")) for r in range(2)], 2)


# ---- the whole dynamical core, one PROCESS per tile over torch.distributed ----
_DYCORE_WORKER = r"""
import os, sys, pickle
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch
torch.set_num_threads(1)
import torch.distributed as dist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=6)
from pace_amd import _lib
from pace_amd.util import TorchDistComm
import helpers
lib = _lib.Library(os.path.join({root!r}, "tests", "emu", "libpace_emu.so"))
r = int(sys.argv[1])
out = helpers.run_dycore_tile(TorchDistComm(), lib, "cpu", helpers.golden(f"acoustic_c12_tile{{r}}.npz"),
                              helpers.golden(f"dycore_c12_tile{{r}}.npz"), 12, 79)
pickle.dump(out, open(sys.argv[2], "wb"))
dist.barrier(); dist.destroy_process_group()
"""


def test_dynamical_core_step_six_processes_gloo(tmp_path):
    """One whole DynamicalCore.step_dynamics with one process per tile over torch.distributed (gloo here; the identical code
    path runs over RCCL with one process per GPU): every halo-update group of the acoustic loop, the tracer advection, the
    omega update and CubedToLatLon goes through TorchDistComm.  Checked against the run of the reference."""
    import pickle

    from helpers import check_dycore, golden

    build_emu()
    port = 33500 + os.getpid() % 2000
    script = tmp_path / "dycore_worker.py"
    script.write_text(_DYCORE_WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(tmp_path / f"dy{r}.pkl")]) for r in range(6)]
    _wait_all(procs, 900)
    outs = [pickle.load(open(tmp_path / f"dy{r}.pkl", "rb")) for r in range(6)]
    check_dycore([golden(f"dycore_c12_tile{t}.npz") for t in range(6)], outs)


def test_bench_six_ranks_dry_run_over_gloo(tmp_path):
    """`bench.py --gpus 6` as the driver launches it (torch.distributed.run, one rank per tile), here on the CPU: gloo instead
    of RCCL and the emulation build of the kernels (--emulate).  Checks that the six-rank path runs end to end -- cubed-sphere
    partitioner, HIP-source pack / grouped exchange / unpack of delp, pt, q_con inside every step, barrier, max-over-ranks
    reduction -- and that rank 0 prints ONE JSON line of the contract's shape."""
    import json

    build_emu()
    port = 35500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "6", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "6", "--steps", "2", "--warmup", "1", "--tile-size", "12",
           "--nz", "8", "--emulate", "--watchdog", "240"]
    env = dict(os.environ, OMP_NUM_THREADS="1", PACE_TRACE_CALLS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-3000:]
    # the ORDER of a step on every rank (the entry points named on stderr): the winds' halo is packed and posted, the interior of
    # the flux preparation (phases 16) is launched BEFORE the wait / unpack, the rest of d_sw (a mask with 32) after it, then the
    # exchange of delp / pt / q_con is posted, the column solver (compute domain only) runs, and only then is it waited for
    for rank in range(6):
        calls = [ln.split("] ", 1)[1] for ln in p.stderr.splitlines() if ln.startswith(f"[pace r{rank}] ")]
        seq = [c for c in calls if c.startswith(("pace_halo_pack", "pace_halo_unpack", "pace_d_sw_phases", "pace_riem_solver3"))]
        steps = 0
        for n in range(len(seq) - 6):
            w = seq[n:n + 7]
            if (w[0] == "pace_halo_pack" and w[1] == "pace_d_sw_phases 16" and w[2] == "pace_halo_unpack" and w[3].startswith("pace_d_sw_phases ")
                    and int(w[3].split()[1]) & 32 and w[4] == "pace_halo_pack" and w[5].startswith("pace_riem_solver3") and w[6] == "pace_halo_unpack"):
                steps += 1
        assert steps >= 3, (rank, seq[:24])  # warm-up + two timed steps (+ the synchronised diagnosis pass)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config"):
        assert key in out, key
    assert out["n_gpus"] == 6 and out["steps"] == 2 and out["scaling"] == "weak" and out["dtype"] == "f64"
    assert out["config"]["tiles"] == 6 and "cubed sphere" in out["config"]["halo_exchange"]
    assert out["value"] > 0 and "EMULATION" in out["data"]
    # what a first real 6-GPU run needs for its diagnosis: the collective library's view and the per-phase times
    assert out["comm"]["world_size"] == 6 and out["comm"]["backend"] == "gloo"
    # the bytes per neighbour of the two exchanges of the step: four neighbours each; scalars: 3 fields x 3 halo rows x n x nz
    ex = out["comm"]["exchanges_per_step"]
    assert len(ex) == 2 and all(len(v) == 4 and min(v.values()) > 0 for v in ex.values()), ex
    ph = out["comm"]["phase_ms_max_over_ranks_synchronised"]
    assert set(ph) == {"uc_vc_start(pack+post)", "flux_prep_interior", "uc_vc_wait(+unpack)", "d_sw_rest", "delp_pt_qcon_start(pack+post)",
                       "delp_pt_qcon_wait(+unpack)", "riem_solver3"}
    assert all(v >= 0.0 for v in ph.values()) and ph["d_sw_rest"] > 0.0


def test_bench_full_loop_diagnosis_two_ranks_over_gloo(tmp_path):
    """`bench.py --gpus 2 --full-loop` on the CPU (gloo, emulation build): after the timed region the whole acoustic loop body
    (AcousticDynamics, all of a substep's halo exchanges over torch.distributed on the ring of tiles) runs and reports ms per
    substep and the host time of every halo updater -- what a first multi-GPU run needs to diagnose the whole pattern."""
    import json

    build_emu()
    port = 37500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--tile-size", "12",
           "--nz", "8", "--emulate", "--watchdog", "240", "--full-loop"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"), cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    fl = json.loads(lines[0])["comm"]["full_loop"]
    assert "error" not in fl, fl
    assert fl["finite"] and fl["ms_per_substep_max_over_ranks"] > 0 and "ring of 2" in fl["topology"]
    # the seven exchanges of a substep (dyn_core.py:720-942) + the three before the loop, gz on the first substep, heat_source after
    per = fl["updaters"]
    assert {"w", "divgd", "uc__vc", "delp__pt__q_con", "zh", "pkc", "u__v"} <= set(per)
    for k in ("w", "divgd", "uc__vc", "delp__pt__q_con", "zh", "pkc", "u__v"):
        assert per[k]["calls_per_substep"] == 1.0, (k, per[k])
    for k in ("q_con__cappa", "delp__pt", "gz", "heat_source"):
        assert per[k]["calls_per_substep"] == 0.25, (k, per[k])


def test_bench_gpus_flag_starts_the_ranks_itself(tmp_path):
    """`python bench.py --gpus 2` WITHOUT a launcher around it (the shape of the command the driver runs at N = 1): the flag
    starts the two ranks as a child `torch.distributed.run`, relays the one JSON line and the exit code -- it does not silently
    run one rank.  And a launcher whose rank count differs from --gpus is refused with a non-zero exit code."""
    import json

    build_emu()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--tile-size", "12", "--nz", "8",
           "--emulate", "--watchdog", "240"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["comm"]["world_size"] == 2
    # WORLD_SIZE from a launcher that disagrees with --gpus: refused before anything runs
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--emulate", "--tile-size", "12", "--nz", "8"],
                       capture_output=True, text=True, timeout=120, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), cwd=str(tmp_path))
    assert p.returncode != 0 and "--gpus 4" in p.stderr
