"""Out-of-bounds accesses of the kernels, caught on the CPU: the emulated kernels run in a child process in which every array
the host layer allocates ends at (mode "over") or starts right after (mode "under") an inaccessible page (tests/guard.py).
A kernel that reads or writes outside an array it was handed dies there, and the emulator names it (PACE_EMU_GUARD=1).
GPU sanitizers are not available on the pool; on the device such an access is a memory fault only when the array happens to
end at the end of an allocator segment (found that way in round 2: a metric row of a tile that sticks out of the storage)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PRELUDE = f"""
import sys
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import guard, helpers
from pace_amd import _lib
lib = _lib.Library(helpers.build_emu())
"""


def _run(body, mode):
    env = dict(os.environ, PACE_EMU_GUARD="1")
    code = _PRELUDE + f"with guard.guarded({mode!r}):\n" + "".join("    " + line + "\n" for line in body.strip().splitlines())
    p = subprocess.run([sys.executable, "-X", "faulthandler", "-c", code], capture_output=True, text=True, timeout=1200, env=env)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[:1500], p.stderr[-1500:])


@pytest.mark.parametrize("mode", ["over", "under"])
def test_operator_chain_with_guard_pages(mode):
    """All 15 operators of the acoustic loop + the stand-alone PPM / divergence-damping classes at C40 x 8: a size at which
    the 32 x 24 tiles of the transport family stick out of the domain by part of a tile (C12: by more than a tile, covered
    below; the 4 x 4-tile build covers tiles that divide the domain)."""
    _run("""
from opchain import Chain, ProductOps, check_case, check_standalone_operators
chain = Chain(40, 8)
ops = ProductOps(lib, "cpu", chain)
names = [case.name for case in chain.cases() if check_case(ops, case) is not None]
assert len(names) == 15
check_standalone_operators(lib, "cpu", 40, 8, exact=True)
""", mode)


def test_dynamical_core_step_with_guard_pages():
    """One whole DynamicalCore.step_dynamics on six C12 tiles (acoustic loop, tracer advection, remapping, c2l, all halo
    updates) with the grid from pace_amd.util.gridgen and the reference run's initial state: no access outside any array, and
    the reference run's output at the usual tolerances."""
    _run("""
fixes, outs = helpers.run_dycore_six_tiles(lib, "cpu", generated="metrics")
helpers.check_dycore(fixes, outs)
""", "over")
