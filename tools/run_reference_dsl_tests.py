"""Run the reference's OWN data-free stencil tests (tests/main/dsl/test_stencil_factory.py, test_stencil_wrapper.py,
test_stencil.py, test_stencil_config.py, test_compilation_config.py, tests/main/fv3core/test_selective_validation.py) with tools/gtinterp standing in for GT4Py (tools/refshim).

These tests hold literal expected values for what a stencil writes given an origin / domain (write windows of API fields),
for `horizontal(region[i_start, :])` under get_stencils_with_varied_bounds (axis offsets relative to the stencil origin),
in-place updates through a temporary, positional / keyword / parameter arguments and the FrozenStencil call contract.  They
are the reference-held check of the interpreter that produced the fixtures under tests/golden/ (dev container only:
needs /root/reference).

    python tools/run_reference_dsl_tests.py [-k expr]      (result of the last run: tools/reference_dsl_tests.log)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()
import pytest  # noqa: E402

REF = refshim.REFERENCE_ROOT
# stencil execution (literal expected values) + the configuration / naming contracts the host mirror follows; the DaCe tests
# (test_dace_config.py) and the GT4Py compiler-pass test (test_skip_passes.py) have no meaning under an interpreter
FILES = ["dsl/test_stencil_factory.py", "dsl/test_stencil_wrapper.py", "dsl/test_stencil.py", "dsl/test_stencil_config.py",
         "dsl/test_compilation_config.py", "fv3core/test_selective_validation.py"]

if __name__ == "__main__":
    args = [os.path.join(REF, "tests", "main", f) for f in FILES]
    args += ([] if "-v" in sys.argv else ["-q"]) + ["-p", "no:cacheprovider", "--rootdir", os.path.join(REF, "tests", "main"), "-c", os.devnull,
             "--confcutdir", os.path.join(REF, "tests", "main")] + sys.argv[1:]
    sys.exit(pytest.main(args))
