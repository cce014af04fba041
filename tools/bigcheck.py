"""Large-shape cross-check on the GPU box (tests/stress_cases.py::big_case, also run by
tests/test_gpu_stress.py): the list-driven and the dense permutation kernels must give
identical exceedance counts for every (gene, trait) pair.

    python tools/bigcheck.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import stress_cases as sc  # noqa: E402
from scoary_amd.engine import AssociationEngine  # noqa: E402

eng = AssociationEngine(0)
bad = 0
for shape in sc.BIG_SHAPES:
    ok, what = sc.big_case(eng, shape)
    bad += not ok
    print(what, "equal:", ok)
sys.exit(1 if bad else 0)
