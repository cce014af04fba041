"""Longer soak of tests/stress_cases.py::seglists_case on a GPU box (the first cases are what
`pytest -m gpu` runs as tests/test_gpu_stress.py).

    python tools/stress_seglists.py [cases] [first case]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import stress_cases as sc  # noqa: E402
from scoary_amd.engine import AssociationEngine  # noqa: E402

eng = AssociationEngine(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # cases are seeded by their number: a later range = new cases
bad = 0
for case in range(first, first + cases):
    ok, what = sc.seglists_case(eng, case)
    bad += not ok
    print(case, what, "ok" if ok else "MISMATCH")
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
