"""The randomised cross-checks that used to run only by hand (tools/stress_*.py,
tools/bigcheck.py), with fixed seeds, under `pytest -m gpu`.
Each test has a wall-clock budget and stops drawing new cases when it is spent; the
minimum number of cases is asserted, so a slow box fails loudly instead of testing nothing.
"""
import time

import pytest

import stress_cases as sc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    from scoary_amd.engine import AssociationEngine
    e = AssociationEngine(0)
    yield e
    e.close()


def _soak(fn, eng, cases, min_cases, budget_s):
    t0, done, bad = time.time(), 0, []
    for case in range(cases):
        ok, what = fn(eng, case)
        done += 1
        if not ok:
            bad.append("case %d: %s" % (case, what))
        if time.time() - t0 > budget_s:
            break
    assert not bad, bad
    assert done >= min_cases, "only %d cases in %.0f s" % (done, budget_s)


def test_stress_label_tiles_vs_label_rows(eng):
    _soak(sc.tiles_case, eng, cases=60, min_cases=20, budget_s=60)


def test_stress_list_kernel_vs_dense_kernel(eng):
    _soak(sc.lists_case, eng, cases=60, min_cases=20, budget_s=60)


def test_stress_device_list_builder_vs_host_builder(eng):
    _soak(sc.listbuild_case, eng, cases=80, min_cases=25, budget_s=60)


def test_stress_segmented_list_builder_positions(eng):
    _soak(sc.seglists_case, eng, cases=40, min_cases=12, budget_s=60)


def test_stress_counts_every_instance_and_mask_pattern(eng):
    _soak(sc.counts_case, eng, cases=120, min_cases=40, budget_s=60)


@pytest.mark.parametrize("shape", sc.BIG_SHAPES, ids=lambda s: "%dx%dx%d" % s[:3])
def test_big_shapes_list_kernel_vs_dense_kernel(eng, shape):
    ok, what = sc.big_case(eng, shape)
    assert ok, what
