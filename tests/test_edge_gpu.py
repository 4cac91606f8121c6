"""GPU tier: edge cases through the C ABI on cuda:0."""
from tests import edge_cases
import pytest

pytestmark = pytest.mark.gpu


def test_giant_and_needle_gaussians(gpu):
    edge_cases.check_giant_and_needle_gaussians(gpu)


def test_invisible_opacity_and_behind_camera(gpu):
    edge_cases.check_invisible_opacity_and_behind_camera(gpu)


def test_saturating_opacity_early_termination(gpu):
    edge_cases.check_saturating_opacity_early_termination(gpu)


def test_mark_visible(gpu):
    edge_cases.check_mark_visible(gpu)


def test_python_flag_paths(gpu):
    edge_cases.check_python_flag_paths(gpu)


def test_create_from_pcd_scales(gpu):
    edge_cases.check_create_from_pcd_scales(gpu)


@pytest.mark.parametrize("n", [2500, 9000, 20000])
def test_long_tile_lists(gpu, n):
    edge_cases.check_long_tile_lists(gpu, n)


def test_multi_chunk_backward_units(gpu):
    edge_cases.check_multi_chunk_units(gpu)


@pytest.mark.parametrize("min_units", [None, 12])
def test_backward_launch_order(gpu, min_units):
    n_units, n_short, chunks = edge_cases.check_backward_launch_order(gpu, min_units=min_units)
    assert n_units > n_short > 0 and (chunks == 1 if min_units is None else chunks > 1), (n_units, n_short, chunks)


@pytest.mark.parametrize("n,longer_than", [(600, 0), (3000, 2048), (12000, 8192), (40000, 8192)])
def test_tile_lists_sorted(gpu, n, longer_than):
    assert edge_cases.check_tile_lists_sorted(gpu, n) > longer_than


@pytest.mark.parametrize("n", [850, 1100, 1600, 1900, 2200])
def test_tile_lists_sorted_as_two_runs(gpu, n):
    edge_cases.check_tile_lists_sorted(gpu, n)
    lengths = edge_cases.check_tile_lists_sorted.lengths
    assert any((512 < x <= 768) if n < 1200 else (1024 < x <= 1536) for x in lengths), lengths


@pytest.mark.parametrize("n", [400, 5000])
def test_tile_lists_are_the_oracles_minus_invisible_instances(gpu, n):
    R, R_ref, worst = edge_cases.check_tile_lists_against_oracle(gpu, n)
    print("tile lists: device R = %d, oracle (3-sigma rectangles) R = %d, largest alpha of a dropped instance = %.3e (< 1/255 = %.3e)"
          % (R, R_ref, worst, 1 / 255))
    assert R < R_ref   # the tighter rectangles do drop instances on this scene, or the test would be vacuous


def test_operator_error_behaviour(gpu, tmp_path):
    edge_cases.check_operator_error_behaviour(gpu, tmp_path)


@pytest.mark.parametrize("binding", ["compiled", "ctypes"])
def test_count_slots_survive_unpolled_forwards(gpu, binding):
    from tests.ops_util import _with_binding
    with _with_binding(binding):
        edge_cases.check_count_slots_survive_unpolled_forwards(gpu)


def test_speculative_stage2_overflow_is_rerendered(gpu):
    edge_cases.check_speculative_stage2_overflow_is_rerendered(gpu)


def test_deterministic_toggle_between_forward_and_backward_is_refused(gpu):
    edge_cases.check_deterministic_toggle_between_forward_and_backward_is_refused(gpu)
