"""CPU tier: edge cases on the emulated kernels."""
import pytest

from tests import edge_cases


def test_giant_and_needle_gaussians(emu):
    edge_cases.check_giant_and_needle_gaussians(emu)


def test_invisible_opacity_and_behind_camera(emu):
    edge_cases.check_invisible_opacity_and_behind_camera(emu)


def test_saturating_opacity_early_termination(emu):
    edge_cases.check_saturating_opacity_early_termination(emu)


def test_mark_visible(emu):
    edge_cases.check_mark_visible(emu)


def test_python_flag_paths(emu):
    edge_cases.check_python_flag_paths(emu)


def test_create_from_pcd_scales(emu):
    edge_cases.check_create_from_pcd_scales(emu)


@pytest.mark.parametrize("n", [2500, 8500, 12000])
def test_long_tile_lists(emu, n):
    edge_cases.check_long_tile_lists(emu, n)


@pytest.mark.parametrize("n,longer_than", [(600, 0), (3000, 2048), (12000, 8192)])
def test_tile_lists_sorted(emu, n, longer_than):
    assert edge_cases.check_tile_lists_sorted(emu, n) > longer_than


def test_operator_error_behaviour(emu, tmp_path):
    edge_cases.check_operator_error_behaviour(emu, tmp_path)
