"""GPU tier, run last: the HIP path against trajectories produced by the reference's OWN `training()` and
`render_set_optimize()` (tests/golden/make_golden.py executes them around the C oracle as the rasterizer operator).
The emulator-tier twins of these tests are in tests/test_ops_emu.py."""
import pytest

from tests import ops_util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused_step", [False, True])
def test_training_loop_matches_reference_function(gpu, fused_step):
    ops_util.check_training_loop_matches_reference_function(gpu, fused_step)


def test_training_loop_plain_adam_fixed_poses_matches_reference_function(gpu):
    ops_util.check_training_loop_matches_reference_function(gpu, False, run="loopb")


@pytest.mark.parametrize("run,fused_loss", [("loop", True), ("loop", False), ("loop", "train_py"), ("loopb", True)])
def test_teacher_forced_gradients_match_reference_function(gpu, run, fused_loss):
    ops_util.check_teacher_forced_gradients_match_reference_function(gpu, run, fused_loss)


def test_pose_tracking_matches_reference_function(gpu):
    ops_util.check_pose_tracking_matches_reference_function(gpu)


def test_capture_matches_reference_class(gpu):
    ops_util.check_capture_matches_reference_class(gpu)


def test_checkpoint_save_and_resume(gpu, tmp_path):
    ops_util.check_checkpoint_save_and_resume(gpu, tmp_path)
