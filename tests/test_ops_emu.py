"""CPU tier: SSIM / kNN / Adam kernels and the training loop, real kernel sources under the emulator."""
import ctypes
import os
import re

import pytest
import torch

from tests import ops_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ssim_matches_reference_golden(emu):
    ops_util.check_ssim_golden(emu)


def test_ssim_ragged_sizes(emu):
    ops_util.check_ssim_random(emu, 7, 5)     # smaller than the window
    ops_util.check_ssim_random(emu, 33, 17)   # not multiples of the 16x16 tile


def test_ssim_valid_padding(emu):
    ops_util.check_ssim_random(emu, 33, 40, padding="valid")
    ops_util.check_ssim_random(emu, 12, 11, padding="valid")   # a 2x1 valid region
    with pytest.raises(RuntimeError, match="larger than the 11x11 window"):
        ops_util.check_ssim_random(emu, 10, 40, padding="valid")
    with pytest.raises(ValueError):
        ops_util.check_ssim_random(emu, 20, 20, padding="reflect")


@pytest.mark.parametrize("n,dup", [(1, False), (3, False), (300, False), (700, True), (9000, False)])
def test_knn_matches_kdtree(emu, n, dup):
    ops_util.check_knn(emu, n, duplicates=dup)


def test_pose_activations_match_autograd(emu):
    ops_util.check_pose_activations(emu)


@pytest.mark.parametrize("degree", [0, 2])
def test_fused_render_equals_unfused(emu, degree):
    ops_util.check_fused_render_equals_unfused(emu, degree)


def test_run_ahead_equals_sync_loop(emu):
    ops_util.check_run_ahead_equals_sync_loop(emu, iters=7, Wm=12, W=32)


def test_run_ahead_ring_stays_a_leaf_with_the_loss_as_written(emu):
    ops_util.check_run_ahead_ring_stays_a_leaf_with_the_loss_as_written(emu)


def test_run_ahead_overflow_is_replayed_exactly(emu):
    ops_util.check_run_ahead_equals_sync_loop(emu, iters=7, force_overflow=True, Wm=12, W=32)


def test_pose_tracking_reduces_masked_l1(emu):
    ops_util.check_pose_tracking(emu, num_iter=3, min_gain=0.0)


def test_fused_train_step_equals_autograd_path(emu):
    ops_util.check_fused_train_step_equals_autograd_path(emu)


def test_gated_off_tensor_keeps_moving(emu):
    ops_util.check_gated_off_tensor_keeps_moving(emu)


def test_run_ahead_crosses_sh_degree_step(emu):
    ops_util.check_run_ahead_crosses_sh_degree_step(emu)


@pytest.mark.parametrize("overflow", [False, True])
def test_fused_synced_loop_equals_autograd_loop(emu, overflow):
    ops_util.check_fused_synced_loop_equals_autograd_loop(emu, force_overflow=overflow)


def test_dropin_node_housekeeping(emu):
    ops_util.check_dropin_node_housekeeping(emu)


def test_synced_one_call_loop_can_be_left_and_reentered(emu):
    ops_util.check_synced_one_call_loop_can_be_left_and_reentered(emu)


def test_commit_gate_leaves_an_overflowed_step_uncommitted(emu):
    ops_util.check_commit_gate_leaves_an_overflowed_step_uncommitted(emu)


def test_adam_matches_reference_trajectory(emu):
    ops_util.check_adam_golden(emu)


def test_adam_step_keeps_the_optimizer_contract(emu):
    """PerPointAdam.step opts out of torch.optim.Optimizer's per-call wrapper (profiler scope + hook dispatch); step hooks, the
    profiler scope, closures and load_state_dict must behave as for any torch optimizer, with the same update either way."""
    from instantsplat_amd.optim import PerPointAdam

    def fresh():
        torch.manual_seed(0)
        p = torch.nn.Parameter(torch.randn(5, 3))
        p.grad = torch.randn(5, 3)
        return p, PerPointAdam([{"params": [p], "lr": 1e-2, "name": "x"}], lr=0.0, eps=1e-15)

    p0, o0 = fresh()
    assert getattr(PerPointAdam.step, "hooked", False) and not hasattr(PerPointAdam.step, "__wrapped__")
    o0.step()
    p1, o1 = fresh()
    calls = []
    h1 = o1.register_step_pre_hook(lambda opt, args, kwargs: calls.append("pre"))
    h2 = o1.register_step_post_hook(lambda opt, args, kwargs: calls.append("post"))
    assert o1.step(lambda: torch.tensor(3.0)) == 3.0 and calls == ["pre", "post"]
    assert torch.equal(p0.detach(), p1.detach())
    h1.remove(); h2.remove()
    with torch.profiler.profile() as prof:
        o1.step()
    assert any("Optimizer.step#PerPointAdam.step" in e.key for e in prof.key_averages())
    o0.step()
    assert torch.equal(p0.detach(), p1.detach()) and calls == ["pre", "post"]
    # moments replaced by load_state_dict are the ones the next step uses
    sd = o1.state_dict()
    sd["state"][0]["exp_avg"] = torch.ones(5, 3)
    o1.load_state_dict(sd)
    before = p1.detach().clone()
    p1.grad = torch.zeros(5, 3) + 1e-3
    o1.step()
    m = o1.state[p1]["exp_avg"]
    assert torch.allclose(m, torch.full((5, 3), 0.9 + 0.1 * 1e-3)) and not torch.equal(before, p1.detach())


def test_device_guard_wraps_calls_for_tensors_off_the_current_device(monkeypatch):
    """_lib.on_device: no context for the current device (the common case costs one comparison), torch.cuda.device(dev) for
    any other one — the library launches go to the process's current device."""
    from instantsplat_amd import _lib
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    assert _lib.on_device(torch.device("cuda", 0)) is _lib._NO_GUARD
    assert _lib.on_device(torch.device("cpu")) is _lib._NO_GUARD and _lib.on_device(None) is _lib._NO_GUARD
    guard = _lib.on_device(torch.device("cuda", 1))
    assert isinstance(guard, torch.cuda.device) and guard.idx == 1


def test_two_train_iterations_match_cpu_oracle(emu):
    ops_util.check_train_matches_cpu_oracle(emu, iters=2)


def test_c_abi_library_exports_every_declared_symbol():
    """No compute: the hipcc-built library must load and export exactly what include/mi355gs.h declares."""
    import __graft_entry__ as ge
    from instantsplat_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    hdr = open(os.path.join(ROOT, "include", "mi355gs.h")).read()
    declared = set(re.findall(r"\b(mi355gs_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.mi355gs_abi_version.restype = ctypes.c_int
    assert lib.mi355gs_abi_version() == 10


def test_product_path_refuses_cpu_tensors_and_missing_library(monkeypatch):
    from instantsplat_amd import _lib
    from instantsplat_amd.simple_knn._C import distCUDA2
    _lib._use_library_for_testing(None)
    with pytest.raises(RuntimeError, match="GPU only"):
        distCUDA2(torch.zeros(4, 3))
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmi355gs.so")
    monkeypatch.setattr(_lib, "_LIB", None)
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        _lib.lib()


@pytest.mark.parametrize("degree", [1, 3])
def test_split_sh_equals_concatenated(emu, degree):
    ops_util.check_split_sh_equals_concatenated(emu, degree=degree)


@pytest.mark.parametrize("degree", [0, 1, 3])
def test_fused_step_gradients_equal_autograd(emu, degree):
    ops_util.check_fused_step_gradients_equal_autograd(emu, degree)


def test_two_one_call_train_iterations_match_cpu_oracle(emu):
    ops_util.check_train_matches_cpu_oracle(emu, iters=2, fused_step=True)


@pytest.mark.parametrize("fused_step,run", [(False, "loop"), (True, "loop"), (False, "loopb")])
def test_training_loop_matches_reference_function(emu, fused_step, run):
    ops_util.check_training_loop_matches_reference_function(emu, fused_step, run)


@pytest.mark.parametrize("run,fused_loss", [("loop", True), ("loopb", True), ("loop", "train_py")])
def test_teacher_forced_gradients_match_reference_function(emu, run, fused_loss):
    ops_util.check_teacher_forced_gradients_match_reference_function(emu, run, fused_loss)


def test_oracle_trainer_matches_reference_function(emu):
    ops_util.check_oracle_trainer_matches_reference_function(emu)


def test_pose_tracking_matches_reference_function(emu):
    ops_util.check_pose_tracking_matches_reference_function(emu)


def test_capture_matches_reference_class(emu):
    ops_util.check_capture_matches_reference_class(emu)


def test_checkpoint_save_and_resume(emu, tmp_path):
    ops_util.check_checkpoint_save_and_resume(emu, tmp_path)


def test_compiled_binding_equals_ctypes_binding(emu):
    ops_util.check_compiled_binding_equals_ctypes(emu)


def test_compiled_adam_takes_gate_flags_only_when_sound(emu):
    ops_util.check_compiled_gate_flags_are_sound(emu)


def test_operator_bindings_agree(emu):
    ops_util.check_operator_bindings_agree(emu)


def test_trainer_keeps_its_unit_length_knob(emu):
    ops_util.check_trainer_keeps_its_unit_length_knob(emu)


def test_pose_row_node(emu):
    ops_util.check_pose_row_node(emu)


def test_run_ahead_sticky_commit_gate(emu):
    ops_util.check_run_ahead_sticky_commit_gate(emu)


def test_loss_utils_against_the_references_own(emu):
    ops_util.check_loss_utils_against_the_references_own(emu)


def test_new_entry_points_reject_bad_arguments():
    """No compute (the hipcc-built library, no GPU needed: every check comes before the first HIP call): the ABI-v8 entry points
    return MI355GS_EINVAL for null pointers, non-positive sizes, a pose row outside its table, a null trainer handle."""
    import __graft_entry__ as ge
    from instantsplat_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    L = _lib._bind(_lib.LIB_PATH)
    EINVAL = -1
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.mi355gs_l1_loss_forward(None, 0, p, p, p, p) == EINVAL
    assert L.mi355gs_l1_loss_forward(None, 16, None, p, p, p) == EINVAL and L.mi355gs_l1_loss_forward(None, 16, p, p, None, p) == EINVAL
    assert L.mi355gs_l1_loss_backward(None, -3, p, p, p, p) == EINVAL and L.mi355gs_l1_loss_backward(None, 16, p, p, None, p) == EINVAL
    assert L.mi355gs_l1_scratch_bytes(0) >= 8 and L.mi355gs_l1_scratch_bytes(3 * 1080 * 1920) >= 8 * ((3 * 1080 * 1920 + 4095) // 4096)
    F7, I7 = ctypes.c_float * 7, ctypes.c_int32 * 7
    assert L.mi355gs_trainer_optimizer_step(None, None, F7(), I7(), 0.9, 0.999, 1e-15, 1) == EINVAL
    # mi355gs_posed_backward(..., d_pose, pose_rows, pose_row, ...): a row outside its table, a negative table
    def posed_backward(rows, row):
        return L.mi355gs_posed_backward(None, 0, 0, 16, 16, p, p, p, p, p, p, 1.0, p, p, p, p, p, 1.0, 1.0, p, p, p, 0, p, p, p, p, p, p, p, p, p, p,
                                        p, p, p, rows, row, 0, 0)
    assert posed_backward(-1, 0) == EINVAL and posed_backward(3, 3) == EINVAL and posed_backward(3, -1) == EINVAL
    assert L.mi355gs_error_string(EINVAL)
    # ---- ABI v9
    I = lambda *v: (ctypes.c_int32 * len(v))(*v)
    Fl = lambda *v: (ctypes.c_float * len(v))(*v)
    assert L.mi355gs_raster_forward_render_only(None, 4, 0, 16, 64, p, p, p, p, p, 0) == EINVAL          # W = 0
    assert L.mi355gs_raster_forward_render_only(None, 4, 16, 16, 64, p, p, p, None, p, 0) == EINVAL      # capacity without a buffer
    assert 0 < L.mi355gs_raster_binning_bytes_render_only(1000, 64, 64) < L.mi355gs_raster_binning_bytes(1000, 64, 64)
    assert L.mi355gs_raster_binning_bytes_render_only(1000, 64, 64) >= 12 * 1000 and L.mi355gs_raster_binning_bytes_render_only(1000, 0, 64) == 0
    assert L.mi355gs_l1_ssim_pair_forward(None, 1, 3, 0, 8, p, p, p, p) == EINVAL and L.mi355gs_l1_ssim_pair_forward(None, 1, 3, 8, 8, p, p, None, p) == EINVAL
    assert L.mi355gs_l1_ssim_pair_backward(None, 0, p, p, p, p, 1.0, p, 1.0, p) == EINVAL
    assert L.mi355gs_l1_ssim_pair_backward(None, 16, p, p, None, p, 1.0, p, 1.0, p) == EINVAL            # an SSIM term without its gradient map
    prog = lambda ops: L.mi355gs_loss_program_eval(None, len(ops), I(*ops), Fl(*([0.5] * len(ops))), 1, 3, 8, 8, p, p, p, p, None, 0.0)
    assert prog([2]) == EINVAL            # MULK on an empty stack
    assert prog([0, 1]) == EINVAL         # two values left
    assert prog([0, 7]) == EINVAL         # ADD with one operand
    assert prog([0, 9]) == EINVAL         # unknown operation
    assert L.mi355gs_loss_program_eval(None, 17, I(*([0] * 17)), Fl(*([0.0] * 17)), 1, 3, 8, 8, p, p, p, p, None, 0.0) == EINVAL
    assert L.mi355gs_loss_program_eval(None, 1, I(0), Fl(0.0), 1, 3, 8, 8, None, p, p, p, None, 0.0) == EINVAL
    assert L.mi355gs_loss_program_eval(None, 1, I(0), Fl(0.0), 1, 3, 8, 8, p, p, p, p, ctypes.c_void_p(p.value + 4), 0.0) == EINVAL   # a host slot that is not 8-byte aligned
    eg = lambda ops, d=p, ds=p, cs=1.0: L.mi355gs_loss_program_eval_grad(None, len(ops), I(*ops), Fl(*([0.5] * len(ops))), 1, 3, 8, 8, p, p, p, p, None, 0.0,
                                                                           p, p, ds, 1.0, cs, d)
    assert eg([0, 1]) == EINVAL and eg([0], d=None) == EINVAL and eg([0], ds=None) == EINVAL   # bad program, no output, an SSIM term without its map
    for knob in (L.mi355gs_tune_scale_grad, L.mi355gs_tune_deterministic):   # query-only calls change nothing
        assert knob(-1) == 0 and knob(1) == 0 and knob(-1) == 1 and knob(0) == 1 and knob(-1) == 0
    assert L.mi355gs_raster_grad_scratch_bytes(1000) == L.mi355gs_raster_grad_gate_offset(1000) + 256
    L.mi355gs_tune_deterministic(1)
    try:   # the deterministic mode enters the size queries
        assert L.mi355gs_raster_grad_scratch_bytes(1000) >= L.mi355gs_raster_grad_gate_offset(1000) + 256 + 4 * 1001
        det_bytes = L.mi355gs_raster_binning_bytes(1000, 64, 64)
    finally:
        L.mi355gs_tune_deterministic(0)
    assert det_bytes >= L.mi355gs_raster_binning_bytes(1000, 64, 64) + 52 * 1000


def test_roctx_ranges_bracket_every_stage_of_a_train_step(emu):
    """ABI v10: with mi355gs_profile_ranges(1) every launching entry point pushes a roctx range
    named after itself — one one-call train step = mi355gs_trainer_step with the projection / binning, composite forward, loss,
    backward and optimizer entry points nested inside it — and with (0) nothing is pushed.  The marker library is opened at run
    time; a box without one (no ROCm install) refuses to switch on and everything else works."""
    from instantsplat_amd import _lib
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import release_trainer, setup_training, train_iteration
    L = _lib.lib()
    st = setup_training(syn_pointmap(3, 8, 8, 24, 24, seed=2), emu, opt=OptimizationParams(iterations=100, pp_optimizer=True, optim_pose=True))
    train_iteration(st, fused_step=True)        # (creates the trainer handle: its exact-count renders are not the step's launches)
    before = L.mi355gs_profile_ranges(-1)
    train_iteration(st, fused_step=True)
    assert L.mi355gs_profile_ranges(-1) == before                  # off: not a single push
    rc = L.mi355gs_profile_ranges(1)
    if rc < 0:
        release_trainer(st)
        pytest.skip("no roctx library can be opened on this box")
    try:
        train_iteration(st, fused_step=True)
        # trainer_step { forward_preprocess, forward_render, l1_ssim_loss_fused, raster_backward } + trainer_optimizer_step { adam_multi_step }
        assert L.mi355gs_profile_ranges(-1) - before >= 6 + 10, L.mi355gs_profile_ranges(-1) - before   # 6 entry points + 10 launch sites (scatter + sort share one)
    finally:
        after = L.mi355gs_profile_ranges(0)
        release_trainer(st)
    train_iteration(st, fused_loss=False)
    assert L.mi355gs_profile_ranges(-1) == after


def test_render_only_forward_is_bit_identical(emu):
    ops_util.check_render_only_forward(emu)


def test_lazy_loss_expression(emu):
    ops_util.check_lazy_loss_expression(emu)


def test_lazy_scalar_behaves_like_a_tensor(emu):
    ops_util.check_lazy_scalar_behaves_like_a_tensor(emu)


def test_late_item_of_an_old_loss(emu):
    ops_util.check_late_item_of_an_old_loss(emu)


def test_deterministic_backward(emu):
    ops_util.check_deterministic_backward(emu, iters=3)


def test_deterministic_backward_multi_chunk_units(emu):
    ops_util.check_deterministic_backward(emu, iters=2, min_units=4)
