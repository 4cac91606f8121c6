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


def test_multi_chunk_backward_units(emu):
    edge_cases.check_multi_chunk_units(emu)


@pytest.mark.parametrize("min_units", [None, 12])
def test_backward_launch_order(emu, min_units):
    n_units, n_short, chunks = edge_cases.check_backward_launch_order(emu, min_units=min_units)
    assert n_units > n_short > 0 and (chunks == 1 if min_units is None else chunks > 1), (n_units, n_short, chunks)


@pytest.mark.parametrize("n,longer_than", [(600, 0), (3000, 2048), (12000, 8192)])
def test_tile_lists_sorted(emu, n, longer_than):
    assert edge_cases.check_tile_lists_sorted(emu, n) > longer_than


def test_operator_error_behaviour(emu, tmp_path):
    edge_cases.check_operator_error_behaviour(emu, tmp_path)


def test_gaussian_model_matches_reference_class(emu):
    """scene.GaussianModel vs vectors produced by the reference's OWN GaussianModel (tests/golden/make_golden.py executes
    reference scene/gaussian_model.py with its CUDA device strings rewritten): create_from_pcd layouts and values, pose
    parameters, activations, packed covariance, optimiser group order / learning rates / schedule, SH degree step."""
    import os
    import numpy as np
    import torch
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.camera import Camera
    from instantsplat_amd.scene import GaussianModel
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))
    T = lambda k: torch.from_numpy(G[k])
    g = GaussianModel(3)
    g.create_from_pcd(T("gm_points"), T("gm_colors"), 2.5, emu)
    for name in ("_xyz", "_features_dc", "_features_rest", "_rotation", "_opacity"):
        assert torch.equal(getattr(g, name).detach(), T("gm" + name)), name
    # scales come from the 3-NN kernel (fp32) on one side and the float64 k-d tree on the other
    assert torch.allclose(g._scaling.detach(), T("gm_scaling"), rtol=0, atol=2e-6)
    cams = []
    for k in range(3):
        w2c = torch.eye(4, dtype=torch.float64)
        w2c[:3, :3] = T("gm_cam_R")[k].t()          # getWorld2View2: rotation stored transposed (reference utils/graphics_utils.py:38-49)
        w2c[:3, 3] = T("gm_cam_t")[k]
        cams.append(Camera(k, w2c, 1.0, 0.8, 64, 48))
    g.init_RT_seq(cams, emu)
    assert torch.allclose(g.P.detach(), T("gm_P"), rtol=0, atol=1e-6)
    g._scaling.data.copy_(T("gm_scaling"))          # compare the activations on identical raw values
    assert torch.equal(g.get_scaling.detach(), T("gm_get_scaling")) and torch.equal(g.get_opacity.detach(), T("gm_get_opacity"))
    assert torch.equal(g.get_features.detach(), T("gm_get_features")) and torch.equal(g.get_rotation.detach(), T("gm_get_rotation"))
    assert torch.allclose(g.get_covariance(1.3).detach(), T("gm_get_covariance"), rtol=1e-5, atol=1e-9)
    g.training_setup_pp(OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True), T("gm_conf_lr"))
    assert [grp["name"] for grp in g.optimizer.param_groups] == list(G["gm_group_names"])
    assert g.optimizer.param_groups[0]["per_point_lr"] is g.per_point_lr
    for it, row in zip(G["gm_lr_iterations"], G["gm_group_lrs"]):
        g.update_learning_rate(int(it))
        ours = np.array([grp["lr"] for grp in g.optimizer.param_groups], dtype=np.float64)
        assert np.allclose(ours, row, rtol=1e-12, atol=0), (it, ours, row)
    g.oneupSHdegree()
    assert g.active_sh_degree == int(G["gm_sh_degree_after_oneup"]) == 1


def test_render_glue_hands_the_operator_what_the_reference_does(emu, monkeypatch):
    """gaussian_renderer.render() vs the arguments the reference's OWN render() passes to the rasterizer operator
    (recorded by tests/golden/make_golden.py with a stand-in operator; reference gaussian_renderer/__init__.py:50-135):
    identity view / zero camera position, camera-frame means, raw Hamilton product of the rotations, activations, SH
    features, and the cov3D / colour tensors of the two python-flag variants.  The default variant runs through the fused
    pose kernel (emulated) and through the op-by-op glue."""
    import os
    import numpy as np
    import torch
    import instantsplat_amd.gaussian_renderer as gr
    from instantsplat_amd.arguments import PipelineParams
    from instantsplat_amd.scene import GaussianModel
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))
    T = lambda k: torch.from_numpy(G[k])
    rec = {}

    class Recording:
        def __init__(self, raster_settings):
            self.s = raster_settings

        def __call__(self, **kw):
            rec.clear()
            rec.update(kw)
            rec["settings"] = self.s
            return torch.zeros(3, self.s.image_height, self.s.image_width), torch.ones(kw["means3D"].shape[0], dtype=torch.int32)

    monkeypatch.setattr(gr, "GaussianRasterizer", Recording)
    g = GaussianModel(3)
    for k in ("_xyz", "_rotation", "_scaling", "_opacity", "_features_dc", "_features_rest"):
        setattr(g, k, torch.nn.Parameter(T("render_in" + k).clone()))
    g.active_sh_degree = 2

    class Cam:
        FoVx, FoVy, image_height, image_width = 1.0, 0.8, 48, 64
        projection_matrix = T("render_default_projmatrix")   # identity view: the settings' projmatrix IS the camera's
        camera_center = T("render_in_camera_center")

    pose, bg = T("render_in_pose"), T("render_in_bg")
    close = lambda a, b: torch.allclose(a.detach().float().cpu(), b, rtol=2e-5, atol=2e-6)
    for tag, cov, shp, fused in (("default", False, False, True), ("default", False, False, False),
                                 ("cov_python", True, False, False), ("sh_python", False, True, False)):
        monkeypatch.setattr(gr, "FUSED_GLUE", fused)
        out = gr.render(Cam, g, PipelineParams(convert_SHs_python=shp, compute_cov3D_python=cov), bg, scaling_modifier=1.2,
                        camera_pose=pose)
        assert set(out) == {"render", "viewspace_points", "visibility_filter", "radii"}
        s = rec["settings"]
        ours = np.array([s.image_height, s.image_width, s.tanfovx, s.tanfovy, s.scale_modifier, s.sh_degree,
                         float(s.prefiltered), float(s.debug)], dtype=np.float64)
        assert np.allclose(ours, G[f"render_{tag}_settings_scalars"], rtol=1e-12), (tag, ours)
        assert torch.equal(s.viewmatrix.cpu(), T(f"render_{tag}_viewmatrix")) and torch.equal(s.campos.cpu(), T(f"render_{tag}_campos"))
        assert torch.equal(s.projmatrix.cpu(), T(f"render_{tag}_projmatrix")) and torch.equal(s.bg.cpu(), T(f"render_{tag}_bg"))
        assert close(rec["means3D"], T(f"render_{tag}_means3D")), tag
        assert close(rec["opacities"], T(f"render_{tag}_opacities")), tag
        assert tuple(rec["means2D"].shape) == tuple(G[f"render_{tag}_means2D"].shape) and float(rec["means2D"].detach().abs().max()) == 0
        for k in ("scales", "rotations", "cov3D_precomp", "colors_precomp"):
            ref = T(f"render_{tag}_{k}")
            if ref.numel() == 0:
                assert rec.get(k) is None, (tag, k)
            else:
                assert close(rec[k], ref), (tag, k)
        ref_shs = T(f"render_{tag}_shs")
        if ref_shs.numel() == 0:
            assert rec.get("shs") is None
        else:
            shs = rec["shs"] if rec.get("shs_rest") is None else torch.cat((rec["shs"], rec["shs_rest"]), dim=1)
            assert torch.equal(shs.detach().cpu(), ref_shs), tag


def test_render_glue_gradients_match_reference_autograd(emu, monkeypatch):
    """Backward of the render() glue: with a stand-in operator whose image is linear in its inputs (fixed coefficients), the
    reference's own render() + autograd gave d(image.sum())/d(raw parameters, camera pose) (make_golden.py).  Ours must
    reproduce them through the fused pose kernels (k_pose_fwd / k_pose_bwd, emulated) and through the op-by-op glue."""
    import os
    import numpy as np
    import torch
    import instantsplat_amd.gaussian_renderer as gr
    from instantsplat_amd.arguments import PipelineParams
    from instantsplat_amd.scene import GaussianModel
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))
    T = lambda k: torch.from_numpy(G[k])
    coef = {k: T("glue_coef_" + k) for k in ("means3D", "rotations", "scales", "opacities")}

    class Linear:
        def __init__(self, raster_settings):
            self.s = raster_settings

        def __call__(self, **kw):
            tot = sum((kw[k] * c).sum() for k, c in coef.items())
            n = 3 * self.s.image_height * self.s.image_width
            return tot.expand(3, self.s.image_height, self.s.image_width) / n, torch.ones(kw["means3D"].shape[0], dtype=torch.int32)

    monkeypatch.setattr(gr, "GaussianRasterizer", Linear)

    class Cam:
        FoVx, FoVy, image_height, image_width = 1.0, 0.8, 48, 64
        projection_matrix = T("render_default_projmatrix")
        camera_center = T("render_in_camera_center")

    for fused in (True, False):
        monkeypatch.setattr(gr, "FUSED_GLUE", fused)
        g = GaussianModel(3)
        for k in ("_xyz", "_rotation", "_scaling", "_opacity", "_features_dc", "_features_rest"):
            setattr(g, k, torch.nn.Parameter(T("render_in" + k).clone()))
        g.active_sh_degree = 2
        pose = T("render_in_pose").clone().requires_grad_(True)
        out = gr.render(Cam, g, PipelineParams(), T("render_in_bg"), scaling_modifier=1.0, camera_pose=pose)
        out["render"].sum().backward()
        for name in ("_xyz", "_rotation", "_scaling", "_opacity"):
            a, b = getattr(g, name).grad, T("glue_grad" + name)
            assert float((a - b).norm() / b.norm()) <= 2e-5, (fused, name, float((a - b).norm() / b.norm()))
        a, b = pose.grad, T("glue_grad_pose")
        assert float((a - b).norm() / b.norm()) <= 5e-5, (fused, "pose", a, b)


def test_tile_lists_are_the_oracles_minus_invisible_instances(emu):
    R, R_ref, worst = edge_cases.check_tile_lists_against_oracle(emu, 400)
    assert R < R_ref and worst < 1 / 255


def test_pending_frames_are_verified_with_their_own_count(emu):
    edge_cases.check_count_slots_survive_unpolled_forwards(emu)   # (CPU: every frame has a word of its own; the bookkeeping is what runs)


def test_speculative_stage2_overflow_is_rerendered(emu):
    edge_cases.check_speculative_stage2_overflow_is_rerendered(emu)


def test_more_tiles_than_scan_threads(emu):
    """A frame with more tiles than the tile scan has threads (1140 > 1024, two tiles per thread with the last threads short):
    the staged form of k_scan_tiles — counts fetched once into LDS, every pass reads them there.  Image, radii and every gradient
    against the C oracle, and the per-tile lists exactly."""
    from tests.util import assert_raster_parity, run_blob_case
    assert_raster_parity(run_blob_case(emu, 700, 608, 480, 1, scale_mean=0.08))
    edge_cases.check_tile_lists_against_oracle(emu, 500, W=608, H=480)


def test_scatter_takes_the_count_kernels_tile_list_or_recounts(emu):
    """The count kernel hands each 512-Gaussian workgroup's touched tiles to the scatter kernel (up to 1024 of them); a workgroup
    that touches more says so and the scatter recounts.  Both ways against the oracle's per-tile lists, on a 1140-tile frame:
    small Gaussians (listed), and the same ones much larger (one workgroup over almost every tile: recount)."""
    n_small, n_big = [], []
    edge_cases.check_tile_lists_against_oracle(emu, 500, W=608, H=480, scale_boost=0.25, entry_counts=n_small)
    edge_cases.check_tile_lists_against_oracle(emu, 500, W=608, H=480, scale_boost=2.0, entry_counts=n_big)
    assert len(n_small) == 1 and 0 < n_small[0] <= 1024, n_small
    assert n_big == [0xffffffff], n_big


@pytest.mark.parametrize("n", [850, 1100, 1600, 1900, 2200])
def test_tile_lists_sorted_as_two_runs(emu, n):
    """Tiles of 513 ... 768 and 1025 ... 1536 keys: two sorted runs (512 + up to 256, 1024 + up to 256 / 512 keys) merged by rank
    (binning.hip sort_tile_two_runs)."""
    edge_cases.check_tile_lists_sorted(emu, n)
    lengths = edge_cases.check_tile_lists_sorted.lengths
    assert any((512 < x <= 768) if n < 1200 else (1024 < x <= 1536) for x in lengths), lengths
    if n == 1900:
        assert any(1024 < x <= 1280 for x in lengths) and any(1280 < x <= 1536 for x in lengths), lengths


def test_deterministic_toggle_between_forward_and_backward_is_refused(emu):
    edge_cases.check_deterministic_toggle_between_forward_and_backward_is_refused(emu)
