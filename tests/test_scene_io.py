"""CPU tier: a trainable scene from the reference's on-disk init layout.

tests/golden/init_scene/ was written by the reference's OWN writers (utils/sfm_utils.py save_extrinsic / save_intrinsics /
save_points3D / storePly) and tests/golden/initdir_vectors.npz holds what the reference's OWN readColmapSceneInfo /
getNerfppNorm / Scene.__init__ / loadCam / GaussianModel / training() made of it (tests/golden/make_golden_initdir.py, run in
the build container).  instantsplat_amd.scene_io + train.training() must reproduce all of it from the same files."""
import json
import os
import random

import numpy as np
import pytest
import torch

from instantsplat_amd import io_formats as iof
from instantsplat_amd import scene_io

HERE = os.path.dirname(os.path.abspath(__file__))
SCENE = os.path.join(HERE, "golden", "init_scene")
G = np.load(os.path.join(HERE, "golden", "initdir_vectors.npz"))
V, WM, IW, IH, ITERS = [int(x) for x in G["initdir_config"]]


def test_scene_info_matches_reference_reader():
    info = scene_io.read_colmap_scene_info(SCENE, None, False, V)
    cams = info.train_cameras
    assert [c.image_name for c in cams] == list(G["initdir_info_names"])            # sorted by image name, not by COLMAP id
    assert [c.uid for c in cams] == list(G["initdir_info_uid"])
    assert np.array_equal(np.stack([c.R for c in cams]), G["initdir_info_R"])         # qvec2rotmat(q)^T, bit for bit
    assert np.array_equal(np.stack([c.T for c in cams]), G["initdir_info_T"])
    assert np.array_equal(np.array([[c.FovX, c.FovY] for c in cams]), G["initdir_info_fov"])
    assert np.array_equal(np.array([[c.width, c.height] for c in cams]), G["initdir_info_wh"])
    assert info.nerf_normalization["radius"] == float(G["initdir_info_radius"])
    assert np.array_equal(info.nerf_normalization["translate"], G["initdir_info_translate"])
    assert np.array_equal(info.points, G["initdir_info_points"]) and np.array_equal(info.colors, G["initdir_info_colors"])
    assert np.array_equal(np.stack(info.train_poses), G["initdir_info_poses"]) and info.test_cameras == [] and info.test_poses == []


def test_eval_camera_order_matches_reference_scene(tmp_path):
    """--eval: the reader hands back ONE list as train and test cameras, the reference's Scene shuffles it twice and builds both
    camera lists from that order (tests/golden/make_golden_eval_order.py ran its Scene.__init__ on this directory)."""
    import shutil
    E = np.load(os.path.join(HERE, "golden", "eval_order_vectors.npz"))
    src = tmp_path / "scene"
    shutil.copytree(SCENE, src)
    os.makedirs(src / f"sparse_{V}" / "1")
    for f in ("cameras.txt", "images.txt"):
        shutil.copyfile(src / f"sparse_{V}" / "0" / f, src / f"sparse_{V}" / "1" / f)
    sc = scene_io.load_init_scene(str(src), V, eval=True, resolution=2, device="cpu")
    for kind, cams in (("train", sc.cameras), ("test", sc.test_cameras)):
        assert [c.image_name for c in cams] == list(E[f"eval_{kind}_names"]), kind
        assert [[c.uid, c.colmap_id] for c in cams] == E[f"eval_{kind}_uid_colmap"].tolist(), kind
    assert [sc.rng.randint(0, 10 ** 6) for _ in range(4)] == list(E["eval_rng_next"])
    assert list(E["eval_train_names"]) != list(G["initdir_cam_r2_names"])   # (the twice-shuffled order is not the once-shuffled one)


@pytest.mark.parametrize("res", [1, 2])
def test_cameras_match_reference_scene(tmp_path, res):
    sc = scene_io.load_init_scene(SCENE, V, resolution=res, device="cpu", model_path=str(tmp_path))
    p = f"initdir_cam_r{res}_"
    cams = sc.cameras
    assert [c.image_name for c in cams] == list(G[p + "names"])                        # the seeded shuffle (random.seed(0))
    assert [[c.uid, c.colmap_id] for c in cams] == G[p + "uid_colmap"].tolist()
    assert [[c.image_width, c.image_height] for c in cams] == G[p + "wh"].tolist()     # from the IMAGE at -r <res>
    assert np.array_equal(np.array([[c.FoVx, c.FoVy] for c in cams]), G[p + "fov"])    # from cameras.txt
    for k in ("world_view_transform", "projection_matrix", "camera_center", "original_image"):
        a, b = np.stack([getattr(c, k).numpy() for c in cams]), G[p + k]
        if k == "camera_center":   # an inverse of a 4x4 in float32: last-bit differences between two LAPACK call layouts
            assert np.allclose(a, b, rtol=0, atol=1e-6)
            continue
        assert a.dtype == b.dtype and np.array_equal(a, b), k                          # PIL resize + /255, bit for bit
    assert sc.cameras_extent == float(G["initdir_cameras_extent"])
    assert [sc.rng.randint(0, 10 ** 6) for _ in range(4)] == list(G["initdir_rng_next"])   # the stream the view sampling continues on
    # what Scene.__init__ leaves in the model directory
    assert open(tmp_path / "input.ply", "rb").read() == open(os.path.join(SCENE, f"sparse_{V}", "0", "points3D.ply"), "rb").read()
    ours, ref = json.load(open(tmp_path / "cameras.json")), json.loads(str(G["initdir_cameras_json"]))
    assert len(ours) == len(ref)
    for a, b in zip(ours, ref):
        assert a.keys() == b.keys() and a["id"] == b["id"] and a["img_name"] == b["img_name"] and (a["width"], a["height"]) == (b["width"], b["height"])
        for k in ("position", "rotation", "fx", "fy"):
            assert np.allclose(np.array(a[k]), np.array(b[k]), rtol=1e-13, atol=1e-15), k


def test_resolution_rules():
    """reference utils/camera_utils.py:24-42"""
    assert scene_io.image_resolution(1000, 750, 1) == (1000, 750) and scene_io.image_resolution(1001, 751, 2) == (round(1001 / 2), round(751 / 2))
    assert scene_io.image_resolution(1000, 750, -1) == (1000, 750) and scene_io.image_resolution(3200, 2400, -1) == (1600, 1200)
    assert scene_io.image_resolution(1000, 750, 500) == (500, 375) and scene_io.image_resolution(1000, 750, 8, 2.0) == (62, 47)


def test_missing_scene_and_missing_confidence(tmp_path):
    with pytest.raises(FileNotFoundError):
        scene_io.load_init_scene(str(tmp_path), 3, device="cpu")
    assert scene_io.load_confidence_lr(str(tmp_path), 3, "cpu") is None


def test_gaussians_from_directory_match_reference_scene(emu):
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.train import setup_training
    for view_depth in (False, True):
        sc = scene_io.load_init_scene(SCENE, V, resolution=2, device="cpu", init_scale_from_view_depth=view_depth)
        st = setup_training(sc, "cpu", opt=OptimizationParams(iterations=ITERS, pp_optimizer=True, optim_pose=True))
        g = st.gaussians
        if view_depth:
            assert np.allclose(g._scaling.detach().numpy(), G["initdir_gm_scaling_view_depth"], rtol=0, atol=2e-6)
            continue
        for n in ("_xyz", "_features_dc", "_rotation", "_opacity", "P"):
            assert np.array_equal(getattr(g, n).detach().numpy(), G["initdir_gm" + (n if n.startswith("_") else "_" + n)]), n
        assert np.allclose(g._scaling.detach().numpy(), G["initdir_gm_scaling"], rtol=0, atol=2e-6)   # 3-NN: fp32 grid vs float64 k-d tree
        assert g.spatial_lr_scale == float(G["initdir_cameras_extent"])
        assert [tuple(x.shape) for x in st.gt_images] == [(3, IH // 2, IW // 2)] * V and st.rng is sc.rng
        conf = np.load(os.path.join(SCENE, f"sparse_{V}", "0", "confidence_dsp.npy"))
        assert np.allclose(g.per_point_lr.numpy(), (1.0 - 1.0 / (1.0 + np.exp(-conf))) * 99.0 + 1.0, rtol=1e-5)


@pytest.mark.parametrize("loop", ["reference_shape", "one_call", "run_ahead"])
def test_training_from_directory_matches_reference_training(emu, tmp_path, loop):
    """training(<source_path>) end to end vs the reference's own training() on the same directory (real Scene, real save_pose,
    real scene.save): view order, losses, final parameters, every output file."""
    from instantsplat_amd.train import train_iteration, training

    def generic_start(st):   # the recorded run's start (make_golden_initdir.py::_SceneFromDisk)
        with torch.no_grad():
            st.gaussians._scaling.add_(torch.from_numpy(G["initdir_init_scaling_delta"]))
            st.gaussians._rotation.copy_(torch.from_numpy(G["initdir_init_rotation"]))

    losses, uids = [], []
    if loop == "run_ahead":
        r = training(SCENE, "cpu", iterations=ITERS, n_views=V, resolution=2, model_path=str(tmp_path), saving_iterations=[ITERS], after_setup=generic_start)
    else:
        from instantsplat_amd import train as T
        orig = T.train_iteration

        def tracked(st, **kw):
            out = orig(st, fused_step=(loop == "one_call"), **{k: v for k, v in kw.items() if k != "fused_step"})
            losses.append(float(out))
            return out
        T.train_iteration = tracked
        try:
            r = training(SCENE, "cpu", iterations=ITERS, n_views=V, resolution=2, model_path=str(tmp_path), saving_iterations=[ITERS], run_ahead=False,
                         after_setup=generic_start)
        finally:
            T.train_iteration = orig
        assert np.allclose(losses, G["initdir_loop_losses"], rtol=1e-3, atol=0), (losses, G["initdir_loop_losses"])
    g = r["state"].gaussians
    for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P"):
        a, b = getattr(g, n).detach(), torch.from_numpy(G["initdir_loop_final" + (n if n.startswith("_") else "_" + n)])
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        assert rel <= 2e-5, (n, rel)
    assert [g.optimizer.state[grp["params"][0]]["step"] for grp in g.optimizer.param_groups] == list(G["initdir_loop_final_steps"])
    # the files
    outs = sorted(os.path.relpath(os.path.join(d, f), tmp_path) for d, _, fs in os.walk(tmp_path) for f in fs)
    # (the recorded run had the reference's `save_time` stubbed out — utils/sfm_utils.py needs cv2 / open3d —: its real runs also leave
    # train_time.txt, whose lines are pinned by the function's own output below)
    assert outs == sorted(list(G["initdir_loop_outputs"]) + ["train_time.txt"]), outs
    import re
    lines = (tmp_path / "train_time.txt").read_text().splitlines()
    assert len(lines) == 2 and re.fullmatch(r"\[2\] train_joint_TrainTime: \d+ min \d+ sec", lines[0]) and re.fullmatch(r"\[2\] train_joint: \d+ min \d+ sec", lines[1]), lines
    # cfg_args is the text the reference's render.py / metrics.py evaluate (arguments/__init__.py:96-116) to find the scene again
    from argparse import Namespace   # noqa: F401  (the name the text refers to)
    cfg = eval((tmp_path / "cfg_args").read_text())
    assert isinstance(cfg, Namespace) and cfg.source_path == SCENE and cfg.n_views == V and cfg.resolution == 2 and cfg.images == "images"
    assert cfg.model_path == str(tmp_path) and cfg.iterations == ITERS and cfg.save_iterations == [ITERS] and cfg.sh_degree == 3 and not cfg.eval
    assert cfg.pp_optimizer and cfg.optim_pose
    assert np.allclose(np.load(tmp_path / "pose" / f"ours_{ITERS}" / "pose_org.npy"), G["initdir_loop_pose_org"], rtol=0, atol=1e-7)
    assert np.allclose(np.load(tmp_path / "pose" / f"ours_{ITERS}" / "pose_optimized.npy"), G["initdir_loop_pose_optimized"], rtol=0, atol=2e-6)
    v = iof.read_ply_vertices(tmp_path / "point_cloud" / f"iteration_{ITERS}" / "point_cloud.ply")
    assert list(v.dtype.names) == list(G["initdir_loop_ply_names"])
    cols = np.stack([v[n] for n in v.dtype.names], axis=1)
    assert np.allclose(cols, G["initdir_loop_ply_columns"], rtol=2e-5, atol=2e-6)
    # ... and they load back into a model that renders what the trained one renders
    from instantsplat_amd.scene import GaussianModel
    from instantsplat_amd.gaussian_renderer import render
    st = r["state"]
    g2 = GaussianModel(3)
    g2.load_ply(str(tmp_path / "point_cloud" / f"iteration_{ITERS}" / "point_cloud.ply"), device="cpu")
    with torch.no_grad():
        cam = st.cameras[0]
        a = render(cam, st.gaussians, st.pipe, st.background, camera_pose=st.gaussians.get_RT(cam.uid))["render"]
        g2.active_sh_degree = st.gaussians.active_sh_degree
        b = render(cam, g2, st.pipe, st.background, camera_pose=st.gaussians.get_RT(cam.uid))["render"]
    assert float((a - b).abs().max()) <= 1e-6


def test_write_init_scene_round_trip(tmp_path):
    """The package's own writer of the layout (what the GPU test and `tools/` use to export a synthetic scene): everything it
    writes comes back through the loader — poses through the quaternion to 1e-12, images and colours to 8 bits."""
    from instantsplat_amd.synthetic import syn_pointmap
    sc = syn_pointmap(3, 5, 5, 40, 30, seed=3)
    g = torch.Generator().manual_seed(0)
    imgs = [torch.rand(3, 30, 40, generator=g) for _ in range(3)]
    w2c = [c.world_view_transform.t().double().numpy() for c in sc.cameras]
    scene_io.write_init_scene(str(tmp_path), w2c, [(c.FoVx, c.FoVy) for c in sc.cameras], imgs, sc.points, sc.colors, sc.confidence)
    back = scene_io.load_init_scene(str(tmp_path), 3, device="cpu", shuffle=False)
    assert [c.colmap_id for c in back.cameras] == [1, 2, 3] and [c.uid for c in back.cameras] == [0, 1, 2]
    for c, m, im in zip(back.cameras, w2c, imgs):
        assert np.allclose(c.world_view_transform.t().numpy(), m, atol=1e-6) and abs(c.FoVx - sc.cameras[0].FoVx) < 1e-12
        assert float((c.original_image - im).abs().max()) <= 0.5 / 255 + 1e-6
    assert torch.equal(back.points, sc.points) and float((back.colors - sc.colors).abs().max()) <= 0.5 / 255 + 1e-6
    lr = (1.0 - torch.sigmoid(sc.confidence)) * 99.0 + 1.0
    assert torch.allclose(back.confidence_lr, lr)
    q = scene_io.rotmat2qvec(w2c[1][:3, :3])
    assert q[0] >= 0 and np.allclose(iof.qvec2rotmat(q), w2c[1][:3, :3], atol=1e-12)


def test_optimizer_relays_out_a_column_major_parameter(emu):
    """The reference's `_xyz` is column-major whenever the points come from a PLY (np.vstack(...).T through torch.tensor keeps the
    strides); the kernels update row-major memory in place, so PerPointAdam makes such a parameter contiguous — once, in place."""
    from instantsplat_amd.optim import PerPointAdam
    g = torch.Generator().manual_seed(0)
    base = torch.randn(3, 40, generator=g)
    p_col = torch.nn.Parameter(torch.tensor(base.numpy().T))           # [40,3] with strides (1, 40), as the reference builds it
    assert not p_col.is_contiguous()
    p_row = torch.nn.Parameter(base.t().contiguous())
    lr_mod = 1.0 + torch.rand(40, 1, generator=g)
    grad = torch.randn(40, 3, generator=g)
    outs = []
    for p in (p_col, p_row):
        opt = PerPointAdam([{"params": [p], "lr": 1e-2, "name": "xyz", "per_point_lr": lr_mod}], lr=0, eps=1e-15)
        assert p.is_contiguous()
        for _ in range(3):
            p.grad = grad.clone()
            opt.step()
        outs.append(p.detach().clone())
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], base.t())


def test_eval_layout_reads_poses_from_sparse_1(tmp_path):
    """`--eval` (reference scene/dataset_readers.py:317-322,336-340): cameras and poses come from sparse_<n>/1, the points still
    from sparse_<n>/0, and train and test camera lists are the same cameras."""
    from instantsplat_amd.synthetic import syn_pointmap
    sc = syn_pointmap(3, 4, 4, 24, 16, seed=8)
    g = torch.Generator().manual_seed(1)
    imgs = [torch.rand(3, 16, 24, generator=g) for _ in range(3)]
    w2c = [c.world_view_transform.t().double().numpy() for c in sc.cameras]
    fovs = [(c.FoVx, c.FoVy) for c in sc.cameras]
    scene_io.write_init_scene(str(tmp_path), w2c, fovs, imgs, sc.points, sc.colors, sc.confidence)
    shifted = [m.copy() for m in w2c]
    for m in shifted:
        m[:3, 3] += 0.25                                   # the "test" poses differ from the training ones
    scene_io.write_init_scene(str(tmp_path), shifted, fovs, imgs, sc.points, sc.colors, None, subdir="1")
    assert not os.path.exists(tmp_path / "sparse_3" / "1" / "points3D.ply")
    tr = scene_io.load_init_scene(str(tmp_path), 3, device="cpu", shuffle=False)
    ev = scene_io.load_init_scene(str(tmp_path), 3, device="cpu", shuffle=False, eval=True)
    assert tr.test_cameras == [] and len(ev.test_cameras) == len(ev.cameras) == 3
    for a, b, c in zip(tr.cameras, ev.cameras, ev.test_cameras):
        assert np.allclose(b.world_view_transform.t().numpy()[:3, 3], a.world_view_transform.t().numpy()[:3, 3] + 0.25, atol=1e-6)
        assert torch.equal(b.world_view_transform, c.world_view_transform) and torch.equal(b.original_image, a.original_image)
    assert torch.equal(ev.points, tr.points)                       # ... the points of sparse_<n>/0
    assert ev.cameras_extent == float(scene_io.get_nerfpp_norm(ev.info.train_cameras)["radius"]) != tr.cameras_extent   # extent of the cameras read


def test_load_cameras_matches_reference_loadCameras():
    """Stored poses back into the cameras (reference scene/dataset_readers.py:75-104, as render.py does with pose_optimized.npy
    and with an interpolated path), against the reference's own function (tests/golden/make_golden_loadcameras.py)."""
    from instantsplat_amd.camera import Camera
    L = np.load(os.path.join(HERE, "golden", "loadcameras_vectors.npz"))
    for tag in ("same", "longer"):
        cams = [Camera(i, torch.from_numpy(scene_io.get_world2view2(L[f"lc_{tag}_in_R"][i], L[f"lc_{tag}_in_T"][i])), 0.9, 0.7, 8, 6,
                       image=torch.zeros(3, 6, 8), colmap_id=i + 1, image_name=f"v{i}") for i in range(3)]
        res = scene_io.load_cameras(L[f"lc_{tag}_poses"], cams)
        assert [[c.uid, c.colmap_id] for c in res] == L[f"lc_{tag}_uid_colmap"].tolist() and [c.image_name for c in res] == list(L[f"lc_{tag}_names"])
        assert np.array_equal(np.stack([c.R for c in res]), L[f"lc_{tag}_R"]) and np.array_equal(np.stack([c.T for c in res]), L[f"lc_{tag}_T"])
        assert np.array_equal(np.stack([c.world_view_transform.numpy() for c in res]), L[f"lc_{tag}_world_view_transform"])
        for k in ("full_proj_transform", "camera_center"):
            assert np.allclose(np.stack([getattr(c, k).numpy() for c in res]), L[f"lc_{tag}_{k}"], rtol=0, atol=2e-6), k
        if tag == "longer":
            assert len(res) == 7 and res[3] is not cams[0]             # copies, not aliases of the three originals


def test_simple_pinhole_cameras(tmp_path):
    """reference scene/dataset_readers.py:129-132: one focal length for both axes"""
    L = np.load(os.path.join(HERE, "golden", "loadcameras_vectors.npz"))
    with open(tmp_path / "cameras.txt", "w") as f:
        f.write("# header\n1 SIMPLE_PINHOLE 8 6 7.5 4.0 3.0\n")
    cams = iof.read_cameras_text(tmp_path / "cameras.txt")
    assert np.allclose(iof.camera_fovs(cams[1]), L["simple_pinhole_fov"], rtol=0, atol=1e-15)
    with open(tmp_path / "bad.txt", "w") as f:
        f.write("1 OPENCV 8 6 7.5 7.5 4.0 3.0 0 0 0 0\n")
    with pytest.raises(ValueError):
        iof.read_cameras_text(tmp_path / "bad.txt")


def test_training_report_at_the_last_iteration(emu, tmp_path, capsys):
    """reference train.py:218,253-295: at the run's last iteration, if it is a testing iteration, mean L1 / PSNR of the clamped
    renders over the training cameras (and the test cameras of an --eval scene) — printed in the reference's words, returned under
    "report"; silent at other iterations; with --eval test cameras the reference's own TypeError (get_RT_test reads a table nothing
    fills, scene/gaussian_model.py:138-140)."""
    import dataclasses
    from instantsplat_amd.arguments import ModelParams
    from instantsplat_amd.train import training
    r = training(SCENE, emu, iterations=3, n_views=V, resolution=2, testing_iterations=[3], log_every=1)
    assert set(r["report"]) == {"train"} and abs(r["report"]["train"][1] - r["psnr_after"]) < 1e-3 and 0 < r["report"]["train"][0] < 1
    assert "[ITER 3] Evaluating train: L1 " in capsys.readouterr().out
    assert training(SCENE, emu, iterations=3, n_views=V, resolution=2, testing_iterations=[7])["report"] == {}
    import shutil
    src = tmp_path / "scene"
    shutil.copytree(SCENE, src)
    os.makedirs(src / f"sparse_{V}" / "1")
    for f in ("cameras.txt", "images.txt"):
        shutil.copyfile(src / f"sparse_{V}" / "0" / f, src / f"sparse_{V}" / "1" / f)
    with pytest.raises(TypeError):
        training(str(src), emu, iterations=2, n_views=V, resolution=2, testing_iterations=[2], model=dataclasses.replace(ModelParams(), eval=True))
