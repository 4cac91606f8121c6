"""CPU tier: on-disk formats. COLMAP text parsing is checked against vectors produced by the
reference's own loader (tests/golden/make_golden.py); PLY and pose files by round trip + header layout."""
import os

import numpy as np
import torch

from instantsplat_amd import io_formats as io

HERE = os.path.dirname(__file__)
G = np.load(os.path.join(HERE, "golden", "reference_vectors.npz"))


def test_colmap_text_matches_reference_loader(tmp_path):
    cams = io.read_cameras_text(os.path.join(HERE, "golden", "colmap_cameras.txt"))
    imgs = io.read_images_text(os.path.join(HERE, "golden", "colmap_images.txt"))
    ids = sorted(cams)
    assert ids == list(G["colmap_cam_ids"])
    assert np.array_equal(np.array([[cams[i].width, cams[i].height] for i in ids]), G["colmap_cam_wh"])
    assert np.allclose(np.stack([cams[i].params for i in ids]), G["colmap_cam_params"], rtol=0, atol=0)
    iid = sorted(imgs)
    assert iid == list(G["colmap_img_ids"])
    assert np.array_equal(np.stack([imgs[i].qvec for i in iid]), G["colmap_img_qvec"])
    assert np.array_equal(np.stack([imgs[i].tvec for i in iid]), G["colmap_img_tvec"])
    assert np.array_equal(np.array([imgs[i].camera_id for i in iid]), G["colmap_img_camid"])
    assert np.allclose(np.stack([io.qvec2rotmat(imgs[i].qvec) for i in iid]), G["colmap_img_R"], atol=1e-15)
    # writer -> reader round trip
    io.write_cameras_text(tmp_path / "c.txt", cams)
    io.write_images_text(tmp_path / "i.txt", imgs)
    cams2, imgs2 = io.read_cameras_text(tmp_path / "c.txt"), io.read_images_text(tmp_path / "i.txt")
    assert all(np.array_equal(cams[i].params, cams2[i].params) for i in ids)
    assert all(np.array_equal(imgs[i].qvec, imgs2[i].qvec) and imgs[i].name == imgs2[i].name for i in iid)
    fx, fy = io.camera_fovs(cams[1])
    assert abs(fx - 2 * np.arctan(1280 / (2 * 1108.5125))) < 1e-12


def test_point_cloud_ply_round_trip(tmp_path):
    g = torch.Generator().manual_seed(0)
    xyz, rgb = torch.randn(100, 3, generator=g), torch.randint(0, 256, (100, 3), generator=g).float() / 255.0
    io.write_point_cloud_ply(tmp_path / "points3D.ply", xyz, rgb)
    xyz2, rgb2 = io.read_point_cloud_ply(tmp_path / "points3D.ply")
    assert torch.equal(xyz, xyz2) and torch.allclose(rgb, rgb2, atol=1e-7)
    hdr = open(tmp_path / "points3D.ply", "rb").read(400).decode("ascii", "replace")
    assert "property float nx" in hdr and "property uchar red" in hdr and "element vertex 100" in hdr


def test_gaussian_ply_layout_and_round_trip(tmp_path):
    class GM:
        pass
    g = torch.Generator().manual_seed(1)
    m = GM()
    m._xyz, m._features_dc, m._features_rest = torch.randn(50, 3, generator=g), torch.randn(50, 1, 3, generator=g), torch.randn(50, 15, 3, generator=g)
    m._opacity, m._scaling, m._rotation = torch.randn(50, 1, generator=g), torch.randn(50, 3, generator=g), torch.randn(50, 4, generator=g)
    io.save_gaussian_ply(tmp_path / "point_cloud.ply", m)
    v = io.read_ply_vertices(tmp_path / "point_cloud.ply")
    # attribute order of reference scene/gaussian_model.py:247-260
    assert list(v.dtype.names) == io.gaussian_ply_attributes() and len(v.dtype.names) == 62
    # f_rest is stored channel-major (transpose(1,2) before flatten, reference :267)
    assert np.allclose(v["f_rest_0"], m._features_rest[:, 0, 0].numpy()) and np.allclose(v["f_rest_15"], m._features_rest[:, 0, 1].numpy())
    d = io.load_gaussian_ply(tmp_path / "point_cloud.ply")
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert torch.equal(d[k], getattr(m, k)), k


def test_save_pose_orders_by_colmap_id(tmp_path):
    poses = torch.tensor([[1.0, 0, 0, 0, 1, 2, 3], [0.0, 1, 0, 0, 4, 5, 6]])
    io.save_pose(tmp_path / "pose_optimized.npy", poses, colmap_ids=[2, 1])
    a = np.load(tmp_path / "pose_optimized.npy")
    assert a.shape == (2, 4, 4) and np.allclose(a[0, :3, 3], [4, 5, 6]) and np.allclose(a[1, :3, 3], [1, 2, 3])


def test_save_pose_matches_reference_function(tmp_path):
    """io.save_pose vs the output of the reference's own save_pose (train.py:46-60, executed by make_golden.py)."""
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))
    io.save_pose(tmp_path / "pose_optimized.npy", torch.from_numpy(G["save_pose_in"]), colmap_ids=[int(i) for i in G["save_pose_colmap_ids"]])
    ours = np.load(tmp_path / "pose_optimized.npy")
    assert ours.shape == G["save_pose_out"].shape and ours.dtype == G["save_pose_out"].dtype
    assert np.allclose(ours, G["save_pose_out"], rtol=0, atol=1e-7)


def test_cfg_args_is_the_text_the_references_own_parser_writes():
    """<model_path>/cfg_args (reference train.py:245-246): character for character what the reference's argument classes and
    train.py's parser produce for its scripts' command line (tests/golden/make_golden_cfg_args.py), so that the reference's
    get_combined_args — render.py, metrics.py — finds source_path / n_views / resolution / sh_degree of a model trained here."""
    import json
    from argparse import Namespace
    from instantsplat_amd.arguments import ModelParams, OptimizationParams, PipelineParams, cfg_args_text
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg_args_reference.json")))
    text = cfg_args_text(ModelParams(source_path="/data/scene", model_path="/out/scene_3_views", resolution=1, n_views=3),
                         OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True), PipelineParams(), save_iterations=[1000])
    assert text == g["cfg_args_text"]
    ns = eval(text)
    assert isinstance(ns, Namespace)
    for k, v in g["render_dataset"].items():   # what the reference's render.py extracts from it
        assert getattr(ns, k) == v, k


def test_save_time_appends_the_references_lines(tmp_path):
    """<model_path>/train_time.txt: the reference's `save_time` (utils/sfm_utils.py:43-50), pinned by what the function itself wrote
    for the same calls (tests/golden/make_golden_cfg_args.py)."""
    import json
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg_args_reference.json")))
    for name, sec in g["save_time_calls"]:
        io.save_time(str(tmp_path / "model"), name, sec)
    assert (tmp_path / "model" / "train_time.txt").read_text() == g["train_time_txt"]
