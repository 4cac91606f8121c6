"""GPU tier: BASELINE configs[0] — "3-view sora, 50 iterations, plumbing" — on the reference's own example frames at their native
1280 x 720 (tests/sora_util.py says what replaces MASt3R): init directory with the JPEG files -> load_init_scene -> 50 iterations of
the drop-in loop (train.py's loss as written) on the MI355X against the all-CPU oracle trainer from the same state, loss by loss."""
import numpy as np
import pytest
import torch

from tests import sora_util

pytestmark = pytest.mark.gpu


def test_sora_art_3_views_50_iterations_native_resolution(gpu, tmp_path):
    from instantsplat_amd import scene_io
    from instantsplat_amd.train import evaluate_psnr
    from oracle import gs_ref
    gs_ref.lib().gsref_set_threads(32)
    sora_util.write_sora_init_dir(str(tmp_path / "Art"), Wm=160, Hm=90)
    sc = scene_io.load_init_scene(str(tmp_path / "Art"), 3, resolution=1, device=gpu)
    assert [(c.image_width, c.image_height) for c in sc.cameras] == [(1280, 720)] * 3 and sc.points.shape[0] == 43200
    l_dev, l_cpu, st = sora_util.train_against_cpu_oracle(sc, gpu, iters=50)
    worst = max(abs(a - b) / max(abs(b), 1e-2) for a, b in zip(l_dev, l_cpu))
    print("sora/Art 1280x720, 43,200 Gaussians, 50 iterations: loss %.5f -> %.5f (device), %.5f -> %.5f (CPU oracle); largest relative "
          "difference %.2e; PSNR after %.2f dB" % (l_dev[0], l_dev[-1], l_cpu[0], l_cpu[-1], worst, evaluate_psnr(st)))
    assert abs(l_dev[0] - l_cpu[0]) <= 2e-5 * abs(l_cpu[0]) + 1e-6          # the first iteration: same state, same frame
    assert worst <= 5e-3                                                     # (the same bound as the synthetic C1' loop)
    assert np.mean(l_dev[-5:]) < 0.9 * np.mean(l_dev[:5])                    # it trains


def test_sora_art_at_half_resolution_trains_from_disk(gpu, tmp_path):
    """`-r 2` (640 x 360) through training(<source_path>): the reference's outputs appear, the loss falls"""
    import os
    from instantsplat_amd.train import training
    sora_util.write_sora_init_dir(str(tmp_path / "Art"), Wm=160, Hm=90)
    out = tmp_path / "model"
    r = training(str(tmp_path / "Art"), gpu, iterations=50, n_views=3, resolution=2, model_path=str(out), saving_iterations=[50])
    st = r["state"]
    assert [(c.image_width, c.image_height) for c in st.cameras] == [(640, 360)] * 3
    assert r["psnr_after"] > r["psnr_before"] + 1.0 and np.isfinite(r["last_loss"])
    for f in ("cfg_args", "input.ply", "cameras.json", "point_cloud/iteration_50/point_cloud.ply", "pose/ours_50/pose_optimized.npy"):
        assert (out / f).exists(), f
    assert "resolution=2" in open(out / "cfg_args").read()
