"""CPU tier: BASELINE configs[0] on the reference's own example frames (tests/sora_util.py: what replaces MASt3R) — the init
directory with JPEG frames, load at `-r 8` (160 x 90: non-square, not a multiple of the 16-pixel tile), a few iterations on
the emulated kernels against the CPU oracle trainer."""
import numpy as np
import torch

from tests import sora_util


def test_sora_frames_are_the_references(tmp_path):
    import hashlib
    import os
    sums = dict(l.split()[::-1] for l in open(os.path.join(os.path.dirname(sora_util.FRAMES[0]), "SHA256SUMS")))
    for f in sora_util.FRAMES:
        assert hashlib.sha256(open(f, "rb").read()).hexdigest() == sums[os.path.basename(f)]


def test_sora_init_directory_loads_and_trains(emu, tmp_path):
    from PIL import Image
    from instantsplat_amd import scene_io
    W, H = sora_util.write_sora_init_dir(str(tmp_path / "Art"), Wm=16, Hm=9)
    assert (W, H) == (1280, 720)
    # -r 1 / 2 / 8 and "width 400": sizes by the reference's rule (utils/camera_utils.py:22-42), pixels = PIL's resize of the JPEG / 255
    for res, wh in ((1, (1280, 720)), (2, (640, 360)), (8, (160, 90)), (400, (400, 225))):
        sc = scene_io.load_init_scene(str(tmp_path / "Art"), 3, resolution=res, device="cpu")
        assert [(c.image_width, c.image_height) for c in sc.cameras] == [wh] * 3
        c = sc.cameras[0]
        with Image.open(sora_util.FRAMES[int(c.image_name)]) as im:   # image_name = "0" / "1" / "2": the reference's file names
            ref = torch.from_numpy(np.array(im.resize(wh))).permute(2, 0, 1) / 255.0
        assert torch.equal(c.original_image, ref.float()) and c.original_image.shape == (3, wh[1], wh[0])
        assert abs(np.tan(c.FoVy / 2) / np.tan(c.FoVx / 2) - 720 / 1280) < 1e-6       # the field of view comes from cameras.txt, not from the resized image
    sc = scene_io.load_init_scene(str(tmp_path / "Art"), 3, resolution=8, device=emu)
    assert sc.points.shape == (3 * 16 * 9, 3) and float(sc.colors.min()) >= 0 and float(sc.colors.max()) <= 1
    l_dev, l_cpu, st = sora_util.train_against_cpu_oracle(sc, emu, iters=4)
    assert max(abs(a - b) / max(abs(b), 1e-2) for a, b in zip(l_dev, l_cpu)) <= 2e-3, (l_dev, l_cpu)
    assert all(np.isfinite(l_dev)) and 0.01 < l_dev[0] < 1.0
