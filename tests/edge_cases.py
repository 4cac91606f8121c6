"""Edge cases shared by the CPU (emulated) and GPU tiers."""
import math

import torch

from tests.util import assert_no_worse_than_fp32_oracle, assert_raster_parity, run_custom_case, run_blob_case


def _base(n, seed):
    g = torch.Generator().manual_seed(seed)
    means = torch.stack([0.8 * (2 * torch.rand(n, generator=g) - 1), 0.6 * (2 * torch.rand(n, generator=g) - 1),
                         3.0 + 2.0 * torch.rand(n, generator=g)], dim=1)
    q = torch.randn(n, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    return g, means, q


def check_giant_and_needle_gaussians(dev):
    """A few Gaussians that cover every tile, and extremely thin ones (ill-conditioned conic: culling box disabled)."""
    g, means, q = _base(40, 0)
    scales = torch.full((40, 3), 0.05)
    scales[:4] = 6.0                                         # cover the whole image
    scales[4:12] = torch.tensor([2.0, 1e-4, 1e-4])           # needles
    scales[12:16] = torch.tensor([1e-5, 1e-5, 1e-5])         # sub-pixel: only the 0.3 px^2 low-pass is left
    opac = torch.rand(40, 1, generator=g) * 0.9 + 0.05
    col = torch.rand(40, 3, generator=g)
    out = run_custom_case(dev, means, scales, q, opac, col, 80, 48)
    # the needles' 2-D covariances have condition numbers ~4000: every fp32 evaluation of the conic backward drifts
    # from the float64 result by ~1e-2 (the fp32 oracle included), so the device is held to the oracle's own error
    assert_no_worse_than_fp32_oracle(out)
    d = (out["ref"]["color"] - out["dut"]["color"]).abs()
    assert float(d.max()) <= 5e-3 and float((d > 1e-4).float().mean()) <= 1e-4
    assert bool((out["ref"]["radii"] == out["dut"]["radii"]).all())
    assert int(out["dut"]["radii"][:4].min()) > 100


def check_invisible_opacity_and_behind_camera(dev):
    """opacity < 1/255 (never passes the alpha test: binned to no tile), z <= 0.2 (culled), off-screen."""
    g, means, q = _base(60, 1)
    scales = torch.full((60, 3), 0.1)
    opac = torch.rand(60, 1, generator=g) * 0.9 + 0.05
    opac[:10] = 0.003                     # below 1/255
    means[10:20, 2] = 0.1                 # behind the near cull plane
    means[20:30, 0] = 50.0                # far off-screen
    col = torch.rand(60, 3, generator=g)
    out = run_custom_case(dev, means, scales, q, opac, col, 64, 64)
    assert_raster_parity(out)
    assert bool((out["dut"]["radii"][10:20] == 0).all())
    for k, gr in out["dut"]["grads"].items():
        assert float(gr[:30].abs().max()) == 0.0, k      # no gradient reaches any of them


def check_saturating_opacity_early_termination(dev):
    """Opaque stack: alpha clamps at 0.99 and pixels terminate (T < 1e-4) long before the list ends."""
    g, means, q = _base(300, 2)
    means[:, :2] *= 0.2
    scales = torch.full((300, 3), 0.4)
    opac = torch.full((300, 1), 0.999)
    col = torch.rand(300, 3, generator=g)
    out = run_custom_case(dev, means, scales, q, opac, col, 48, 48)
    assert_raster_parity(out, grad_tol=5e-4)


def check_mark_visible(dev):
    from instantsplat_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from instantsplat_amd.camera import Camera
    from tests.util import settings_for
    cam = Camera(0, torch.eye(4), 1.0, 1.0, 32, 32)
    st = settings_for(cam, 0, GaussianRasterizationSettings, torch.zeros(3), device=dev)
    pts = torch.tensor([[0.0, 0.0, 1.0], [0.0, 0.0, 0.2], [0.0, 0.0, -3.0], [5.0, 0.0, 0.21]])
    vis = GaussianRasterizer(st).markVisible(pts.to(dev))
    assert vis.cpu().tolist() == [True, False, False, True]


def check_python_flag_paths(dev):
    """pipe.convert_SHs_python / compute_cov3D_python (reference gaussian_renderer/__init__.py:97-119): colours and
    covariances precomputed in PyTorch go through colors_precomp / cov3D_precomp and gradients still reach every
    parameter."""
    import instantsplat_amd.gaussian_renderer as gr
    from instantsplat_amd.arguments import PipelineParams
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import setup_training
    st = setup_training(syn_pointmap(3, 12, 12, 32, 32, seed=2), dev)
    g = st.gaussians
    cam = st.cameras[0]
    base = gr.render(cam, g, PipelineParams(), st.background, camera_pose=g.get_RT(cam.uid))["render"]
    pipe = PipelineParams(convert_SHs_python=True, compute_cov3D_python=True)
    img = gr.render(cam, g, pipe, st.background, camera_pose=g.get_RT(cam.uid))["render"]
    assert img.shape == base.shape and bool(torch.isfinite(img).all())
    img.sum().backward()
    for t in (g._xyz, g._features_dc, g._scaling, g._rotation, g._opacity):
        assert t.grad is not None and bool(torch.isfinite(t.grad).all())


def check_create_from_pcd_scales(dev):
    """GaussianModel.create_from_pcd (reference scene/gaussian_model.py:146-172): log sqrt of the clamped mean 3-NN distance."""
    from instantsplat_amd.scene import GaussianModel
    from oracle import knn_ref
    g = torch.Generator().manual_seed(4)
    pts, col = torch.randn(500, 3, generator=g), torch.rand(500, 3, generator=g)
    gm = GaussianModel(3)
    gm.create_from_pcd(pts, col, 1.0, dev)
    ref = torch.log(torch.sqrt(knn_ref.dist2(pts).clamp_min(1e-7)))[:, None].repeat(1, 3)
    assert torch.allclose(gm._scaling.detach().cpu(), ref, rtol=1e-5, atol=1e-6)
    assert gm._features_dc.shape == (500, 1, 3) and gm._features_rest.shape == (500, 15, 3)
    assert torch.allclose(gm.get_opacity.detach().cpu(), torch.full((500, 1), 0.1), atol=1e-6)


def check_long_tile_lists(dev, n):
    """n faint Gaussians stacked on a few tiles: per-tile lists longer than the sort kernel's register network (2048 keys:
    sorted runs + rank placement) and, for n > 8192, than its LDS window (in-place global network); also dozens of backward
    segments per tile.  Everything still has to match the oracle."""
    g = torch.Generator().manual_seed(11)
    z = 2.0 + 4.0 * torch.rand(n, generator=g)
    means = torch.stack([0.12 * (2 * torch.rand(n, generator=g) - 1) * z, 0.12 * (2 * torch.rand(n, generator=g) - 1) * z, z], dim=1)
    q = torch.randn(n, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    scales = 0.01 + 0.02 * torch.rand(n, 3, generator=g)
    opac = torch.full((n, 1), 0.01) + 0.02 * torch.rand(n, 1, generator=g)   # faint: nothing saturates, every instance counts
    col = torch.rand(n, 3, generator=g)
    out = run_custom_case(dev, means, scales, q, opac, col, 48, 32)
    # Thousands of fp32 additions per pixel in a different order than the oracle, and alphas of 0.00-0.03 around the 1/255
    # cut: a (pixel, Gaussian) pair whose alpha rounds to the other side of the threshold moves that pixel by ~4e-4 and
    # the gradients by ~1e-3 (seen for n = 12000: one such pair).  Ordering itself is checked exactly by
    # check_tile_lists_sorted.
    assert_raster_parity(dict(ref=out["ref"], dut=out["dut"]), fwd_tol=5e-4, grad_tol=3e-3)


def check_multi_chunk_units(dev, n=6000, min_units=4):
    """The backward's units are lengthened to 2, 4 or 8 chunks of 64 instances once a frame has more than ~12 k of them
    (csrc/common.h, GS_MIN_UNITS; C4 runs at 8).  Lowering the knob makes small scenes take that path: the forward then
    leaves boundary records only every chunks * 64 instances and a backward wave replays several chunks in a row — image and
    every gradient must be what the one-chunk path gives (bit for bit in the forward; the backward's float atomics land in a
    different order: on the GPU always, under the emulator because the two unit lengths number — and since the XCD transposition of
    the launch index also order — their units differently; measured there: 1.05e-6 of the largest gradient, relative L2 9e-7)."""
    from instantsplat_amd import _lib
    L = _lib.lib()
    res = {}
    old = L.mi355gs_tune_min_units(0)
    try:
        for mu in (1 << 30, min_units):     # never lengthen / lengthen as far as GS_MAX_CHUNKS allows
            L.mi355gs_tune_min_units(mu)
            res[mu] = run_blob_case(dev, n, 96, 64, 1, scale_mean=0.12, seed=9)
            assert_raster_parity(res[mu], grad_tol=2e-4)
    finally:
        L.mi355gs_tune_min_units(old)
    a, b = res[1 << 30]["dut"], res[min_units]["dut"]
    assert torch.equal(a["color"], b["color"]) and torch.equal(a["radii"], b["radii"])
    for k in a["grads"]:
        d = float((a["grads"][k] - b["grads"][k]).abs().max())
        assert d <= 1e-5 * max(1.0, float(a["grads"][k].abs().max())), (k, d)


def check_tile_lists_sorted(dev, n):
    """Exact property of the binning stage, straight through the C ABI: every per-tile list is a duplicate-free set of
    Gaussian indices in ascending (depth bits, index) order — for list lengths on both sides of the sort kernel's
    2048-key register network and 8192-key rank-placement window."""
    from instantsplat_amd import _lib
    from instantsplat_amd.camera import Camera
    import numpy as np
    dev = torch.device(dev)
    L = _lib.lib()
    g = torch.Generator().manual_seed(11)
    z = 2.0 + 4.0 * torch.rand(n, generator=g)
    z[: n // 50] = z[n // 50: 2 * (n // 50)]   # exact depth ties: the index has to break them
    means = torch.stack([0.12 * (2 * torch.rand(n, generator=g) - 1) * z, 0.12 * (2 * torch.rand(n, generator=g) - 1) * z, z], dim=1)
    q = torch.randn(n, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    scales = 0.01 + 0.02 * torch.rand(n, 3, generator=g)
    opac = (torch.full((n, 1), 0.02) + 0.02 * torch.rand(n, 1, generator=g)).reshape(-1)
    col = torch.rand(n, 3, generator=g)
    W, H = 48, 32
    tanx = math.tan(math.radians(60) / 2)
    tany = tanx * H / W
    cam = Camera(0, torch.eye(4), math.radians(60), 2 * math.atan(tany), W, H)
    t = lambda x: x.float().contiguous().to(dev)
    means, q, scales, opac, col = map(t, (means, q, scales, opac, col))
    view, proj, campos = t(torch.eye(4).reshape(-1)), t(cam.projection_matrix.reshape(-1)), t(torch.zeros(3))
    geom = torch.zeros(L.mi355gs_raster_geom_bytes(n), dtype=torch.uint8, device=dev)
    tiles = torch.zeros(L.mi355gs_raster_tiles_bytes(W, H), dtype=torch.uint8, device=dev)
    radii = torch.zeros(n, dtype=torch.int32, device=dev)
    nr = torch.zeros(1, dtype=torch.int32, device=dev)
    p, stream = _lib.ptr, _lib.stream_ptr(dev)
    _lib.check(L.mi355gs_raster_forward_preprocess(stream, n, 0, 0, W, H, p(means), None, None, p(col), p(opac), p(scales), 1.0, p(q), None,
                                                   p(view), p(proj), p(campos), tanx, tany, 0, p(radii), p(geom), p(tiles), p(nr), None, None, 0), "preprocess")
    R = int(nr.item())
    binning = torch.zeros(L.mi355gs_raster_binning_bytes(R, W, H), dtype=torch.uint8, device=dev)
    img, bg = torch.zeros(3, H, W, device=dev), torch.zeros(3, device=dev)
    _lib.check(L.mi355gs_raster_forward_render(stream, n, W, H, R, p(bg), p(geom), p(tiles), p(binning), p(img), 0), "render")
    T, al = 6, (lambda x: (x + 255) & ~255)
    # scratch layouts (csrc/common.h): tiles = count | cursor | start[T+1] ...; binning = keys[R] (8 B) | list[R] (4 B) ...
    start = tiles[2 * al(T * 4): 2 * al(T * 4) + (T + 1) * 4].cpu().view(torch.int32).numpy()
    lst = binning[al(R * 8): al(R * 8) + R * 4].cpu().view(torch.int32).numpy()
    depth = geom[: n * 48].cpu().view(torch.float32).reshape(n, 12).numpy()[:, 11]
    assert start[0] == 0 and start[T] == R and R > 0
    longest = 0
    check_tile_lists_sorted.lengths = [int(start[tile + 1] - start[tile]) for tile in range(T)]
    for tile in range(T):
        seg = lst[start[tile]:start[tile + 1]]
        longest = max(longest, len(seg))
        assert len(np.unique(seg)) == len(seg), tile
        key = (depth[seg].view(np.uint32).astype(np.uint64) << np.uint64(32)) | seg.astype(np.uint64)
        assert bool((key[1:] > key[:-1]).all()), (tile, len(seg))
    return longest



def check_backward_launch_order(dev, n=3000, min_units=None):
    """The unit table the forward leaves for the backward (csrc/composite.hip, common.h), read raw through the C ABI: it is a
    permutation of every (tile, segment) of the frame; every full-length unit comes before every short one (the tiles' last
    units, which are launched last to shorten the kernel's tail); an entry's slot is seg_first[tile] + segment."""
    from instantsplat_amd import _lib
    from instantsplat_amd.camera import Camera
    import numpy as np
    dev = torch.device(dev)
    L = _lib.lib()
    old = L.mi355gs_tune_min_units(0)
    if min_units is not None:
        L.mi355gs_tune_min_units(min_units)
    try:
        g = torch.Generator().manual_seed(3)
        z = 2.0 + 4.0 * torch.rand(n, generator=g)
        means = torch.stack([0.5 * (2 * torch.rand(n, generator=g) - 1) * z, 0.35 * (2 * torch.rand(n, generator=g) - 1) * z, z], dim=1)
        q = torch.randn(n, 4, generator=g)
        q = q / q.norm(dim=1, keepdim=True)
        scales = 0.02 + 0.05 * torch.rand(n, 3, generator=g)
        opac = torch.full((n,), 0.05) + 0.1 * torch.rand(n, generator=g)
        col = torch.rand(n, 3, generator=g)
        W, H = 80, 48
        tanx = math.tan(math.radians(60) / 2)
        tany = tanx * H / W
        cam = Camera(0, torch.eye(4), math.radians(60), 2 * math.atan(tany), W, H)
        t = lambda x: x.float().contiguous().to(dev)
        means, q, scales, opac, col = map(t, (means, q, scales, opac, col))
        view, proj, campos = t(torch.eye(4).reshape(-1)), t(cam.projection_matrix.reshape(-1)), t(torch.zeros(3))
        geom = torch.zeros(L.mi355gs_raster_geom_bytes(n), dtype=torch.uint8, device=dev)
        tiles = torch.zeros(L.mi355gs_raster_tiles_bytes(W, H), dtype=torch.uint8, device=dev)
        radii = torch.zeros(n, dtype=torch.int32, device=dev)
        nr = torch.zeros(1, dtype=torch.int32, device=dev)
        p, stream = _lib.ptr, _lib.stream_ptr(dev)
        _lib.check(L.mi355gs_raster_forward_preprocess(stream, n, 0, 0, W, H, p(means), None, None, p(col), p(opac), p(scales), 1.0, p(q), None,
                                                       p(view), p(proj), p(campos), tanx, tany, 0, p(radii), p(geom), p(tiles), p(nr), None, None, 0), "preprocess")
        R = int(nr.item())
        binning = torch.zeros(L.mi355gs_raster_binning_bytes(R, W, H), dtype=torch.uint8, device=dev)
        img, bg = torch.zeros(3, H, W, device=dev), torch.zeros(3, device=dev)
        _lib.check(L.mi355gs_raster_forward_render(stream, n, W, H, R, p(bg), p(geom), p(tiles), p(binning), p(img), 0), "render")
        gx, gy = (W + 15) // 16, (H + 15) // 16
        T, al = gx * gy, (lambda x: (x + 255) & ~255)
        # scratch layouts (csrc/common.h).  tiles = count | cursor | start[T+1] | final_T | n_contrib | order | seg_first[T+1] |
        # part_first[T+1] | meta; binning = keys[R] (8 B) | list[R] (4 B) | unit table (16 B per entry) | boundary records
        tb = tiles.cpu().numpy()
        o = 2 * al(T * 4)
        start = tb[o: o + (T + 1) * 4].view(np.int32); o += al((T + 1) * 4) + 2 * al(W * H * 4) + al(T * 4)
        seg_first = tb[o: o + (T + 1) * 4].view(np.int32); o += al((T + 1) * 4)
        part_first = tb[o: o + (T + 1) * 4].view(np.int32); o += al((T + 1) * 4)
        meta = tb[o: o + 16].view(np.int32)
        n_units, chunks, n_short = int(meta[1]), int(meta[2]), int(meta[3])
        seg_len = 64 * chunks
        count = np.diff(start)
        assert n_units == int(np.sum((count + seg_len - 1) // seg_len)) == int(seg_first[T]) and n_short == int(np.sum(count % seg_len != 0))
        table = binning.cpu().numpy()[al(R * 8) + al(R * 4):][: n_units * 16].view(np.uint32).reshape(n_units, 4)
        where, seg, slot = table[:, 0], table[:, 1], table[:, 2]
        tile = (where >> 16) * gx + (where & 0xFFFF)
        assert bool((tile < T).all()) and bool((slot == seg_first[tile] + seg).all())
        assert len(np.unique(slot)) == n_units and int(slot.max()) == n_units - 1          # every unit exactly once
        last_seg = (count[tile] + seg_len - 1) // seg_len - 1
        short = (seg == last_seg) & (count[tile] % seg_len != 0)
        assert int(short.sum()) == n_short and not short[: n_units - n_short].any() and short[n_units - n_short:].all()
        return n_units, n_short, chunks
    finally:
        L.mi355gs_tune_min_units(old)


def check_operator_error_behaviour(dev, tmp_path):
    """Error contract of the rasterizer module (operator __init__ as called at reference
    gaussian_renderer/__init__.py:126-135): exactly one colour source and one covariance source or an `Exception`, the
    means3D shape message, and with `debug=True` a failing call leaves `snapshot_fw.dump` (a torch.save of the CPU copies
    of its arguments) in the working directory before the exception propagates."""
    import os
    import pytest
    from instantsplat_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from instantsplat_amd.synthetic import syn_blob
    from tests.util import settings_for
    sc = syn_blob(64, 48, 32, seed=2)
    dev = torch.device(dev)
    t = lambda x: x.to(dev)
    means, shs, op = t(sc.means3D), t(sc.shs), torch.sigmoid(t(sc.opacity_logit))
    scales, rots = torch.exp(t(sc.scaling_logit)), t(sc.rotation)
    m2d = torch.zeros_like(means)
    cols, cov = torch.rand(64, 3, device=dev), torch.rand(64, 6, device=dev)
    bg = torch.zeros(3)
    r = GaussianRasterizer(settings_for(sc.camera, 0, GaussianRasterizationSettings, bg, device=dev))
    with pytest.raises(Exception, match="one of either SHs or precomputed colors"):
        r(means3D=means, means2D=m2d, opacities=op, shs=shs, colors_precomp=cols, scales=scales, rotations=rots)
    with pytest.raises(Exception, match="one of either SHs or precomputed colors"):
        r(means3D=means, means2D=m2d, opacities=op, scales=scales, rotations=rots)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=means, means2D=m2d, opacities=op, shs=shs, scales=scales, rotations=rots, cov3D_precomp=cov)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=means, means2D=m2d, opacities=op, shs=shs, scales=scales)
    with pytest.raises(RuntimeError, match=r"means3D must have dimensions \(num_points, 3\)"):
        r(means3D=means[:, :2].contiguous(), means2D=m2d, opacities=op, shs=shs, scales=scales, rotations=rots)
    # a call the library rejects (SH degree 4), debug on: snapshot + exception; debug off: exception only
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        bad = GaussianRasterizer(settings_for(sc.camera, 4, GaussianRasterizationSettings, bg, device=dev))
        with pytest.raises(RuntimeError):
            bad(means3D=means, means2D=m2d, opacities=op, shs=shs, scales=scales, rotations=rots)
        assert not os.path.exists("snapshot_fw.dump")
        bad = GaussianRasterizer(settings_for(sc.camera, 4, GaussianRasterizationSettings, bg, device=dev, debug=True))
        with pytest.raises(RuntimeError):
            bad(means3D=means, means2D=m2d, opacities=op, shs=shs, scales=scales, rotations=rots)
        dump = torch.load("snapshot_fw.dump", weights_only=False)
        assert torch.equal(dump[0], means.cpu()) and dump[-1][8] == 4   # (means3D, ..., settings tuple with sh_degree)
    finally:
        os.chdir(cwd)
    # and the valid call still works afterwards
    color, radii = r(means3D=means, means2D=m2d, opacities=op, shs=shs, scales=scales, rotations=rots)
    assert color.shape == (3, 32, 48) and radii.shape == (64,) and bool(torch.isfinite(color).all())


def check_tile_lists_against_oracle(dev, n, W=80, H=48, seed=5, scale_boost=1.0, entry_counts=None):
    """The binning stage against the ORACLE's lists (SURVEY.md 8d asked for num_rendered within 0.01 %; the device bins to
    a tighter rectangle on purpose, preprocess.hip "tight tile rects", so the property is stated exactly instead):
      * every per-tile list of the device is a SUBSET of the oracle's list of that tile (the reference's 3-sigma rectangle),
        in the same relative order (depth, then index);
      * every instance the device dropped has alpha < 1/255 at every pixel of that tile — it could not have been blended,
        so image, n_contrib-as-a-set and all gradients are unaffected by dropping it;
      * no instance is dropped that reaches alpha >= 1/255 anywhere (the same statement, from the other side)."""
    from instantsplat_amd import _lib
    from instantsplat_amd.camera import Camera
    from oracle import gs_ref
    from oracle.raster_torch import RasterSettings
    import numpy as np
    dev = torch.device(dev)
    L = _lib.lib()
    g = torch.Generator().manual_seed(seed)
    z = 2.0 + 4.0 * torch.rand(n, generator=g)
    means = torch.stack([0.6 * (2 * torch.rand(n, generator=g) - 1) * z, 0.4 * (2 * torch.rand(n, generator=g) - 1) * z, z], dim=1)
    q = torch.randn(n, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    # a mix of sizes and opacities: faint large Gaussians are the ones whose 3-sigma rectangle is much larger than their
    # alpha >= 1/255 footprint
    scales = (0.01 + 0.08 * torch.rand(n, 3, generator=g) ** 2) * scale_boost
    opac = 0.005 + 0.9 * torch.rand(n, generator=g) ** 3
    col = torch.rand(n, 3, generator=g)
    tanx = math.tan(math.radians(60) / 2)
    tany = tanx * H / W
    cam = Camera(0, torch.eye(4), math.radians(60), 2 * math.atan(tany), W, H)
    t = lambda x: x.float().contiguous().to(dev)
    d_means, d_q, d_scales, d_opac, d_col = map(t, (means, q, scales, opac, col))
    view, proj, campos = t(torch.eye(4).reshape(-1)), t(cam.projection_matrix.reshape(-1)), t(torch.zeros(3))
    geom = torch.zeros(L.mi355gs_raster_geom_bytes(n), dtype=torch.uint8, device=dev)
    tiles = torch.zeros(L.mi355gs_raster_tiles_bytes(W, H), dtype=torch.uint8, device=dev)
    radii = torch.zeros(n, dtype=torch.int32, device=dev)
    nr = torch.zeros(1, dtype=torch.int32, device=dev)
    p, stream = _lib.ptr, _lib.stream_ptr(dev)
    _lib.check(L.mi355gs_raster_forward_preprocess(stream, n, 0, 0, W, H, p(d_means), None, None, p(d_col), p(d_opac), p(d_scales), 1.0,
                                                   p(d_q), None, p(view), p(proj), p(campos), tanx, tany, 0, p(radii), p(geom), p(tiles),
                                                   p(nr), None, None, 0), "preprocess")
    R = int(nr.item())
    binning = torch.zeros(L.mi355gs_raster_binning_bytes(R, W, H), dtype=torch.uint8, device=dev)
    img, bg = torch.zeros(3, H, W, device=dev), torch.zeros(3, device=dev)
    _lib.check(L.mi355gs_raster_forward_render(stream, n, W, H, R, p(bg), p(geom), p(tiles), p(binning), p(img), 0), "render")
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T, al = gx * gy, (lambda x: (x + 255) & ~255)
    start = tiles[2 * al(T * 4): 2 * al(T * 4) + (T + 1) * 4].cpu().view(torch.int32).numpy()
    lst = binning[al(R * 8): al(R * 8) + R * 4].cpu().view(torch.int32).numpy()
    rec = geom[: n * 48].cpu().view(torch.float32).reshape(n, 12).numpy().astype(np.float64)
    if entry_counts is not None:
        # what the count kernel handed the scatter kernel (common.h GeomLayout): per 512-Gaussian workgroup, the number of
        # touched tiles it listed, or ~0 = "more than the list holds, count again"
        chunks = (n + 511) // 512
        off = al(n * 48) + al(n * 24) + al(n * 8) + al(n) + al(n * 4) + al(chunks * 1024 * 4)
        entry_counts.extend(int(v) for v in geom[off: off + chunks * 4].cpu().view(torch.int32).numpy().astype(np.int64) & 0xffffffff)

    settings = RasterSettings(image_height=H, image_width=W, tanfovx=tanx, tanfovy=tany, bg=torch.zeros(3), scale_modifier=1.0,
                              viewmatrix=torch.eye(4), projmatrix=cam.projection_matrix.cpu(), sh_degree=0, campos=torch.zeros(3),
                              prefiltered=False, debug=False)
    _, radii_ref, ctx = gs_ref.forward(means, opac, settings, colors_precomp=col, scales=scales, rotations=q)
    aux = ctx.aux(W, H)
    o_start, o_list = aux["tile_start"].numpy(), aux["list"].numpy()
    R_ref = ctx.num_rendered
    assert start[0] == 0 and start[T] == R and 0 < R <= R_ref
    dropped = kept = 0
    worst = 0.0
    for tile in range(T):
        mine = lst[start[tile]:start[tile + 1]]
        theirs = o_list[o_start[tile]:o_start[tile + 1]]
        in_mine = np.isin(theirs, mine)
        assert len(np.unique(mine)) == len(mine) and bool(np.isin(mine, theirs).all()), tile       # subset of the reference's list
        assert bool((theirs[in_mine] == mine).all()), tile                                          # same relative order
        gone = theirs[~in_mine]
        kept += len(mine)
        dropped += len(gone)
        if len(gone) == 0:
            continue
        tx, ty = tile % gx, tile // gx
        xs = np.arange(tx * 16, min(tx * 16 + 16, W), dtype=np.float64)
        ys = np.arange(ty * 16, min(ty * 16 + 16, H), dtype=np.float64)
        r = rec[gone]   # x, y, hx, hy | A, C, B (log2-scaled conic), opacity | r, g, b, depth
        dx = r[:, 0, None, None] - xs[None, None, :]
        dy = r[:, 1, None, None] - ys[None, :, None]
        power2 = r[:, 4, None, None] * dx * dx + r[:, 5, None, None] * dy * dy + r[:, 6, None, None] * dx * dy
        alpha = r[:, 7, None, None] * np.exp2(np.minimum(power2, 0.0))
        alpha = np.where(power2 > 0, 0.0, alpha)
        worst = max(worst, float(alpha.max()))
        assert float(alpha.max()) < 1.0 / 255.0, (tile, float(alpha.max()))
    assert kept == R and kept + dropped == R_ref
    return R, R_ref, worst


def check_count_slots_survive_unpolled_forwards(dev):
    """A frame rendered without reading its count back (BinningPolicy "bounded") keeps its word of the pinned count ring until
    poll() has read it — however many other forwards (exact mode, inference renders) run in between and take words of the
    ring themselves.  (the ring used to hand every word out round-robin, so 256 later forwards overwrote the
    pending frame's count and poll() verified the wrong number.)"""
    from instantsplat_amd.diff_gaussian_rasterization import (COUNT_RING, BinningPolicy, GaussianRasterizationSettings, GaussianRasterizer,
                                                               binning_hint)
    from instantsplat_amd.synthetic import syn_blob
    from tests.util import settings_for
    dev = torch.device(dev)
    bg = torch.zeros(3)

    def frame(P, seed):
        sc = syn_blob(P, 64, 48, seed=seed, scale_mean=0.05)
        st = settings_for(sc.camera, 0, GaussianRasterizationSettings, bg, device=dev)
        kw = dict(means3D=sc.means3D.to(dev), means2D=torch.zeros(P, 3, device=dev), opacities=torch.sigmoid(sc.opacity_logit).to(dev),
                  shs=sc.shs.to(dev), scales=torch.exp(sc.scaling_logit).to(dev), rotations=sc.rotation.to(dev))
        return lambda: GaussianRasterizer(st)(**kw)

    a, b = frame(900, 1), frame(300, 2)
    try:
        BinningPolicy.reset("exact")
        with torch.no_grad():
            with binning_hint("a"):
                a()
            with binning_hint("b"):
                b()
            ra, rb = BinningPolicy.known["a"], BinningPolicy.known["b"]
            assert ra != rb and ra > 0 and rb > 0
            BinningPolicy.mode = "bounded"
            BinningPolicy.known["a"] = ra + 7          # a stale number: poll() must replace it by the frame's true count
            with binning_hint("a", tag="pending"):
                a()                                     # queued, count not read
            assert len(BinningPolicy.pending) == 1
            BinningPolicy.mode = "exact"
            for _ in range(2 * COUNT_RING + 3 if dev.type == "cuda" else 3):   # unhinted exact forwards of ANOTHER scene walk the whole ring twice
                b()
            assert BinningPolicy.poll(block=True) == [] and BinningPolicy.known["a"] == ra and not BinningPolicy.pending
    finally:
        BinningPolicy.reset("exact")


def check_speculative_stage2_overflow_is_rerendered(dev):
    """The compiled nodes enqueue stage 2 of a forward before the frame's count has arrived, in buffers sized from the previous
    frame of the same shape (1.5 x + 16384).  A frame that outgrows that guess must be projected and rendered again with exact
    buffers: same image, radii and gradients as the ctypes binding (which always sizes exactly)."""
    from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy, GaussianRasterizationSettings, GaussianRasterizer
    from instantsplat_amd.synthetic import syn_blob
    from tests.ops_util import _with_binding
    from tests.util import settings_for
    dev = torch.device(dev)
    P, W, H = 6000, 160, 128
    bg = torch.zeros(3)

    def run(scale_mean, seed):
        sc = syn_blob(P, W, H, seed=seed, scale_mean=scale_mean)
        st = settings_for(sc.camera, 0, GaussianRasterizationSettings, bg, device=dev)
        means = sc.means3D.to(dev).requires_grad_(True)
        color, radii = GaussianRasterizer(st)(means3D=means, means2D=torch.zeros(P, 3, device=dev, requires_grad=True),
                                              opacities=torch.sigmoid(sc.opacity_logit).to(dev), shs=sc.shs.to(dev),
                                              scales=torch.exp(sc.scaling_logit).to(dev), rotations=sc.rotation.to(dev))
        color.sum().backward()
        return color.detach().cpu(), radii.cpu(), means.grad.detach().cpu()

    res = {}
    try:
        for binding in ("ctypes", "compiled"):
            with _with_binding(binding):
                BinningPolicy.reset("exact")
                small = run(0.004, 1)       # a few thousand instances: the hint for the next frame of this shape
                big = run(0.25, 2)          # > 20 x as many: far beyond 1.5 x + 16384
                res[binding] = (small, big)
        from instantsplat_amd import diff_gaussian_rasterization as dgr
        counts = [v for k, v in dgr._LAST_COUNT.items() if k[:3] == (P, W, H)]
        assert counts and max(counts) > 1.5 * 3000 + 16384 + 3000, counts   # the second frame did outgrow any guess from the first
        cuda = dev.type == "cuda"
        for i in (0, 1):
            a, b = res["ctypes"][i], res["compiled"][i]
            assert bool((a[1] == b[1]).all())
            assert float((a[0] - b[0]).abs().max()) <= (1e-5 if cuda else 0.0)
            assert float((a[2] - b[2]).norm() / (a[2].norm() + 1e-30)) <= (1e-5 if cuda else 0.0)
    finally:
        BinningPolicy.reset("exact")


def check_deterministic_toggle_between_forward_and_backward_is_refused(dev):
    """the deterministic-backward mode is a process-wide switch that enters the layout of a frame's `binning` buffer
    (+ 52 B per instance of rows and row indices).  Switched ON between a frame's forward and its backward, the backward would
    write those rows past the end of a buffer laid out without them: both bindings refuse from the buffer sizes, on the host,
    before anything is enqueued.  Switched OFF in between is harmless (the buffer is the larger one) and must keep working."""
    import pytest
    import torch
    from instantsplat_amd import _lib
    from instantsplat_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, set_deterministic
    from instantsplat_amd.synthetic import syn_blob
    from tests.util import settings_for
    sc = syn_blob(400, 64, 48, seed=5, scale_mean=0.08)
    was_binding = _lib.BINDING
    try:
        for binding in ("compiled", "ctypes") if _lib.compiled() is not None else ("ctypes",):
            _lib.BINDING = binding
            for first, second, refused in ((False, True, True), (True, False, False), (True, True, False)):
                set_deterministic(first)
                leaves = [t.to(dev).requires_grad_(True) for t in (sc.means3D, sc.scaling_logit, sc.rotation, sc.opacity_logit, sc.shs)]
                st = settings_for(sc.camera, 1, GaussianRasterizationSettings, sc.bg, device=dev)
                color, _ = GaussianRasterizer(st)(means3D=leaves[0], means2D=torch.zeros(400, 3, device=dev, requires_grad=True),
                                                  opacities=torch.sigmoid(leaves[3]), shs=leaves[4], scales=torch.exp(leaves[1]), rotations=leaves[2])
                set_deterministic(second)
                if refused:
                    with pytest.raises(RuntimeError, match="deterministic-backward mode was switched on"):
                        color.sum().backward()
                else:
                    color.sum().backward()
                    assert all(t.grad is not None and bool(torch.isfinite(t.grad).all()) for t in leaves)
    finally:
        set_deterministic(False)
        _lib.BINDING = was_binding
