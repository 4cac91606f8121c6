"""Edge cases shared by the CPU (emulated) and GPU tiers."""
import math

import torch

from tests.util import assert_no_worse_than_fp32_oracle, assert_raster_parity, run_custom_case


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
