"""GPU tier, BASELINE sizes (200k Gaussians, 512x512): properties that hold at any size, where the dense
oracle would be too slow to be the checker.
  * forward is run-to-run bit-identical (sorted per-tile lists, no atomics in the forward path);
  * backward is linear in dL/dpixel;
  * with a black background the image is linear in precomputed colours (transmittance does not depend on colour);
  * every per-tile list the binning produces is sorted by (depth, index) and its length matches the tile counts."""

import pytest
import torch

from instantsplat_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from instantsplat_amd.synthetic import syn_blob
from tests.util import settings_for

pytestmark = pytest.mark.gpu
P, W, H = 200000, 512, 512


def _scene(gpu, deg=0, seed=0):
    sc = syn_blob(P, W, H, seed=seed)
    st = settings_for(sc.camera, deg, GaussianRasterizationSettings, sc.bg, device=gpu)
    t = dict(means3D=sc.means3D.to(gpu), opacities=torch.sigmoid(sc.opacity_logit).to(gpu), scales=torch.exp(sc.scaling_logit).to(gpu),
             rotations=sc.rotation.to(gpu))
    return sc, st, t


def test_forward_is_bitwise_deterministic(gpu):
    sc, st, t = _scene(gpu, deg=3)
    r = GaussianRasterizer(st)
    with torch.no_grad():
        a, ra = r(means2D=torch.zeros(P, 3, device=gpu), shs=sc.shs.to(gpu), **t)
        b, rb = r(means2D=torch.zeros(P, 3, device=gpu), shs=sc.shs.to(gpu), **t)
    assert torch.equal(a, b) and torch.equal(ra, rb)


def test_backward_is_linear_in_pixel_gradient(gpu):
    sc, st, t = _scene(gpu)
    leaves = {k: v.clone().requires_grad_(True) for k, v in t.items()}
    shs = sc.shs.to(gpu).requires_grad_(True)
    img, _ = GaussianRasterizer(st)(means2D=torch.zeros(P, 3, device=gpu), shs=shs, **leaves)
    g = torch.Generator().manual_seed(5)
    g1, g2 = torch.randn(3, H, W, generator=g).to(gpu), torch.randn(3, H, W, generator=g).to(gpu)
    names = list(leaves) + ["shs"]
    tens = list(leaves.values()) + [shs]
    ga = torch.autograd.grad(img, tens, g1, retain_graph=True)
    gb = torch.autograd.grad(img, tens, g2, retain_graph=True)
    gc = torch.autograd.grad(img, tens, 2.0 * g1 + g2)
    for n, a, b, c in zip(names, ga, gb, gc):
        ref = 2.0 * a + b
        assert float((c - ref).norm() / (ref.norm() + 1e-30)) <= 1e-4, n


def test_image_is_linear_in_precomputed_colours(gpu):
    sc, st, t = _scene(gpu)
    g = torch.Generator().manual_seed(6)
    c1, c2 = torch.rand(P, 3, generator=g).to(gpu), torch.rand(P, 3, generator=g).to(gpu)
    r = GaussianRasterizer(st)
    z = torch.zeros(P, 3, device=gpu)
    with torch.no_grad():
        a = r(means2D=z, colors_precomp=c1, **t)[0]
        b = r(means2D=z, colors_precomp=c2, **t)[0]
        c = r(means2D=z, colors_precomp=c1 + c2, **t)[0]
    assert float((c - (a + b)).abs().max()) <= 2e-5


def test_largest_config_1M_gaussians_1080p(gpu):
    """BASELINE config C4 (1 M Gaussians, 1920x1080, ~7 M instances): forward run-to-run identical, backward linear in
    dL/dpixel and finite — the same size-independent properties, at the largest size the reference quotes."""
    Pn, Wn, Hn = 1000000, 1920, 1080
    sc = syn_blob(Pn, Wn, Hn, seed=3)
    st = settings_for(sc.camera, 0, GaussianRasterizationSettings, sc.bg, device=gpu)
    t = dict(means3D=sc.means3D.to(gpu), opacities=torch.sigmoid(sc.opacity_logit).to(gpu), scales=torch.exp(sc.scaling_logit).to(gpu),
             rotations=sc.rotation.to(gpu))
    r = GaussianRasterizer(st)
    z = torch.zeros(Pn, 3, device=gpu)
    with torch.no_grad():
        a, ra = r(means2D=z, shs=sc.shs.to(gpu), **t)
        b, rb = r(means2D=z, shs=sc.shs.to(gpu), **t)
    assert torch.equal(a, b) and torch.equal(ra, rb)
    assert int((ra > 0).sum()) > Pn // 2
    leaves = {k: v.clone().requires_grad_(True) for k, v in t.items()}
    img, _ = r(means2D=z, shs=sc.shs.to(gpu), **leaves)
    g = torch.Generator().manual_seed(9)
    g1 = torch.randn(3, Hn, Wn, generator=g).to(gpu)
    tens = list(leaves.values())
    ga = torch.autograd.grad(img, tens, g1, retain_graph=True)
    gb = torch.autograd.grad(img, tens, -0.5 * g1)
    for n, x, y in zip(leaves, ga, gb):
        assert bool(torch.isfinite(x).all()), n
        ref = -0.5 * x
        assert float((y - ref).norm() / (ref.norm() + 1e-30)) <= 1e-4, n
