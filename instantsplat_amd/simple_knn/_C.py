"""`simple_knn._C.distCUDA2` — mean squared distance to the 3 nearest other points.
Call site being served: reference scene/gaussian_model.py:156."""
import torch

from .. import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    pts = _lib.f32c(points)
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    dev = _lib.require_device(pts)
    n = pts.shape[0]
    out = torch.zeros(n, dtype=torch.float32, device=dev)
    L = _lib.lib()
    scratch = torch.empty(max(int(L.mi355gs_knn_scratch_bytes(n)), 1), dtype=torch.uint8, device=dev)
    with _lib.on_device(dev):
        _lib.check(L.mi355gs_knn_dist2(_lib.stream_ptr(dev), n, _lib.ptr(pts), _lib.ptr(out), _lib.ptr(scratch)), "knn_dist2")
    return out
