"""ORACLE — TEST INFRASTRUCTURE.  Exact 3-NN mean squared distance, the contract of
simple_knn._C.distCUDA2 as used at reference scene/gaussian_model.py:156 (operator source un-vendored;
"exact 3 nearest OTHER points, mean of squared distances" is independently checkable). float64 k-d tree."""
import numpy as np
import torch
from scipy.spatial import cKDTree


def dist2(points: torch.Tensor) -> torch.Tensor:
    p = points.detach().cpu().double().numpy()
    n = p.shape[0]
    if n == 0:
        return torch.zeros(0)
    k = min(4, n)
    d, idx = cKDTree(p).query(p, k=k)
    d = np.atleast_2d(d).reshape(n, k)
    idx = np.atleast_2d(idx).reshape(n, k)
    out = np.zeros(n)
    for i in range(n):  # drop self by index (duplicates at distance 0 stay)
        keep = [d[i, j] for j in range(k) if idx[i, j] != i]
        if len(keep) == k:
            keep = sorted(d[i])[1:]
        keep = (keep + [0.0, 0.0, 0.0])[:3]
        out[i] = sum(x * x for x in keep) / 3.0
    return torch.from_numpy(out).float()
