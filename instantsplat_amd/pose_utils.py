"""Pose algebra on the hot path, with the semantics of reference utils/pose_utils.py:10-104,183-215:
quaternions are (w, x, y, z); `get_camera_from_tensor` NORMALISES the quaternion before building the
rotation, `quadmultiply` is the plain Hamilton product (no normalisation)."""
from __future__ import annotations

import torch


def quad2rotation(q: torch.Tensor) -> torch.Tensor:
    """[B,4] -> [B,3,3], quaternion normalised first."""
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q.unbind(dim=1)
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=1).reshape(-1, 3, 3)


def get_camera_from_tensor(inputs: torch.Tensor) -> torch.Tensor:
    """7-vector (quat, t) -> 4x4 world-to-camera."""
    if inputs.dim() == 1:
        inputs = inputs.unsqueeze(0)
    R = quad2rotation(inputs[:, :4])[0]
    top = torch.cat([R, inputs[0, 4:].reshape(3, 1)], dim=1)
    bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=inputs.dtype, device=inputs.device)
    return torch.cat([top, bottom], dim=0)


def quadmultiply(q1: torch.Tensor, q2: torch.Tensor) -> torch.Tensor:
    w1, x1, y1, z1 = q1.unbind(dim=-1)
    w2, x2, y2, z2 = q2.unbind(dim=-1)
    return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], dim=-1)


def rotation2quad(R: torch.Tensor) -> torch.Tensor:
    """[3,3] rotation -> (w,x,y,z) with w >= 0 branch selection by largest diagonal term
    (same convention family as reference utils/pose_utils.py:117-180: best-conditioned candidate)."""
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = R.reshape(9).unbind()
    q_abs = torch.sqrt(torch.clamp(torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22,
                                                1 - m00 + m11 - m22, 1 - m00 - m11 + m22]), min=0.0))
    cand = torch.stack([
        torch.stack([q_abs[0] ** 2, m21 - m12, m02 - m20, m10 - m01]),
        torch.stack([m21 - m12, q_abs[1] ** 2, m10 + m01, m02 + m20]),
        torch.stack([m02 - m20, m10 + m01, q_abs[2] ** 2, m12 + m21]),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[3] ** 2])])
    cand = cand / (2.0 * q_abs[:, None].clamp(min=0.1))
    return cand[int(torch.argmax(q_abs))]


def get_tensor_from_camera(RT: torch.Tensor) -> torch.Tensor:
    """4x4 world-to-camera -> 7-vector (quat, t)."""
    return torch.cat([rotation2quad(RT[:3, :3].detach()), RT[:3, 3].detach()])
