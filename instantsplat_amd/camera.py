"""Per-view camera constants for the hot path.

Mirrors the fields `render()` reads from the reference's Camera (reference scene/cameras.py:17-57:
FoVx, FoVy, image_width, image_height, projection_matrix stored TRANSPOSED, camera_center,
original_image) and the projection convention of reference utils/graphics_utils.py:71-91
(z_sign = +1, znear 0.01, zfar 100).
"""
from __future__ import annotations

import math

import torch

ZNEAR, ZFAR = 0.01, 100.0


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> torch.Tensor:
    """OpenGL-style frustum with +z forward; returned UN-transposed (P @ column-vector)."""
    ty, tx = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def fov2focal(fov: float, pixels: int) -> float:
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal: float, pixels: int) -> float:
    return 2 * math.atan(pixels / (2 * focal))


class Camera:
    """Minimal stand-in for the reference Camera: only what the train/render hot path touches."""

    def __init__(self, uid: int, w2c: torch.Tensor, fovx: float, fovy: float, width: int, height: int,
                 image: torch.Tensor | None = None, device="cpu", colmap_id: int | None = None, image_name: str = ""):
        self.uid = uid
        self.colmap_id = uid + 1 if colmap_id is None else colmap_id
        self.image_name = image_name
        self.FoVx, self.FoVy = fovx, fovy
        self.image_width, self.image_height = width, height
        self.znear, self.zfar = ZNEAR, ZFAR
        self.trans, self.scale = (0.0, 0.0, 0.0), 1.0   # reference :19,42-43 (`getWorld2View2`'s recentring: never used by InstantSplat)
        self.data_device = torch.device(device)          # reference :33-38
        w2c = w2c.to(torch.float32)
        self.world_view_transform = w2c.t().contiguous().to(device)
        self.projection_matrix = projection_matrix(ZNEAR, ZFAR, fovx, fovy).t().contiguous().to(device)
        self.full_proj_transform = self.world_view_transform @ self.projection_matrix
        self.camera_center = torch.linalg.inv(self.world_view_transform)[3, :3]
        self.original_image = None if image is None else image.clamp(0.0, 1.0).to(device)

    def to(self, device):
        """Move the per-view constants to `device` (the reference keeps them on the GPU: scene/cameras.py:54-57)."""
        for name in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center", "original_image"):
            t = getattr(self, name)
            if t is not None:
                setattr(self, name, t.to(device))
        self.data_device = torch.device(device)
        return self
