"""On-disk contract between InstantSplat's init stage and the train/render hot path,
implemented without `plyfile` (not installed):

  sparse_<n>/0/cameras.txt, images.txt   COLMAP text, PINHOLE only     (reference scene/colmap_loader.py:159-183,248-276)
  sparse_<n>/0/points3D.ply              x y z nx ny nz (f4) + r g b (u1)  (reference utils/sfm_utils.py:495-510,
                                                                         read at scene/dataset_readers.py:214-220)
  confidence_dsp.npy                     [N,1] float                    (reference train.py:63-85,95-96)
  point_cloud/iteration_<it>/point_cloud.ply   trained Gaussians, raw (pre-activation) values, f4:
        x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3, f_rest stored channel-major
                                                                        (reference scene/gaussian_model.py:247-326)
  pose/ours_<it>/pose_optimized.npy      [V,4,4] world-to-camera ordered by COLMAP id  (reference train.py:46-60)
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------ COLMAP text
@dataclass
class ColmapCamera:
    id: int
    model: str
    width: int
    height: int
    params: np.ndarray  # fx fy cx cy


@dataclass
class ColmapImage:
    id: int
    qvec: np.ndarray  # (w,x,y,z) world->camera
    tvec: np.ndarray
    camera_id: int
    name: str


def qvec2rotmat(q) -> np.ndarray:
    w, x, y, z = q
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def _data_lines(path):
    with open(path, "r") as f:
        for line in f:
            yield line.rstrip("\n")


def read_cameras_text(path) -> Dict[int, ColmapCamera]:
    cams = {}
    for line in _data_lines(path):
        s = line.strip()
        if not s or s[0] == "#":
            continue
        e = s.split()
        if e[1] not in ("PINHOLE", "SIMPLE_PINHOLE"):   # what reference scene/dataset_readers.py:129-138 takes without undistorting
            raise ValueError("only PINHOLE / SIMPLE_PINHOLE cameras are supported on this path (as in the reference loader)")
        cams[int(e[0])] = ColmapCamera(int(e[0]), e[1], int(e[2]), int(e[3]), np.array([float(v) for v in e[4:]]))
    return cams


def read_images_text(path) -> Dict[int, ColmapImage]:
    imgs = {}
    lines = list(_data_lines(path))
    i = 0
    while i < len(lines):
        s = lines[i].strip()
        i += 1
        if not s or s[0] == "#":
            continue
        e = s.split()
        imgs[int(e[0])] = ColmapImage(int(e[0]), np.array([float(v) for v in e[1:5]]), np.array([float(v) for v in e[5:8]]),
                                      int(e[8]), e[9])
        i += 1  # the 2-D points line that follows every image line (empty in InstantSplat's export)
    return imgs


def write_cameras_text(path, cams: Dict[int, ColmapCamera]):
    with open(path, "w") as f:
        f.write("# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n")
        f.write(f"# Number of cameras: {len(cams)}\n")
        for c in cams.values():
            f.write(" ".join([str(c.id), c.model, str(c.width), str(c.height)] + [repr(float(p)) for p in c.params]) + "\n")


def write_images_text(path, imgs: Dict[int, ColmapImage]):
    with open(path, "w") as f:
        f.write("# Image list with two lines of data per image:\n#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME\n")
        f.write("#   POINTS2D[] as (X, Y, POINT3D_ID)\n")
        f.write(f"# Number of images: {len(imgs)}\n")
        for im in imgs.values():
            f.write(" ".join([str(im.id)] + [repr(float(v)) for v in im.qvec] + [repr(float(v)) for v in im.tvec]
                             + [str(im.camera_id), im.name]) + "\n\n")


def camera_fovs(cam: ColmapCamera) -> Tuple[float, float]:
    """PINHOLE intrinsics -> (FoVx, FoVy) as reference scene/dataset_readers.py:129-134 (focal2fov)."""
    fx, fy = cam.params[0], (cam.params[0] if cam.model == "SIMPLE_PINHOLE" else cam.params[1])
    return 2 * math.atan(cam.width / (2 * fx)), 2 * math.atan(cam.height / (2 * fy))


def image_w2c(im: ColmapImage) -> np.ndarray:
    w2c = np.eye(4)
    w2c[:3, :3] = qvec2rotmat(im.qvec)
    w2c[:3, 3] = im.tvec
    return w2c


# ------------------------------------------------------------------------------------------------ PLY (binary LE)
_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "uchar": "u1", "uint8": "u1", "int": "<i4", "int32": "<i4",
              "uint": "<u4", "short": "<i2", "ushort": "<u2", "char": "i1"}
_PLY_NAMES = {"<f4": "float", "<f8": "double", "u1": "uchar", "<i4": "int", "<u4": "uint", "<i2": "short", "<u2": "ushort", "i1": "char"}


def read_ply_vertices(path) -> np.ndarray:
    """Structured array of the `vertex` element (binary_little_endian or ascii)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, count, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("unterminated PLY header")
            t = line.decode("ascii", "replace").split()
            if not t:
                continue
            if t[0] == "format":
                fmt = t[1]
            elif t[0] == "element":
                in_vertex = t[1] == "vertex"
                if in_vertex:
                    count = int(t[2])
            elif t[0] == "property" and in_vertex:
                if t[1] == "list":
                    raise ValueError("list properties on the vertex element are not supported")
                props.append((t[2], _PLY_TYPES[t[1]]))
            elif t[0] == "end_header":
                break
        dt = np.dtype(props)
        if fmt == "binary_little_endian":
            return np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count).copy()
        if fmt == "ascii":
            rows = [tuple(f.readline().split()) for _ in range(count)]
            return np.array([tuple(np.dtype(ty).type(v) for v, (_, ty) in zip(r, props)) for r in rows], dtype=dt)
        raise ValueError(f"unsupported PLY format {fmt}")


def write_ply_vertices(path, arr: np.ndarray):
    with open(path, "wb") as f:
        hdr = ["ply", "format binary_little_endian 1.0", f"element vertex {arr.shape[0]}"]
        for name in arr.dtype.names:
            hdr.append(f"property {_PLY_NAMES[arr.dtype[name].str.replace('|', '')]} {name}")
        hdr.append("end_header")
        f.write(("\n".join(hdr) + "\n").encode("ascii"))
        f.write(arr.tobytes())


def read_point_cloud_ply(path):
    """points3D.ply -> (xyz [N,3] float32, rgb [N,3] float32 in [0,1])."""
    v = read_ply_vertices(path)
    xyz = np.stack([v["x"], v["y"], v["z"]], axis=1).astype(np.float32)
    rgb = np.stack([v["red"], v["green"], v["blue"]], axis=1).astype(np.float32) / 255.0
    return torch.from_numpy(xyz), torch.from_numpy(rgb)


def write_point_cloud_ply(path, xyz: torch.Tensor, rgb01: torch.Tensor):
    n = xyz.shape[0]
    dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"),
                   ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    a = np.zeros(n, dtype=dt)
    p = xyz.detach().cpu().numpy()
    c = (rgb01.detach().cpu().numpy() * 255.0).round().clip(0, 255).astype(np.uint8)
    a["x"], a["y"], a["z"] = p[:, 0], p[:, 1], p[:, 2]
    a["red"], a["green"], a["blue"] = c[:, 0], c[:, 1], c[:, 2]
    write_ply_vertices(path, a)


def gaussian_ply_attributes(n_dc: int = 3, n_rest: int = 45) -> List[str]:
    return (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)]
            + ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])


def save_gaussian_ply(path, g):
    """g: object with _xyz [P,3], _features_dc [P,1,3], _features_rest [P,K,3], _opacity [P,1], _scaling [P,3], _rotation [P,4]."""
    c = lambda t: t.detach().cpu().numpy().astype(np.float32)
    xyz = c(g._xyz)
    f_dc = c(g._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
    f_rest = c(g._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
    cols = np.concatenate([xyz, np.zeros_like(xyz), f_dc, f_rest, c(g._opacity), c(g._scaling), c(g._rotation)], axis=1)
    names = gaussian_ply_attributes(f_dc.shape[1], f_rest.shape[1])
    a = np.zeros(xyz.shape[0], dtype=np.dtype([(n, "<f4") for n in names]))
    for i, n in enumerate(names):
        a[n] = cols[:, i]
    write_ply_vertices(path, a)


def load_gaussian_ply(path, max_sh_degree: int = 3, device="cpu") -> dict:
    v = read_ply_vertices(path)
    n = v.shape[0]
    col = lambda names: np.stack([v[k] for k in names], axis=1).astype(np.float32)
    rest_names = sorted([k for k in v.dtype.names if k.startswith("f_rest_")], key=lambda s: int(s.split("_")[-1]))
    if len(rest_names) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError("f_rest count does not match the SH degree")
    scale_names = sorted([k for k in v.dtype.names if k.startswith("scale_")], key=lambda s: int(s.split("_")[-1]))
    rot_names = sorted([k for k in v.dtype.names if k.startswith("rot")], key=lambda s: int(s.split("_")[-1]))
    T = lambda a: torch.from_numpy(a).to(device)
    f_dc = col(["f_dc_0", "f_dc_1", "f_dc_2"]).reshape(n, 3, 1)
    f_rest = col(rest_names).reshape(n, 3, (max_sh_degree + 1) ** 2 - 1)
    return dict(_xyz=T(col(["x", "y", "z"])), _features_dc=T(f_dc).transpose(1, 2).contiguous(),
                _features_rest=T(f_rest).transpose(1, 2).contiguous(), _opacity=T(col(["opacity"])),
                _scaling=T(col(scale_names)), _rotation=T(col(rot_names)))


# ------------------------------------------------------------------------------------------------ poses
def save_pose(path, quat_pose: torch.Tensor, colmap_ids: List[int]):
    """[V,7] poses -> [V,4,4] world-to-camera ordered by COLMAP id 1..V (reference train.py:46-60)."""
    from .pose_utils import get_camera_from_tensor
    w2c = [get_camera_from_tensor(q) for q in quat_pose.detach().cpu()]
    ordered = [w2c[colmap_ids.index(i + 1)] for i in range(len(colmap_ids))]
    np.save(path, torch.stack(ordered).numpy())


def save_time(model_path, process_name: str, seconds: float):
    """One line of the stage log the reference's pipeline keeps next to a model (utils/sfm_utils.py:43-50; train.py:217,231 append
    '[2] train_joint_TrainTime' and '[2] train_joint'): `<name>: <m> min <s> sec`, appended to <model_path>/train_time.txt."""
    os.makedirs(model_path, exist_ok=True)
    minutes, secs = divmod(seconds, 60)
    with open(os.path.join(model_path, "train_time.txt"), "a") as f:
        f.write(f"{process_name}: {int(minutes)} min {int(secs)} sec\n")
