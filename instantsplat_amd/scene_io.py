"""A trainable scene from the on-disk layout InstantSplat's init stage leaves behind.

What the reference does between `init_geo.py`'s output directory and the first training iteration, composed from
`io_formats` (the file formats) — and nothing else: no MASt3R, no dataset conventions beyond this layout.

    <source_path>/sparse_<n>/0/{cameras.txt, images.txt, points3D.ply, confidence_dsp.npy}     (sparse_<n>/1/ for --eval poses)
    <source_path>/images/<name>

  read_colmap_scene_info   reference scene/dataset_readers.py:315-369 (readColmapSceneInfo) + :106-160 (readColmapCameras):
                           one CameraInfo per images.txt entry, R = qvec2rotmat(q)^T, FoV from the PINHOLE focals and the
                           cameras.txt size, cameras sorted by image name, points / colours from points3D.ply (:214-220)
  get_nerfpp_norm          :50-73 — `radius` = 1.1 x the largest distance of a camera centre from their mean: `cameras_extent`,
                           the scale of the position learning rate (scene/__init__.py:71,94)
  load_cam                 utils/camera_utils.py:21-54 + utils/general_utils.py:21-27 (PILtoTorch): the image at `-r <resolution>`
                           as float32 [3,H,W] in [0,1]; width / height of the camera come from the IMAGE, FoV from cameras.txt
  scale_from_view_depth    utils/graphics_utils.py:107-135 (`--init_scale_from_view_depth`, off in the reference's scripts)
  load_cameras             scene/dataset_readers.py:75-104 (loadCameras): stored poses (pose_optimized.npy, an interpolated path) back
                           into the cameras, as render.py:206-207,235-236 does before it renders with them
  load_init_scene          scene/__init__.py:28-101 (Scene.__init__): input.ply + cameras.json into the model directory, the
                           seeded shuffle of the training cameras (uid = position after it), cameras on the device, and the
                           per-point learning-rate multipliers of train.py:63-85,95-96
  write_init_scene         the inverse, for tests and for exporting a synthetic scene: the same files utils/sfm_utils.py:202-316
                           writes (text models, points3D.ply, confidence_dsp.npy) + images/*.png

Host-side only: nothing here launches a kernel; `train.setup_training()` takes the result to the device path.
"""
from __future__ import annotations

import json
import os
import random
import shutil
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import io_formats as iof
from .camera import Camera, focal2fov, fov2focal


@dataclass
class CameraInfo:
    """reference scene/dataset_readers.py:29-39"""
    uid: int                 # the COLMAP camera id (becomes Camera.colmap_id)
    R: np.ndarray            # [3,3] camera-to-world rotation = qvec2rotmat(q)^T
    T: np.ndarray            # [3] world-to-camera translation
    FovY: float
    FovX: float
    image: object            # PIL.Image
    image_path: str
    image_name: str
    width: int
    height: int


@dataclass
class SceneInfo:
    """reference scene/dataset_readers.py:42-49"""
    points: np.ndarray       # [N,3] float32
    colors: np.ndarray       # [N,3] in [0,1]
    normals: np.ndarray
    train_cameras: List[CameraInfo]
    test_cameras: List[CameraInfo]
    nerf_normalization: dict
    ply_path: str
    train_poses: list        # [4,4] blocks [[R, T], [0, 1]] in camera order (R as stored: transposed)
    test_poses: list


@dataclass
class InitScene:
    """What reference Scene.__init__ + train.py:95-96 leave in memory before `training_setup`."""
    source_path: str
    n_views: int
    cameras: List[Camera]                 # training cameras on `device`, uid = position in this list
    test_cameras: List[Camera]
    cameras_extent: float
    points: torch.Tensor                  # [N,3] float32 (host)
    colors: torch.Tensor                  # [N,3] float32 in [0,1] (host)
    confidence_lr: Optional[torch.Tensor]  # [N,1] per-point LR multipliers on `device` (None without confidence_dsp.npy)
    scale_gaussian: Optional[np.ndarray] = None
    info: Optional[SceneInfo] = None
    rng: random.Random = field(default_factory=lambda: random.Random(0))   # the stream the reference's `random` module is at

    @property
    def gt_images(self):
        return [c.original_image for c in self.cameras]


def get_world2view2(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0) -> np.ndarray:
    """reference utils/graphics_utils.py:38-49 — incl. the round trip through the inverse in float64 before the float32 cast"""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R).transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + translate) * scale
    return np.float32(np.linalg.inv(C2W))


def get_nerfpp_norm(cam_infos) -> dict:
    centers = [np.linalg.inv(get_world2view2(c.R, c.T))[:3, 3:4] for c in cam_infos]
    centers = np.hstack(centers)
    center = np.mean(centers, axis=1, keepdims=True)
    diagonal = np.max(np.linalg.norm(centers - center, axis=0, keepdims=True))
    return {"translate": -center.flatten(), "radius": diagonal * 1.1}


def read_colmap_cameras(extrinsics: dict, intrinsics: dict, images_folder: str):
    from PIL import Image
    infos, poses = [], []
    for key in extrinsics:
        extr = extrinsics[key]
        intr = intrinsics[extr.camera_id]
        R = np.transpose(iof.qvec2rotmat(extr.qvec))
        T = np.array(extr.tvec)
        poses.append(np.block([[R, T.reshape(3, 1)], [np.zeros((1, 3)), 1]]))
        image_path = os.path.join(images_folder, os.path.basename(extr.name))
        image_name = os.path.basename(image_path).split(".")[0]
        if intr.model == "SIMPLE_PINHOLE":     # reference :129-132: one focal for both axes
            fovy, fovx = focal2fov(intr.params[0], intr.height), focal2fov(intr.params[0], intr.width)
        elif intr.model == "PINHOLE":          # :133-137 (what InstantSplat's init writes)
            fovy, fovx = focal2fov(intr.params[1], intr.height), focal2fov(intr.params[0], intr.width)
        else:
            raise ValueError("Colmap camera model not handled: only undistorted datasets (PINHOLE or SIMPLE_PINHOLE cameras) supported")
        infos.append(CameraInfo(uid=intr.id, R=R, T=T, FovY=fovy, FovX=fovx, image=Image.open(image_path), image_path=image_path,
                                image_name=image_name, width=intr.width, height=intr.height))
    return infos, poses


def read_colmap_scene_info(path: str, images: Optional[str], eval: bool, n_views: int) -> SceneInfo:
    sub = "1" if eval else "0"
    extr = iof.read_images_text(os.path.join(path, f"sparse_{n_views}/{sub}", "images.txt"))
    intr = iof.read_cameras_text(os.path.join(path, f"sparse_{n_views}/{sub}", "cameras.txt"))
    unsorted_infos, poses = read_colmap_cameras(extr, intr, os.path.join(path, "images" if images is None else images))
    order = sorted(range(len(unsorted_infos)), key=lambda i: unsorted_infos[i].image_name)
    cam_infos = [unsorted_infos[i] for i in order]
    sorted_poses = [poses[i] for i in order]
    ply_path = os.path.join(path, f"sparse_{n_views}/0/points3D.ply")   # (also under --eval: the points live in /0)
    v = iof.read_ply_vertices(ply_path)
    points = np.vstack([v["x"], v["y"], v["z"]]).T
    colors = np.vstack([v["red"], v["green"], v["blue"]]).T / 255.0
    normals = np.vstack([v["nx"], v["ny"], v["nz"]]).T
    return SceneInfo(points=points, colors=colors, normals=normals, train_cameras=cam_infos, test_cameras=cam_infos if eval else [],
                     nerf_normalization=get_nerfpp_norm(cam_infos), ply_path=ply_path, train_poses=sorted_poses,
                     test_poses=sorted_poses if eval else [])


def pil_to_torch(pil_image, resolution) -> torch.Tensor:
    resized = torch.from_numpy(np.array(pil_image.resize(resolution))) / 255.0
    return resized.permute(2, 0, 1) if resized.dim() == 3 else resized.unsqueeze(dim=-1).permute(2, 0, 1)


def image_resolution(orig_w: int, orig_h: int, resolution, resolution_scale: float = 1.0):
    """reference utils/camera_utils.py:22-42: `-r 1|2|4|8` divides, `-r -1` caps the width at 1600, any other value is a target width"""
    if resolution in [1, 2, 4, 8]:
        return round(orig_w / (resolution_scale * resolution)), round(orig_h / (resolution_scale * resolution))
    if resolution == -1:
        global_down = orig_w / 1600 if orig_w > 1600 else 1
    else:
        global_down = orig_w / resolution
    scale = float(global_down) * float(resolution_scale)
    return int(orig_w / scale), int(orig_h / scale)


def load_cam(info: CameraInfo, uid: int, resolution=1, resolution_scale: float = 1.0, device="cuda") -> Camera:
    img = pil_to_torch(info.image, image_resolution(*info.image.size, resolution, resolution_scale))
    # (the reference looks for an alpha channel with `shape[1] == 4` — the image HEIGHT — so RGBA inputs lose their alpha without
    # being masked by it; only the first three channels are ever used)
    gt = img[:3, ...].to(torch.float32)
    cam = Camera(uid, torch.from_numpy(get_world2view2(info.R, info.T)), info.FovX, info.FovY, int(gt.shape[2]), int(gt.shape[1]),
                 image=gt, device=device, colmap_id=info.uid, image_name=info.image_name)
    cam.R, cam.T = info.R, info.T
    return cam


def camera_to_json(idx: int, info: CameraInfo) -> dict:
    """reference utils/camera_utils.py:65-85 (an entry of cameras.json)"""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = info.R.transpose()
    Rt[:3, 3] = info.T
    Rt[3, 3] = 1.0
    W2C = np.linalg.inv(Rt)
    return {"id": idx, "img_name": info.image_name, "width": info.width, "height": info.height, "position": W2C[:3, 3].tolist(),
            "rotation": [x.tolist() for x in W2C[:3, :3]], "fy": fov2focal(info.FovY, info.height), "fx": fov2focal(info.FovX, info.width)}


def scale_from_view_depth(points: np.ndarray, extrins: np.ndarray, intrins) -> np.ndarray:
    """reference utils/graphics_utils.py:107-135: per-point scale cap = smallest depth over the views / mean focal — with the
    reference's own conventions: `extrins` are the inverses of the stored pose blocks, and the focal pair of the LAST view is
    used for every point (the loop variables outlive the loop)."""
    depth_z = []
    for extrin, intrin in zip(extrins, intrins):
        R, t = extrin[:3, :3], extrin[:3, 3]
        depth_z.append((R @ points.T + t[:, np.newaxis])[2, :])
        fx, fy = intrin
    depth_z = np.min(np.array(depth_z), 0)
    depth_z = np.clip(depth_z, 0.01, depth_z.max())
    return depth_z / ((fx + fy) / 2)


def load_confidence_lr(source_path: str, n_views: int, device, scale=(1.0, 100.0)) -> Optional[torch.Tensor]:
    from .scene import confidence_to_lr_modifiers
    p = os.path.join(source_path, f"sparse_{n_views}/0", "confidence_dsp.npy")
    if not os.path.exists(p):
        return None
    return confidence_to_lr_modifiers(torch.from_numpy(np.load(p)).float().to(device), scale=scale)


def load_init_scene(source_path: str, n_views: int, images: Optional[str] = None, eval: bool = False, resolution=1, shuffle: bool = True,
                    device="cuda", model_path: Optional[str] = None, init_scale_from_view_depth: bool = False,
                    rng: Optional[random.Random] = None) -> InitScene:
    """Everything reference train.py:90-97 has in hand before `training_setup`, from the init directory.  `rng` stands for the
    reference's process-wide `random` module (seeded 0 by `safe_state`): the camera shuffle draws from it first, the training
    loop's view sampling continues on the same stream (`InitScene.rng`)."""
    rng = rng if rng is not None else random.Random(0)
    if not os.path.exists(os.path.join(source_path, f"sparse_{n_views}")):
        raise FileNotFoundError(f"Could not recognize scene type: {source_path}/sparse_{n_views} does not exist")
    info = read_colmap_scene_info(source_path, images, eval, n_views)
    if model_path:   # scene/__init__.py:52-65
        os.makedirs(model_path, exist_ok=True)
        shutil.copyfile(info.ply_path, os.path.join(model_path, "input.ply"))
        with open(os.path.join(model_path, "cameras.json"), "w") as f:
            json.dump([camera_to_json(i, c) for i, c in enumerate(list(info.test_cameras) + list(info.train_cameras))], f)
    # reference scene/__init__.py:67-69 shuffles scene_info.train_cameras, then scene_info.test_cameras.  Under --eval the reader
    # returns ONE list object as both (dataset_readers.py:343-346), so that list is shuffled twice and both camera lists are built
    # from the twice-shuffled order (uid, the pose-table row and the view sampling follow it).
    if info.test_cameras is info.train_cameras:
        train_infos = test_infos = list(info.train_cameras)
        if shuffle:
            rng.shuffle(train_infos)
            rng.shuffle(train_infos)
    else:
        train_infos, test_infos = list(info.train_cameras), list(info.test_cameras)
        if shuffle:
            rng.shuffle(train_infos)
            rng.shuffle(test_infos)
    extent = float(info.nerf_normalization["radius"])
    cams = [load_cam(c, i, resolution, 1.0, device) for i, c in enumerate(train_infos)]
    test_cams = [load_cam(c, i, resolution, 1.0, device) for i, c in enumerate(test_infos)]
    scale_gaussian = None
    if init_scale_from_view_depth:
        scale_gaussian = scale_from_view_depth(info.points, np.linalg.inv(info.train_poses),
                                               [[fov2focal(c.FovX, c.width), fov2focal(c.FovY, c.height)] for c in train_infos])
    return InitScene(source_path=source_path, n_views=n_views, cameras=cams, test_cameras=test_cams, cameras_extent=extent,
                     points=torch.from_numpy(np.ascontiguousarray(info.points)).float(),
                     colors=torch.from_numpy(np.ascontiguousarray(info.colors)).float(),
                     confidence_lr=load_confidence_lr(source_path, n_views, device), scale_gaussian=scale_gaussian, info=info, rng=rng)


def load_cameras(poses: np.ndarray, cameras: List[Camera]) -> List[Camera]:
    """reference scene/dataset_readers.py:75-104 (`loadCameras`): give the cameras the stored world-to-camera matrices — the
    optimised poses of `pose/ours_<it>/pose_optimized.npy` ([V,4,4], one per camera: in place) or a longer interpolated path
    ([N > V,4,4]: the camera list is repeated to N copies, renumbered uid 0.., colmap_id 1.., image_name "00000".. like the
    reference's).  Sets R, T, world_view_transform, full_proj_transform, camera_center."""
    import copy
    poses = np.asarray(poses)

    def place(cam, m):
        R, T = np.transpose(m[:3, :3]), m[:3, 3]
        dev = cam.world_view_transform.device
        cam.R, cam.T = R, T
        cam.world_view_transform = torch.tensor(get_world2view2(R, T)).transpose(0, 1).to(dev)
        cam.full_proj_transform = cam.world_view_transform.unsqueeze(0).bmm(cam.projection_matrix.unsqueeze(0)).squeeze(0)
        cam.camera_center = cam.world_view_transform.inverse()[3, :3]

    if poses.shape[0] == len(cameras):
        for cam, m in zip(cameras, poses):
            place(cam, m)
    elif poses.shape[0] > len(cameras):
        repeat = int(np.ceil(poses.shape[0] / len(cameras)))
        cameras = [copy.deepcopy(c) for c in cameras * repeat][:poses.shape[0]]
        for idx, (cam, m) in enumerate(zip(cameras, poses)):
            cam.uid, cam.colmap_id, cam.image_name = idx, idx + 1, str(idx).zfill(5)
            place(cam, m)
    return cameras


# ---------------------------------------------------------------------------------------------------- writer (tests / export)
def rotmat2qvec(R: np.ndarray) -> np.ndarray:
    """COLMAP's rotation matrix -> (w,x,y,z), largest-eigenvector form (the convention the layout's writer uses)."""
    Rxx, Ryx, Rzx, Rxy, Ryy, Rzy, Rxz, Ryz, Rzz = np.asarray(R, dtype=np.float64).flat
    K = np.array([[Rxx - Ryy - Rzz, 0, 0, 0], [Ryx + Rxy, Ryy - Rxx - Rzz, 0, 0], [Rzx + Rxz, Rzy + Ryz, Rzz - Rxx - Ryy, 0],
                  [Ryz - Rzy, Rzx - Rxz, Rxy - Ryx, Rxx + Ryy + Rzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    return -q if q[0] < 0 else q


def write_init_scene(source_path: str, w2c: List[np.ndarray], fovs, images: List[torch.Tensor], points: torch.Tensor, colors: torch.Tensor,
                     confidence: Optional[torch.Tensor], names: Optional[List[str]] = None, subdir: str = "0") -> None:
    """The layout of Appendix F for V views: `w2c[v]` [4,4], `fovs[v]` = (FoVx, FoVy), `images[v]` float [3,H,W] in [0,1]
    (stored as 8-bit PNG) or the path of an image file (copied under `names[v]`), points / colours [N,3], confidence [N,1] or None."""
    from PIL import Image
    V = len(w2c)
    names = names or [f"{v:04d}.png" for v in range(V)]
    sparse = os.path.join(source_path, f"sparse_{V}", subdir)
    os.makedirs(sparse, exist_ok=True)
    os.makedirs(os.path.join(source_path, "images"), exist_ok=True)
    cams, imgs = {}, {}
    for v in range(V):
        if isinstance(images[v], (str, os.PathLike)):   # an image FILE (a photograph, a video frame): copied as it is, decoded by the loader
            with Image.open(images[v]) as im:
                W, H = im.size
        else:
            H, W = int(images[v].shape[1]), int(images[v].shape[2])
        m = np.asarray(w2c[v], dtype=np.float64)
        cams[v + 1] = iof.ColmapCamera(v + 1, "PINHOLE", W, H, np.array([fov2focal(fovs[v][0], W), fov2focal(fovs[v][1], H), W / 2, H / 2]))
        imgs[v + 1] = iof.ColmapImage(v + 1, rotmat2qvec(m[:3, :3]), m[:3, 3].copy(), v + 1, names[v])
        if isinstance(images[v], (str, os.PathLike)):
            shutil.copyfile(images[v], os.path.join(source_path, "images", names[v]))
            continue
        arr = (images[v].detach().cpu().clamp(0, 1).permute(1, 2, 0).numpy() * 255.0).round().astype(np.uint8)
        Image.fromarray(arr).save(os.path.join(source_path, "images", names[v]))
    iof.write_cameras_text(os.path.join(sparse, "cameras.txt"), cams)
    iof.write_images_text(os.path.join(sparse, "images.txt"), imgs)
    if subdir == "0":
        iof.write_point_cloud_ply(os.path.join(sparse, "points3D.ply"), points, colors)
        if confidence is not None:
            np.save(os.path.join(sparse, "confidence_dsp.npy"), confidence.detach().cpu().numpy())
