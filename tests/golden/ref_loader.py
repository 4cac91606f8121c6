"""Loads the reference's scene-building stack from /root/reference for the build container's golden generator
(make_golden_initdir.py) and drop-in test: `Scene` (scene/__init__.py), `readColmapSceneInfo` / `getNerfppNorm`
(scene/dataset_readers.py), `loadCam` (utils/camera_utils.py), `Camera`, `GaussianModel`, `render()`, and function
definitions of train.py / utils/sfm_utils.py (files that cannot be imported whole here: torchvision, cv2, open3d, roma ...).

Everything is the reference's own source, executed from its files.  What is substituted, and why:
  * "cuda" device strings / `.cuda()` -> CPU, in memory (no GPU in the build container);
  * `plyfile` -> tests/golden/plyfile_standin.py (not installed);  `matplotlib` -> empty stub (imported, never used on this path);
  * the three operator packages -> whatever the caller passes (the C oracle for goldens, instantsplat_amd's for the drop-in test).
Test infrastructure; never imported by the product."""
import ast
import importlib
import os
import sys
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def cpu(src: str) -> str:
    return src.replace('device="cuda"', 'device="cpu"').replace("device='cuda'", "device='cpu'").replace(".cuda()", "")


def exec_file(path, name, rewrite=True, setitem=None):
    mod = types.ModuleType(name)
    mod.__file__ = None
    src = open(path).read()
    if setitem is not None:
        setitem(sys.modules, name, mod)
    exec(compile(cpu(src) if rewrite else src, path, "exec"), mod.__dict__)
    return mod


def function_sources(path, names=None):
    """{name: source} of the top-level function definitions of a file that cannot be imported as a whole"""
    src = open(path).read()
    return {n.name: ast.get_source_segment(src, n) for n in ast.parse(src).body
            if isinstance(n, ast.FunctionDef) and (names is None or n.name in names)}


def assignment_sources(path, targets):
    src = open(path).read()
    out = []
    for n in ast.parse(src).body:
        if isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id in targets for t in n.targets):
            out.append(ast.get_source_segment(src, n))
    return "\n".join(out)


def load(setitem, delitem, rasterizer_module, knn_module, fused_ssim_module=None):
    """`setitem(mapping, key, value)` / `delitem(mapping, key)`: monkeypatch's in a test, plain dict operations in the generator.
    REF must already be first on sys.path.  Returns the loaded modules."""
    for k in [k for k in list(sys.modules) if k == "utils" or k.startswith("utils.") or k == "arguments" or k == "scene" or k.startswith("scene.")]:
        delitem(sys.modules, k)   # the reference's `utils` / `arguments` / `scene` packages, not anything cached under those names
    sk = types.ModuleType("simple_knn")
    sk._C = knn_module
    for name, mod in (("diff_gaussian_rasterization", rasterizer_module), ("simple_knn", sk), ("simple_knn._C", knn_module)):
        setitem(sys.modules, name, mod)
    if fused_ssim_module is not None:
        setitem(sys.modules, "fused_ssim", fused_ssim_module)
    spec = importlib.util.spec_from_file_location("plyfile", os.path.join(HERE, "plyfile_standin.py"))
    ply = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ply)
    setitem(sys.modules, "plyfile", ply)
    mpl = types.ModuleType("matplotlib")
    mpl.pyplot = types.ModuleType("matplotlib.pyplot")
    setitem(sys.modules, "matplotlib", mpl)
    setitem(sys.modules, "matplotlib.pyplot", mpl.pyplot)
    scene_pkg = types.ModuleType("scene")            # bare package: scene/__init__.py is executed separately below
    scene_pkg.__path__ = [os.path.join(REF, "scene")]
    setitem(sys.modules, "scene", scene_pkg)
    from utils import graphics_utils, loss_utils, pose_utils          # reference
    from arguments import OptimizationParams                          # reference
    gm = exec_file(os.path.join(REF, "scene", "gaussian_model.py"), "scene.gaussian_model", setitem=setitem)
    cm = exec_file(os.path.join(REF, "scene", "cameras.py"), "scene.cameras", setitem=setitem)
    cl = importlib.import_module("scene.colmap_loader")
    dr = exec_file(os.path.join(REF, "scene", "dataset_readers.py"), "scene.dataset_readers", setitem=setitem)
    cu = exec_file(os.path.join(REF, "utils", "camera_utils.py"), "utils.camera_utils", setitem=setitem)
    sc = exec_file(os.path.join(REF, "scene", "__init__.py"), "ref_scene_init", setitem=setitem)
    gr = exec_file(os.path.join(REF, "gaussian_renderer", "__init__.py"), "ref_gaussian_renderer", setitem=setitem)
    return types.SimpleNamespace(gm=gm, cm=cm, cl=cl, dr=dr, cu=cu, Scene=sc.Scene, scene_module=sc, gr=gr, ply=ply, loss_utils=loss_utils,
                                 pose_utils=pose_utils, graphics_utils=graphics_utils, OptimizationParams=OptimizationParams,
                                 train_functions=function_sources(os.path.join(REF, "train.py")))


def sfm_writers(ply):
    """The writers of the init layout (reference utils/sfm_utils.py:202-316,495-510) as executable functions: save_extrinsic,
    save_intrinsics, save_points3D, storePly, around the reference's own colmap_loader text/binary writers."""
    import collections
    from pathlib import Path
    import numpy as np
    cl = importlib.import_module("scene.colmap_loader")
    path = os.path.join(REF, "utils", "sfm_utils.py")
    ns = {"np": np, "Path": Path, "collections": collections, "PlyData": ply.PlyData, "PlyElement": ply.PlyElement,
          "to_numpy": lambda x: None if x is None else np.asarray(x)}
    for k in ("rotmat2qvec", "write_cameras_binary", "write_cameras_text", "write_images_text", "write_images_binary"):
        ns[k] = getattr(cl, k)
    exec(compile(assignment_sources(path, {"CameraModel", "Camera", "BaseImage", "Point3D", "CAMERA_MODELS", "CAMERA_MODEL_IDS",
                                           "CAMERA_MODEL_NAMES"}), path, "exec"), ns)
    for name, src in function_sources(path, {"save_extrinsic", "save_intrinsics", "save_points3D", "storePly"}).items():
        exec(compile(src, path, "exec"), ns)
    return types.SimpleNamespace(**{k: ns[k] for k in ("save_extrinsic", "save_intrinsics", "save_points3D", "storePly")})
