"""Parameter groups with the names and defaults of reference arguments/__init__.py:47-94
(ModelParams / PipelineParams / OptimizationParams), and the `cfg_args` record a run leaves next to its outputs
(reference train.py:242-246), in the form the reference's own `get_combined_args` (arguments/__init__.py:96-116) evaluates."""
import dataclasses
from argparse import Namespace
from dataclasses import dataclass


@dataclass
class ModelParams:
    sh_degree: int = 3
    source_path: str = ""
    model_path: str = ""
    images: str = "images"
    resolution: int = -1
    white_background: bool = False
    data_device: str = "cuda"
    eval: bool = False
    n_views: int = 3     # (the reference's default is 0: its scripts always pass --n_views; the synthetic scenes here have three)
    init_scale_from_view_depth: bool = False


@dataclass
class PipelineParams:
    convert_SHs_python: bool = False
    compute_cov3D_python: bool = False
    debug: bool = False


@dataclass
class OptimizationParams:
    iterations: int = 30_000
    position_lr_init: float = 0.00016
    position_lr_final: float = 0.0000016
    position_lr_delay_mult: float = 0.01
    position_lr_max_steps: int = 30_000
    feature_lr: float = 0.0025
    opacity_lr: float = 0.05
    scaling_lr: float = 0.005
    rotation_lr: float = 0.001
    percent_dense: float = 0.01
    lambda_dssim: float = 0.2
    # densification is switched off in the reference's loop (train.py:195-206 are comments); the fields exist for its cfg_args
    densification_interval: int = 100
    opacity_reset_interval: int = 3000
    densify_from_iter: int = 500
    densify_until_iter: int = 15_000
    densify_grad_threshold: float = 0.0002
    random_background: bool = False
    pp_optimizer: bool = False
    optim_pose: bool = False


def cfg_args_text(model: ModelParams, opt: OptimizationParams, pipe: PipelineParams, **script_args) -> str:
    """What reference train.py:245-246 writes to <model_path>/cfg_args: `str(Namespace(**vars(args)))` over the three parameter
    groups and the training script's own options — the text the reference's render.py / metrics.py read back with `eval` to find
    the scene a model directory was trained on (source_path, n_views, images, resolution, sh_degree, white_background, eval)."""
    fields = dict(dataclasses.asdict(model))   # (source_path as given: the reference makes it absolute on the extracted group, not here)
    fields.update(dataclasses.asdict(opt))
    fields.update(dataclasses.asdict(pipe))
    # the script's own options with its defaults (train.py:303-312); `save_iterations` always ends with `iterations` (:314)
    fields.update(ip="127.0.0.1", port=6009, debug_from=-1, detect_anomaly=False, test_iterations=[], save_iterations=[opt.iterations],
                  quiet=False, disable_viewer=True, checkpoint_iterations=[], start_checkpoint=None)
    fields.update(script_args)
    return str(Namespace(**fields))
