"""Parameter groups with the names and defaults of reference arguments/__init__.py:47-94
(ModelParams / PipelineParams / OptimizationParams) — only the fields the train/render hot path reads."""
from dataclasses import dataclass


@dataclass
class ModelParams:
    sh_degree: int = 3
    white_background: bool = False
    n_views: int = 3


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
    random_background: bool = False
    pp_optimizer: bool = False
    optim_pose: bool = False
