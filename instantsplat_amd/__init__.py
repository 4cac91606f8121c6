"""instantsplat_amd — MI355X-native (gfx950) implementation of InstantSplat's train/render hot path.

Sub-packages mirror the reference's operator surface for that path:
  diff_gaussian_rasterization, simple_knn, fused_ssim  (drop-in operator packages)
  gaussian_renderer.render                             (reference gaussian_renderer/__init__.py:23)
All compute goes through the C ABI of libmi355gs.so (include/mi355gs.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
