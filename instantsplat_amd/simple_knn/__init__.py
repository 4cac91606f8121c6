"""Drop-in for the `simple_knn` package (reference scene/gaussian_model.py:20: `from simple_knn._C import distCUDA2`)."""
from . import _C  # noqa: F401
