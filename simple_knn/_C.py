"""`from simple_knn._C import distCUDA2` (scene/Gaussians.py:8) -> HIP implementation (gaus_slam_amd/csrc/sknn.hip)."""
from gaus_slam_amd.knn import distCUDA2  # noqa: F401
