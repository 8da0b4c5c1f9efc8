"""Drop-in import surface: `from gaus_2dgs_rasterization import GaussianRasterizationSettings, GaussianRasterizer`
(render/render_2dgs.py:3-4 of the reference) resolves to the MI355X-native implementation."""
from gaus_slam_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                      _RasterizeGaussians, rasterize_gaussians_apply as rasterize_gaussians)
from . import _C  # noqa: F401
