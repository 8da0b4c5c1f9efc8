"""Mirror of the reference extension module `gaus_2dgs_rasterization._C` (RAST/ext.cpp:15-19)."""
from gaus_slam_amd.rasterizer import (mark_visible, rasterize_gaussians,  # noqa: F401
                                      rasterize_gaussians_backward)
