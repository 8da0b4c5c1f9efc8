"""Drop-in for the third-party `simple_knn` package the reference imports (scene/Gaussians.py:8)."""
