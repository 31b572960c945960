from gaussiangrasper_amd.ops import RasterizeGaussians  # noqa: F401
