from gaussiangrasper_amd.ops import NDRasterizeGaussians  # noqa: F401
