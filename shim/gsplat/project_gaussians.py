from gaussiangrasper_amd.ops import ProjectGaussians  # noqa: F401
