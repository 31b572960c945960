from gaussiangrasper_amd.ops import quat_to_rotmat  # noqa: F401
