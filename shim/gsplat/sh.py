from gaussiangrasper_amd.constants import deg_from_sh, num_sh_bases  # noqa: F401
from gaussiangrasper_amd.ops import SphericalHarmonics  # noqa: F401
