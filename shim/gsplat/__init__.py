"""Drop-in `gsplat` package (API of gsplat==0.1.0 as far as leejaehot/GaussianGrasper uses it)
backed by gaussiangrasper_amd / libgg_raster.so on MI355X.

Put <repo>/shim and <repo> on PYTHONPATH; the reference's imports
(nerfstudio/models/gaussian_splatting.py:46-50, nerfstudio/scripts/update.py:74) then resolve here:
    gsplat._torch_impl.quat_to_rotmat
    gsplat.nd_rasterize.NDRasterizeGaussians
    gsplat.project_gaussians.ProjectGaussians
    gsplat.rasterize.RasterizeGaussians
    gsplat.sh.SphericalHarmonics, gsplat.sh.num_sh_bases
"""
__version__ = "0.1.0+gg.amd"
