"""Wire formats either side of the path: the reference's training checkpoints and
its Gaussian-splat PLY export, so scenes trained with the reference can be rendered/benchmarked here
and vice versa.  Host-side Python (the reference's is, too); nothing here touches the GPU.

Checkpoint (nerfstudio/engine/trainer.py:428-456): `step-%09d.ckpt` is `torch.save` of
    {"step", "pipeline": state_dict, "optimizers", "schedulers", "scalers"}
and the splatting model's entries of `pipeline` are `_model.{means,scales,quats,opacities,colors_all,
feature}` plus `_model.fea_up.layers.{0,2}.{weight,bias}` (nerfstudio/models/gaussian_splatting.py:271-281,
258; `load_state_dict` at :301-313 resizes the parameters to the checkpoint's point count).  Files are
read with `torch.load(..., weights_only=True)` only.

PLY (nerfstudio/scripts/exporter.py:481-530, written there through open3d's tensor PointCloud):
per-point properties positions (x, y, z), normals (nx, ny, nz; zeros), colors (uchar red, green,
blue = 255 * f_dc), f_dc_0..2 = `model.colors` = SH2RGB(colors_all[:, 0]) = 0.5 + C0 * dc (this
exporter writes the RGB value, not the raw SH coefficient; gaussian_splatting.py:80-85,294-295),
f_rest_0..3(K-1)-1 (colors_all[:, 1:, :] flattened in (band, rgb) order), opacity (logit),
scale_0..2 (log), rot_0..3 (wxyz).  Written here as binary_little_endian 1.0
with that property order.  PARITY UNPINNED: the tree holds no file produced by the reference and
open3d is not installed, so byte-compatibility with open3d's writer is untested; the round trip and
the header are."""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple, Union

import numpy as np
import torch

from .constants import SH_C0
from .scene import Scene

MODEL_PREFIX = "_model."
PARAM_KEYS = ("means", "scales", "quats", "opacities", "colors_all", "feature")
MLP_KEYS = ("fea_up.layers.0.weight", "fea_up.layers.0.bias", "fea_up.layers.2.weight", "fea_up.layers.2.bias")


# ------------------------------------------------------------------------------------------------
# checkpoints
# ------------------------------------------------------------------------------------------------
def scene_from_state_dict(pipeline_state: Dict[str, torch.Tensor]) -> Tuple[Scene, Dict[str, torch.Tensor]]:
    """(Scene, fea_up state) from the `pipeline` state dict of a reference checkpoint.  The second
    item loads into `gaussiangrasper_amd.mlp.MLP` with `load_state_dict` (keys `layers.*`)."""
    missing = [k for k in PARAM_KEYS if MODEL_PREFIX + k not in pipeline_state]
    if missing:
        raise KeyError(f"not a GaussianGrasper splatting checkpoint: missing {missing}")
    t = {k: pipeline_state[MODEL_PREFIX + k].detach().float().contiguous() for k in PARAM_KEYS}
    n = t["means"].shape[0]
    expect = {"means": (n, 3), "scales": (n, 3), "quats": (n, 4), "opacities": (n, 1)}
    for k, shp in expect.items():
        if tuple(t[k].shape) != shp:
            raise ValueError(f"{k} has shape {tuple(t[k].shape)}, expected {shp}")
    if t["colors_all"].ndim != 3 or t["colors_all"].shape[0] != n or t["colors_all"].shape[2] != 3:
        raise ValueError(f"colors_all has shape {tuple(t['colors_all'].shape)}, expected (N, K, 3)")
    if t["feature"].ndim != 2 or t["feature"].shape[0] != n:
        raise ValueError(f"feature has shape {tuple(t['feature'].shape)}, expected (N, D)")
    mlp = {k[len("fea_up."):]: pipeline_state[MODEL_PREFIX + k].detach().float().contiguous()
           for k in MLP_KEYS if MODEL_PREFIX + k in pipeline_state}
    return Scene(*[t[k] for k in PARAM_KEYS]), mlp


def load_checkpoint(path: Union[str, os.PathLike]) -> Tuple[Scene, Dict[str, torch.Tensor], int]:
    """Read a reference `step-%09d.ckpt` -> (Scene, fea_up state, step).  Tensors only: the file is
    opened with weights_only=True, which refuses anything that would execute code."""
    blob = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(blob, dict) or "pipeline" not in blob:
        raise KeyError("checkpoint has no 'pipeline' entry (nerfstudio/engine/trainer.py:437-449)")
    scene, mlp = scene_from_state_dict(blob["pipeline"])
    return scene, mlp, int(blob.get("step", 0))


def state_dict_from_scene(scene: Scene, mlp_state: Optional[Dict[str, torch.Tensor]] = None
                          ) -> Dict[str, torch.Tensor]:
    out = {MODEL_PREFIX + k: getattr(scene, k).detach().cpu() for k in PARAM_KEYS}
    for k, v in (mlp_state or {}).items():
        out[MODEL_PREFIX + "fea_up." + k] = v.detach().cpu()
    return out


def save_checkpoint(path: Union[str, os.PathLike], scene: Scene,
                    mlp_state: Optional[Dict[str, torch.Tensor]] = None, step: int = 0) -> None:
    """Write a file the reference's trainer can resume from (model weights; optimizer, scheduler and
    scaler states are empty — the reference rebuilds them, trainer.py:405-420)."""
    torch.save({"step": int(step), "pipeline": state_dict_from_scene(scene, mlp_state),
                "optimizers": {}, "schedulers": {}, "scalers": {}}, path)


# ------------------------------------------------------------------------------------------------
# PLY
# ------------------------------------------------------------------------------------------------
def ply_properties(num_sh_bases: int, reference_literal: bool = False):
    """(name, numpy dtype) in the order the reference's exporter fills its map (exporter.py:499-525).

    Where this writer deliberately differs from the reference's literal behaviour: the reference
    reshapes `shs_rest` to (N, 3(K-1), 1) and loops over `shs.shape[-1]` == 1 (exporter.py:508-512), so
    its PLY has exactly ONE higher-band property, `f_rest_0` (= colors_all[:, 1, 0]), and every other
    higher-band coefficient is lost.  The default here writes all 3(K-1) `f_rest_*` columns (the
    3DGS viewer convention the loop evidently aimed at), so that a scene survives the round trip;
    `reference_literal=True` reproduces the reference's property set exactly."""
    props = [(n, "<f4") for n in ("x", "y", "z", "nx", "ny", "nz")]
    props += [(n, "u1") for n in ("red", "green", "blue")]
    props += [(f"f_dc_{i}", "<f4") for i in range(3)]
    nrest = 3 * (num_sh_bases - 1)
    if reference_literal:
        nrest = min(nrest, 1)
    props += [(f"f_rest_{i}", "<f4") for i in range(nrest)]
    props += [("opacity", "<f4")]
    props += [(f"scale_{i}", "<f4") for i in range(3)]
    props += [(f"rot_{i}", "<f4") for i in range(4)]
    return props


def export_ply(path: Union[str, os.PathLike], scene: Scene, reference_literal: bool = False) -> None:
    """Splat PLY with the reference exporter's property names (exporter.py:499-525); see
    `ply_properties` for the one deliberate difference and the `reference_literal` switch.  Byte
    compatibility with open3d's writer (property order, header comments) is unpinned: open3d is not
    available offline and the reference ships no PLY."""
    n, k = scene.num_points, scene.colors_all.shape[1]
    props = ply_properties(k, reference_literal)
    rec = np.zeros(n, dtype=np.dtype(props))
    means = scene.means.detach().cpu().numpy()
    dc = (scene.colors_all[:, 0, :].detach().cpu() * SH_C0 + 0.5).numpy()    # model.colors = SH2RGB(dc)
    rest = scene.colors_all[:, 1:, :].detach().cpu().numpy().reshape(n, -1)    # (band, rgb) order
    for i, a in enumerate("xyz"):
        rec[a] = means[:, i]
    for i, a in enumerate(("red", "green", "blue")):
        rec[a] = (dc[:, i] * 255).astype(np.uint8)          # `(colors * 255).astype(np.uint8)`, :503
    for i in range(3):
        rec[f"f_dc_{i}"] = dc[:, i]
        rec[f"scale_{i}"] = scene.scales[:, i].detach().cpu().numpy()
    for i in range(sum(1 for name, _ in props if name.startswith("f_rest_"))):
        rec[f"f_rest_{i}"] = rest[:, i]
    rec["opacity"] = scene.opacities[:, 0].detach().cpu().numpy()
    for i in range(4):
        rec[f"rot_{i}"] = scene.quats[:, i].detach().cpu().numpy()
    names = {"<f4": "float", "u1": "uchar"}
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {n}"]
    header += [f"property {names[d]} {name}" for name, d in props] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(rec.tobytes())


def load_ply(path: Union[str, os.PathLike], feature_dim: int = 32, sh_degree: int = 4) -> Scene:
    """Read a Gaussian-splat PLY (this module's or the reference exporter's property names; any
    property order; binary_little_endian).  The `f_rest_*` columns that are present fill the
    higher SH bands in (band, rgb) order and the missing ones are zero — a PLY written by the
    reference's exporter has only `f_rest_0` (exporter.py:508-512, see `ply_properties`), a PLY
    without any has only the DC band.  The scene gets max((sh_degree+1)^2, bands present) bases.
    The PLY carries no feature field: `feature` comes back as zeros (N, feature_dim)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, n, props = None, None, []
        types = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1",
                 "uint8": "u1", "int": "<i4", "int32": "<i4", "uint": "<u4", "short": "<i2",
                 "ushort": "<u2", "char": "i1"}
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PLY header without end_header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if tok[1] != "vertex":
                    raise ValueError(f"unsupported PLY element '{tok[1]}'")
                n = int(tok[2])
            elif tok[0] == "property":
                if tok[1] == "list":
                    raise ValueError("list properties are not part of a splat PLY")
                props.append((tok[2], types[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt != "binary_little_endian" or n is None:
            raise ValueError("only binary_little_endian vertex PLYs are supported")
        rec = np.frombuffer(f.read(n * np.dtype(props).itemsize), dtype=np.dtype(props), count=n)
    col = lambda name: torch.from_numpy(np.ascontiguousarray(rec[name]).astype(np.float32))
    stack = lambda names: torch.stack([col(a) for a in names], dim=1)
    rest_idx = sorted(int(name[len("f_rest_"):]) for name, _ in props if name.startswith("f_rest_"))
    bands = (sh_degree + 1) ** 2 - 1
    if rest_idx:
        bands = max(bands, rest_idx[-1] // 3 + 1)
    flat = torch.zeros(n, 3 * bands)
    for i in rest_idx:
        flat[:, i] = col(f"f_rest_{i}")
    rest = flat.reshape(n, bands, 3)
    dc = stack([f"f_dc_{i}" for i in range(3)])
    dc = (dc - 0.5) / SH_C0                                   # RGB2SH: back to the SH coefficient
    return Scene(stack("xyz"), stack([f"scale_{i}" for i in range(3)]),
                 stack([f"rot_{i}" for i in range(4)]), col("opacity")[:, None],
                 torch.cat([dc[:, None, :], rest], dim=1), torch.zeros(n, feature_dim))
