"""Safe loading of nnU-Net v1 model folders (SURVEY.md 8a row L, section 5 "checkpoint").

Layout the reference expects (run_brats2021_inference_singlethread.py:263-264,178-183;
scripts/check_compatibility.py:104-136):

    <model_dir>/plans.pkl
    <model_dir>/fold_k/model_final_checkpoint.model       torch.save({'state_dict': ..., ...})
    <model_dir>/fold_k/model_final_checkpoint.model.pkl   pickle {'init': ..., 'name': trainer, 'plans': ...}

Both pickles are read through allow-listed unpicklers (numpy arrays/scalars, OrderedDict and
torch tensor rebuilders only) - a checkpoint is data, never code.  The trainer class name
decides the normalisation (the state_dict cannot: GroupNorm and InstanceNorm weights look the
same): ``...Groupnorm...`` -> GroupNorm(16), ``..._BN...`` -> BatchNorm (eval),
otherwise InstanceNorm (generic nnU-Net default).
"""
from __future__ import annotations

import os
import pickle
from collections import OrderedDict
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Sequence

import numpy as np

_ALLOWED = {
    ("collections", "OrderedDict"), ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
    ("builtins", "slice"), ("builtins", "set"), ("builtins", "frozenset"),
}


class _SafeUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        if module.startswith("numpy") and name in ("int64", "int32", "float64", "float32", "bool_"):
            return getattr(np, name)
        raise pickle.UnpicklingError(f"refusing to unpickle {module}.{name}")


def safe_pickle_load(path):
    with open(path, "rb") as f:
        return _SafeUnpickler(f).load()


def load_state_dict_file(path) -> Dict[str, np.ndarray]:
    """torch.load(weights_only=True) with numpy scalars allow-listed (nnU-Net checkpoints hold
    them, PROJECT_DOCUMENTATION.md:346-360); returns {key: np.ndarray} of the network weights."""
    import torch
    safe = []
    try:
        import numpy.core.multiarray as ncm  # noqa
        safe += [ncm.scalar, ncm._reconstruct]
    except Exception:
        pass
    safe += [np.ndarray, np.dtype, OrderedDict]
    for n in ("float64", "float32", "int64", "int32", "bool_"):
        safe.append(getattr(np, n))
    for n in ("Float64DType", "Float32DType", "Int64DType", "Int32DType", "BoolDType"):
        if hasattr(np.dtypes, n):
            safe.append(getattr(np.dtypes, n))
    with torch.serialization.safe_globals(safe):
        ckpt = torch.load(str(path), map_location="cpu", weights_only=True)
    sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    out = OrderedDict()
    for k, v in sd.items():
        k = k[7:] if k.startswith("module.") else k
        out[k] = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
    return out


def norm_from_trainer_name(name: str):
    low = name.lower()
    if "groupnorm" in low:
        return "group", 16
    if "_bn" in low or "batchnorm" in low:
        return "batch", 16
    return "instance", 16


@dataclass
class ModelFolder:
    path: Path
    trainer: str
    norm: str
    num_groups: int
    plans: Dict
    fold_state_dicts: List[Dict[str, np.ndarray]] = field(default_factory=list)
    regions: bool = True  # BraTSRegions trainers: sigmoid region heads

    @property
    def patch_size(self):
        st = self.plans["plans_per_stage"]
        return tuple(int(v) for v in st[max(st.keys())]["patch_size"])


def load_model_folder(model_dir, folds: Sequence[int] = (0, 1, 2, 3, 4),
                      checkpoint_name: str = "model_final_checkpoint") -> ModelFolder:
    """Mirror of nnunet.training.model_restore.load_model_and_checkpoint_files as the driver calls it."""
    model_dir = Path(model_dir)
    if not model_dir.is_dir():
        raise FileNotFoundError(f"model folder not found: {model_dir}")
    fold_dirs = [model_dir / f"fold_{f}" for f in folds]
    for fd in fold_dirs:
        if not (fd / f"{checkpoint_name}.model").is_file():
            raise FileNotFoundError(f"missing checkpoint {fd / (checkpoint_name + '.model')}")
    info = safe_pickle_load(fold_dirs[0] / f"{checkpoint_name}.model.pkl")
    trainer = str(info.get("name", model_dir.name.split("__")[0]))
    plans = info.get("plans")
    if plans is None and (model_dir / "plans.pkl").is_file():
        plans = safe_pickle_load(model_dir / "plans.pkl")
    if plans is None:
        raise ValueError(f"{model_dir}: no plans in the checkpoint pkl and no plans.pkl")
    norm, groups = norm_from_trainer_name(trainer)
    mf = ModelFolder(path=model_dir, trainer=trainer, norm=norm, num_groups=groups, plans=plans,
                     regions="region" in trainer.lower())
    for fd in fold_dirs:
        mf.fold_state_dicts.append(load_state_dict_file(fd / f"{checkpoint_name}.model"))
    return mf


def default_brats_plans(patch=(128, 128, 128)) -> Dict:
    """The fields of data/temp_inference_output1 (Task500_BraTS2021 plans) the path reads."""
    return {
        "num_stages": 1, "num_modalities": 4, "modalities": {0: "T1", 1: "T1ce", 2: "T2", 3: "FLAIR"},
        "normalization_schemes": OrderedDict((i, "nonCT") for i in range(4)),
        "use_mask_for_norm": OrderedDict((i, True) for i in range(4)),
        "num_classes": 3, "all_classes": [1, 2, 3], "base_num_features": 32, "conv_per_stage": 2,
        "transpose_forward": [0, 1, 2], "transpose_backward": [0, 1, 2],
        "plans_per_stage": {0: {"patch_size": np.array(patch), "current_spacing": np.array([1.0, 1.0, 1.0]),
                                "pool_op_kernel_sizes": [[2, 2, 2]] * 5, "conv_kernel_sizes": [[3, 3, 3]] * 6}},
    }


def save_model_folder(model_dir, trainer: str, fold_state_dicts: Sequence[Dict[str, np.ndarray]], plans: Dict = None,
                      checkpoint_name: str = "model_final_checkpoint") -> Path:
    """Writes the nnU-Net v1 layout (used with synthetic weights by tests, smoke and the demo)."""
    import torch
    model_dir = Path(model_dir)
    plans = plans or default_brats_plans()
    model_dir.mkdir(parents=True, exist_ok=True)
    with open(model_dir / "plans.pkl", "wb") as f:
        pickle.dump(plans, f)
    for k, sd in enumerate(fold_state_dicts):
        fd = model_dir / f"fold_{k}"
        fd.mkdir(exist_ok=True)
        torch.save({"state_dict": OrderedDict((key, torch.from_numpy(np.array(v))) for key, v in sd.items()),
                    "epoch": 1000}, fd / f"{checkpoint_name}.model")
        with open(fd / f"{checkpoint_name}.model.pkl", "wb") as f:
            pickle.dump({"init": (), "name": trainer, "plans": plans}, f)
    return model_dir
