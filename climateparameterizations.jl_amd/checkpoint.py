"""Training logs and checkpoints with the reference's JLD2 GROUP LAYOUT and selection rules.

    write_metadata_NDE_training / write_data_NDE_training      wind_mixing/src/data_writing.jl:4-78
    write_metadata_NN_training / write_data_NN_training / write_data_NN   data_writing.jl:80-116
    extract_NN(FILE_PATH, OUTPUT_PATH, type)                    wind_mixing/src/data_extraction.jl:1-149 (arg-min of the LAST stage)
    resume of the optimiser (β, state) from an extracted file   wind_mixing/train_NDE_args.jl:124-147

What is mirrored is the tree — every key path ("training_data/loss/∂u∂z/$stage/$count", "training_data/optimizer/state/…", …), the
stage/count bookkeeping (`count = length(keys(group)) + 1`), what `extract_NN` selects and what a resumed run reads back.  What is NOT
mirrored is JLD2's byte format: writing it needs HDF5 plus Julia's type serialisation of `Flux.Chain` objects, neither of which exists
in this image (no h5py, no Julia).  The container here is a directory tree, one `.npy` / `.json` leaf per key (a group = a directory, so
`keys(file[group])` = its entries); INTEGRATION.md holds the 15-line Julia loop that copies such a tree into a `.jld2` and back.

Values: a neural network is stored as its `Flux.destructure` vector plus (layer_sizes, activations) — what `re(θ)` needs; Flux's ADAM
`state::IdDict(param => (mt, vt, βp))` is stored for the flat θ as (m, v, beta_t).  Host-side only: no GPU, no oracle imports.
"""
from __future__ import annotations

import json
import os
import shutil
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from .flux_compat import ADAM

LOSS_KEYS = ("u", "v", "T", "∂u∂z", "∂v∂z", "∂T∂z")
_ASCII = {"dudz": "∂u∂z", "dvdz": "∂v∂z", "dTdz": "∂T∂z"}


class GroupFile:
    """`jldopen(path, mode)`: mode "w" truncates, "a" appends, "r" reads.  file[key] with '/'-separated group paths."""

    def __init__(self, path: str, mode: str = "r"):
        if mode not in ("r", "w", "a"):
            raise ValueError("mode must be r, w or a")
        self.path, self.mode = path, mode
        if mode == "w" and os.path.isdir(path):
            shutil.rmtree(path)
        if mode in ("w", "a"):
            os.makedirs(path, exist_ok=True)
        elif not os.path.isdir(path):
            raise FileNotFoundError(path)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def _p(self, key: str) -> str:
        parts = [q for q in key.split("/") if q]
        if any(q in (".", "..") for q in parts):
            raise KeyError(key)
        return os.path.join(self.path, *parts)

    def group(self, key: str) -> None:
        """`JLD2.Group(parent, name)`: an (empty) group."""
        os.makedirs(self._p(key), exist_ok=True)

    def haskey(self, key: str) -> bool:
        p = self._p(key)
        return os.path.isdir(p) or os.path.exists(p + ".npy") or os.path.exists(p + ".json")

    __contains__ = haskey

    def keys(self, key: str = "") -> List[str]:
        """Entries of a group; numeric names in numeric order (JLD2 keeps insertion order: 1, 2, 3, …)."""
        p = self._p(key)
        if not os.path.isdir(p):
            raise KeyError(key)
        names = sorted({os.path.splitext(n)[0] if not os.path.isdir(os.path.join(p, n)) else n for n in os.listdir(p)})
        return sorted(names, key=lambda n: (0, int(n)) if n.isdigit() else (1, n))

    def __setitem__(self, key: str, value: Any) -> None:
        if self.mode == "r":
            raise IOError("file opened read-only")
        p = self._p(key)
        if self.haskey(key):
            raise KeyError("%s already exists (JLD2 does not overwrite datasets)" % key)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        if isinstance(value, np.ndarray) or np.isscalar(value) and not isinstance(value, str):
            np.save(p + ".npy", np.asarray(value))
        else:
            with open(p + ".json", "w") as f:
                json.dump(_to_json(value), f)

    def __getitem__(self, key: str) -> Any:
        p = self._p(key)
        if os.path.exists(p + ".npy"):
            a = np.load(p + ".npy")
            return a[()] if a.ndim == 0 else a
        if os.path.exists(p + ".json"):
            with open(p + ".json") as f:
                return _from_json(json.load(f))
        if os.path.isdir(p):
            return {k: self[key + "/" + k] for k in self.keys(key)}
        raise KeyError(key)


def _to_json(v):
    if isinstance(v, np.ndarray):
        return {"__ndarray__": v.tolist(), "dtype": str(v.dtype)}
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, dict):
        return {str(k): _to_json(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_to_json(x) for x in v]
    if isinstance(v, ADAM):
        return optimizer_record(v)
    return v


def _from_json(v):
    if isinstance(v, dict) and "__ndarray__" in v:
        return np.asarray(v["__ndarray__"], dtype=v["dtype"])
    if isinstance(v, dict):
        return {k: _from_json(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_from_json(x) for x in v]
    return v


def network_record(theta, layer_sizes: Sequence[int], activations: Sequence[str]) -> Dict[str, Any]:
    """One `Chain(Dense…)`: its Flux.destructure vector and what `re` needs to rebuild it."""
    return dict(theta=np.asarray(theta, np.float32), layer_sizes=[int(s) for s in layer_sizes], activations=list(activations))


def optimizer_record(opt: ADAM) -> Dict[str, Any]:
    """`optimizer.eta`, `.beta`, `.state` (data_writing.jl:52-54) of a Flux ADAM; state = (mt, vt, βp) of the flat θ."""
    return dict(eta=opt.eta, beta=list(opt.beta),
                state=None if opt.m is None else dict(m=np.asarray(opt.m, np.float64), v=np.asarray(opt.v, np.float64),
                                                      beta_t=[float(b) for b in opt.beta_t]))


def restore_optimizer(rate: float, beta, state) -> ADAM:
    """train_NDE_args.jl:141-143: `ADAM(rate)` with `.beta` and `.state` taken from the extracted file."""
    opt = ADAM(rate, tuple(beta))
    if state is not None:
        opt.m, opt.v = np.array(state["m"], np.float64), np.array(state["v"], np.float64)
        opt.beta_t = [float(b) for b in state["beta_t"]]
    return opt


def _losses(losses: Dict[str, float]) -> Dict[str, float]:
    out = {_ASCII.get(k, k): float(v) for k, v in losses.items() if _ASCII.get(k, k) in LOSS_KEYS}
    missing = [k for k in LOSS_KEYS if k not in out]
    if missing:
        raise KeyError("losses lack %s" % missing)
    return out


# ---- NDE training (data_writing.jl:4-78) -------------------------------------------------------------------------------------
def write_metadata_NDE_training(FILE_PATH, train_files, train_epochs, train_tranges, train_parameters, opts, uw_NN, vw_NN, wT_NN):
    with GroupFile(FILE_PATH, "w") as file:
        file["training_info/train_files"] = list(train_files)
        file["training_info/train_epochs"] = list(train_epochs)
        file["training_info/train_tranges"] = [list(r) for r in train_tranges]
        file["training_info/optimizers"] = [[optimizer_record(o) for o in (stage if isinstance(stage, (list, tuple)) else [stage])] for stage in opts]
        file["training_info/parameters"] = dict(train_parameters)
        file["training_info/uw_neural_network"] = uw_NN
        file["training_info/vw_neural_network"] = vw_NN
        file["training_info/wT_neural_network"] = wT_NN
        for g in ("training_data/loss", "training_data/neural_network/uw", "training_data/neural_network/vw",
                  "training_data/neural_network/wT", "training_data/η", "training_data/β", "training_data/state"):
            file.group(g)


def write_data_NDE_training(FILE_PATH, losses, loss_scalings, uw_NN, vw_NN, wT_NN, stage, optimizer):
    L = _losses(losses)
    profile_loss = L["u"] + L["v"] + L["T"]
    gradient_loss = L["∂u∂z"] + L["∂v∂z"] + L["∂T∂z"]
    total_loss = profile_loss + gradient_loss
    with GroupFile(FILE_PATH, "a") as file:
        if not file.haskey("training_data/loss/total/%s" % stage):
            if not file.haskey("training_info/loss_scalings"):          # (the reference writes it at every new stage: JLD2 would refuse the second)
                file["training_info/loss_scalings"] = {_ASCII.get(k, k): float(v) for k, v in dict(loss_scalings).items()}
            count = 1
        else:
            count = len(file.keys("training_data/loss/total/%s" % stage)) + 1
        file["training_data/loss/total/%s/%d" % (stage, count)] = np.float32(total_loss)
        file["training_data/loss/profile/%s/%d" % (stage, count)] = np.float32(profile_loss)
        file["training_data/loss/gradient/%s/%d" % (stage, count)] = np.float32(gradient_loss)
        for k in LOSS_KEYS:
            file["training_data/loss/%s/%s/%d" % (k, stage, count)] = np.float32(L[k])
        file["training_data/neural_network/uw/%s/%d" % (stage, count)] = uw_NN
        file["training_data/neural_network/vw/%s/%d" % (stage, count)] = vw_NN
        file["training_data/neural_network/wT/%s/%d" % (stage, count)] = wT_NN
        rec = optimizer_record(optimizer)
        file["training_data/optimizer/η/%s/%d" % (stage, count)] = np.float64(rec["eta"])
        file["training_data/optimizer/β/%s/%d" % (stage, count)] = np.asarray(rec["beta"], np.float64)
        file["training_data/optimizer/state/%s/%d" % (stage, count)] = rec["state"]


# ---- flux-NN pre-training (data_writing.jl:80-116) ---------------------------------------------------------------------------
def write_metadata_NN_training(FILE_PATH, train_files, train_parameters, train_epochs, opts, NN, NN_type):
    with GroupFile(FILE_PATH, "w") as file:
        file["training_info/train_files"] = list(train_files)
        file["training_info/train_epochs"] = list(train_epochs)
        file["training_info/optimizers"] = [optimizer_record(o) for o in opts]
        file["training_info/parameters"] = dict(train_parameters)
        file["training_info/%s_neural_network" % NN_type] = NN
        file.group("training_data")


def write_data_NN_training(FILE_PATH, loss, NN):
    with GroupFile(FILE_PATH, "a") as file:
        count = len(file.keys("training_data/loss")) + 1 if file.haskey("training_data/loss") else 1
        file["training_data/loss/%d" % count] = np.float32(loss)
        count = len(file.keys("training_data/neural_network")) + 1 if file.haskey("training_data/neural_network") else 1
        file["training_data/neural_network/%d" % count] = NN


def write_data_NN(FILE_PATH, uw_NN, vw_NN, wT_NN):
    with GroupFile(FILE_PATH, "w") as file:
        file["neural_network/uw"] = uw_NN
        file["neural_network/vw"] = vw_NN
        file["neural_network/wT"] = wT_NN


# ---- extract_NN (data_extraction.jl:1-149) -----------------------------------------------------------------------------------
def extract_NN(FILE_PATH, OUTPUT_PATH, type: str):
    """Of the LAST stage's entries, keep the one with the smallest total loss (`NN_index = argmin(total_losses)`, :72-75) together
    with its optimiser (η, β, state) and the whole loss history of that stage; type "NDE" or anything else for a flux-NN log.

    FORMAT: both paths are `GroupFile` directory trees (this module's container), NOT `.jld2` files — a log written here cannot be
    opened by the reference's Julia `extract_NN` / `train_NDE_args.jl`, nor the reverse; the key paths, the stage/count bookkeeping and
    the selection rule are the reference's (tests/test_checkpoint.py holds the key list its reader asks for), and INTEGRATION.md has
    an (untested: no Julia in the build image) copy loop between the two containers.  Like the reference, the last stage is the group
    named `"$N_stages"` with N_stages = the number of stage groups (data_extraction.jl:6-8): stages must be named 1..N."""
    with GroupFile(FILE_PATH, "r") as file:
        train_files = file["training_info/train_files"]
        if type == "NDE":
            N_stages = len(file.keys("training_data/neural_network/uw"))
            stage = str(N_stages)
            if not file.haskey("training_data/neural_network/uw/%s" % stage):
                raise KeyError("training_data/neural_network/uw/%s: the reference indexes the last stage as \"$N_stages\" (data_extraction.jl:8); "
                               "stage groups must be named 1..N, found %s" % (stage, file.keys("training_data/neural_network/uw")))
            N_data = len(file.keys("training_data/neural_network/uw/%s" % stage))
            train_parameters = file["training_info/parameters"] if "parameters" in file.keys("training_info") else None
            loss_scalings = file["training_info/loss_scalings"] if "loss_scalings" in file.keys("training_info") else None
            names = ("total", "profile", "gradient") + LOSS_KEYS
            losses = {n: np.array([file["training_data/loss/%s/%s/%d" % (n, stage, i)] for i in range(1, N_data + 1)], np.float32)
                      for n in names}
            NN_index = int(np.argmin(losses["total"])) + 1                      # first minimum, 1-based like Julia's argmin
            nets = {k: file["training_data/neural_network/%s/%s/%d" % (k, stage, NN_index)] for k in ("uw", "vw", "wT")}
            optimizer = None
            if "optimizer" in file.keys("training_data"):
                optimizer = dict(η=file["training_data/optimizer/η/%s/%d" % (stage, NN_index)],
                                 β=file["training_data/optimizer/β/%s/%d" % (stage, NN_index)],
                                 state=file["training_data/optimizer/state/%s/%d" % (stage, NN_index)])
        else:
            train_parameters = file["training_info/parameters"]
            N_data = len(file.keys("training_data/loss"))
            losses = np.array([file["training_data/loss/%d" % i] for i in range(1, N_data + 1)], np.float64)
            NN_index = int(np.argmin(losses)) + 1
            NN = file["training_data/neural_network/%d" % NN_index]
    with GroupFile(OUTPUT_PATH, "w") as out:
        out["training_info/train_files"] = train_files
        out["training_info/parameters"] = train_parameters
        if type == "NDE":
            out["training_info/loss_scalings"] = loss_scalings
            for n, v in losses.items():
                out["losses/%s" % n] = v
            for k in ("uw", "vw", "wT"):
                out["neural_network/%s" % k] = nets[k]
            if optimizer is not None:
                out["optimizer/η"] = np.float64(optimizer["η"])
                out["optimizer/β"] = np.asarray(optimizer["β"], np.float64)
                out["optimizer/state"] = optimizer["state"]
        else:
            out["losses"] = losses
            out["neural_network"] = NN
    return NN_index


def load_extracted_NDE(EXTRACTED_PATH, rate: Optional[float] = None):
    """What train_NDE_args.jl:124-147 reads back to resume: the three networks, the training parameters and — when `rate` is given —
    `ADAM(rate)` carrying the stored β and state.  EXTRACTED_PATH is a `GroupFile` tree written by this module's `extract_NN`, not a
    `.jld2` (see `extract_NN`): networks come back as (θ, layer sizes, activations) records, the ADAM state as flat (m, v, βᵗ)."""
    with GroupFile(EXTRACTED_PATH, "r") as file:
        nets = {k: file["neural_network/%s" % k] for k in ("uw", "vw", "wT")}
        params = file["training_info/parameters"]
        opt = None
        if rate is not None and file.haskey("optimizer/β"):
            opt = restore_optimizer(rate, file["optimizer/β"], file["optimizer/state"])
    weights = np.concatenate([np.asarray(nets[k]["theta"], np.float32) for k in ("uw", "vw", "wT")])
    return weights, nets, params, opt
