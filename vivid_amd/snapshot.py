"""Read the reference's network snapshots (`network-snapshot-*.pkl`, `vivid-*.pkl`) WITHOUT executing them.

SURVEY.md 8(f) rank 4.  The reference writes `pickle.dump(EasyDict(encoder=..., dataset_kwargs=..., loss_fn=...,
ema=<NVPrecond, fp16>))` (`training/training_loop.py:485-496`) and reads it back with a bare `pickle.load`
(`generate_images.py:164-169`).  Its networks are `torch_utils.persistence` classes: each pickles as
`_reconstruct_persistent_obj(meta)` with `meta = dict(type, version, module_src, class_name, state)`, and
unpickling `exec`s `module_src` — the whole source file of the module, as text — to rebuild the class
(`torch_utils/persistence.py:129-137,189-237`).  Loading a snapshot therefore runs whatever code is inside it.

Here the pickle stream is decoded by a restricted `pickle.Unpickler`: only the handful of globals such a file
legitimately names are resolved (table below), every other global raises, and `module_src` is never looked at.
A persistent object comes back as a plain record (class name, constructor kwargs, module tree); the weights are
collected from the module tree into a `state_dict` with the reference's key names and loaded into
`vivid_amd.NVPrecond`, which has the reference's constructor arguments.

    data = read_snapshot("network-snapshot-0001000.pkl")     # {'ema': SnapshotNet, 'encoder': ..., ...}
    net  = load_network_pkl("network-snapshot-0001000.pkl").to("cuda")
"""
from __future__ import annotations

import collections
import io
import pickle
from typing import Any, BinaryIO, Dict, Optional, Union

import torch

PERSISTENCE_VERSION = 6          # torch_utils/persistence.py:20


class EasyDict(dict):
    """dnnlib.EasyDict as data: attribute access to dict entries (dnnlib/util.py:41-56)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


class _Record:
    """Stand-in for an instance of a class named in the stream (e.g. torch.nn.ModuleDict): keeps the pickled state only."""

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        self.__dict__["state"] = state


class SnapshotNet:
    """A persistent object of the reference, as data."""

    def __init__(self, class_name: str, state: Dict[str, Any]):
        self.class_name = class_name
        self.state = state

    @property
    def init_kwargs(self) -> Dict[str, Any]:
        return dict(self.state.get("_init_kwargs") or {})

    @property
    def init_args(self):
        return list(self.state.get("_init_args") or [])

    def state_dict(self) -> "collections.OrderedDict[str, torch.Tensor]":
        """Parameters and persistent buffers of the module tree, named as `nn.Module.state_dict()` names them."""
        out: "collections.OrderedDict[str, torch.Tensor]" = collections.OrderedDict()
        _collect(self.state, "", out)
        return out


def _module_state(obj) -> Optional[Dict[str, Any]]:
    if obj is None:
        return None
    if isinstance(obj, SnapshotNet):
        return obj.state
    if isinstance(obj, _Record):
        return obj.__dict__.get("state")
    raise pickle.UnpicklingError(f"snapshot: unexpected object of type {type(obj).__name__} in a module tree")


def _collect(state: Dict[str, Any], prefix: str, out) -> None:
    skip = state.get("_non_persistent_buffers_set") or set()
    for name, p in (state.get("_parameters") or {}).items():
        if p is not None:
            out[prefix + name] = p.detach()
    for name, b in (state.get("_buffers") or {}).items():
        if b is not None and name not in skip:
            out[prefix + name] = b.detach()
    for name, m in (state.get("_modules") or {}).items():
        sub = _module_state(m)
        if sub is not None:
            _collect(sub, prefix + name + ".", out)


def _reconstruct_persistent_obj(meta):
    """Replacement for torch_utils.persistence._reconstruct_persistent_obj: no exec of meta['module_src']."""
    if not isinstance(meta, dict) or meta.get("type") != "class":
        raise pickle.UnpicklingError("snapshot: malformed persistent-object record")
    if meta.get("version") != PERSISTENCE_VERSION:
        raise pickle.UnpicklingError(f"snapshot: persistence version {meta.get('version')} (expected {PERSISTENCE_VERSION})")
    state = meta.get("state")
    return SnapshotNet(str(meta.get("class_name")), dict(state) if state is not None else {})


def _load_from_bytes(b):
    """torch.storage._load_from_bytes, with torch's own restricted loader."""
    return torch.load(io.BytesIO(b), weights_only=True)


def _allowed():
    import torch._utils
    table = {
        ("collections", "OrderedDict"): collections.OrderedDict,
        ("dnnlib.util", "EasyDict"): EasyDict,
        ("torch_utils.persistence", "_reconstruct_persistent_obj"): _reconstruct_persistent_obj,
        ("torch.storage", "_load_from_bytes"): _load_from_bytes,
        ("torch._utils", "_rebuild_tensor_v2"): torch._utils._rebuild_tensor_v2,
        ("torch._utils", "_rebuild_parameter"): torch._utils._rebuild_parameter,
        ("torch", "Size"): torch.Size,
        ("builtins", "set"): set,
        ("builtins", "frozenset"): frozenset,
    }
    for cls in ("ModuleDict", "ModuleList", "Sequential"):
        table[("torch.nn.modules.container", cls)] = _Record
    return table


class _Unpickler(pickle.Unpickler):
    _table = None

    def find_class(self, module, name):
        if _Unpickler._table is None:
            _Unpickler._table = _allowed()
        try:
            return _Unpickler._table[(module, name)]
        except KeyError:
            raise pickle.UnpicklingError(f"snapshot: global '{module}.{name}' is not allowed "
                                         "(this loader resolves only tensors, containers and the reference's persistent-object records)")


def read_snapshot(f: Union[str, BinaryIO]) -> Dict[str, Any]:
    """Decode a snapshot file into plain data.  Persistent objects become `SnapshotNet` records."""
    if isinstance(f, (str, bytes)) or hasattr(f, "__fspath__"):
        with open(f, "rb") as fh:
            return _Unpickler(fh).load()
    return _Unpickler(f).load()


def network_from_snapshot(data, key: Optional[str] = None, *, dual_source: bool = True, precision: Optional[str] = None):
    """`vivid_amd.NVPrecond` from decoded snapshot data (:func:`read_snapshot`): `data['ema']` (or `data['net']`), as
    `generate_images.py:169` picks it.  Weights are stored in fp16 by the reference (`training_loop.py:489`) and are
    widened to fp32 here, as the reference's `MPConv` does on every forward (`training/models.py:115`)."""
    from .net import NVPrecond
    if isinstance(data, SnapshotNet):
        rec = data
    else:
        if key is None:
            key = "ema" if "ema" in data else "net"
        rec = data[key]
    if not isinstance(rec, SnapshotNet) or rec.class_name != "NVPrecond":
        raise TypeError(f"snapshot entry {key!r} is {getattr(rec, 'class_name', type(rec).__name__)}, not an NVPrecond")
    if rec.init_args:
        raise TypeError("snapshot: positional constructor arguments are not supported (the reference constructs NVPrecond by keyword)")
    kw = rec.init_kwargs
    if precision is not None:
        kw["precision"] = precision
    net = NVPrecond(**kw, dual_source=dual_source)
    net.load_state_dict({k: v.to(torch.float32) for k, v in rec.state_dict().items()}, strict=True)
    return net.eval()


def load_network_pkl(f: Union[str, BinaryIO], key: Optional[str] = None, *, dual_source: bool = True,
                     precision: Optional[str] = None):
    """Read a reference snapshot file and build the network it holds (see :func:`network_from_snapshot`)."""
    return network_from_snapshot(read_snapshot(f), key, dual_source=dual_source, precision=precision)


def snapshot_encoder(data: Dict[str, Any]):
    """The pixel codec stored beside the network (`generate_images.py:170-173`): only StandardRGBEncoder exists here."""
    from .encoders import StandardRGBEncoder
    enc = data.get("encoder") if isinstance(data, dict) else None
    if enc is None or (isinstance(enc, SnapshotNet) and enc.class_name == "StandardRGBEncoder"):
        return StandardRGBEncoder()
    raise TypeError(f"snapshot: unsupported encoder {getattr(enc, 'class_name', type(enc).__name__)}")
