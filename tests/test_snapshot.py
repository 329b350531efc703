"""Reading the reference's network snapshots without executing them (vivid_amd/snapshot.py).

The snapshot is produced at test time by the reference's own persistence code (tests/golden/make_snapshot.py, in a
subprocess) — it embeds reference source text, so it is never committed; without /root/reference the tests that need it skip.
"""
import io
import os
import pickle
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("VIVID_REFERENCE", "/root/reference")

from tests.golden.cases import CASES                     # noqa: E402
from vivid_amd import snapshot                           # noqa: E402
from vivid_amd.weights import synth_state_dict           # noqa: E402


@pytest.fixture(scope="module")
def snap_path(tmp_path_factory):
    if not os.path.isdir(REF):
        pytest.skip("needs the reference checkout to write a snapshot")
    path = str(tmp_path_factory.mktemp("snap") / "network-snapshot-0000001.pkl")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_snapshot.py"), path, "tiny_dual"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return path


def test_read_snapshot_is_data_only(snap_path):
    data = snapshot.read_snapshot(snap_path)
    assert set(data) == {"encoder", "dataset_kwargs", "loss_fn", "ema"}
    assert data.dataset_kwargs == dict(path="nowhere", split="train") and data.loss_fn is None
    ema = data["ema"]
    assert isinstance(ema, snapshot.SnapshotNet) and ema.class_name == "NVPrecond"
    # nothing of the reference was imported or executed to get here
    assert not any(m == "training" or m.startswith("training.") or m.startswith("torch_utils") or m == "dnnlib" for m in sys.modules)
    cfg = CASES["tiny_dual"]["cfg"]
    kw = ema.init_kwargs
    assert kw["img_resolution"] == cfg.img_resolution and kw["model_channels"] == cfg.model_channels
    # weights: the reference stores fp16 (training_loop.py:489); names and values = what was loaded into the reference net
    want = synth_state_dict(cfg, seed=CASES["tiny_dual"]["seed"])
    got = ema.state_dict()
    assert list(got) == list(want)
    for k, v in want.items():
        assert got[k].dtype == torch.float16 and got[k].shape == v.shape, k
        assert torch.equal(got[k], v.to(torch.float16)), k
    assert isinstance(snapshot.snapshot_encoder(data), __import__("vivid_amd").StandardRGBEncoder)


def test_load_network_pkl_builds_the_net(snap_path):
    import vivid_amd
    net = snapshot.load_network_pkl(snap_path)
    assert isinstance(net, vivid_amd.NVPrecond) and not net.training
    cfg = CASES["tiny_dual"]["cfg"]
    assert net.img_resolution == cfg.img_resolution
    want = synth_state_dict(cfg, seed=CASES["tiny_dual"]["seed"])
    sd = net.state_dict()
    assert list(sd) == list(want)
    for k, v in want.items():
        assert torch.equal(sd[k], v.to(torch.float16).to(torch.float32)), k
    with open(snap_path, "rb") as f:                                  # file objects work too; `net` key fallback
        assert isinstance(snapshot.load_network_pkl(f, key="ema"), vivid_amd.NVPrecond)
    with pytest.raises(TypeError):
        snapshot.load_network_pkl(snap_path, key="encoder")


class _Boom:
    def __reduce__(self):
        return (os.system, ("echo pwned > /dev/null",))


@pytest.mark.parametrize("payload", [
    pickle.dumps(_Boom()),                                            # classic os.system gadget
    pickle.dumps({"ema": torch.nn.Linear(2, 2)}),                     # arbitrary module classes are not resolved either
    b"cbuiltins\neval\n(V1+1\ntR.",                                   # protocol-0 GLOBAL builtins.eval
])
def test_hostile_pickles_are_refused(payload):
    with pytest.raises(pickle.UnpicklingError, match="not allowed"):
        snapshot.read_snapshot(io.BytesIO(payload))


def test_persistent_record_never_execs_module_src():
    """A record whose module_src would raise if executed decodes fine: the text is ignored."""
    meta = dict(type="class", version=6, module_src="raise SystemExit('executed!')", class_name="NVPrecond",
                state=dict(_parameters={}, _buffers={}, _modules={}, _init_args=[], _init_kwargs=dict(img_resolution=16)))
    stream = io.BytesIO()
    p = pickle.Pickler(stream, protocol=4)
    # hand-assemble `_reconstruct_persistent_obj(meta)` as the reference's __reduce__ emits it
    stream.write(b"\x80\x04")
    stream.write(b"\x8c\x17torch_utils.persistence\x8c\x1b_reconstruct_persistent_obj\x93")
    body = pickle.dumps(meta, protocol=4)
    stream.write(body[2:-1])                                          # strip PROTO header and STOP
    stream.write(b"\x85R.")                                           # TUPLE1, REDUCE, STOP
    del p
    stream.seek(0)
    rec = snapshot.read_snapshot(stream)
    assert isinstance(rec, snapshot.SnapshotNet) and rec.init_kwargs == dict(img_resolution=16)
    bad = dict(meta, version=5)
    s2 = io.BytesIO(b"\x80\x04\x8c\x17torch_utils.persistence\x8c\x1b_reconstruct_persistent_obj\x93" + pickle.dumps(bad, protocol=4)[2:-1] + b"\x85R.")
    with pytest.raises(pickle.UnpicklingError, match="version"):
        snapshot.read_snapshot(s2)
