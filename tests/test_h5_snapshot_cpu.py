"""Policy snapshots in the reference's HDF5 container (es_distributed/policies.py:49-67) written / read through libhdf5
directly (dne_hip/h5lite.py).  Checked three ways: h5dump's view of a file we wrote against the layout h5py gives the
reference (recorded below from `h5dump -H` of the snapshot the reference ships), a read of that reference-written file when
/root/reference is present, and round trips through Policy.save / Load / initialize_from."""
import os
import pickle
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import h5lite, policies  # noqa: E402

pytestmark = pytest.mark.skipif(not h5lite.available(), reason="no libhdf5 on this machine")

REF_SNAPSHOT = "/root/reference/visual_inspector/sample_data/mujoco/final_xy_bc/snapshots/snapshot_gen_0097/snapshot_parent_0097.h5"

# `h5dump -H` of REF_SNAPSHOT (written by the reference's Policy.save through h5py): the two root attributes
REF_ATTR_LAYOUT = '''   ATTRIBUTE "args_and_kwargs" {
      DATATYPE  H5T_OPAQUE {
         OPAQUE_TAG "";
      }
      DATASPACE  SCALAR
   }
   ATTRIBUTE "name" {
      DATATYPE  H5T_STRING {
         STRSIZE H5T_VARIABLE;
         STRPAD H5T_STR_NULLTERM;
         CSET H5T_CSET_UTF8;
         CTYPE H5T_C_S1;
      }
      DATASPACE  SCALAR
   }
'''


def _h5dump():
    for c in (shutil.which("h5dump"), "/opt/conda/bin/h5dump"):
        if c and os.path.exists(c):
            return c
    return None


def test_write_read_roundtrip(tmp_path):
    rng = np.random.RandomState(0)
    arrays = {"P/conv1/weights:0": rng.randn(8, 8, 4, 16).astype(np.float32), "P/conv1/biases:0": rng.randn(16).astype(np.float32),
              "P/out/weights:0": rng.randn(256, 18).astype(np.float32), "P/scalar:0": np.float32(3.5), "P/empty:0": np.zeros((0,), np.float32)}
    blob = pickle.dumps((((84, 84, 4), 18), {"k": 1}), protocol=-1)
    fn = str(tmp_path / "a.h5")
    h5lite.write_snapshot(fn, arrays, "Pé", blob)          # non-ASCII: the attribute is UTF-8
    name, blob2, got = h5lite.read_snapshot(fn)
    assert name == "Pé" and blob2 == blob and sorted(got) == sorted(arrays)
    for k, v in arrays.items():
        assert got[k].dtype == np.float32 and got[k].shape == np.shape(v) and np.array_equal(got[k], v)
    h5lite.write_snapshot(fn, {"Q/x:0": np.ones(3, np.float32)}, "Q", b"\x00\x01")     # truncates like h5py's mode 'w'
    assert list(h5lite.read_snapshot(fn)[2]) == ["Q/x:0"]
    with pytest.raises(h5lite.H5Error):
        h5lite.read_snapshot(str(tmp_path / "missing.h5"))
    with pytest.raises(h5lite.H5Error):
        h5lite.write_snapshot(str(tmp_path / "no_such_dir" / "a.h5"), arrays, "P", blob)


def test_layout_matches_h5py_written_reference_file(tmp_path):
    h5dump = _h5dump()
    if h5dump is None:
        pytest.skip("no h5dump")
    pol = policies.GAAtariPolicy(policies._Space(shape=(84, 84, 4)), policies._Space(n=18), nonlin_type="relu")
    pol.set_trainable_flat(np.random.RandomState(3).randn(pol.num_params).astype(np.float32))
    fn = str(tmp_path / "ga.h5")
    pol.save(fn)
    out = subprocess.run([h5dump, "-H", fn], check=True, capture_output=True, text=True).stdout
    assert REF_ATTR_LAYOUT in out                              # same attribute names, classes, string flavour, dataspace
    for name, (off, shape) in pol.spec.items():                # 'GAAtariPolicy/conv1/w:0' -> nested groups, F32LE dataset
        leaf = name.split("/")[-1] + ":0"
        dims = ", ".join(str(d) for d in shape)
        assert 'DATASET "%s" {\n            DATATYPE  H5T_IEEE_F32LE\n            DATASPACE  SIMPLE { ( %s ) / ( %s ) }' % (leaf, dims, dims) in out
    assert 'GROUP "GAAtariPolicy" {' in out and 'GROUP "conv1" {' in out
    # and the values, as h5dump (an independent reader) prints them
    data = subprocess.run([h5dump, "-d", "/GAAtariPolicy/conv1/b:0", "-y", "-w", "0", fn], check=True, capture_output=True, text=True).stdout
    o, shape = pol.spec["conv1/b"]
    want = pol.get_trainable_flat()[o:o + int(np.prod(shape))]
    body = data[data.index("DATA {") + 6:data.rindex("}")]
    got = np.array([float(x) for x in body.replace("}", "").replace("\n", " ").split(",") if x.strip()], np.float32)
    assert np.allclose(got, want, rtol=1e-5, atol=1e-7)


@pytest.mark.skipif(not os.path.exists(REF_SNAPSHOT), reason="the reference's sample snapshot is not on this machine")
def test_reads_reference_written_snapshot():
    h5dump = _h5dump()
    if h5dump is not None:
        out = subprocess.run([h5dump, "-H", REF_SNAPSHOT], check=True, capture_output=True, text=True).stdout
        assert REF_ATTR_LAYOUT in out                          # the recorded layout is still what the reference ships
    name, (ob_shape, nact), kwargs, arrays = policies.Policy._read_snapshot(REF_SNAPSHOT)
    assert name == "MujocoPolicy" and ob_shape == (376,) and nact == 17            # gym Boxes unpickled without gym
    assert kwargs["hidden_dims"] == [256, 256] and kwargs["nonlin_type"] == "tanh"
    shapes = {k: v.shape for k, v in arrays.items()}
    assert shapes == {"MujocoPolicy/l0/b:0": (256,), "MujocoPolicy/l0/w:0": (376, 256), "MujocoPolicy/l1/b:0": (256,),
                      "MujocoPolicy/l1/w:0": (256, 256), "MujocoPolicy/ob_mean:0": (376,), "MujocoPolicy/ob_std:0": (376,),
                      "MujocoPolicy/out/b:0": (17,), "MujocoPolicy/out/w:0": (256, 17)}
    assert all(v.dtype == np.float32 and np.isfinite(v).all() for v in arrays.values())
    # the parent's flat parameters are also in the .dat next to it (es_modified.py:149-160 writes both from the same theta)?
    # no: the .dat holds the behaviour characterisation; the weights are pinned by their recorded sums instead
    assert abs(float(arrays["MujocoPolicy/l0/w:0"].sum()) - (-9.911567687988281)) < 1e-3
    assert abs(float(arrays["MujocoPolicy/ob_std:0"].sum()) - 2062.794677734375) < 1e-1


def test_policy_save_load_and_initialize_from_h5(tmp_path):
    ob, a18, a14 = policies._Space(shape=(84, 84, 4)), policies._Space(n=18), policies._Space(n=14)
    pol = policies.GAAtariPolicy(ob, a18, nonlin_type="relu")
    pol.set_trainable_flat(np.random.RandomState(5).randn(pol.num_params).astype(np.float32))
    fn = str(tmp_path / "p.h5")
    pol.save(fn)
    back = policies.GAAtariPolicy.Load(fn)
    assert np.array_equal(back.get_trainable_flat(), pol.get_trainable_flat()) and back.num_actions == 18
    small = policies.GAAtariPolicy(ob, a14, nonlin_type="relu")
    small.set_trainable_flat(np.random.RandomState(6).randn(small.num_params).astype(np.float32))
    small.save(str(tmp_path / "s.h5"))
    before = pol.get_trainable_flat()
    pol.initialize_from(str(tmp_path / "s.h5"))
    after, sf = pol.get_trainable_flat(), small.get_trainable_flat()
    o18, o14 = pol.spec["out/w"][0], small.spec["out/w"][0]
    assert np.array_equal(after[:o18], sf[:o14])
    assert np.array_equal(after[o18:o18 + 256 * 18].reshape(256, 18)[:, :14], sf[o14:o14 + 256 * 14].reshape(256, 14))
    assert np.array_equal(after[o18:o18 + 256 * 18].reshape(256, 18)[:, 14:], before[o18:o18 + 256 * 18].reshape(256, 18)[:, 14:])
    assert policies.snapshot_extension() == ".h5"


def test_args_pickle_is_what_a_stock_checkout_unpickles(tmp_path, monkeypatch):
    """policies.py:59-67: Load does `cls(*pickle.loads(attr))`; the attribute must therefore rebuild gym 0.9.4's Box / Discrete
    (requirements.txt:3).  A stand-in gym with that version's class shapes (shape is a property of low) checks it."""
    import types
    blob = policies._dumps_spaces((84, 84, 4), 18, {"nonlin_type": "relu"})
    assert len(blob) < 400 and b"numpy._core" not in blob and b"dne_hip" not in blob and "gym" not in sys.modules

    class Box:
        shape = property(lambda self: self.low.shape)

    class Discrete:
        pass
    Box.__module__, Discrete.__module__ = "gym.spaces.box", "gym.spaces.discrete"
    for name in ("gym", "gym.spaces", "gym.spaces.box", "gym.spaces.discrete"):
        monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    sys.modules["gym.spaces.box"].Box, sys.modules["gym.spaces.discrete"].Discrete = Box, Discrete
    (ob, ac), kwargs = pickle.loads(blob)                      # plain pickle, as the reference does
    assert type(ob) is Box and ob.shape == (84, 84, 4) and ob.low.dtype == np.float64 and ob.low.max() == 0 and ob.high.min() == 1
    assert type(ac) is Discrete and ac.n == 18 and kwargs == {"nonlin_type": "relu"}
    pol = policies.GAAtariPolicy(policies._Space(shape=(84, 84, 4)), policies._Space(n=18), nonlin_type="relu")
    pol.set_trainable_flat(np.zeros(pol.num_params, np.float32))
    fn = str(tmp_path / "g.h5")
    pol.save(fn)
    assert h5lite.read_snapshot(fn)[1] == policies._dumps_spaces((84, 84, 4), 18, pol.kwargs)
    assert policies.GAAtariPolicy.Load(fn).num_actions == 18
