"""The reference's policy snapshot files (es_distributed/policies.py:49-67) through the HDF5 C library, without h5py.

Policy.save writes, with h5py: one float32 dataset per TF variable under its full name ('ESAtariPolicy/conv1/weights:0' -- the
slashes make nested groups), and two attributes on the root group: 'name' (a variable-length UTF-8 string) and
'args_and_kwargs' (np.void of a pickle = an opaque scalar with an empty tag).  `h5dump -H` of the snapshot the reference ships
(visual_inspector/sample_data/.../snapshot_parent_0097.h5) shows exactly that layout; this module produces and reads it by
calling libhdf5 (1.10 / 1.12 / 1.14 API subset common to all) through ctypes.  h5py is used instead when it is installed
(policies.py decides); neither exists -> the drivers fall back to the .npz container and tools/npz_to_h5.py.
"""
import ctypes as C
import ctypes.util
import glob
import os
from collections import OrderedDict

import numpy as np

hid_t = C.c_int64
herr_t = C.c_int
hsize_t = C.c_uint64
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5S_SCALAR, H5T_OPAQUE, H5T_STRING_CLASS, H5T_FLOAT_CLASS = 0, 5, 3, 1
H5T_CSET_UTF8 = 1
H5T_VARIABLE = C.c_size_t(-1).value
H5_INDEX_NAME, H5_ITER_INC = 0, 0
H5F_LIBVER_LATEST_NAMES = ("H5F_LIBVER_LATEST",)

_lib = None


def _candidates():
    env = os.environ.get("DNE_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for pat in ("/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*",
                "/opt/conda/lib/libhdf5.so*", "/usr/local/lib/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            yield p


def lib():
    """libhdf5 with argtypes set, or None when no usable library is on this machine"""
    global _lib
    if _lib is not None:
        return _lib or None
    for path in _candidates():
        try:
            L = C.CDLL(path)
            L.H5open.restype = herr_t
            if L.H5open() < 0:
                continue
            sig = {
                "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
                "H5Fclose": (herr_t, [hid_t]), "H5Pcreate": (hid_t, [hid_t]), "H5Pclose": (herr_t, [hid_t]),
                "H5Pset_create_intermediate_group": (herr_t, [hid_t, C.c_uint]), "H5Pset_libver_bounds": (herr_t, [hid_t, C.c_int, C.c_int]),
                "H5Screate": (hid_t, [C.c_int]), "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
                "H5Sclose": (herr_t, [hid_t]), "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
                "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
                "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]), "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
                "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]), "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
                "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]), "H5Dclose": (herr_t, [hid_t]),
                "H5Tcopy": (hid_t, [hid_t]), "H5Tcreate": (hid_t, [C.c_int, C.c_size_t]), "H5Tset_size": (herr_t, [hid_t, C.c_size_t]),
                "H5Tset_cset": (herr_t, [hid_t, C.c_int]), "H5Tget_size": (C.c_size_t, [hid_t]), "H5Tget_class": (C.c_int, [hid_t]),
                "H5Tis_variable_str": (C.c_int, [hid_t]), "H5Tclose": (herr_t, [hid_t]),
                "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), "H5Awrite": (herr_t, [hid_t, hid_t, C.c_void_p]),
                "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Aread": (herr_t, [hid_t, hid_t, C.c_void_p]),
                "H5Aget_type": (hid_t, [hid_t]), "H5Aget_space": (hid_t, [hid_t]), "H5Aclose": (herr_t, [hid_t]),
                "H5Dvlen_reclaim": (herr_t, [hid_t, hid_t, hid_t, C.c_void_p]),
                "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
            }
            for name, (res, args) in sig.items():
                f = getattr(L, name)
                f.restype, f.argtypes = res, args
            L.g = lambda n, L=L: hid_t.in_dll(L, n).value           # the macro constants are globals set by H5open()
            L.H5Eset_auto2(0, None, None)                            # errors come back as return codes, not as stderr dumps
            _lib = L
            return L
        except (OSError, AttributeError, ValueError):
            continue
    _lib = False
    return None


def available():
    return lib() is not None


class H5Error(RuntimeError):
    pass


def _ck(v, what):
    if v < 0:
        raise H5Error("libhdf5: %s failed" % what)
    return v


def write_snapshot(filename, arrays, name, args_blob):
    """policies.py:49-57: `for v in all_variables: f[v.name] = v.eval(); f.attrs['name'] = ...; f.attrs['args_and_kwargs'] = np.void(...)`"""
    L = lib()
    if L is None:
        raise H5Error("no libhdf5 on this machine")
    fapl = _ck(L.H5Pcreate(L.g("H5P_CLS_FILE_ACCESS_ID_g")), "H5Pcreate(fapl)")
    L.H5Pset_libver_bounds(fapl, 0, 2 if hasattr(L, "H5Pset_libver_bounds") else 0)   # earliest .. v110+: readable by every h5py
    f = _ck(L.H5Fcreate(filename.encode(), H5F_ACC_TRUNC, 0, fapl), "H5Fcreate(%s)" % filename)
    try:
        lcpl = _ck(L.H5Pcreate(L.g("H5P_CLS_LINK_CREATE_ID_g")), "H5Pcreate(lcpl)")
        _ck(L.H5Pset_create_intermediate_group(lcpl, 1), "H5Pset_create_intermediate_group")
        for key, val in arrays.items():
            a = np.require(np.asarray(val, np.float32), requirements='C')    # (ascontiguousarray would make a scalar 1-d)
            dims = (hsize_t * max(a.ndim, 1))(*a.shape)
            space = _ck(L.H5Screate_simple(a.ndim, dims, None) if a.ndim else L.H5Screate(H5S_SCALAR), "H5Screate_simple")
            d = _ck(L.H5Dcreate2(f, key.encode(), L.g("H5T_IEEE_F32LE_g"), space, lcpl, 0, 0), "H5Dcreate2(%s)" % key)
            _ck(L.H5Dwrite(d, L.g("H5T_NATIVE_FLOAT_g"), 0, 0, 0, a.ctypes.data_as(C.c_void_p)), "H5Dwrite(%s)" % key)
            L.H5Dclose(d); L.H5Sclose(space)
        L.H5Pclose(lcpl)
        scalar = _ck(L.H5Screate(H5S_SCALAR), "H5Screate")
        st = _ck(L.H5Tcopy(L.g("H5T_C_S1_g")), "H5Tcopy")                      # 'name': variable-length UTF-8 string, as h5py stores a str
        L.H5Tset_size(st, H5T_VARIABLE); L.H5Tset_cset(st, H5T_CSET_UTF8)
        at = _ck(L.H5Acreate2(f, b"name", st, scalar, 0, 0), "H5Acreate2(name)")
        s = C.c_char_p(name.encode("utf-8"))
        _ck(L.H5Awrite(at, st, C.byref(s)), "H5Awrite(name)")
        L.H5Aclose(at); L.H5Tclose(st)
        ot = _ck(L.H5Tcreate(H5T_OPAQUE, len(args_blob)), "H5Tcreate(opaque)")  # 'args_and_kwargs': np.void -> opaque, empty tag
        at = _ck(L.H5Acreate2(f, b"args_and_kwargs", ot, scalar, 0, 0), "H5Acreate2(args_and_kwargs)")
        buf = C.create_string_buffer(bytes(args_blob), len(args_blob))
        _ck(L.H5Awrite(at, ot, buf), "H5Awrite(args_and_kwargs)")
        L.H5Aclose(at); L.H5Tclose(ot); L.H5Sclose(scalar)
    finally:
        L.H5Fclose(f)
        L.H5Pclose(fapl)


_VISIT_CB = C.CFUNCTYPE(herr_t, hid_t, C.c_char_p, C.c_void_p, C.c_void_p)


def _object_names(L, f):
    names = []

    def cb(obj, name, info, data):
        names.append(name.decode())
        return 0
    fn = _VISIT_CB(cb)
    visit = L.H5Ovisit1 if hasattr(L, "H5Ovisit1") else L.H5Ovisit
    visit.restype, visit.argtypes = herr_t, [hid_t, C.c_int, C.c_int, _VISIT_CB, C.c_void_p]
    _ck(visit(f, H5_INDEX_NAME, H5_ITER_INC, fn, None), "H5Ovisit")
    return [n for n in names if n != "."]


def read_snapshot(filename):
    """-> (name, args_blob, OrderedDict{dataset path: float32 array}) of a snapshot written by the reference (h5py), by h5py on
    our side, or by write_snapshot"""
    L = lib()
    if L is None:
        raise H5Error("no libhdf5 on this machine")
    f = _ck(L.H5Fopen(filename.encode(), H5F_ACC_RDONLY, 0), "H5Fopen(%s)" % filename)
    try:
        arrays = OrderedDict()
        for n in _object_names(L, f):
            d = L.H5Dopen2(f, n.encode(), 0)
            if d < 0:
                continue                                   # a group
            space = L.H5Dget_space(d)
            nd = L.H5Sget_simple_extent_ndims(space)
            dims = (hsize_t * max(nd, 1))()
            if nd > 0:
                L.H5Sget_simple_extent_dims(space, dims, None)
            a = np.empty(tuple(int(x) for x in dims[:nd]), np.float32)
            _ck(L.H5Dread(d, L.g("H5T_NATIVE_FLOAT_g"), 0, 0, 0, a.ctypes.data_as(C.c_void_p)), "H5Dread(%s)" % n)
            L.H5Sclose(space); L.H5Dclose(d)
            arrays[n] = a
        at = _ck(L.H5Aopen(f, b"args_and_kwargs", 0), "H5Aopen(args_and_kwargs)")
        t = L.H5Aget_type(at)
        blob = C.create_string_buffer(L.H5Tget_size(t))
        _ck(L.H5Aread(at, t, blob), "H5Aread(args_and_kwargs)")
        L.H5Tclose(t); L.H5Aclose(at)
        at = _ck(L.H5Aopen(f, b"name", 0), "H5Aopen(name)")
        t = L.H5Aget_type(at)
        if L.H5Tis_variable_str(t) > 0:
            p = C.c_char_p()
            _ck(L.H5Aread(at, t, C.byref(p)), "H5Aread(name)")
            name = p.value.decode("utf-8")
            sp = L.H5Aget_space(at)
            L.H5Dvlen_reclaim(t, sp, 0, C.byref(p))
            L.H5Sclose(sp)
        else:
            sb = C.create_string_buffer(L.H5Tget_size(t) + 1)
            _ck(L.H5Aread(at, t, sb), "H5Aread(name)")
            name = sb.value.decode("utf-8")
        L.H5Tclose(t); L.H5Aclose(at)
        return name, blob.raw, arrays
    finally:
        L.H5Fclose(f)
