"""Flat model-blob container ("GRPM" format).

The compiled model is a list of named little-endian arrays so that the HIP
product library, the C oracle and Python can all read the same file without
sharing any code:

    header : magic "GRPM" | u32 version | u32 n_entries
    entry  : char name[24] | u32 dtype (0 = f64, 1 = i32) | u32 ndim |
             u32 dims[4] | payload (row-major, padded to 8 bytes)

Nothing in here comes from the reference tree: the blob is this repo's own
interchange format for the quantities a MuJoCo compile of
``xmls/*_env.xml`` would hold in ``mjModel``.
"""
import struct

import numpy as np

MAGIC = b"GRPM"
VERSION = 3
_DT = {0: np.dtype("<f8"), 1: np.dtype("<i4")}


def write_blob(path, arrays):
    """arrays: dict name -> ndarray (float64 or int32)."""
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<II", VERSION, len(arrays)))
        for name, a in arrays.items():
            a = np.asarray(a)
            if a.dtype.kind == "f":
                code, a = 0, np.ascontiguousarray(a, dtype="<f8")
            elif a.dtype.kind in "iub":
                code, a = 1, np.ascontiguousarray(a, dtype="<i4")
            else:
                raise TypeError(f"{name}: unsupported dtype {a.dtype}")
            if a.ndim == 0:
                a = a.reshape(1)
            if a.ndim > 4:
                raise ValueError(f"{name}: ndim > 4")
            nb = name.encode()
            if len(nb) > 23:
                raise ValueError(f"name too long: {name}")
            f.write(nb.ljust(24, b"\0"))
            dims = list(a.shape) + [1] * (4 - a.ndim)
            f.write(struct.pack("<II4I", code, a.ndim, *dims))
            payload = a.tobytes()
            f.write(payload)
            f.write(b"\0" * ((-len(payload)) % 8))


def read_blob(path):
    out = {}
    with open(path, "rb") as f:
        if f.read(4) != MAGIC:
            raise ValueError(f"{path}: not a GRPM blob")
        version, n = struct.unpack("<II", f.read(8))
        if version != VERSION:
            raise ValueError(f"{path}: blob version {version}, expected {VERSION}")
        for _ in range(n):
            name = f.read(24).rstrip(b"\0").decode()
            code, ndim, *dims = struct.unpack("<II4I", f.read(24))
            shape = tuple(dims[:ndim])
            dt = _DT[code]
            count = int(np.prod(shape)) if shape else 1
            nbytes = count * dt.itemsize
            a = np.frombuffer(f.read(nbytes), dtype=dt).reshape(shape).copy()
            f.read((-nbytes) % 8)
            out[name] = a
    return out
