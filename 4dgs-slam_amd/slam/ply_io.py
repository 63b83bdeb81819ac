"""Binary PLY reader / writer for the Gaussian map: the file layout ``plyfile`` produces for the reference's ``GaussianModel.save_ply``
(gaussian_splatting/scene/gaussian_model.py:584-620: one ``vertex`` element, every property ``float`` (f4), native byte order =
little endian here, columns x y z nx ny nz f_dc_* f_rest_* opacity scale_* rot_* dygs) and that ``load_ply`` (:640-733) parses.
``plyfile`` is not available in this image; the format is small enough to state directly."""
import numpy as np

_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1", "char": "i1", "int8": "i1",
          "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2", "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def write_ply(path, names, data):
    """data float32 [P, len(names)] -> binary little-endian PLY with one float property per column (plyfile's PlyData([el]).write)."""
    data = np.ascontiguousarray(data, dtype="<f4")
    assert data.ndim == 2 and data.shape[1] == len(names)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {data.shape[0]}"]
    header += [f"property float {n}" for n in names]
    header.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(data.tobytes())


def read_ply(path):
    """-> (property names, float32 array [P, n]) of the ``vertex`` element. Handles binary little-endian and ascii files whose vertex
    properties are scalars (what 3DGS-style writers produce); list properties and big-endian files raise."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
                elif count is None:
                    raise ValueError(f"{path}: an element precedes `vertex`; not supported")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported")
                props.append((tok[2], _TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if count is None:
            raise ValueError(f"{path}: no vertex element")
        names = [n for n, _ in props]
        if fmt == "binary_little_endian":
            rec = np.dtype([(n, t) for n, t in props])
            raw = np.frombuffer(f.read(count * rec.itemsize), dtype=rec, count=count)
            out = np.stack([raw[n].astype(np.float32) for n in names], axis=1) if names else np.zeros((count, 0), np.float32)
        elif fmt == "ascii":
            out = np.loadtxt(f, dtype=np.float32, max_rows=count).reshape(count, len(names))
        else:
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
    return names, out
