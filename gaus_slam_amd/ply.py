"""On-disk map layout of the reference (scene/Gaussians.py:435-464 `construct_list_of_attributes` / `save_ply`, `:466-`
`load_ply`): one binary little-endian PLY `vertex` element of float32 properties
    x y z nx ny nz opacity scale_0..scale_{S-1} rot_0..rot_3  then  r g b   or   f_dc_* f_rest_*
The reference goes through the `plyfile` package (not installed here); this is a dependency-free reader/writer of the
same byte layout (what `PlyData([PlyElement.describe(elements, 'vertex')]).write(path)` produces), so maps saved by the
reference can be benchmarked and maps saved here load in the reference."""
import os

import numpy as np


def attribute_names(n_scale=2, n_rot=4, use_sh=False, n_dc=3, n_rest=0):
    names = ["x", "y", "z", "nx", "ny", "nz", "opacity"]
    names += [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)]
    if use_sh:
        names += [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)]
    else:
        names += list("rgb")
    return names


def save_ply(path, xyz, opacity, scaling, rotation, rgb=None, f_dc=None, f_rest=None):
    """Arrays are [P,k] (raw, pre-activation values exactly as the reference stores them).  Either `rgb` [P,3] or
    `f_dc` [P,1,3] (+ `f_rest` [P,M-1,3]) as held by the reference's parameters; SH columns are written channel-major
    ([P,3,M-1] flattened), the ordering `load_ply` (scene/Gaussians.py:484-496) expects.  (The reference's own `save_ply`
    cannot run with use_sh=True -- it concatenates 2-D and 3-D arrays -- so its loader defines the layout.)"""
    xyz = np.asarray(xyz, np.float32)
    P = xyz.shape[0]
    cols = [xyz, np.zeros_like(xyz), np.asarray(opacity, np.float32).reshape(P, 1), np.asarray(scaling, np.float32).reshape(P, -1),
            np.asarray(rotation, np.float32).reshape(P, -1)]
    use_sh = rgb is None
    if use_sh:
        dc = np.asarray(f_dc, np.float32).reshape(P, -1, 3).transpose(0, 2, 1).reshape(P, -1)
        rest = (np.zeros((P, 0), np.float32) if f_rest is None
                else np.asarray(f_rest, np.float32).reshape(P, -1, 3).transpose(0, 2, 1).reshape(P, -1))
        cols += [dc, rest]
        names = attribute_names(cols[3].shape[1], cols[4].shape[1], True, dc.shape[1], rest.shape[1])
    else:
        cols.append(np.asarray(rgb, np.float32).reshape(P, 3))
        names = attribute_names(cols[3].shape[1], cols[4].shape[1], False)
    table = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")
    assert table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as fh:
        fh.write(header.encode("ascii"))
        fh.write(table.tobytes())


_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1", "char": "i1",
              "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2", "int": "<i4", "int32": "<i4",
              "uint": "<u4", "uint32": "<u4"}


def read_vertex_table(path):
    """-> (names, structured numpy array) of the first element of a binary little-endian or ascii PLY."""
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, count, props, in_first, n_elements = None, None, [], False, 0
        while True:
            line = fh.readline()
            if not line:
                raise ValueError("PLY header not terminated")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                n_elements += 1
                in_first = n_elements == 1
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError("list properties are not supported in the vertex element")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "binary_little_endian":
            data = np.fromfile(fh, dtype=np.dtype(props), count=count)
        elif fmt == "ascii":
            rows = np.loadtxt(fh, max_rows=count, ndmin=2)
            data = np.empty(count, dtype=np.dtype(props))
            for j, (n, _) in enumerate(props):
                data[n] = rows[:, j]
        else:
            raise ValueError(f"unsupported PLY format {fmt}")
    if data.shape[0] != count:
        raise ValueError("PLY vertex data truncated")
    return [n for n, _ in props], data


def load_ply(path):
    """-> dict(xyz [P,3], opacity [P,1], scaling [P,S], rotation [P,4], and rgb [P,3] or f_dc [P,1,3] + f_rest [P,M-1,3]),
    float32, with the column ordering rules of load_ply (scale_/rot_/f_rest_ sorted by their integer suffix)."""
    names, d = read_vertex_table(path)

    def cols(prefix):
        ns = sorted([n for n in names if n.startswith(prefix)], key=lambda s: int(s.split("_")[-1]))
        return np.stack([d[n].astype(np.float32) for n in ns], axis=1) if ns else np.zeros((d.shape[0], 0), np.float32)

    out = {"xyz": np.stack([d["x"], d["y"], d["z"]], axis=1).astype(np.float32),
           "opacity": d["opacity"].astype(np.float32)[:, None], "scaling": cols("scale_"), "rotation": cols("rot_")}
    if "f_dc_0" in names:
        P = d.shape[0]
        out["f_dc"] = cols("f_dc_").reshape(P, 3, 1).transpose(0, 2, 1).copy()
        rest = cols("f_rest_")
        out["f_rest"] = rest.reshape(P, 3, rest.shape[1] // 3).transpose(0, 2, 1).copy()
    else:
        out["rgb"] = np.stack([d["r"], d["g"], d["b"]], axis=1).astype(np.float32)
    return out
