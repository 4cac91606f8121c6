"""Minimal stand-in for the `plyfile` package (not installed here), for tests/golden/make_golden_initdir.py and the drop-in
test ONLY: enough of its API for the reference's own storePly / fetchPly / GaussianModel.save_ply / load_ply to run — one
element, scalar properties, `binary_little_endian` on write (plyfile's default on this machine), binary or ascii on read.
It follows the PLY specification, not instantsplat_amd.io_formats, so that the reference's readers and writers meet the
product's through the FILE FORMAT and not through shared code.  Test infrastructure."""
import numpy as np

_NAMES = {"f4": "float", "f8": "double", "u1": "uchar", "i1": "char", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint"}
_TYPES = {v: k for k, v in _NAMES.items()}
_TYPES.update({"float32": "f4", "float64": "f8", "uint8": "u1", "int8": "i1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4"})


class _Property:
    def __init__(self, name):
        self.name = name


class PlyElement:
    def __init__(self, name, data):
        self.name, self.data = name, data
        self.properties = [_Property(n) for n in data.dtype.names]

    @staticmethod
    def describe(data, name):
        return PlyElement(name, np.asarray(data))

    def __getitem__(self, key):
        return self.data[key]

    def __len__(self):
        return len(self.data)


class PlyData:
    def __init__(self, elements):
        self.elements = list(elements)

    def __getitem__(self, name):
        return next(e for e in self.elements if e.name == name)

    def write(self, path):
        with open(path, "wb") as f:
            hdr = ["ply", "format binary_little_endian 1.0"]
            for e in self.elements:
                hdr.append("element %s %d" % (e.name, len(e.data)))
                for n in e.data.dtype.names:
                    dt = e.data.dtype[n]
                    hdr.append("property %s %s" % (_NAMES[dt.kind + str(dt.itemsize)], n))
            hdr.append("end_header")
            f.write(("\n".join(hdr) + "\n").encode("ascii"))
            for e in self.elements:
                f.write(e.data.astype(e.data.dtype.newbyteorder("<")).tobytes())

    @staticmethod
    def read(path):
        with open(path, "rb") as f:
            assert f.readline().strip() == b"ply"
            fmt, elems = None, []
            while True:
                t = f.readline().decode("ascii").split()
                if not t or t[0] == "comment":
                    continue
                if t[0] == "format":
                    fmt = t[1]
                elif t[0] == "element":
                    elems.append([t[1], int(t[2]), []])
                elif t[0] == "property":
                    assert t[1] != "list"
                    elems[-1][2].append((t[2], "<" + _TYPES[t[1]]))
                elif t[0] == "end_header":
                    break
            out = []
            for name, count, props in elems:
                dt = np.dtype(props)
                if fmt == "binary_little_endian":
                    data = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count).copy()
                else:
                    assert fmt == "ascii"
                    data = np.array([tuple(f.readline().split()) for _ in range(count)], dtype=dt)
                out.append(PlyElement(name, data))
        return PlyData(out)
