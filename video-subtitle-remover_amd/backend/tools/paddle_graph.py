"""Reader for the PaddlePaddle PIR inference programs the reference ships for its text detector
(backend/models/V5/{ch_det,ch_det_fast}/inference.json, PP-OCRv5 server / mobile detection; loaded by paddleocr's
TextDetection in backend/tools/subtitle_detect.py:41-54).  Host-side, pure Python: turns the JSON program into a flat
list of ops over integer value ids that the GPU runner (ocr_det.py) and the test oracle (oracle/ppocr_det.py) both walk.
"""
import json


class Graph:
    """params: {value_id: (name, shape)}; ops: [(type, [input ids], [output ids], {attr: value})] in program order;
    input_id / output_id: the data and fetch values."""

    def __init__(self, params, ops, input_id, output_id):
        self.params, self.ops, self.input_id, self.output_id = params, ops, input_id, output_id

    def to_json(self):
        return {"params": {str(k): [v[0], list(v[1])] for k, v in self.params.items()}, "ops": [[t, i, o, a] for t, i, o, a in self.ops],
                "input": self.input_id, "output": self.output_id}

    @staticmethod
    def from_json(d):
        return Graph({int(k): (v[0], tuple(v[1])) for k, v in d["params"].items()}, [(t, i, o, a) for t, i, o, a in d["ops"]], d["input"], d["output"])


def _attr_value(at):
    d = at.get("D")
    if isinstance(d, list) and d and isinstance(d[0], dict):
        return [e.get("D") for e in d]
    return d


def load_graph(path_or_dict):
    """inference.json (PIR program) or the condensed form written by Graph.to_json -> Graph"""
    d = path_or_dict
    if not isinstance(d, dict):
        with open(path_or_dict) as f:
            d = json.load(f)
    if "ops" in d and "params" in d:
        return Graph.from_json(d)
    block = d["program"]["regions"][0]["blocks"][0]
    params, ops, input_id, output_id = {}, [], None, None
    for o in block["ops"]:
        t = o["#"]
        if t == "p":                                   # builtin.parameter: A[3] = name
            params[o["O"]["%"]] = (o["A"][3], tuple(o["O"]["TT"]["D"][1]))
            continue
        kind = t.split(".", 1)[1]
        ins = [i["%"] for i in o.get("I", [])]
        outs = [v["%"] for v in o.get("O", [])]
        attrs = {a["N"]: _attr_value(a["AT"]) for a in o.get("A", []) if isinstance(a, dict) and "N" in a and a["N"] != "struct_name"}
        if kind == "data":
            input_id = outs[0]
            continue
        if kind == "fetch":
            output_id = ins[0]
            continue
        ops.append((kind, ins, outs, attrs))
    return Graph(params, ops, input_id, output_id)


# ---- inference.pdiparams ------------------------------------------------------------------------------------------
# The weights file paddleocr reads beside inference.json.  It is a plain concatenation of serialized dense tensors
# (Paddle's save_combine: per tensor  u32 version | u64 lod levels (+ each level: u64 byte size, data) | u32 tensor version |
# i32 descriptor size | VarType.TensorDesc protobuf {1: data_type enum, 2: repeated int64 dims} | raw little-endian data),
# without names: the order is that of the parameter list handed to save_combine.  The file is absent from the reference
# checkout and Paddle is absent from this image, so this reader is restated from the published format and exercised only
# against files written by tests/test_ocr_det_host.py's writer of the same layout -- unverified against a real one.
_PD_DTYPES = {0: "bool", 1: "<i2", 2: "<i4", 3: "<i8", 4: "<f2", 5: "<f4", 6: "<f8", 20: "u1", 21: "i1"}


def _varint(buf, p):
    v, s = 0, 0
    while True:
        b = buf[p]
        p += 1
        v |= (b & 0x7F) << s
        if not b & 0x80:
            return v, p
        s += 7


def _tensor_desc(buf):
    dtype, dims, p = None, [], 0
    while p < len(buf):
        key, p = _varint(buf, p)
        field, wire = key >> 3, key & 7
        if wire == 0:
            v, p = _varint(buf, p)
            if field == 1:
                dtype = v
            elif field == 2:
                dims.append(v - (1 << 64) if v >> 63 else v)
        elif wire == 2:                                  # packed dims
            n, p = _varint(buf, p)
            end = p + n
            while p < end:
                v, p = _varint(buf, p)
                if field == 2:
                    dims.append(v - (1 << 64) if v >> 63 else v)
        else:
            raise ValueError(f"unexpected wire type {wire} in a TensorDesc")
    if dtype not in _PD_DTYPES:
        raise ValueError(f"unsupported Paddle dtype {dtype}")
    return _PD_DTYPES[dtype], tuple(dims)


def read_pdiparams_tensors(path):
    """-> [ndarray] in file order"""
    import struct

    import numpy as np
    with open(path, "rb") as f:
        buf = f.read()
    out, p = [], 0
    while p < len(buf):
        (ver,) = struct.unpack_from("<I", buf, p)
        p += 4
        (levels,) = struct.unpack_from("<Q", buf, p)
        p += 8
        if ver != 0 or levels > 8:
            raise ValueError(f"{path}: not a save_combine stream at byte {p - 12}")
        for _ in range(levels):
            (nbytes,) = struct.unpack_from("<Q", buf, p)
            p += 8 + nbytes
        (tver, dsize) = struct.unpack_from("<Ii", buf, p)
        p += 8
        if tver != 0 or dsize <= 0 or p + dsize > len(buf):
            raise ValueError(f"{path}: bad tensor header at byte {p - 8}")
        dt, dims = _tensor_desc(buf[p:p + dsize])
        p += dsize
        n = 1
        for d in dims:
            n *= d
        a = np.frombuffer(buf, dtype=np.dtype(dt), count=n, offset=p).reshape(dims)
        p += a.nbytes
        out.append(a)
    return out


def read_pdiparams(path, graph):
    """{parameter name: float32 array} for `graph`.  The stream carries no names, so the tensors are matched to the
    program's parameters by order -- sorted by name (what save_inference_model passes) or, failing that, program order --
    and the assignment is accepted only if every shape agrees."""
    import numpy as np
    tensors = read_pdiparams_tensors(path)
    plist = [graph.params[k] for k in sorted(graph.params)]           # program order
    if len(tensors) != len(plist):
        raise ValueError(f"{path}: {len(tensors)} tensors for a program with {len(plist)} parameters")
    for order in (sorted(plist, key=lambda nv: nv[0]), plist):
        if all(tuple(t.shape) == tuple(shape) for t, (_, shape) in zip(tensors, order)):
            return {name: np.ascontiguousarray(t, dtype=np.float32) for t, (name, _) in zip(tensors, order)}
    raise ValueError(f"{path}: tensor shapes match the program's parameters in neither name nor program order")
