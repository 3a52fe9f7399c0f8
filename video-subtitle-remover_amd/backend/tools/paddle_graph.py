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
