"""Prints the detector program op by op with tensor shapes (a reading aid for the NHWC plan of ocr_det.py)."""
import sys, importlib
import numpy as np
sys.path.insert(0, '.')
pg = importlib.import_module('video-subtitle-remover_amd.backend.tools.paddle_graph')
path = sys.argv[1] if len(sys.argv) > 1 else '/root/reference/backend/models/V5/ch_det/inference.json'
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = pg.load_graph(path)
shape = {g.input_id: (8, 3, 544, 960)}
for v, (n, s) in g.params.items():
    shape[v] = tuple(s)
val = {}
for i, (kind, ins, outs, a) in enumerate(g.ops):
    s = None
    line = None
    if kind in ('conv2d', 'depthwise_conv2d'):
        n, c, h, w = shape[ins[0]]; co, ci, kh, kw = shape[ins[1]]; sh, sw = a['strides']; pt, pl = a['paddings'][:2]
        if a.get('padding_algorithm') == 'SAME': ho, wo = -(-h // sh), -(-w // sw)
        else: ho, wo = (h + 2 * pt - kh) // sh + 1, (w + 2 * pl - kw) // sw + 1
        s = (n, co, ho, wo)
        line = f"in {ins[0]} {shape[ins[0]]} w {shape[ins[1]]} s {sh} p {pt} {a.get('padding_algorithm')} g {a['groups']}"
    elif kind == 'conv2d_transpose':
        n, c, h, w = shape[ins[0]]; ci, co, kh, kw = shape[ins[1]]; s = (n, co, 2 * h, 2 * w)
        line = f"in {ins[0]} {shape[ins[0]]} w {shape[ins[1]]} g {a['groups']}"
    elif kind == 'full_int_array': val[outs[0]] = a['value']; continue
    elif kind == 'full': val[outs[0]] = a['value']; continue
    elif kind == 'reshape':
        s = tuple(val[ins[1]]); line = f"{ins[0]}"
    elif kind in ('add', 'multiply'):
        sa, sb = shape[ins[0]], shape[ins[1]]
        s = sa if np.prod(sa) >= np.prod(sb) else sb
        line = f"{ins} {sa} {sb}"
    elif kind in ('batch_norm_', 'relu', 'hardswish', 'hardsigmoid', 'sigmoid', 'scale'):
        s = shape[ins[0]]; line = f"{ins[0]}"
    elif kind == 'pool2d':
        n, c, h, w = shape[ins[0]]
        if a['adaptive']: s = (n, c, 1, 1)
        else:
            ks = val[ins[1]]; sh, sw = a['strides']; pt, pl = a['paddings'][:2]
            if a.get('padding_algorithm') == 'SAME': ho, wo = -(-h // sh), -(-w // sw)
            elif a['ceil_mode']: ho, wo = -(-(h + 2 * pt - ks[0]) // sh) + 1, -(-(w + 2 * pl - ks[1]) // sw) + 1
            else: ho, wo = (h + 2 * pt - ks[0]) // sh + 1, (w + 2 * pl - ks[1]) // sw + 1
            s = (n, c, ho, wo)
        line = f"{a['pooling_type']} {'adaptive' if a['adaptive'] else val[ins[1]]} s {a['strides']} {ins[0]}"
    elif kind == 'nearest_interp':
        n, c, h, w = shape[ins[0]]; sc = int(a['scale'][0]); s = (n, c, h * sc, w * sc); line = f"{ins[0]}"
    elif kind == 'combine':
        val[outs[0]] = ins; continue
    elif kind == 'concat':
        parts = val[ins[0]]; ss = [shape[p] for p in parts]; s = (ss[0][0], sum(x[1] for x in ss), ss[0][2], ss[0][3]); line = f"{parts}"
    else:
        line = '??'
    if s is not None: shape[outs[0]] = s
    if i >= lo: print(i, kind, line, '->', outs[0], s)
print('output', g.output_id)
