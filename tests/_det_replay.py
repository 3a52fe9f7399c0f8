"""CPU execution of a compiled detector plan (vsr_amd/backend/tools/ocr_det_nhwc.py): every step kind with the semantics of the launcher it
maps to (include/vsr_hip.h: vsr_gemm_plan_run's GGProblem, vsr_det_launch_*), on flat float32 numpy buffers.  Test infrastructure: the
plan's tables, folds, slices and halos are checked against the program interpreter (oracle/ppocr_det.py) without a GPU."""
import numpy as np
import torch
import torch.nn.functional as F


def _act(v, code):
    if code == 1:
        return np.maximum(v, 0)
    if code == 2:
        return v * np.clip(v + 3.0, 0.0, 6.0) / 6.0
    if code == 3:
        return 1.0 / (1.0 + np.exp(-v))
    return v


def _view(buf, off, n, H, W, C, img, row, cs):
    """[n][H][W][C] strided window on a flat buffer (no copy); negative halo reach is the caller's arithmetic"""
    return np.lib.stride_tricks.as_strided(buf[off:], shape=(n, H, W, C), strides=(img * 4, row * 4, cs * 4, 4))


def run_plan(plan, x, after_step=None):
    """x: float32 [n,3,H,W] -> output array of plan.output's shape"""
    bufs = {k: (np.zeros(sz, np.float32) if zero else np.full(sz, np.nan, np.float32)) for k, (sz, zero) in plan.buffers.items()}
    C_ = plan.consts
    bufs[plan.input][:] = np.asarray(x, np.float32).reshape(-1)
    len_done = 0
    for kind, p in plan.steps:
        if kind == "gemm":
            t = {k: C_[v].astype(np.int64) for k, v in p["tables"].items()}
            M, N, K = p["M"], p["N"], p["K"]
            A, Cb = bufs[p["A"]], bufs[p["C"]]
            ra, ca = t["rowA"][:M], t["colA"]
            idx = (ra[:, None, None] + ca[None, :, None] + np.arange(32)[None, None, :]).reshape(M, K)
            Am = A[idx]
            assert np.isfinite(Am).all(), f"{p['tag']}: the A operand reads memory nothing wrote"
            B = C_[p["B"]]
            bidx = (t["rowB"][:N, None, None] + t["colB"][None, :, None] + np.arange(32)[None, None, :]).reshape(N, K)
            acc = torch.from_numpy(Am) @ torch.from_numpy(B[bidx]).t()
            acc = acc.numpy()
            if p["bias"] is not None:
                acc = acc + C_[p["bias"]][None, :N]
            if p["act"] == 2:
                acc = np.maximum(acc, 0)
            else:
                assert p["act"] == 0
            cidx = (t["rowC"][:M, None, None] + t["colC"][None, :N // 32, None] + np.arange(32)[None, None, :]).reshape(M, N)
            if p["R"] is not None:
                ridx = (t["rowR"][:M, None, None] + t["colC"][None, :N // 32, None] + np.arange(32)[None, None, :]).reshape(M, N)
                r = bufs[p["R"]][ridx]
                assert np.isfinite(r).all(), f"{p['tag']}: the residual reads memory nothing wrote"
                acc = acc + r
            assert len(np.unique(cidx)) == cidx.size, f"{p['tag']}: two outputs share an address"
            Cb[cidx] = acc
        elif kind == "to_view":
            n, C, H, W, Cw = p["n"], p["C"], p["H"], p["W"], p["Cw"]
            src = bufs[p["x"]][:n * C * H * W].reshape(n, C, H, W)
            dst = _view(bufs[p["out"]], p["out_off"], n, H, W, Cw, p["img_stride"], p["row_stride"], p["Cs"])
            dst[..., :C] = src.transpose(0, 2, 3, 1)
            dst[..., C:] = 0
        elif kind == "from_view":
            n, C, H, W = p["n"], p["C"], p["H"], p["W"]
            src = _view(bufs[p["inp"]], p["in_off"], n, H, W, C, p["img_stride"], p["row_stride"], p["Cs"])
            out = bufs[p["out"]]
            for b in range(n):
                o = p["out_off"] + b * p["out_img_stride"]
                out[o:o + C * H * W] = src[b].transpose(2, 0, 1).reshape(-1)
        elif kind == "dwconv_view":
            n, C, kh, kw, sh, sw, pt, pl, Ho, Wo = (p[k] for k in ("n", "C", "kh", "kw", "sh", "sw", "pt", "pl", "Ho", "Wo"))
            w = C_[p["w"]].reshape(kh * kw, C)
            acc = np.zeros((n, Ho, Wo, C), np.float32)
            base = p["in_off"] - pt * p["in_row"] - pl * p["in_cs"]
            assert base >= 0
            for ky in range(kh):
                for kx in range(kw):
                    tap = _view(bufs[p["inp"]], base + ky * p["in_row"] + kx * p["in_cs"], n, Ho, Wo, C, p["in_img"], p["in_row"] * sh, p["in_cs"] * sw)
                    assert np.isfinite(tap).all(), "depthwise conv reads memory nothing wrote"
                    acc += tap * w[ky * kw + kx][None, None, None, :]
            if p["scale"] is not None:
                acc = acc * C_[p["scale"]] + C_[p["shift"]]
            _view(bufs[p["out"]], p["out_off"], n, Ho, Wo, C, p["out_img"], p["out_row"], p["out_cs"])[...] = _act(acc, p["act"])
        elif kind == "nearest_view":
            n, C, Ho, Wo, s = p["n"], p["C"], p["Ho"], p["Wo"], p["s"]
            src = _view(bufs[p["inp"]], p["in_off"], n, Ho // s, Wo // s, C, p["in_img"], p["in_row"], p["in_cs"])
            _view(bufs[p["out"]], p["out_off"], n, Ho, Wo, C, p["out_img"], p["out_row"], p["out_cs"])[...] = src.repeat(s, axis=1).repeat(s, axis=2)
        elif kind == "im2col_view":
            n, C, H, W, kh, kw, pt, pl = (p[k] for k in ("n", "C", "H", "W", "kh", "kw", "pt", "pl"))
            src = np.pad(bufs[p["x"]][:n * C * H * W].reshape(n, C, H, W), ((0, 0), (0, 0), (pt, kh - 1 - pt), (pl, kw - 1 - pl)))
            dst = _view(bufs[p["out"]], p["out_off"], n, H, W, 32, p["out_img"], p["out_row"], p["out_cs"])
            dst[...] = 0
            for c in range(C):
                for ky in range(kh):
                    for kx in range(kw):
                        dst[..., (c * kh + ky) * kw + kx] = src[:, c, ky:ky + H, kx:kx + W]
        elif kind == "dots_view":
            n, C, H, W, no = p["n"], p["C"], p["H"], p["W"], p["n_out"]
            src = _view(bufs[p["inp"]], p["in_off"], n, H, W, C, p["in_img"], p["in_row"], p["in_cs"])
            assert np.isfinite(src).all()
            w = C_[p["w"]].reshape(no, C)
            r = src.reshape(-1, C) @ w.T                                           # [pixels][no]
            if p["bias"] is not None:
                r = r + C_[p["bias"]][0]
            r = _act(r, p["act"]).astype(np.float32)
            if no == 1:
                bufs[p["out"]][:n * H * W] = r[:, 0]
            else:
                bufs[p["out"]][:n * 4 * H * W] = r.reshape(n, H, W, 2, 2).transpose(0, 1, 3, 2, 4).reshape(-1)
        elif kind == "conv_nchw":
            x_ = torch.from_numpy(bufs[p["x"]][:p["n"] * p["cin"] * p["h"] * p["wd"]].reshape(p["n"], p["cin"], p["h"], p["wd"]))
            w = torch.from_numpy(C_[p["w"]].reshape(p["cout"], 1 if p["dw"] else p["cin"], p["kh"], p["kw"]))
            pb = max(0, (p["ho"] - 1) * p["sh"] + p["kh"] - p["pt"] - p["h"])
            pr = max(0, (p["wo"] - 1) * p["sw"] + p["kw"] - p["pl"] - p["wd"])
            y = F.conv2d(F.pad(x_, (p["pl"], pr, p["pt"], pb)), w, stride=(p["sh"], p["sw"]), groups=p["cin"] if p["dw"] else 1)
            y = y[:, :, :p["ho"], :p["wo"]]
            bufs[p["out"]][:y.numel()] = y.numpy().reshape(-1)
        elif kind == "deconv_nchw":
            x_ = torch.from_numpy(bufs[p["x"]][:p["n"] * p["cin"] * p["h"] * p["wd"]].reshape(p["n"], p["cin"], p["h"], p["wd"]))
            w = torch.from_numpy(C_[p["w"]].reshape(p["cin"], 1 if p["dw"] else p["cout"], 2, 2))
            y = F.conv_transpose2d(x_, w, stride=2, groups=p["cin"] if p["dw"] else 1)
            bufs[p["out"]][:y.numel()] = y.numpy().reshape(-1)
        elif kind == "affine":
            tot, C, HW = p["total"], p["C"], p["HW"]
            v = bufs[p["x"]][:tot].reshape(-1, C, HW)
            bufs[p["out"]][:tot] = (v * C_[p["scale"]][None, :, None] + C_[p["shift"]][None, :, None]).reshape(-1)
        elif kind == "binary":
            tot, C, HW, mode = p["total"], p["C"], p["HW"], p["mode"]
            a = bufs[p["a"]][:tot].reshape(-1, C, HW)
            src = C_[p["b"][1]] if p["b"][0] == "const" else bufs[p["b"][1]]
            if mode == 0:
                b = src[:tot].reshape(-1, C, HW)
            elif mode == 1:
                b = src[:C].reshape(1, C, 1)
            elif mode == 2:
                b = src[:a.shape[0] * C].reshape(-1, C, 1)
            else:
                b = src[0]
            bufs[p["out"]][:tot] = (a + b if p["op"] == 0 else a * b).reshape(-1)
        elif kind == "unary":
            v = bufs[p["x"]][:p["total"]]
            k = p["kind"]
            r = (np.maximum(v, 0) if k == 0 else _act(v, 2) if k == 1 else np.clip(v * np.float32(p["p0"]) + np.float32(p["p1"]), 0, 1) if k == 2
                 else _act(v, 3) if k == 3 else v * np.float32(p["p0"]) + np.float32(p["p1"]))
            bufs[p["out"]][:p["total"]] = r
        elif kind == "gap":
            v = bufs[p["x"]][:p["planes"] * p["HW"]].reshape(p["planes"], p["HW"])
            bufs[p["out"]][:p["planes"]] = v.mean(axis=1)
        elif kind == "maxpool":
            v = torch.from_numpy(bufs[p["x"]][:p["planes"] * p["H"] * p["W"]].reshape(1, p["planes"], p["H"], p["W"]))
            pb = max(0, (p["Ho"] - 1) * p["sh"] + p["kh"] - p["pt"] - p["H"])
            pr = max(0, (p["Wo"] - 1) * p["sw"] + p["kw"] - p["pl"] - p["W"])
            y = F.max_pool2d(F.pad(v, (p["pl"], pr, p["pt"], pb), value=float("-inf")), (p["kh"], p["kw"]), stride=(p["sh"], p["sw"]))
            bufs[p["out"]][:y.numel()] = y.numpy().reshape(-1)
        elif kind == "nearest_nchw":
            v = bufs[p["x"]][:p["planes"] * p["H"] * p["W"]].reshape(p["planes"], p["H"], p["W"])
            r = v.repeat(p["s"], axis=1).repeat(p["s"], axis=2)
            bufs[p["out"]][:r.size] = r.reshape(-1)
        elif kind == "copy":
            for r in range(p["rows"]):
                bufs[p["dst"]][p["dst_off"] + r * p["dst_pitch"]: p["dst_off"] + r * p["dst_pitch"] + p["width"]] = \
                    bufs[p["src"]][r * p["src_pitch"]: r * p["src_pitch"] + p["width"]]
        else:
            raise NotImplementedError(kind)
        if after_step is not None:
            after_step(len_done, kind, p, bufs)
        len_done += 1
    name, shape = plan.output
    out = bufs[name][:int(np.prod(shape))].reshape(shape).copy()
    return out, bufs
