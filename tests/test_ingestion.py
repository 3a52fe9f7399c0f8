"""Real artefacts meeting the loaders: the reference's split checkpoints, DataParallel / TorchScript key layouts.
Every case must either load exactly or fail loudly -- never load something else."""
import os

import numpy as np
import pytest
import torch

import vsr_amd  # noqa: F401
from vsr_amd import synth
from vsr_amd.backend.tools import common_tools as ct


def _split(tmp_path, name, blob, part):
    """what `filesplit` leaves behind: <stem>_<n><ext> parts of `part` bytes and fs_manifest.csv"""
    stem, ext = os.path.splitext(name)
    rows = ["filename,filesize,encoding,header"]
    for i in range(0, len(blob), part):
        pn = f"{stem}_{i // part + 1}{ext}"
        (tmp_path / pn).write_bytes(blob[i:i + part])
        rows.append(f"{pn},{len(blob[i:i + part])},,")
    (tmp_path / "fs_manifest.csv").write_text("\n".join(rows) + "\n")


def test_split_checkpoint_is_reassembled(tmp_path):
    """reference common_tools.py:40-45 + models/*/fs_manifest.csv: parts concatenated in manifest order into <stem><ext>"""
    blob = np.random.default_rng(0).integers(0, 256, 230_017, dtype=np.uint8).tobytes()
    _split(tmp_path, "ProPainter.pth", blob, 50_000)
    assert sorted(os.listdir(tmp_path))[:2] == ["ProPainter_1.pth", "ProPainter_2.pth"]
    out = ct.merge_big_file_if_not_exists(str(tmp_path), "ProPainter.pth")
    assert out == str(tmp_path / "ProPainter.pth") and (tmp_path / "ProPainter.pth").read_bytes() == blob
    assert ct.merge_big_file_if_not_exists(str(tmp_path), "ProPainter.pth") is None          # nothing to do the second time
    # the reference asks for 'bit-lama.pt' (a typo, model_config.py:24): the parts still assemble what their names say
    assert ct.merged_name("big-lama_5.pt") == "big-lama.pt"


def test_split_checkpoint_failures_are_loud(tmp_path):
    blob = bytes(range(256)) * 400
    _split(tmp_path, "big-lama.pt", blob, 30_000)
    os.remove(tmp_path / "big-lama_2.pt")
    with pytest.raises(FileNotFoundError, match="big-lama_2.pt"):
        ct.merge_big_file_if_not_exists(str(tmp_path), "big-lama.pt")
    assert not (tmp_path / "big-lama.pt").exists() and not (tmp_path / "big-lama.pt.partial").exists()
    (tmp_path / "big-lama_2.pt").write_bytes(b"short")
    with pytest.raises(ValueError, match="the manifest says 30000"):
        ct.merge_big_file_if_not_exists(str(tmp_path), "big-lama.pt")
    empty = tmp_path / "none"
    empty.mkdir()
    with pytest.raises(FileNotFoundError, match="no fs_manifest.csv"):
        ct.merge_big_file_if_not_exists(str(empty), "big-lama.pt")


def test_propainter_checkpoint_loads_from_parts(tmp_path):
    """_load() of the propainter plugin: ProPainter.pth only exists as parts, as in the reference tree"""
    from vsr_amd.backend.inpaint.propainter_inpaint import _load

    sd = {k: torch.from_numpy(v) for k, v in list(synth.make_propainter_state_dict(0).items())[:12]}
    whole = tmp_path / "whole.pth"
    torch.save(sd, whole)
    _split(tmp_path, "ProPainter.pth", whole.read_bytes(), 40_000)
    os.remove(whole)
    got = _load(str(tmp_path), "propainter", "ProPainter.pth")
    assert list(got) == list(sd) and all(torch.equal(got[k], sd[k]) for k in sd)


def test_sttn_checkpoint_layouts(tmp_path, built_lib):
    """torch.load(path)['netG'] as the reference reads it (sttn_auto_inpaint.py:34); DataParallel's 'module.' prefix is unwrapped;
    a file without 'netG', a missing key or a wrong shape is refused"""
    from vsr_amd.backend.inpaint.sttn_auto_inpaint import _load_state_dict
    from vsr_amd.engine import SttnEngine

    sd = synth.make_state_dict(0, "auto")
    plain, wrapped, alien = tmp_path / "a.pth", tmp_path / "b.pth", tmp_path / "c.pth"
    torch.save({"netG": {k: torch.from_numpy(v) for k, v in sd.items()}}, plain)
    torch.save({"netG": {"module." + k: torch.from_numpy(v) for k, v in sd.items()}, "netD": {}}, wrapped)
    torch.save({"state_dict": {}}, alien)
    for p in (plain, wrapped):
        got = _load_state_dict(str(p))
        assert list(got) == list(sd)
        SttnEngine({k: np.asarray(v) for k, v in got.items()}, "auto", device=None).close()          # strict host-side load
    with pytest.raises(KeyError, match="no 'netG' entry"):
        _load_state_dict(str(alien))
    bad = {k: np.asarray(v) for k, v in _load_state_dict(str(plain)).items()}
    bad["encoder.0.weight"] = bad["encoder.0.weight"][:, :2]
    with pytest.raises(built_lib.VsrError, match="shape mismatch"):
        SttnEngine(bad, "auto", device=None)


def test_lama_torchscript_blob_key_mapping(tmp_path):
    """big-lama.pt is a TorchScript export whose parameters sit under `generator.model.N...` (lama_inpaint.py:13 torch.jit.load); the
    loader must hand the engine `model.N...` -- checked against a REAL scripted module with that attribute tree (round 2 only met
    dicts), and an export without a generator must be refused"""
    from vsr_amd.backend.inpaint.lama_inpaint import _load_lama_state_dict

    sd = synth.make_lama_state_dict(1, 1)
    keep = {k: v for k, v in sd.items() if k.split(".")[1] in ("1", "2")}           # a few modules are enough for the mapping

    def build(prefix_root):
        root = torch.nn.Module()
        for k, v in keep.items():
            node = root
            parts = k.split(".")
            for a in parts[:-1]:
                if not hasattr(node, a):
                    node.add_module(a, torch.nn.Module())
                node = getattr(node, a)
            if parts[-1] in ("running_mean", "running_var", "num_batches_tracked"):
                node.register_buffer(parts[-1], torch.from_numpy(np.asarray(v)).clone())
            else:
                node.register_parameter(parts[-1], torch.nn.Parameter(torch.from_numpy(np.asarray(v)).clone(), requires_grad=False))

        class Export(torch.nn.Module):
            def __init__(self):
                super().__init__()
                if prefix_root:
                    self.generator = root
                else:
                    self.other = torch.nn.Linear(2, 2)

            def forward(self, image: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
                return image * (1 - mask)

        return Export()

    path = str(tmp_path / "big-lama.pt")
    torch.jit.script(build(True)).save(path)
    got = _load_lama_state_dict(path)
    assert sorted(got) == sorted(keep)
    assert all(np.array_equal(np.asarray(got[k]), keep[k]) for k in keep)
    bad = str(tmp_path / "other.pt")
    torch.jit.script(build(False)).save(bad)
    with pytest.raises(KeyError, match="no generator entries"):
        _load_lama_state_dict(bad)
