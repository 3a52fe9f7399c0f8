"""Checkpoint assembly (reference backend/tools/common_tools.py:40-45, called from tools/model_config.py:24-25).

The reference ships its two largest checkpoints cut into 50 MB parts -- models/big-lama/big-lama_{1..5}.pt and
models/propainter/ProPainter_{1..4}.pth -- next to a `fs_manifest.csv` (filename, filesize, encoding, header) written by the
`filesplit` package, and re-assembles them on first use with `Filesplit().merge(input_dir=dir)`: the parts are concatenated in
manifest order into `<stem>.<ext>` (the part name without its `_<n>` suffix).  The package is not in the image; this is that merge,
with the checks a silent concatenation lacks: every part must exist with the size the manifest states, the result must have the
summed size, and a failure leaves no half-written checkpoint behind.
"""
import csv
import os
import re

MANIFEST = "fs_manifest.csv"


def merged_name(part_name):
    """'big-lama_3.pt' -> 'big-lama.pt' (filesplit's naming: stem + '_' + running number + extension)"""
    stem, ext = os.path.splitext(part_name)
    m = re.match(r"^(.*)_(\d+)$", stem)
    if m is None:
        raise ValueError(f"{part_name}: not a split part name (<stem>_<n><ext>)")
    return m.group(1) + ext


def merge_big_file_if_not_exists(dir, file, man_filename=None):
    """If `file` is not in `dir`, merge the parts listed in the directory's manifest (common_tools.py:40-45).  Returns the path of the
    merged file, or None when there was nothing to do (the file exists) -- and raises when the parts cannot give it."""
    if file in os.listdir(dir):
        return None
    man = os.path.join(dir, man_filename or MANIFEST)
    if not os.path.isfile(man):
        raise FileNotFoundError(f"{os.path.join(dir, file)} is missing and there is no {man_filename or MANIFEST} to assemble it from")
    with open(man, newline="") as f:
        rows = [r for r in csv.DictReader(f) if r.get("filename")]
    if not rows:
        raise ValueError(f"{man}: no parts listed")
    targets = {merged_name(r["filename"]) for r in rows}
    if len(targets) != 1:
        raise ValueError(f"{man}: parts of more than one file: {sorted(targets)}")
    target = os.path.join(dir, targets.pop())
    total = 0
    for r in rows:
        part = os.path.join(dir, r["filename"])
        if not os.path.isfile(part):
            raise FileNotFoundError(f"{part}: part listed in {man} is missing")
        size = os.path.getsize(part)
        if r.get("filesize") and size != int(r["filesize"]):
            raise ValueError(f"{part}: {size} bytes, the manifest says {r['filesize']}")
        total += size
    tmp = target + ".partial"
    try:
        with open(tmp, "wb") as out:
            for r in rows:
                with open(os.path.join(dir, r["filename"]), "rb") as src:
                    while True:
                        buf = src.read(1 << 24)
                        if not buf:
                            break
                        out.write(buf)
        if os.path.getsize(tmp) != total:
            raise IOError(f"{tmp}: {os.path.getsize(tmp)} bytes written, {total} expected")
        os.replace(tmp, target)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return target


def checkpoint_path(path):
    """`path` as given when it exists; otherwise assembled from the split parts next to it (model_config.py:24-25 does this for
    big-lama and ProPainter when the process starts)"""
    if os.path.exists(path):
        return path
    d, f = os.path.split(os.path.abspath(path))
    if os.path.isdir(d) and os.path.isfile(os.path.join(d, MANIFEST)):
        merged = merge_big_file_if_not_exists(d, f)
        if merged is not None and os.path.basename(merged) != f:
            raise FileNotFoundError(f"{path}: the parts in {d} assemble {os.path.basename(merged)}, not {f}")
    return path
