"""Mirror of the image-loading part of fl_tissue_model_tools.helper (reference helper.py:23-139): `load_image` returns the
pixel array and the physical pixel sizes found in the file's metadata.  The reference reads files through aicsimageio;
this build reads TIFF / PNG with Pillow and parses the two metadata conventions aicsimageio's TIFF readers honour:

* OME-TIFF: `PhysicalSizeX/Y/Z` (+ `PhysicalSize?Unit`, default µm) of the first `Pixels` element of the OME-XML held in
  the ImageDescription tag;
* ImageJ TIFF: `unit=micron|um|µm` in the ImageDescription together with the X/YResolution tags (pixels per unit).

Anything else yields PhysicalPixelSizes(None, None, None), for which compute_branches.py asks for --image-width-microns
exactly as the reference does (compute_branches.py:184-212).
"""
from __future__ import annotations

import re
from collections import namedtuple
from typing import Optional

import numpy as np

PhysicalPixelSizes = namedtuple("PhysicalPixelSizes", ["Z", "Y", "X"])

_UNIT_TO_MICRON = {"µm": 1.0, "um": 1.0, "micron": 1.0, "microns": 1.0, "micrometer": 1.0, "nm": 1e-3, "mm": 1e3, "cm": 1e4,
                   "m": 1e6, "pm": 1e-6, "inch": 25400.0, "in": 25400.0}


def _ome_sizes(desc: str):
    m = re.search(r"<Pixels\b[^>]*>", desc)
    if not m:
        return None
    attrs = dict(re.findall(r'(\w+)="([^"]*)"', m.group(0)))
    out = []
    for ax in "ZYX":
        v = attrs.get("PhysicalSize" + ax)
        if v is None:
            out.append(None)
            continue
        unit = attrs.get(f"PhysicalSize{ax}Unit", "µm").replace("μ", "µ")
        scale = _UNIT_TO_MICRON.get(unit)
        out.append(float(v) * scale if scale is not None else None)
    return PhysicalPixelSizes(*out)


def _imagej_sizes(desc: str, tags):
    if "ImageJ=" not in desc:
        return None
    m = re.search(r"unit=(\S+)", desc)
    if not m:
        return None
    unit = m.group(1).strip().replace("\\u00B5", "µ").replace("μ", "µ")
    scale = _UNIT_TO_MICRON.get(unit)
    if scale is None:
        return None

    def res(tag):
        v = tags.get(tag)
        if v is None:
            return None
        v = v[0] if isinstance(v, tuple) and len(v) == 1 else v
        try:
            v = float(v[0]) / float(v[1]) if isinstance(v, tuple) else float(v)
        except (TypeError, ZeroDivisionError, ValueError):
            return None
        return scale / v if v > 0 else None
    sp = re.search(r"spacing=([0-9.eE+-]+)", desc)
    return PhysicalPixelSizes(float(sp.group(1)) * scale if sp else None, res(283), res(282))


def physical_pixel_sizes(path) -> PhysicalPixelSizes:
    """PhysicalPixelSizes(Z, Y, X) in microns per pixel, None where the file does not say"""
    from PIL import Image
    try:
        with Image.open(path) as im:
            tags = dict(getattr(im, "tag_v2", {}) or {})
    except OSError:
        return PhysicalPixelSizes(None, None, None)
    desc = tags.get(270, "")
    if isinstance(desc, (tuple, list)):
        desc = desc[0] if desc else ""
    if isinstance(desc, bytes):
        desc = desc.decode("utf8", "replace")
    for parse in (lambda: _ome_sizes(desc), lambda: _imagej_sizes(desc, tags)):
        got = parse()
        if got is not None:
            return got
    return PhysicalPixelSizes(None, None, None)


def load_image(file_path, T: Optional[int] = None, C: Optional[int] = None):
    """(ZYX or YX array, PhysicalPixelSizes): the reference's contract (helper.py:23-120) for single files and for
    lists of slice files.  Time series are not supported (T must be None or 0)."""
    if isinstance(file_path, (list, tuple)):
        imgs, sizes = zip(*[load_image(fp, T, C) for fp in file_path])
        return np.array(imgs), sizes[0]
    if T not in (None, 0):
        raise ValueError(f"{file_path}: time-series files are not part of the accelerated path")
    from PIL import Image
    with Image.open(file_path) as im:
        pages = []
        for i in range(getattr(im, "n_frames", 1)):
            im.seek(i)
            pages.append(np.array(im))
    out = []
    for a in pages:
        if a.ndim == 3:
            if C is None:
                raise ValueError(f"{file_path} is a multi channel image but no channel index was specified.")
            a = a[..., C] if a.shape[-1] <= 4 else a[C]
        out.append(a)
    arr = out[0] if len(out) == 1 else np.stack(out)
    return arr, physical_pixel_sizes(file_path)
