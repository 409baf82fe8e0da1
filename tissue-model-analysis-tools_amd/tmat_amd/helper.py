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


def page_layout(desc: str, n_pages: int):
    """(size_t, size_z, size_c, order) of a multi-page TIFF from its ImageDescription, order = the page axes from the fastest
    to the slowest varying one (e.g. "ZCT").  OME-XML: SizeT / SizeZ / SizeC / DimensionOrder of the first Pixels element;
    ImageJ hyperstacks: frames= / slices= / channels=, always channel-fastest ("CZT").  Files without either convention are
    one Z stack, as aicsimageio's plain TIFF reader presents them."""
    m = re.search(r"<Pixels\b[^>]*>", desc or "")
    if m:
        attrs = dict(re.findall(r'(\w+)="([^"]*)"', m.group(0)))
        try:
            st, sz, sc = int(attrs.get("SizeT", 1)), int(attrs.get("SizeZ", 1)), int(attrs.get("SizeC", 1))
        except ValueError:
            st = sz = sc = 0
        order = attrs.get("DimensionOrder", "XYZCT")[2:]
        if st * sz * sc == n_pages and sorted(order) == ["C", "T", "Z"]:
            return st, sz, sc, order
    if "ImageJ=" in (desc or ""):
        def num(key):
            mm = re.search(key + r"=(\d+)", desc)
            return int(mm.group(1)) if mm else 1
        st, sz, sc = num("frames"), num("slices"), num("channels")
        if st * sz * sc == n_pages:
            return st, sz, sc, "CZT"
    return 1, n_pages, 1, "ZCT"


def load_image(file_path, T: Optional[int] = None, C: Optional[int] = None):
    """(ZYX or YX array, PhysicalPixelSizes): the reference's contract (helper.py:23-95) for single files and for lists of
    slice files, with its error messages: T / C must be given for time series / multi-channel files and lie in range.  The
    page layout of multi-page files (which page is which T, Z, C) comes from the OME-XML or ImageJ description."""
    if isinstance(file_path, (list, tuple)):
        imgs, sizes = zip(*[load_image(fp, T, C) for fp in file_path])
        return np.array(imgs), sizes[0]
    from PIL import Image
    with Image.open(file_path) as im:
        n_pages = getattr(im, "n_frames", 1)
        tags = dict(getattr(im, "tag_v2", {}) or {})
        desc = tags.get(270, "")
        if isinstance(desc, (tuple, list)):
            desc = desc[0] if desc else ""
        if isinstance(desc, bytes):
            desc = desc.decode("utf8", "replace")
        st, sz, sc, order = page_layout(desc, n_pages)
        im.seek(0)
        first = np.array(im)
        interleaved = first.ndim == 3              # RGB(A) pages: the channel axis is inside the page
        n_c = first.shape[-1] if interleaved else sc
        if T is None:
            if st > 1:
                raise ValueError(f"{file_path} is a time series image but no time index was specified.")
            T = 0
        elif T >= st or T < 0:
            raise ValueError(f"Time {T} is out of range for {file_path} with times: 0 - {st - 1}")
        if C is None:
            if n_c > 1:
                raise ValueError(f"{file_path} is a multi channel image but no color channel index was specified.")
            C = 0
        elif C >= n_c or C < 0:
            raise ValueError(f"Color channel {C} is out of range for {file_path} with color channels: 0 - {n_c - 1}")
        sizes = {"T": st, "Z": sz, "C": sc}
        stride, strides = 1, {}
        for ax in order:
            strides[ax] = stride
            stride *= sizes[ax]
        planes = []
        for z in range(sz):
            page = T * strides["T"] + z * strides["Z"] + (0 if interleaved else C) * strides["C"]
            im.seek(page)
            a = np.array(im)
            planes.append(a[..., C] if interleaved else a)
    arr = planes[0] if len(planes) == 1 else np.stack(planes)
    return arr, physical_pixel_sizes(file_path)
