"""CPU: image loading and physical pixel sizes from TIFF metadata (tmat_amd/helper.py; reference helper.py:23-139 and
its use in compute_branches.py:184-212)."""
import sys
from pathlib import Path

import numpy as np
import pytest
from PIL import Image, TiffImagePlugin

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO / "tissue-model-analysis-tools_amd"))

from tmat_amd import helper  # noqa: E402


def _save(path, arr, description=None, xres=None):
    info = TiffImagePlugin.ImageFileDirectory_v2()
    if description is not None:
        info[270] = description
    if xres is not None:
        info[282] = TiffImagePlugin.IFDRational(*xres)
        info[283] = TiffImagePlugin.IFDRational(*xres)
    Image.fromarray(arr).save(path, tiffinfo=info)


def test_ome_and_imagej_pixel_sizes(tmp_path):
    a = np.arange(12, dtype=np.uint16).reshape(3, 4)
    ome = ('<?xml version="1.0"?><OME xmlns="http://www.openmicroscopy.org/Schemas/OME/2016-06"><Image ID="Image:0">'
           '<Pixels ID="Pixels:0" DimensionOrder="XYZCT" SizeX="4" SizeY="3" PhysicalSizeX="0.65" PhysicalSizeY="0.66" '
           'PhysicalSizeZ="2.0" Type="uint16"/></Image></OME>')
    _save(tmp_path / "a.ome.tif", a, ome)
    assert helper.physical_pixel_sizes(tmp_path / "a.ome.tif") == helper.PhysicalPixelSizes(2.0, 0.66, 0.65)
    _save(tmp_path / "nm.tif", a, ome.replace('PhysicalSizeX="0.65"', 'PhysicalSizeX="650" PhysicalSizeXUnit="nm"'))
    assert helper.physical_pixel_sizes(tmp_path / "nm.tif").X == pytest.approx(0.65)
    _save(tmp_path / "ij.tif", a, "ImageJ=1.53t\nunit=micron\nspacing=3.5\n", xres=(2000000, 1300000))
    got = helper.physical_pixel_sizes(tmp_path / "ij.tif")
    assert got.X == pytest.approx(0.65) and got.Y == pytest.approx(0.65) and got.Z == pytest.approx(3.5)
    _save(tmp_path / "inch.tif", a, "ImageJ=1.53t\nunit=inch\n", xres=(300, 1))
    assert helper.physical_pixel_sizes(tmp_path / "inch.tif").X == pytest.approx(25400.0 / 300)
    _save(tmp_path / "plain.tif", a)
    assert helper.physical_pixel_sizes(tmp_path / "plain.tif") == helper.PhysicalPixelSizes(None, None, None)
    Image.fromarray(a.astype(np.uint8)).save(tmp_path / "p.png")
    assert helper.physical_pixel_sizes(tmp_path / "p.png").X is None


def test_load_image_contract(tmp_path):
    a = np.arange(12, dtype=np.uint16).reshape(3, 4)
    _save(tmp_path / "s0.tif", a)
    _save(tmp_path / "s1.tif", a + 1)
    img, sizes = helper.load_image(str(tmp_path / "s0.tif"))
    assert np.array_equal(img, a) and sizes.X is None
    st, _ = helper.load_image([str(tmp_path / "s0.tif"), str(tmp_path / "s1.tif")])
    assert st.shape == (2, 3, 4) and np.array_equal(st[1], a + 1)
    Image.fromarray(a).save(tmp_path / "multi.tif", save_all=True, append_images=[Image.fromarray(a + 2)])
    st, _ = helper.load_image(str(tmp_path / "multi.tif"))
    assert st.shape == (2, 3, 4) and np.array_equal(st[1], a + 2)
    rgb = np.zeros((3, 4, 3), np.uint8)
    rgb[..., 1] = 7
    Image.fromarray(rgb).save(tmp_path / "rgb.png")
    with pytest.raises(ValueError, match="multi channel"):
        helper.load_image(str(tmp_path / "rgb.png"))
    g, _ = helper.load_image(str(tmp_path / "rgb.png"), C=1)
    assert g.shape == (3, 4) and np.all(g == 7)
    with pytest.raises(ValueError, match="time-series"):
        helper.load_image(str(tmp_path / "s0.tif"), T=1)
