"""Mirror of fl_tissue_model_tools.zstacks (reference zstacks.py): Z-stack discovery and the projection methods
`compute_zproj.py` offers.  The projections run in HIP kernels behind the C-ABI (tmat_zproj_batch,
csrc/zproj_kernels.hip); there is no CPU fallback.

proj_* keep the reference signatures (`stack` is a (Z, H, W) array, `axis` the axis to project along) and add a batch
form: a 4-D (n, Z, H, W) array projects every stack in one launch.
"""
from __future__ import annotations

import os.path as osp
import re
from difflib import SequenceMatcher
from glob import glob

import numpy as np

from . import _lib


class ZStackInputException(Exception):
    """reference exceptions.py: raised for unrecognised Z-slice naming"""


_handle = None


def default_handle() -> "_lib.Handle":
    """a model-less handle on device 0 (tmat_create_plain), created on first use"""
    global _handle
    if _handle is None:
        _handle = _lib.Handle(None, 0)
    return _handle


def _project(stack, axis, method, handle):
    a = np.asarray(stack)
    if a.ndim == 4:                       # batch form (n, Z, H, W)
        if axis not in (0, 1):
            raise ValueError("batch form projects along axis 1 (Z)")
        return (handle or default_handle()).zproj(a, method)
    if a.ndim != 3:
        raise ValueError(f"expected a (Z, H, W) stack, got shape {a.shape}")
    if axis != 0:
        a = np.moveaxis(a, axis, 0)
    return (handle or default_handle()).zproj(a[None], method)[0]


def proj_focus_stacking(stack, axis: int = 0, kernel_size: int = 5, handle=None):
    """Focus stacking: per pixel the value of the slice with the largest |Laplacian of the blurred slice|
    (reference zstacks.py:153-189; kernel_size 5 is the only size the reference ever passes)."""
    if kernel_size != 5:
        raise NotImplementedError("only kernel_size=5 (the reference default) is implemented")
    return _project(stack, axis, "fs", handle)


def proj_avg(stack, axis: int = 0, handle=None):
    return _project(stack, axis, "avg", handle)      # zstacks.py:192-204


def proj_med(stack, axis: int = 0, handle=None):
    return _project(stack, axis, "med", handle)      # zstacks.py:207-219


def proj_max(stack, axis: int = 0, handle=None):
    return _project(stack, axis, "max", handle)      # zstacks.py:222-235


def proj_min(stack, axis: int = 0, handle=None):
    return _project(stack, axis, "min", handle)      # zstacks.py:238-249


# ---------------------------------------------------------------------------------------------
# Z-stack discovery (host logic; reference zstacks.py:17-131)
# ---------------------------------------------------------------------------------------------
def _unique_or_keep(candidate, current):
    return candidate if len(set(candidate)) == len(candidate) else current


def clean_zstack_ids(zstack_ids):
    """Shorten stack identifiers (zstacks.py:17-61): drop a directory part that only repeats the file name, then strip
    leading / trailing underscores and collapse double underscores, each step only if the identifiers stay distinct.
    As in the reference, path separators are turned into underscores only when doing so makes two identifiers collide
    (then it is applied to the original identifiers); otherwise they are kept."""
    original = list(zstack_ids)
    ids = []
    for zid in original:
        name, dir_name = osp.basename(zid), osp.dirname(zid)
        if len(dir_name) > len(name) / 2:
            sm = SequenceMatcher(a=dir_name.lower(), b=name.lower())
            if sum(m.size for m in sm.get_matching_blocks()) == len(dir_name):
                zid = name
        ids.append(zid)
    cur = _unique_or_keep(ids, original)
    flat = [z.replace("/", "_").replace("\\", "_") for z in cur]
    if len(set(flat)) != len(flat):
        cur = [z.replace("/", "_").replace("\\", "_") for z in original]
    cur = _unique_or_keep([z.lstrip("_") for z in cur], cur)
    cur = _unique_or_keep([z.rstrip("_") for z in cur], cur)
    cur = _unique_or_keep([z.replace("__", "_") for z in cur], cur)
    return cur


def find_zstack_image_sequences(input_dir: str):
    """{stack id: [slice paths in Z order]} for slices stored one per file with a `z<number>` token in the name,
    either directly in `input_dir` or one directory level below (zstacks.py:64-117)."""
    paths = [p for p in glob(osp.join(input_dir, "*")) if osp.isfile(p)]
    if not paths:
        paths = [p for p in glob(osp.join(input_dir, "*", "*")) if osp.isfile(p)]
    stack_of, numbers = [], []
    for rel in (osp.relpath(p, input_dir) for p in paths):
        name, dir_name = osp.basename(rel), osp.dirname(rel)
        sid = osp.splitext(osp.join(dir_name, re.sub(r"z\d+", "", name, flags=re.IGNORECASE)))[0]
        stack_of.append(sid)
        numbers.append([int(v) for v in re.findall(r"(?<=z)\d+", name, re.IGNORECASE)][::-1])
    originals = list(set(stack_of))
    renamed = dict(zip(originals, clean_zstack_ids(originals)))
    stack_of = [renamed[s] for s in stack_of]
    out = {}
    for sid in set(stack_of):
        members = [i for i, s in enumerate(stack_of) if s == sid]
        nums = [numbers[i] for i in members]
        if any(len(v) != len(nums[0]) for v in nums):
            raise ZStackInputException("Unrecognized Z slice naming convention")
        if len({tuple(v) for v in nums}) != len(members):
            raise ZStackInputException("Unrecognized Z slice numbering convention in image names")
        out[sid] = [paths[k[-1]] for k in sorted(v + [i] for i, v in zip(members, nums))]
    return out


def find_zstack_files(input_dir: str):
    """{file stem: path} for stacks stored one multi-page file each (zstacks.py:120-131)."""
    paths = [p for p in glob(osp.join(input_dir, "*")) if osp.isfile(p)]
    return {osp.splitext(osp.basename(p))[0]: p for p in paths}
