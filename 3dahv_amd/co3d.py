"""CO3D-v2 annotation reader for the evaluation harness.

Counterpart of what ``test_co3d.py`` uses from ``Co3dDataset`` in evaluation mode
(``data_loader_co3d.py:74-372`` with ``random_aug=False, eval_time=True, normalize_cameras=False``):

* ``<CO3D_ANNOTATION_DIR>/<category>_<split>.jgz`` = gzip'd JSON ``{sequence: [frame, ...]}``, frame keys
  ``filepath, bbox (xyxy), R (3x3), T (3), focal_length, principal_point``                    (:123-152)
* sequences with fewer frames than ``num_images`` are skipped; a sequence with any frame whose
  ``T[0] + T[1] + T[2] > 1e5`` is dropped as a whole                                          (:128-139,155-156)
* per requested frame: RGB image from ``<CO3D_DIR>/<filepath>``; square box around ``bbox``, scaled by 1.15
  about its centre, corners rounded to integers (:200-214 with the eval-time jitter of :181-183); crop with
  zero fill outside the image (:226-233); ToTensor -> Resize(OBJ_SIZE) -> Normalize(PIXEL_MEAN, PIXEL_STD)
  (test_co3d.py:57-66)
* ``R`` is the raw annotation rotation (pytorch3d row-vector convention); the harness forms
  ``R_gt = R1^T R2`` from it (test_co3d.py:121-124).

Images are decoded with PIL; nothing here needs torchvision or pytorch3d.  The resize is bilinear without
anti-aliasing (what ``transforms.Resize`` does to a tensor in the torchvision 0.14 the reference pins).
This module is host-side glue: it feeds ``harness.evaluate_category`` lazily (two frames per sequence are
decoded, as in the reference) and never touches the GPU.
"""
from __future__ import annotations

import gzip
import json
import os.path as osp
from typing import Dict, Iterator, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

TEST_CATEGORIES = ["ball", "book", "couch", "frisbee", "hotdog", "kite", "remote", "sandwich", "skateboard", "suitcase"]

_FRAME_KEYS = ("filepath", "bbox", "R", "T", "focal_length", "principal_point")


def square_bbox(bbox: np.ndarray) -> np.ndarray:
    """Smallest centred square containing an xyxy box (utils.py:194-214, padding 0), float32."""
    bbox = np.asarray(bbox, dtype=np.float32)
    center = (bbox[:2] + bbox[2:]) / 2
    s = np.max((bbox[2:] - bbox[:2]) / 2)
    return np.array([center[0] - s, center[1] - s, center[0] + s, center[1] + s], dtype=np.float32)


def eval_crop_box(bbox, scale: float = 1.15) -> np.ndarray:
    """Integer xyxy crop used at evaluation time (data_loader_co3d.py:200-214 with s = 1.15, t = 0)."""
    sq = square_bbox(bbox)
    side = sq[2] - sq[0]
    center = (sq[:2] + sq[2:]) / 2
    extent = side / 2 * scale
    ul = (center - extent).round().astype(int)
    lr = ul + np.round(2 * extent).astype(int)
    return np.concatenate((ul, lr))


def read_annotations(path: str, num_images: int = 2) -> Dict[str, List[dict]]:
    """``{sequence: [frame dicts]}`` after the reference's two filters."""
    with gzip.open(path, "r") as f:
        annotation = json.loads(f.read())
    kept = {}
    for seq_name, seq_data in annotation.items():
        if len(seq_data) < num_images:
            continue
        if any(d["T"][0] + d["T"][1] + d["T"][2] > 1e5 for d in seq_data):
            continue
        kept[seq_name] = [{k: d[k] for k in _FRAME_KEYS} for d in seq_data]
    return kept


class Co3dSequences:
    """Iterable of ``{"n", "model_id", "category", "get_data"}``; ``get_data(ids)`` decodes those frames and
    returns ``{"image": (k,3,S,S), "R": (k,3,3), "T": (k,3)}``."""

    def __init__(self, cfg: dict, category: str, split: str = "test", num_images: int = 2):
        self.cfg, self.category = cfg, category
        self.size = int(cfg["DATA"]["OBJ_SIZE"])
        self.mean = torch.tensor(cfg["DATA"]["PIXEL_MEAN"], dtype=torch.float32).view(3, 1, 1)
        self.std = torch.tensor(cfg["DATA"]["PIXEL_STD"], dtype=torch.float32).view(3, 1, 1)
        path = osp.join(cfg["CO3D"]["CO3D_ANNOTATION_DIR"], "%s_%s.jgz" % (category, split))
        self.frames = read_annotations(path, num_images)
        self.sequence_list = list(self.frames.keys())

    def __len__(self):
        return len(self.sequence_list)

    def load_image(self, anno: dict) -> torch.Tensor:
        from PIL import Image
        image = Image.open(osp.join(self.cfg["CO3D"]["CO3D_DIR"], anno["filepath"])).convert("RGB")
        x0, y0, x1, y1 = (int(v) for v in eval_crop_box(np.array(anno["bbox"])))
        image = image.crop((x0, y0, x1, y1))  # PIL fills what lies outside the picture with zeros
        t = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
        h, w = t.shape[1:]
        # transforms.Resize(int): the shorter side becomes `size`; crops are square up to rounding
        if h <= w:
            nh, nw = self.size, max(int(self.size * w / h), 1)
        else:
            nh, nw = max(int(self.size * h / w), 1), self.size
        t = F.interpolate(t[None], size=(nh, nw), mode="bilinear", align_corners=False, antialias=False)[0]
        return (t - self.mean) / self.std

    def get_data(self, sequence_name: str, ids: Sequence[int]) -> dict:
        annos = [self.frames[sequence_name][int(i)] for i in ids]
        return {"image": torch.stack([self.load_image(a) for a in annos]),
                "R": torch.stack([torch.tensor(a["R"], dtype=torch.float32) for a in annos]),
                "T": torch.stack([torch.tensor(a["T"], dtype=torch.float32) for a in annos])}

    def __iter__(self) -> Iterator[dict]:
        for name in self.sequence_list:
            yield {"n": len(self.frames[name]), "model_id": name, "category": self.category,
                   "get_data": (lambda ids, name=name: self.get_data(name, ids))}


def load_categories(cfg: dict, categories: Sequence[str] = TEST_CATEGORIES, split: str = "test") -> Dict[str, Co3dSequences]:
    """``{category: Co3dSequences}`` for ``harness.evaluate_pairwise`` / ``run_co3d`` (test_co3d.py:201-216)."""
    return {c: Co3dSequences(cfg, c, split) for c in categories}
