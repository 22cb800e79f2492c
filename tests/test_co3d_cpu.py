"""CO3D-v2 annotation reader (3dahv_amd/co3d.py) on a tiny synthetic dataset written by the test itself: the
`.jgz` schema, the two sequence filters, the evaluation-time crop and normalisation, and the lazy hook through
which the harness decodes only the key frames (data_loader_co3d.py:123-233, test_co3d.py:57-66,110-124)."""
import gzip
import json
import os

import numpy as np
import pytest
import torch


def write_dataset(root, n_frames=3):
    from PIL import Image
    rng = np.random.RandomState(0)
    ann_dir, img_dir = os.path.join(root, "ann"), os.path.join(root, "img")
    os.makedirs(ann_dir), os.makedirs(os.path.join(img_dir, "ball", "seq_a", "images"))
    pics = []

    def frame(k, T=(0.1, 0.2, 3.0), bbox=(30, 40, 130, 100)):
        rel = "ball/seq_a/images/frame%06d.png" % k
        if not os.path.exists(os.path.join(img_dir, rel)):
            pic = rng.randint(0, 256, size=(120, 160, 3)).astype(np.uint8)   # H=120, W=160
            Image.fromarray(pic).save(os.path.join(img_dir, rel))
            pics.append(pic)
        a = 0.3 * (k + 1)
        R = [[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]]
        return {"filepath": rel, "bbox": list(bbox), "R": R, "T": list(T), "focal_length": [2.0, 2.0],
                "principal_point": [0.0, 0.0], "extra_key_the_reader_drops": 1}
    ann = {"seq_a": [frame(k) for k in range(n_frames)],
           "seq_short": [frame(0)],                                            # fewer frames than num_images
           "seq_badT": [frame(0), frame(1, T=(1e5, 1.0, 1.0))]}                # translation sum > 1e5
    with gzip.open(os.path.join(ann_dir, "ball_test.jgz"), "w") as f:
        f.write(json.dumps(ann).encode())
    cfg = {"RUN_NAME": "t", "CO3D": {"CO3D_DIR": img_dir, "CO3D_ANNOTATION_DIR": ann_dir},
           "DATA": {"NUM_ROTA": 32, "OBJ_SIZE": 64, "PIXEL_MEAN": [0.5, 0.5, 0.5], "PIXEL_STD": [0.5, 0.5, 0.5],
                    "BG": True, "SIZE_THR": 25}}
    return cfg, pics


def test_crop_box_numbers(ahv):
    # bbox 100x60 centred (60,50) -> square [10,0,110,100], side 100, extent 57.5; corners round half to even
    box = ahv.co3d.eval_crop_box(np.array([10, 20, 110, 80]))
    assert box.tolist() == [2, -8, 117, 107]
    assert ahv.co3d.square_bbox(np.array([0, 0, 10, 30])).tolist() == [-10.0, 0.0, 20.0, 30.0]


def test_reader_filters_and_frames(ahv, tmp_path):
    cfg, pics = write_dataset(str(tmp_path))
    ds = ahv.co3d.Co3dSequences(cfg, "ball", "test")
    assert ds.sequence_list == ["seq_a"] and len(ds) == 1
    meta = next(iter(ds))
    assert meta["n"] == 3 and meta["model_id"] == "seq_a" and meta["category"] == "ball"
    assert set(ds.frames["seq_a"][0]) == {"filepath", "bbox", "R", "T", "focal_length", "principal_point"}
    batch = meta["get_data"]([2, 0])
    assert batch["image"].shape == (2, 3, 64, 64) and batch["R"].shape == (2, 3, 3)
    assert torch.allclose(batch["R"][1], torch.tensor(ds.frames["seq_a"][0]["R"], dtype=torch.float32))
    # independent evaluation of the crop: box of bbox (30,40,130,100) = square [30,20,130,120] scaled 1.15 about
    # (80,70) -> [22,12,137,127]; rows 120..126 lie below the 120-row picture and must read as zeros
    x0, y0, x1, y1 = ahv.co3d.eval_crop_box(np.array([30, 40, 130, 100])).tolist()
    assert (x0, y0, x1, y1) == (22, 12, 137, 127)
    canvas = np.zeros((y1 - y0, x1 - x0, 3), np.float32)
    canvas[:120 - y0] = pics[0][y0:120, x0:x1] / 255.0
    ref = torch.nn.functional.interpolate(torch.from_numpy(canvas).permute(2, 0, 1)[None], size=(64, 64), mode="bilinear",
                                          align_corners=False)[0]
    ref = (ref - 0.5) / 0.5
    assert torch.allclose(batch["image"][1], ref, atol=1e-6)
    assert torch.allclose(batch["image"][1][:, -1, :], torch.full((3, 64), -1.0), atol=1e-6)  # zero fill, normalised


def test_harness_decodes_only_key_frames(ahv, tmp_path):
    cfg, _ = write_dataset(str(tmp_path), n_frames=5)
    cats = ahv.co3d.load_categories(cfg, ["ball"])
    calls = []
    ds = cats["ball"]
    orig = ds.get_data
    ds.get_data = lambda name, ids: (calls.append(list(map(int, ids))) or orig(name, ids))

    class Model:  # stands in for Estimator.forward: (1,3,S,S) x2 -> two volumes
        def __call__(self, a, b):
            assert a.shape == (1, 3, 64, 64)
            return a.mean().expand(1, 16, 8, 8, 8), b.mean().expand(1, 16, 8, 8, 8)
    np.random.seed(0)
    P = ahv.rotations.random_rotations(32, generator=torch.Generator().manual_seed(0))
    errs = ahv.harness.evaluate_category(cfg, Model(), ds, proposals=P, device=torch.device("cpu"),
                                         verify_fn=lambda vs, vt, pr: (torch.zeros(1), torch.tensor([3])))
    assert len(calls) == 1 and len(calls[0]) == 2 and len(errs) == 2
    # GT = R1^T R2 of the two decoded frames; prediction = proposal 3 for both orders
    R = torch.stack([torch.tensor(ds.frames["seq_a"][i]["R"], dtype=torch.float32) for i in calls[0]])
    gt01 = R[0].T @ R[1]
    assert abs(errs[0] - ahv.rotations.geodesic_deg(P[3][None], gt01[None]).item()) < 1e-4
    assert abs(errs[1] - ahv.rotations.geodesic_deg(P[3][None], gt01.T[None]).item()) < 1e-4
