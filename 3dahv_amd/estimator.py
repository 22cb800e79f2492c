"""Host-side mirror of the reference ``Estimator`` call surface.

Two variants exist in the reference, both Lightning modules:

* ``modules/model_co3d.py:26-101``  -- ``forward(img_src, img_tgt)``           (test_co3d.py)
* ``modules/model.py:28-218``       -- ``forward(img_src, mask_src, img_tgt, mask_tgt)``,
  ``validation_step`` / ``test_step``                                          (test_objaverse.py, test_linemod.py)

Mirrored here as plain ``nn.Module`` s: constructor ``Estimator(cfg)``, attributes
``feature_extractor / feature_aligner / num_rota / step_outputs / gt_dis / pred_Rs``,
``feature_extraction``, ``forward``, ``test_step``, ``validation_step``, ``eval()``,
``load_from_checkpoint(path, cfg=cfg)`` and the ``state_dict`` prefixes
``feature_extractor.*`` / ``feature_aligner.*``.  The per-hypothesis loop inside the steps runs
as ONE fused HIP launch.  ``training_step`` / ``infoNCE_loss`` / ``configure_optimizers`` are here as well: the
loss back-propagates through the HIP backward of the fused scorer (``ops.score_hypotheses_autograd``,
``ops.forward_3d2d_autograd``) and through the stock-torch encoder; Lightning's trainer machinery (DDP
launcher, logging, checkpoint callbacks) is not mirrored -- a plain loop over ``training_step`` +
``optimizer.step()`` is the counterpart.

The MiDaS DPT/Swin-V2 backbone (``feature_extractor``) is stock timm code that is neither
installed nor downloadable offline; it is injected (``feature_extractor=`` or
``backbone_factory=``) and only its ``layer_4`` contract is relied on: images
``(B,3,256,256)`` -> ``(B,768,8,8)`` (MiDaS/midas/backbones/swin_common.py:47-50).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn as nn

from . import checkpoint as _ckpt
from . import ops
from .aligner import Feature_Aligner
from .rotations import geodesic_deg, random_rotations


class PatchifyBackbone(nn.Module):
    """Stand-in ``feature_extractor`` with the same I/O contract as the reference's Swin stage-4 hook
    (images (B,3,256,256) -> layer_4 (B,768,8,8)): one stride-32 patch embedding.  For synthetic
    runs and tests only; real accuracy needs the real MiDaS weights."""

    def __init__(self, out_channels: int = 768, patch: int = 32, seed: int = 0):
        super().__init__()
        self.proj = nn.Conv2d(3, out_channels, patch, stride=patch)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            self.proj.weight.copy_(torch.randn(self.proj.weight.shape, generator=g) * 0.02)
            self.proj.bias.zero_()

    def forward(self, img):
        # a stride = kernel convolution is a reshape + one GEMM; written that way because MIOpen has only its naive
        # fp32 kernels for this shape on gfx950 (4.4 ms forward + 2.8 ms weight gradient per step at B = 12)
        p = self.proj.kernel_size[0]
        B, C, H, W = img.shape
        x = img.reshape(B, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).reshape(B, (H // p) * (W // p), C * p * p)
        y = x @ self.proj.weight.reshape(self.proj.out_channels, -1).t() + self.proj.bias
        return y.transpose(1, 2).reshape(B, -1, H // p, W // p)


class _EstimatorBase(nn.Module):
    def __init__(self, cfg, feature_extractor: Optional[nn.Module] = None,
                 backbone_factory: Optional[Callable[[], nn.Module]] = None):
        super().__init__()
        self.cfg = cfg
        self.num_rota = cfg["DATA"]["NUM_ROTA"]
        self.mid_channel = 256
        if feature_extractor is None and backbone_factory is not None:
            feature_extractor = backbone_factory()
        if feature_extractor is None:
            feature_extractor = _try_midas_backbone(cfg)
        self.feature_extractor = feature_extractor
        self.feature_aligner = Feature_Aligner(in_channel=768, mid_channel=256, out_channel=32, n_heads=4, depth=4)
        self.step_outputs = []
        self.gt_dis = []
        self.pred_Rs = []
        self.logged = {}  # stands in for Lightning's self.log
        self._proposal_draws = 0

    # -- reference: modules/model_co3d.py:37-39, modules/model.py:39-41
    def feature_extraction(self, img):
        fx = self.feature_extractor
        if fx is None:
            raise RuntimeError("no feature_extractor: the reference's DPT_SwinV2_T_256 factory could not be used (%s); pass "
                               "feature_extractor= / backbone_factory=, or feed layer_4 features to forward_features()"
                               % (midas_unavailable_reason or "not attempted"))
        if hasattr(fx, "forward_transformer"):  # the real MiDaS DPT object
            return fx.forward_transformer(fx.pretrained, img)[3]
        return fx(img)

    def forward_features(self, layer4_src, layer4_tgt):
        """Everything after the backbone: (B,768,8,8) x2 -> volumes (B,16,8,8,8) x2."""
        return self.feature_aligner.forward_2d3d(layer4_src, layer4_tgt, random_mask=False, mask_ratio=0.0)

    def fresh_proposals(self, device):
        """`random_rotations(self.num_rota)` of the reference steps (modules/model.py:184): a fresh Haar set per
        call, generated on the GPU (counter-based, seeded by torch's global seed and a draw counter)."""
        self._proposal_draws += 1
        if torch.device(device).type == "cuda":
            return ops.random_rotations(self.num_rota, seed=torch.initial_seed() + self._proposal_draws, device=device)
        return random_rotations(self.num_rota, device=device)

    def log(self, name, value, **_):
        self.logged.setdefault(name, []).append(float(value))

    # -- the verify step shared by test_step / validation_step / the harness
    @torch.no_grad()
    def verify(self, img_feat_src, img_feat_tgt, proposals, want_scores: bool = False, want_feat_tgt: bool = False):
        """For B volume pairs and shared proposals (N,3,3): scores (optional), best score, best index,
        R_pred = proposals[idx]  (test_co3d.py:137-146, modules/model.py:186-196): one fused launch + one select.
        ``want_feat_tgt``: also return forward_3d2d(img_feat_tgt) as the launch built it (fifth element)."""
        res = ops.verify_pair(img_feat_src, img_feat_tgt, proposals, *self.feature_aligner.head_weights(),
                              want_scores=want_scores, want_feat_tgt=want_feat_tgt)
        best, idx, R_pred = ops.select_rotation(res[1], proposals)   # torch.max + proposals[idx] in one launch
        return (res[0], best, idx, R_pred, res[2]) if want_feat_tgt else (res[0], best, idx, R_pred)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, cfg=None, map_location="cpu", strict: bool = True,
                             trusted: bool = False, **kw):
        """Reads a Lightning ``.ckpt`` written by the reference (``state_dict`` with prefixes
        ``feature_aligner.*`` / ``feature_extractor.*``).  ``cfg`` is required, as in test_co3d.py:218.
        ``trusted=True`` allows full unpickling (Lightning files that carry non-tensor objects) -- only for files
        whose origin you trust."""
        if cfg is None:
            raise TypeError("load_from_checkpoint() missing cfg (the reference passes cfg=cfg, test_co3d.py:218)")
        model = cls(cfg, **kw)
        sd = _ckpt.read_state_dict(checkpoint_path, map_location=map_location, trusted=trusted)
        _ckpt.load_into(model, sd, strict=strict)
        return model

    def infoNCE_loss(self, img_feat_1, img_feat_2, sampled_R, gt_delta_R, reduce_mean: Optional[bool] = None):
        """The reference's InfoNCE loss (modules/model_co3d.py:41-61, modules/model.py:43-63): per-sample
        hypothesis sets ``sampled_R (B,N,3,3)``, positives = hypotheses within ``DATA.ACC_THR`` degrees of
        ``gt_delta_R``, ``-log(sum_pos exp(s/0.1) / sum_all exp(s/0.1))``.  The (B,N) similarities come from ONE
        fused HIP launch with per-sample rotations; under autograd their backward is the three-kernel HIP backward
        (``ops.score_hypotheses_autograd``), so the loss is differentiable w.r.t. both volumes and the head.
        ``reduce_mean`` defaults to the variant's behaviour: mean (model_co3d.py:59) or per-sample (model.py:61)."""
        import math
        with torch.no_grad():
            gt_sim = (torch.sum(sampled_R.flatten(2) * gt_delta_R.reshape(-1, 1, 9), dim=-1).clamp(-1, 3) - 1) / 2
            positive = 180 * (torch.arccos(gt_sim) / math.pi) <= self.cfg["DATA"]["ACC_THR"]      # (B, N)
        f_tgt = self.feature_aligner.forward_3d2d(img_feat_2)
        W1, W2, b2 = self.feature_aligner.head_weights()
        if torch.is_grad_enabled() and (img_feat_1.requires_grad or f_tgt.requires_grad or W1.requires_grad):
            sim = ops.score_hypotheses_autograd(img_feat_1, f_tgt, sampled_R.contiguous(), W1, W2, b2)
        else:
            sim, _ = ops.score_hypotheses(img_feat_1, f_tgt, sampled_R.contiguous(), W1, W2, b2)
        e = torch.exp(sim / 0.1)
        loss = -torch.log((e * positive).sum(dim=-1) / e.sum(dim=-1).clamp(min=1e-8))
        if reduce_mean is None:
            reduce_mean = isinstance(self, EstimatorCo3d)
        return loss.mean() if reduce_mean else loss

    def sample_training_rotations(self, gt_src_2_tgt_R):
        """``cat([gt, random_rotations(B*(num_rota-1))])`` (modules/model_co3d.py:84-86): the ground truth is
        hypothesis 0 of every sample, the rest are fresh Haar samples (generated on the GPU)."""
        B, dev = gt_src_2_tgt_R.shape[0], gt_src_2_tgt_R.device
        self._proposal_draws += 1
        n = B * (self.num_rota - 1)
        if dev.type == "cuda":
            R = ops.random_rotations(n, seed=torch.initial_seed() + self._proposal_draws, device=dev)
        else:
            R = random_rotations(n, device=dev)
        return torch.cat([gt_src_2_tgt_R[:, None], R.reshape(B, self.num_rota - 1, 3, 3)], dim=1)

    def training_loss(self, vol_src, vol_tgt, gt_src_2_tgt_R):
        """Everything of ``training_step`` after the encoder: sample rotations, InfoNCE (mean over the batch)."""
        with torch.no_grad():
            sampled_R = self.sample_training_rotations(gt_src_2_tgt_R)
        return self.infoNCE_loss(vol_src, vol_tgt, sampled_R, gt_src_2_tgt_R, reduce_mean=True)

    def configure_optimizers(self):
        """AdamW(eps=1e-5) on the aligner (+ backbone when present) and StepLR(200, 0.1)
        (modules/model_co3d.py:93-99)."""
        lr = float(self.cfg["TRAIN"]["LR"])
        groups = [{"params": self.feature_aligner.parameters(), "lr": lr}]
        if isinstance(self.feature_extractor, nn.Module):
            groups.append({"params": self.feature_extractor.parameters(), "lr": lr})
        optimizer = torch.optim.AdamW(groups, eps=1e-5)
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=200, gamma=0.1)
        return [optimizer], [scheduler]


class EstimatorCo3d(_EstimatorBase):
    """modules/model_co3d.py::Estimator."""

    def forward(self, img_src, img_tgt):
        f_src, f_tgt = self.feature_extraction(img_src), self.feature_extraction(img_tgt)
        return self.forward_features(f_src, f_tgt)

    def training_step(self, batch, batch_idx):
        """modules/model_co3d.py:71-91: batch keys ``image (B,2,3,S,S)``, ``relative_rotation (B,1,3,3)``."""
        img_src, img_tgt = batch["image"][:, 0], batch["image"][:, 1]
        gt = batch["relative_rotation"].squeeze(1)
        vol_src, vol_tgt = self.feature_aligner.forward_2d3d(
            self.feature_extraction(img_src), self.feature_extraction(img_tgt),
            random_mask=self.cfg["TRAIN"]["MASK"], mask_ratio=self.cfg["TRAIN"]["MASK_RATIO"])
        loss = self.training_loss(vol_src, vol_tgt, gt)
        self.log("train_loss", loss.item())
        return loss


class EstimatorObjaverse(_EstimatorBase):
    """modules/model.py::Estimator (Objaverse / LINEMOD)."""

    def forward(self, img_src, mask_src, img_tgt, mask_tgt):
        if self.cfg["DATA"]["BG"] is False:  # modules/model.py:67-69
            img_src, img_tgt = img_src * mask_src, img_tgt * mask_tgt
        f_src, f_tgt = self.feature_extraction(img_src), self.feature_extraction(img_tgt)
        return self.forward_features(f_src, f_tgt)

    def training_step(self, batch, batch_idx):
        """modules/model.py:78-116: masks out small objects / far views from the per-sample losses."""
        mask_src, mask_tgt = batch["src_mask"], batch["ref_mask"]
        img_src, img_tgt = batch["src_img"], batch["ref_img"]
        if self.cfg["DATA"]["BG"] is False:
            img_src, img_tgt = img_src * mask_src, img_tgt * mask_tgt
        with torch.no_grad():
            gt = torch.bmm(batch["ref_R"], torch.inverse(batch["src_R"]))
        vol_src, vol_tgt = self.feature_aligner.forward_2d3d(
            self.feature_extraction(img_src), self.feature_extraction(img_tgt),
            random_mask=self.cfg["TRAIN"]["MASK"], mask_ratio=self.cfg["TRAIN"]["MASK_RATIO"])
        with torch.no_grad():
            self.Rs = self.sample_training_rotations(gt)
        thr = self.cfg["DATA"]["SIZE_THR"]
        valid = (mask_src.flatten(1).sum(dim=-1) > thr) * (mask_tgt.flatten(1).sum(dim=-1) > thr)
        if "dis_init" in batch:
            valid = valid * (batch["dis_init"] < self.cfg["DATA"]["VIEW_THR"]).float()
        loss = self.infoNCE_loss(vol_src, vol_tgt, self.Rs, gt, reduce_mean=False) * valid
        loss = loss.sum() / valid.sum().clamp(min=1e-8)
        self.log("train_loss", loss.item())
        return loss

    def configure_optimizers(self):
        """modules/model.py:212-218: the backbone trains at a tenth of the aligner's rate, StepLR(20, 0.1)."""
        lr = float(self.cfg["TRAIN"]["LR"])
        groups = [{"params": self.feature_aligner.parameters(), "lr": lr}]
        if isinstance(self.feature_extractor, nn.Module):
            groups.append({"params": self.feature_extractor.parameters(), "lr": 0.1 * lr})
        optimizer = torch.optim.AdamW(groups, eps=1e-5)
        return [optimizer], [torch.optim.lr_scheduler.StepLR(optimizer, step_size=20, gamma=0.1)]

    def _too_small(self, mask_src, mask_tgt):
        thr = self.cfg["DATA"]["SIZE_THR"]
        return bool(torch.any(mask_src.flatten(1).sum(dim=-1) < thr) or torch.any(mask_tgt.flatten(1).sum(dim=-1) < thr))

    @torch.no_grad()
    def test_step(self, batch, batch_idx, proposals=None):
        """modules/model.py:168-209.  ``proposals`` defaults to fresh Haar samples shared by the batch."""
        mask_src, mask_tgt = batch["src_mask"], batch["ref_mask"]
        img_src, img_tgt = batch["src_img"], batch["ref_img"]
        R_src, R_tgt = batch["src_R"], batch["ref_R"]
        if self._too_small(mask_src, mask_tgt):
            print("Skip bad case")
            return 0
        vol_src, vol_tgt = self.forward(img_src, mask_src, img_tgt, mask_tgt)
        gt_src_2_tgt_R = torch.bmm(R_tgt, torch.inverse(R_src))
        if proposals is None:
            proposals = self.fresh_proposals(img_src.device)
        _, _, _, pred_R = self.verify(vol_src, vol_tgt, proposals)
        geo_dis = geodesic_deg(pred_R, gt_src_2_tgt_R)
        gt_dis = geodesic_deg(R_src, R_tgt)
        self.step_outputs.append(geo_dis)
        self.gt_dis.append(gt_dis)
        self.pred_Rs.append(pred_R.cpu().numpy().reshape(-1))
        self.log("test_error", geo_dis.mean().item())
        return geo_dis

    @torch.no_grad()
    def validation_step(self, batch, batch_idx, proposals=None):
        """modules/model.py:118-158: test_step plus the GT-rotation score and Acc@15/30 (<=)."""
        mask_src, mask_tgt = batch["src_mask"], batch["ref_mask"]
        img_src, img_tgt = batch["src_img"], batch["ref_img"]
        R_src, R_tgt = batch["src_R"], batch["ref_R"]
        vol_src, vol_tgt = self.forward(img_src, mask_src, img_tgt, mask_tgt)
        gt_src_2_tgt_R = torch.bmm(R_tgt, torch.inverse(R_src))
        if proposals is None:
            proposals = self.fresh_proposals(img_src.device)
        _, pred_sim, _, pred_R, f_tgt = self.verify(vol_src, vol_tgt, proposals, want_feat_tgt=True)
        # gt_sim: each sample's own GT rotation = per-sample R with N = 1 (modules/model.py:137-143), against the target
        # features the verify launch built
        gt_sim, _ = ops.score_hypotheses(vol_src, f_tgt, gt_src_2_tgt_R[:, None].contiguous(),
                                         *self.feature_aligner.head_weights())
        geo_dis = geodesic_deg(pred_R, gt_src_2_tgt_R)
        self.log("val_acc_15", (geo_dis <= 15).float().mean().item())
        self.log("val_acc_30", (geo_dis <= 30).float().mean().item())
        self.step_outputs.append(geo_dis)
        return {"geo_dis": geo_dis, "pred_sim": pred_sim, "gt_sim": gt_sim[:, 0]}

    def on_validation_epoch_end(self):
        geo_dis = torch.cat(self.step_outputs)
        out = (100 * (geo_dis <= 15).float().mean(), 100 * (geo_dis <= 30).float().mean())
        self.step_outputs.clear()
        return out


Estimator = EstimatorCo3d  # the class test_co3d.py imports from modules.model_co3d


#: why the last ``_try_midas_backbone`` call returned None (one line; ``Estimator(cfg)`` callers can print it)
midas_unavailable_reason = ""


def _try_midas_backbone(cfg=None):
    """What the reference's constructor does -- ``self.feature_extractor = DPT_SwinV2_T_256(pretrained=True)``
    (modules/model_co3d.py:32, modules/model.py:34; factory at MiDaS/hubconf.py:124-145) -- attempted for real:
    the MiDaS directory (``cfg["MODEL"]["MIDAS_DIR"]``, else ``$AHV_MIDAS_DIR``, else ``MiDaS/`` next to the working
    directory, which is where the reference's scripts run from) goes on ``sys.path`` and its ``hubconf`` is imported.
    The factory needs timm == 0.6.12 and downloads ``dpt_swin2_tiny_256.pt`` unless torch.hub already has it cached;
    ``$AHV_MIDAS_PRETRAINED=0`` builds the architecture without weights (a checkpoint's ``feature_extractor.*`` tensors
    then fill it, ``load_from_checkpoint``).  Any failure -- no directory, no timm, no network -- returns None with the
    reason in ``estimator.midas_unavailable_reason``: the caller injects a backbone or feeds ``layer_4`` features."""
    global midas_unavailable_reason
    import importlib
    import os
    import sys
    cand = []
    if cfg is not None:
        try:
            cand.append(cfg["MODEL"]["MIDAS_DIR"])
        except (KeyError, TypeError):
            pass
    if os.environ.get("AHV_MIDAS_DIR"):
        cand.append(os.environ["AHV_MIDAS_DIR"])
    cand.append(os.path.join(os.getcwd(), "MiDaS"))
    midas_dir = next((d for d in cand if d and os.path.isfile(os.path.join(d, "hubconf.py"))), None)
    if midas_dir is None:
        midas_unavailable_reason = "no MiDaS/hubconf.py in %s" % ", ".join(repr(d) for d in cand)
        return None
    added = midas_dir not in sys.path
    if added:
        sys.path.append(midas_dir)  # as the reference does (modules/model_co3d.py:11)
    try:
        hubconf = sys.modules.get("hubconf")
        if hubconf is None or os.path.dirname(os.path.abspath(getattr(hubconf, "__file__", ""))) != os.path.abspath(midas_dir):
            sys.modules.pop("hubconf", None)
            hubconf = importlib.import_module("hubconf")
        model = hubconf.DPT_SwinV2_T_256(pretrained=os.environ.get("AHV_MIDAS_PRETRAINED", "1") != "0")
    except Exception as e:  # ImportError (timm), URLError (download), a changed factory ... : say which, carry on without
        midas_unavailable_reason = "%s: %s: %s" % (os.path.join(midas_dir, "hubconf.py"), type(e).__name__, e)
        if added:
            sys.path.remove(midas_dir)
        return None
    midas_unavailable_reason = ""
    return model
