"""Reader for the reference's Lightning checkpoints ("checkpoint / resume").

The reference saves with ``ModelCheckpoint`` (train_estimator_co3d.py:20) and loads with
``Estimator.load_from_checkpoint(path, cfg=cfg)`` (test_co3d.py:218).  A Lightning ``.ckpt`` is a
``torch.save``d dict whose ``"state_dict"`` holds ``feature_aligner.<k>`` and
``feature_extractor.<k>`` tensors (key list: SURVEY.md section 8b).  Only that dict is needed here;
optimizer / loop state is ignored.
"""
from __future__ import annotations

import torch


def read_state_dict(path: str, map_location="cpu", trusted: bool = False) -> dict:
    """Returns the flat ``state_dict`` of a Lightning ``.ckpt`` (or of a bare state-dict file).

    Loaded with ``weights_only=True`` (tensors and plain containers only).  Lightning checkpoints that pickle other
    objects (hyper-parameters, callbacks) need the full unpickler, which can execute arbitrary code from the file:
    that is an explicit opt-in, ``trusted=True``, for files whose origin you trust -- never a silent retry."""
    try:
        blob = torch.load(path, map_location=map_location, weights_only=not trusted)
    except Exception as e:
        if trusted:
            raise
        raise RuntimeError("%s could not be read with weights_only=True (%s: %s).  If the file comes from a source "
                           "you trust, pass trusted=True to allow full unpickling." % (path, type(e).__name__, e)) from e
    if isinstance(blob, dict) and "state_dict" in blob:
        blob = blob["state_dict"]
    if not isinstance(blob, dict) or not all(isinstance(k, str) for k in blob):
        raise ValueError("%s does not hold a state_dict" % path)
    return blob


def split_prefix(state_dict: dict, prefix: str) -> dict:
    p = prefix if prefix.endswith(".") else prefix + "."
    return {k[len(p):]: v for k, v in state_dict.items() if k.startswith(p)}


def load_into(model, state_dict: dict, strict: bool = True, strict_backbone: bool = False) -> None:
    """Loads ``feature_aligner.*`` (always) and ``feature_extractor.*`` (when the model has a backbone
    with matching keys).  ``strict`` applies to the aligner: it is the part this build owns.  A checkpoint that
    carries backbone weights the model's backbone cannot take (different key set) is never skipped silently: it
    warns -- the aligner would otherwise run on a randomly initialised backbone unnoticed -- or raises with
    ``strict_backbone``."""
    aligner = split_prefix(state_dict, "feature_aligner")
    if not aligner:
        raise KeyError("checkpoint has no feature_aligner.* tensors")
    model.feature_aligner.load_state_dict(aligner, strict=strict)
    backbone = split_prefix(state_dict, "feature_extractor")
    fx = getattr(model, "feature_extractor", None)
    if backbone and fx is not None:
        own = fx.state_dict() if hasattr(fx, "state_dict") else {}
        if set(own) == set(backbone):
            fx.load_state_dict(backbone, strict=True)
        else:
            msg = ("checkpoint holds %d feature_extractor.* tensors that do not match the model's backbone (%d keys, "
                   "%d in common): backbone weights NOT loaded" % (len(backbone), len(own), len(set(own) & set(backbone))))
            if strict_backbone:
                raise KeyError(msg)
            import warnings
            warnings.warn(msg, RuntimeWarning, stacklevel=2)


def save_lightning_style(path: str, model, extra: dict | None = None) -> None:
    """Writes ``{"state_dict": ...}`` with the reference's prefixes (used by tests and synthetic runs)."""
    sd = {"feature_aligner." + k: v.detach().cpu() for k, v in model.feature_aligner.state_dict().items()}
    fx = getattr(model, "feature_extractor", None)
    if fx is not None:
        sd.update({"feature_extractor." + k: v.detach().cpu() for k, v in fx.state_dict().items()})
    blob = {"state_dict": sd, "epoch": 0, "global_step": 0, "pytorch-lightning_version": "2.0.0"}
    if extra:
        blob.update(extra)
    torch.save(blob, path)
