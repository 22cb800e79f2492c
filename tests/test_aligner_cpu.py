"""Host mirror of Feature_Aligner: state-dict key parity with the reference and forward_2d3d numerics
(SURVEY.md 8a rows A6/A7), pinned by the encoder_small golden fixture.  CPU (torch ops only)."""
import numpy as np
import pytest
import torch

from .conftest import load_golden


@pytest.fixture(scope="module")
def small(ahv):
    g = load_golden("encoder_small")
    fa = ahv.aligner.Feature_Aligner(in_channel=64, mid_channel=32, out_channel=32, n_heads=4, depth=1).eval()
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    return g, fa, sd


def test_state_dict_keys_match_reference(small):
    g, fa, sd = small
    mine = fa.state_dict()
    assert sorted(mine.keys()) == sorted(sd.keys())
    for k in sd:
        assert tuple(mine[k].shape) == tuple(sd[k].shape), k
    fa.load_state_dict(sd, strict=True)


def test_full_size_key_set(ahv):
    fa = ahv.aligner.Feature_Aligner(in_channel=768, mid_channel=256, out_channel=32, n_heads=4, depth=4)
    sd = fa.state_dict()
    assert sd["feature_embedding.0.weight"].shape == (256, 768, 1, 1)
    assert sd["feature_embedding_2d.0.weight"].shape == (32, 384, 1, 1)
    assert sd["feature_embedding_2d.2.bias"].shape == (32,)
    assert sd["feature_embedding_3d.downsample.0.weight"].shape == (16, 32, 1, 1, 1)
    assert sd["att.transformer_blocks.3.attn_cross_2.ff.net.0.proj.weight"].shape == (4096, 512)
    assert sd["att.transformer_blocks.0.attn_self_1.attn.to_out.0.bias"].shape == (256,)
    n_param = sum(p.numel() for p in fa.parameters())
    assert n_param == 47902016  # SURVEY.md section 2 row 4 [probed on the reference]


def test_forward_2d3d_matches_reference(small):
    g, fa, sd = small
    fa.load_state_dict(sd, strict=True)
    T = lambda k: torch.from_numpy(g[k])
    with torch.no_grad():
        e_src = fa.feature_embedding(T("x_src"))
        assert torch.allclose(e_src, T("emb_src"), atol=1e-5, rtol=1e-5)
        pe = fa.posemb_sincos_2d(e_src, channel=32)
        assert torch.allclose(pe, T("posemb"), atol=1e-6)
        a_src, a_tgt = fa.att(e_src + pe[None], fa.feature_embedding(T("x_tgt")) + pe[None])
        assert torch.allclose(a_src, T("att_src"), atol=2e-5, rtol=1e-4)
        assert torch.allclose(a_tgt, T("att_tgt"), atol=2e-5, rtol=1e-4)
        v_src, v_tgt = fa.forward_2d3d(T("x_src"), T("x_tgt"), random_mask=False, mask_ratio=0)
    assert v_src.shape == (2, 16, 8, 8, 8)
    for got, ref in ((v_src, g["vol_src"]), (v_tgt, g["vol_tgt"])):
        rel = np.max(np.abs(got.numpy() - ref)) / np.max(np.abs(ref))
        assert rel < 1e-5, rel


def test_random_masking_statistics(ahv):
    torch.manual_seed(0)
    x = torch.randn(4000, 16, 8, 8, 8)
    m = ahv.aligner.random_masking(x, 0.25)
    assert m.shape == (4000, 512) and set(m.unique().tolist()) <= {0.0, 1.0}
    full = (m.sum(1) == 512).float().mean().item()     # gate: about half the samples keep everything
    assert 0.45 < full < 0.55
    part = m[m.sum(1) < 512]
    assert torch.all(part.sum(1) == int(512 * 0.75))    # the others keep exactly 75 %


def test_forward_3d2d_has_no_cpu_path(ahv):
    fa = ahv.aligner.Feature_Aligner(64, 32, 32, 4, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fa.forward_3d2d(torch.zeros(1, 16, 8, 8, 8))


def test_conv_as_matmul_equals_conv_forward_and_backward(ahv):
    """aligner.conv_mm swaps the aligner's convolutions for unfold + matmul under autograd on the GPU; the GEMM form
    must be the same function (values and gradients) for every convolution shape the aligner has."""
    import torch.nn as nn
    torch.manual_seed(0)
    cases = [(nn.Conv2d(24, 8, 1, bias=False), (2, 24, 8, 8)), (nn.Conv2d(8, 8, 1), (2, 8, 8, 8)),
             (nn.Conv2d(8, 8, 3, padding=1, bias=False), (3, 8, 8, 8)),
             (nn.Conv3d(4, 2, 3, padding=1, bias=False), (2, 4, 8, 8, 8)), (nn.Conv3d(4, 2, 1, bias=False), (2, 4, 8, 8, 8))]
    for conv, shape in cases:
        conv = conv.double()
        x1 = torch.randn(*shape, dtype=torch.float64, requires_grad=True)
        x2 = x1.detach().clone().requires_grad_(True)
        y1, y2 = conv(x1), ahv.aligner.conv_as_matmul(conv, x2)
        assert y1.shape == y2.shape and torch.allclose(y1, y2, atol=1e-12)
        g = torch.randn_like(y1)
        params = list(conv.parameters())
        r1 = torch.autograd.grad(y1, [x1] + params, grad_outputs=g)
        r2 = torch.autograd.grad(y2, [x2] + params, grad_outputs=g)
        for a, b in zip(r1, r2):
            assert torch.allclose(a, b, atol=1e-11)
    assert ahv.aligner.conv_mm(cases[0][0], torch.randn(2, 24, 8, 8, dtype=torch.float64)).shape == (2, 8, 8, 8)  # CPU: the module itself


# ---- G7 `encoder_full`: the mirror at the size the HIP kernels are built for (768 / 256 / 4 x 64 / depth 4) ----
@pytest.fixture(scope="module")
def full(ahv):
    from .procfill import procedural_state_dict
    g = load_golden("encoder_full")
    fa = ahv.aligner.Feature_Aligner(in_channel=768, mid_channel=256, out_channel=32, n_heads=4, depth=4).eval()
    fa.load_state_dict(procedural_state_dict(fa.state_dict()), strict=True)
    assert sum(p.numel() for p in fa.parameters()) == int(g["n_param"])
    return g, fa


def test_full_size_mirror_matches_reference_fixture(full):
    """Reference Feature_Aligner(768,256,32,4,4).forward_2d3d (modules/modules.py:86-110) -> mirror, <= 1e-5 of the
    tensor's largest entry on both volumes and on the tokens after BidirectionTransformerBlock 0."""
    g, fa = full
    grabbed = {}
    hook = fa.att.transformer_blocks[0].register_forward_hook(
        lambda m, i, o: grabbed.update(tin=(i[0].detach(), i[1].detach()), tout=(o[0].detach(), o[1].detach())))
    x_src, x_tgt = torch.from_numpy(g["x_src"]).float(), torch.from_numpy(g["x_tgt"]).float()
    with torch.no_grad():
        v_src, v_tgt = fa.forward_2d3d(x_src, x_tgt, random_mask=False, mask_ratio=0)
    hook.remove()
    rel = lambda a, b: float(np.max(np.abs(a.numpy() - b)) / np.max(np.abs(b)))
    assert rel(grabbed["tin"][0][:, :8], g["tok_in_src_first8"]) < 1e-5
    assert rel(grabbed["tin"][1][:, :8], g["tok_in_tgt_first8"]) < 1e-5
    assert rel(grabbed["tout"][0], g["tok0_src"]) < 1e-5 and rel(grabbed["tout"][1], g["tok0_tgt"]) < 1e-5
    assert rel(v_src, g["vol_src"]) < 1e-5 and rel(v_tgt, g["vol_tgt"]) < 1e-5


def test_procedural_fill_is_stable():
    """The fill must not drift with the numpy version: a few pinned values (generated when the fixture was)."""
    from .procfill import procedural_tensor
    a = procedural_tensor("att.transformer_blocks.0.attn_self_1.ff.net.0.proj.weight", (4096, 512))
    b = procedural_tensor("att.norm.weight", (256,))
    c = procedural_tensor("feature_embedding_2d.2.bias", (32,))
    assert a.dtype == np.float32 and abs(float(np.abs(a).max()) - 1 / np.sqrt(512)) < 1e-6
    assert 0.75 <= float(b.min()) and float(b.max()) <= 1.25 and abs(float(c.max())) <= 0.1
    import hashlib
    h = hashlib.sha256(a.tobytes() + b.tobytes() + c.tobytes()).hexdigest()
    assert h == PROCFILL_SHA, h


PROCFILL_SHA = "64658f846e85ec49b4fc90c1ded4fda17c2d5e5ad0592ce1e5770ee6503354d8"


def test_packed_weight_key_sees_reregistered_parameters(ahv):
    """the staleness key of the packed-weight tables is computed over a CACHED parameter list.  An in-place
    update, a storage move and a parameter RE-REGISTERED as a new object (the old object's version and address do not
    change) must all change the key; an untouched module must not."""
    import torch.nn as nn
    al = ahv.aligner
    m = nn.Sequential(nn.Linear(4, 4), nn.Sequential(nn.Linear(4, 2)))
    params = al._param_list(m)
    k0 = al._version_key(params)
    assert al._version_key(params) == k0
    with torch.no_grad():
        m[0].weight.add_(1.0)                                   # in-place update: version counter
    k1 = al._version_key(params)
    assert k1 != k0
    m[1][0].bias = nn.Parameter(torch.ones(2))                  # re-registration: only the owner's dict shows it
    k2 = al._version_key(params)
    assert k2 != k1 and k2[0] == k1[0] and k2[1] == k1[1]       # versions and addresses of the cached objects unchanged
    m[0].weight.data = m[0].weight.data.clone()                 # storage moved
    assert al._version_key(params) != k2
