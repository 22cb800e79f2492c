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
