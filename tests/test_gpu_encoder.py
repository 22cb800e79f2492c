"""HIP token stage of the 3D-aware encoder (csrc/ahv_encoder.hip) against the stock-torch operator path of
the host mirror -- itself pinned to the reference by the encoder_small fixture (tests/test_aligner_cpu.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa(ahv):
    torch.manual_seed(0)
    m = ahv.aligner.Feature_Aligner(in_channel=768, mid_channel=256, out_channel=32, n_heads=4, depth=4).cuda().eval()
    return m


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item()


@pytest.mark.parametrize("B", [1, 3, 16, 17, 24, 32, 33])   # 16: the 128x128-tiled FF kernels (M >= 1024, M % 128 == 0); 17: back on
                                                         # the skinny ones; 24: attention per (sample, head) (from 192 pairs);
                                                         # 32: the 64x64-tiled q / k / v projections too (M >= 2048); 33: attention
                                                         # per (sample, head) beside the skinny linears (M % 128 != 0)
def test_transformer_blocks_match_torch_ops(fa, B):
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 256, 8, 8, generator=g).cuda()
    c = torch.randn(B, 256, 8, 8, generator=g).cuda()
    with torch.no_grad():
        fa.att.use_hip = False
        ref_x, ref_c = fa.att(x, c)
        fa.att.use_hip = True
        got_x, got_c = fa.att(x, c)
    assert rel(got_x, ref_x) < 2e-5 and rel(got_c, ref_c) < 2e-5


def test_single_block_and_attention_against_modules(ahv, fa):
    """depth-1 slice: isolates one BidirectionTransformerBlock."""
    import copy
    att1 = copy.deepcopy(fa.att)
    att1.transformer_blocks = torch.nn.ModuleList([att1.transformer_blocks[2]])
    att1.invalidate_packed()
    g = torch.Generator().manual_seed(9)
    xs, cs = torch.randn(2, 64, 256, generator=g).cuda(), torch.randn(2, 64, 256, generator=g).cuda()
    with torch.no_grad():
        rx, rc = att1.transformer_blocks[0](xs, cs)
        gx, gc = att1._hip_blocks(xs, cs)
    assert rel(gx, rx) < 1e-5 and rel(gc, rc) < 1e-5
    assert torch.equal(xs, xs.clone())  # inputs untouched (the kernels work on copies)


def test_attention_with_saturated_softmax(ahv, fa):
    """attention_heads_kernel merges four per-key-tile softmaxes with the online-softmax identity: with q/k weights
    scaled 20x the logits spread over hundreds (one key dominates, partial sums underflow) and the merge must still
    equal a one-pass softmax; also one stream of tokens made constant (all logits equal: uniform attention)."""
    import copy
    att1 = copy.deepcopy(fa.att)
    att1.transformer_blocks = torch.nn.ModuleList([att1.transformer_blocks[1]])
    with torch.no_grad():
        for blk in (att1.transformer_blocks[0].attn_self_1, att1.transformer_blocks[0].attn_cross_2):
            blk.attn.to_q.weight.mul_(20.0)
            blk.attn.to_k.weight.mul_(20.0)
    att1.invalidate_packed()
    g = torch.Generator().manual_seed(11)
    xs, cs = torch.randn(3, 64, 256, generator=g).cuda(), torch.randn(3, 64, 256, generator=g).cuda()
    cs[1] = cs[1, :1]                      # sample 1: 64 identical context tokens
    with torch.no_grad():
        rx, rc = att1.transformer_blocks[0](xs, cs)
        gx, gc = att1._hip_blocks(xs, cs)
        blk64 = copy.deepcopy(att1.transformer_blocks[0]).double()
        tx, tc = blk64(xs.double(), cs.double())          # fp64 truth
    assert torch.isfinite(gx).all() and torch.isfinite(gc).all()
    # logits of several hundred carry ~3e-5 of absolute fp32 rounding each, so two fp32 evaluations legitimately
    # differ by ~1e-4 here: the HIP path must be as close to the fp64 result as stock fp32 operators are
    for got, ref32, truth in ((gx, rx, tx), (gc, rc, tc)):
        e_hip, e_t32 = rel(got.double(), truth), rel(ref32.double(), truth)
        assert e_hip < 3e-4 and e_hip <= 3.0 * e_t32 + 1e-5, (e_hip, e_t32)


@pytest.mark.parametrize("B", [1, 2])
def test_forward_2d3d_hip_vs_torch_and_graph(fa, B):
    g = torch.Generator().manual_seed(4 + B)
    a, b = torch.randn(B, 768, 8, 8, generator=g).cuda(), torch.randn(B, 768, 8, 8, generator=g).cuda()
    with torch.no_grad():
        fa.use_hip_encoder = fa.att.use_hip = False
        ref = fa.forward_2d3d(a, b, random_mask=False, mask_ratio=0.0)   # stock torch operators
        fa.att.use_hip = True
        mid = fa.forward_2d3d(a, b, random_mask=False, mask_ratio=0.0)   # HIP token stage only
        fa.use_hip_encoder = True
        got = fa.forward_2d3d(a, b, random_mask=False, mask_ratio=0.0)   # everything on the HIP kernels
    for x, y, z in zip(got, ref, mid):
        assert x.shape == (B, 16, 8, 8, 8)
        assert rel(z, y) < 2e-5 and rel(x, y) < 2e-5
    if B != 1:
        return
    run = fa.graphed_forward_2d3d(batch=1)  # the HIP launches are captured into the hipGraph too
    out = run(a, b)
    for x, y in zip(out, ref):
        assert rel(x, y) < 2e-5


def test_state_dict_reload_invalidates_packed_weights(fa):
    import copy
    m = copy.deepcopy(fa)
    x = torch.randn(1, 256, 8, 8, device="cuda")
    with torch.no_grad():
        y0, _ = m.att(x, x)
        sd = {k: (v * 1.01 if "ff.net.2.weight" in k else v) for k, v in m.state_dict().items()}
        m.load_state_dict(sd)
        y1, _ = m.att(x, x)
        m.att.use_hip = False
        y2, _ = m.att(x, x)
    assert rel(y1, y2) < 2e-5 and rel(y0, y2) > 1e-4


def test_patch_rebinds_forward_2d3d_of_a_reference_shaped_module(ahv, fa):
    """patch.install() on a stand-in `modules.modules`: a class with the reference's attribute names whose own
    forward_2d3d runs stock torch operators; after patching, inference calls go through ahv_forward_2d3d_f32."""
    import copy
    import types

    class RefAligner(ahv.aligner.Feature_Aligner):  # torch-operator forward, like the reference's class
        def forward_2d3d(self, a, b, random_mask=True, mask_ratio=0.25):
            self.use_hip_encoder = self.att.use_hip = False
            return super().forward_2d3d(a, b, random_mask, mask_ratio)

    um, mm = types.ModuleType("utils"), types.ModuleType("modules.modules")
    um.rotate_volume = lambda *a, **k: None
    mm.Feature_Aligner = RefAligner
    ref = RefAligner(768, 256, 32, 4, 4).cuda().eval()
    ref.load_state_dict(fa.state_dict())
    g = torch.Generator().manual_seed(11)
    a, b = torch.randn(1, 768, 8, 8, generator=g).cuda(), torch.randn(1, 768, 8, 8, generator=g).cuda()
    with torch.no_grad():
        want = ref.forward_2d3d(a, b, random_mask=False, mask_ratio=0)
        ahv.patch.install(um, mm)
        try:
            assert RefAligner.forward_2d3d is ahv.patch._hip_forward_2d3d
            got = ref.forward_2d3d(a, b, random_mask=False, mask_ratio=0)
            masked = ref.forward_2d3d(a, b)  # training-style call falls through to the original
        finally:
            ahv.patch.uninstall()
    for x, y in zip(got, want):
        assert rel(x, y) < 2e-5
    assert masked[0].shape == (1, 16, 8, 8, 8)


def test_forward_2d3d_strided_batch_inputs(fa):
    """Inputs that need a contiguous copy (a strided batch such as feats[:, 0] of a (P, 2, C, 8, 8) tensor): the
    copies must outlive the launch -- regression for temporaries freed (and their block reused) before the kernel ran."""
    g = torch.Generator().manual_seed(11)
    feats = torch.randn(3, 2, 768, 8, 8, generator=g).cuda()
    with torch.no_grad():
        a, b = fa.forward_2d3d(feats[:, 0], feats[:, 1], random_mask=False, mask_ratio=0.0)
        ra, rb = fa.forward_2d3d(feats[:, 0].contiguous(), feats[:, 1].contiguous(), random_mask=False, mask_ratio=0.0)
    assert torch.equal(a, ra) and torch.equal(b, rb)


# ---- G7 `encoder_full`: reference -> HIP, one hop, at the size the kernels are built for -----------------------
@pytest.fixture(scope="module")
def full(ahv):
    """Mirror module carrying the key-seeded procedural weights the REFERENCE's Feature_Aligner carried when
    tools/gen_golden.py produced tests/golden/encoder_full.npz (modules/modules.py:86-110)."""
    from .conftest import load_golden
    from .procfill import procedural_state_dict
    g = load_golden("encoder_full")
    m = ahv.aligner.Feature_Aligner(in_channel=768, mid_channel=256, out_channel=32, n_heads=4, depth=4).eval()
    m.load_state_dict(procedural_state_dict(m.state_dict()), strict=True)
    return g, m.cuda()


@pytest.mark.parametrize("B", [1, 2, 17, 16, 32, 64])  # 17 = the stored pair tiled (skinny kernels, odd row count); 32 = the
                                               # tiled kernels (128-tiles for the FF projections, 64-tiles for the 256-wide
                                               # ones); 16 / 32 / 64 = the 3 x 3 convolutions as implicit GEMMs on 64-tiles
                                               # with K split 4 / 2 / 1 ways
def test_hip_forward_2d3d_matches_reference_fixture(full, B):
    """ahv_forward_2d3d_f32 vs the reference's own outputs: <= 1e-4 of the largest entry (measured ~1e-6)."""
    g, m = full
    xs = torch.from_numpy(g["x_src"]).float().cuda()
    xt = torch.from_numpy(g["x_tgt"]).float().cuda()
    pick = [i % 2 for i in range(B)]
    with torch.no_grad():
        assert m._hip_2d3d_eligible(xs[pick])
        v_src, v_tgt = m.forward_2d3d(xs[pick], xt[pick], random_mask=False, mask_ratio=0.0)
    want_s = torch.from_numpy(g["vol_src"]).cuda()[pick]
    want_t = torch.from_numpy(g["vol_tgt"]).cuda()[pick]
    assert v_src.shape == (B, 16, 8, 8, 8)
    assert rel(v_src, want_s) < 1e-4 and rel(v_tgt, want_t) < 1e-4, (rel(v_src, want_s), rel(v_tgt, want_t))


def test_hip_block0_tokens_match_reference_fixture(full):
    """ahv_transformer_blocks_f32 restricted to BidirectionTransformerBlock 0 (transformer/attention.py:269-274)
    vs the tokens a forward hook grabbed from the reference; block input = GroupNorm + proj_in of the embedded
    features, evaluated by the mirror's stock operators (pinned to the same fixture on CPU, first 8 tokens here)."""
    import copy
    g, m = full
    xs = torch.from_numpy(g["x_src"]).float().cuda()
    xt = torch.from_numpy(g["x_tgt"]).float().cuda()
    att1 = copy.deepcopy(m.att)
    att1.transformer_blocks = torch.nn.ModuleList([att1.transformer_blocks[0]])
    att1.invalidate_packed()
    with torch.no_grad():
        embed = lambda t: m.feature_embedding[1](m.feature_embedding[0](t))
        pe = m.posemb_sincos_2d(xs, channel=256)[None]
        tok = lambda t: t.flatten(2).transpose(1, 2)
        ts = tok(att1.proj_in(att1.norm(embed(xs) + pe)))
        tc = tok(att1.proj_context_in(att1.norm(embed(xt) + pe)))
        assert rel(ts[:, :8], torch.from_numpy(g["tok_in_src_first8"]).cuda()) < 1e-4
        assert rel(tc[:, :8], torch.from_numpy(g["tok_in_tgt_first8"]).cuda()) < 1e-4
        gx, gc = att1._hip_blocks(ts, tc)
    assert rel(gx, torch.from_numpy(g["tok0_src"]).cuda()) < 1e-4
    assert rel(gc, torch.from_numpy(g["tok0_tgt"]).cuda()) < 1e-4
