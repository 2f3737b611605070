"""Per-kernel parity through the C ABI on a real MI355X.

Integer work (masking, index selection) must be bit-exact against the oracle and the
reference-generated golden vectors; floating point is compared with the tolerances of
the task: 1e-5 relative in fp32, 2e-2 relative in bf16 (norm-relative per tensor).
"""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-5
BF16_TOL = 2e-2


# ------------------------------------------------------------------ K1 masking
def test_mask_golden_bit_exact(dev):
    from src.masking_strategy import RandomMaskingStrategy
    g = load_golden("masking.npz")
    st = RandomMaskingStrategy(101, 102, 0, 103)
    ids = torch.from_numpy(g["ka_ids"]).to(dev)
    # reference tests/test_masking_strategy.py:22-49
    m, i = st.apply_random_mask(ids, mask_prob=1.0)
    assert np.array_equal(m.cpu().numpy(), g["ka_masked_p1"]) and np.array_equal(i.cpu().numpy(), g["ka_mask_p1"])
    assert i.dtype == torch.bool and m.dtype == torch.int64
    m, i = st.apply_random_mask(ids, mask_prob=0.0)
    assert torch.equal(m, ids) and not i.any()
    for j in range(5):
        ids = torch.from_numpy(g[f"r{j}_ids"]).to(dev)
        before = ids.clone()
        m, i = st.apply_random_mask(ids, float(g[f"r{j}_p"][0]), draw=torch.from_numpy(g[f"r{j}_draw"]).to(dev))
        assert np.array_equal(m.cpu().numpy(), g[f"r{j}_masked"]), j
        assert np.array_equal(i.cpu().numpy(), g[f"r{j}_mask"]), j
        assert torch.equal(ids, before)          # input is not modified


def test_mask_torch_rng_stream_matches_reference_semantics(dev):
    """Default mode consumes torch's generator exactly like the reference line
    `rand = torch.rand_like(input_ids, dtype=torch.float)`."""
    from oracle import step_oracle as so
    from src.masking_strategy import RandomMaskingStrategy
    ids, _ = so.synthetic_batch(8, 512, 30522, seed=3, ragged=True, dirty=True)
    ids = ids.to(dev)
    torch.manual_seed(77)
    m, i = RandomMaskingStrategy().apply_random_mask(ids, 0.37)
    torch.manual_seed(77)
    draw = torch.rand_like(ids, dtype=torch.float)
    om, oi = so.forward_mask(ids.cpu(), 0.37, draw.cpu())
    assert torch.equal(m.cpu(), om) and torch.equal(i.cpu(), oi)


@pytest.mark.parametrize("B,S", [(1, 1), (3, 10), (5, 127), (4, 512), (64, 512), (2, 4096)])
def test_mask_fused_philox_and_per_sequence_t(dev, B, S):
    from oracle import philox
    from oracle import step_oracle as so
    from src.masking_strategy import RandomMaskingStrategy
    ids, _ = so.synthetic_batch(B, S, 30522, seed=B * 1000 + S, ragged=S > 8, dirty=S > 8)
    st = RandomMaskingStrategy(fused_rng=True, seed=0x1234567_89ABCDEF)
    st.calls = 5
    m, i = st.apply_random_mask(ids.to(dev), 0.42)
    draw = torch.from_numpy(philox.token_draws(B * S, st.seed, 5)).view(B, S)
    om, oi = so.forward_mask(ids, 0.42, draw)
    assert torch.equal(m.cpu(), om) and torch.equal(i.cpu(), oi)
    # per-sequence t ~ U(eps,1) drawn in-kernel
    m, i, t = st.apply_per_sequence_mask(ids.to(dev), eps=1e-3, offset=9)
    t_ref = torch.from_numpy(philox.seq_t(B, st.seed, 9, 1e-3))
    assert torch.equal(t.cpu(), t_ref)
    draw = torch.from_numpy(philox.token_draws(B * S, st.seed, 9)).view(B, S)
    om, oi = so.forward_mask(ids, t_ref, draw)
    assert torch.equal(m.cpu(), om) and torch.equal(i.cpu(), oi)
    # per-sequence t given as a tensor, unaligned view (scalar kernel path)
    if S > 3:
        view = ids.to(dev)[:, 1:]
        tt = torch.linspace(0.1, 0.9, B)
        d2 = torch.rand(B, S - 1)
        m, i = RandomMaskingStrategy().apply_random_mask(view, tt.to(dev), draw=d2.to(dev))
        om, oi = so.forward_mask(ids[:, 1:], tt, d2)
        assert torch.equal(m.cpu(), om) and torch.equal(i.cpu(), oi)


def test_mask_empty(dev):
    from src.masking_strategy import RandomMaskingStrategy
    m, i = RandomMaskingStrategy().apply_random_mask(torch.zeros(0, 7, dtype=torch.int64, device=dev), 0.5)
    assert m.shape == (0, 7) and i.shape == (0, 7)


# ------------------------------------------------------------------ K4 masked CE
def test_ce_reference_fixture(dev):
    from src.loss import masked_cross_entropy_loss, variational_lower_bound_loss
    g = load_golden("loss.npz")
    logits = torch.from_numpy(g["t_logits"]).to(dev).requires_grad_(True)
    targets = torch.from_numpy(g["t_targets"]).to(dev)
    mask = torch.from_numpy(g["t_mask"]).to(dev)
    mids = torch.from_numpy(g["t_masked_ids"]).to(dev)
    loss = masked_cross_entropy_loss(logits, targets, mask)
    # reference tests/test_loss.py:69-98
    assert loss.dim() == 0 and loss.item() > 0 and loss.requires_grad
    assert rel_err(loss, g["t_loss_mce"]) < FP32_TOL
    loss.backward()
    assert rel_err(logits.grad, g["t_grad_mce"]) < FP32_TOL
    empty = masked_cross_entropy_loss(logits, targets, torch.zeros_like(mask))
    assert empty.item() == 0.0 and empty.requires_grad
    # reference tests/test_loss.py:100-157
    vlb = variational_lower_bound_loss(logits, targets, mids, 103)
    assert rel_err(vlb, g["t_loss_vlb"]) < FP32_TOL
    assert abs(vlb.item() - masked_cross_entropy_loss(logits, targets, mids == 103).item()) < 1e-6
    v1 = variational_lower_bound_loss(logits, targets, mids, 103, torch.from_numpy(g["t_att"]).to(dev))
    v2 = variational_lower_bound_loss(logits, targets, mids, 103, torch.from_numpy(g["t_att2"]).to(dev))
    assert rel_err(v1, g["t_loss_vlb_att"]) < FP32_TOL and rel_err(v2, g["t_loss_vlb_att2"]) < FP32_TOL
    assert v1.item() != v2.item()


def test_ce_large_vocab_golden_and_bf16(dev):
    from src.loss import variational_lower_bound_loss
    g = load_golden("loss.npz")
    logits = torch.from_numpy(g["v_logits"]).to(dev).requires_grad_(True)
    targets = torch.from_numpy(g["v_targets"]).to(dev)
    mids = torch.from_numpy(g["v_masked_ids"]).to(dev)
    att = torch.from_numpy(g["v_att"]).to(dev)
    loss = variational_lower_bound_loss(logits, targets, mids, 103, att)
    assert rel_err(loss, g["v_loss"]) < FP32_TOL
    loss.backward()
    rows = torch.from_numpy(g["v_rows"]).to(dev)
    assert rel_err(logits.grad[rows], g["v_grad_rows"]) < FP32_TOL
    assert float(logits.grad[~rows].abs().max()) == 0.0
    lb = logits.detach().bfloat16().requires_grad_(True)
    loss_b = variational_lower_bound_loss(lb, targets, mids, 103, att)
    ref = torch.nn.functional.cross_entropy(lb.detach().float()[rows], targets[rows])
    assert rel_err(loss_b, ref) < 1e-5          # same bf16 inputs, fp32 arithmetic
    loss_b.backward()
    assert lb.grad.dtype == torch.bfloat16 and rel_err(lb.grad[rows].float(), g["v_grad_rows"]) < BF16_TOL


def test_ce_weights_and_token_normalisation(dev):
    """Paper-form options (not reference behaviour): per-sequence weight and B*L normaliser."""
    from src.loss import variational_lower_bound_loss
    torch.manual_seed(1)
    B, S, V = 3, 9, 77
    logits = torch.randn(B, S, V, device=dev, requires_grad=True)
    targets = torch.randint(0, V, (B, S), device=dev)
    mids = targets.clone()
    mids[:, 2::3] = 103
    w = torch.tensor([2.0, 0.5, 4.0], device=dev)
    loss = variational_lower_bound_loss(logits, targets, mids, 103, seq_weight=w, normalize_by="tokens")
    rows = mids == 103
    ce = torch.nn.functional.cross_entropy(logits.detach().reshape(-1, V), targets.reshape(-1), reduction="none").view(B, S)
    ref = (ce * rows * w[:, None]).sum() / (B * S)
    assert rel_err(loss, ref) < FP32_TOL
    loss.backward()
    lr = logits.detach().clone().requires_grad_(True)
    ce2 = torch.nn.functional.cross_entropy(lr.reshape(-1, V), targets.reshape(-1), reduction="none").view(B, S)
    ((ce2 * rows * w[:, None]).sum() / (B * S)).backward()
    assert rel_err(logits.grad, lr.grad) < FP32_TOL


# ------------------------------------------------------------------ row kernels
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T,d", [(30, 32), (257, 768), (64, 4096), (5, 200)])
def test_rmsnorm(dev, dtype, T, d):
    from src import ops
    torch.manual_seed(0)
    x = torch.randn(T, d, device=dev).to(dtype).requires_grad_(True)
    w = (1 + 0.1 * torch.randn(d, device=dev)).requires_grad_(True)
    gy = torch.randn(T, d, device=dev).to(dtype)
    y = ops.rms_norm(x, w, 1e-12)
    y.backward(gy)
    xr = x.detach().double().requires_grad_(True)
    wr = w.detach().double().requires_grad_(True)
    yr = wr * (xr / torch.sqrt((xr ** 2).mean(-1, keepdim=True) + 1e-12))
    yr.backward(gy.double())
    tol = FP32_TOL if dtype == torch.float32 else BF16_TOL
    assert rel_err(y, yr) < tol and rel_err(x.grad, xr.grad) < tol and rel_err(w.grad, wr.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embedding_and_swiglu(dev, dtype):
    from src import ops
    torch.manual_seed(0)
    V, d, F = 300, 64, 96
    table = torch.randn(V, d, device=dev, requires_grad=True)
    ids = torch.randint(0, V, (4, 50), device=dev)
    ids[:, ::2] = 103                      # hot row
    out = ops.embedding(ids, table, dtype, 103)
    g = torch.randn(4, 50, d, device=dev).to(dtype)
    out.backward(g)
    assert rel_err(out, table.detach()[ids].to(dtype)) == 0.0
    ref = torch.zeros(V, d, device=dev, dtype=torch.float64)
    ref.index_add_(0, ids.reshape(-1), g.double().reshape(-1, d))
    assert rel_err(table.grad, ref) < 1e-5
    ag = torch.randn(33, 2 * F, device=dev).to(dtype).requires_grad_(True)
    gh = torch.randn(33, F, device=dev).to(dtype)
    h = ops.swiglu_act(ag)
    h.backward(gh)
    agr = ag.detach().double().requires_grad_(True)
    hr = agr[:, :F] * torch.nn.functional.silu(agr[:, F:])
    hr.backward(gh.double())
    tol = FP32_TOL if dtype == torch.float32 else BF16_TOL
    assert rel_err(h, hr) < tol and rel_err(ag.grad, agr.grad) < tol


# ------------------------------------------------------------------ GEMM
def _gemm_call(a, b, ta, tb, M, N, K, out_dtype, bias=None, res=None):
    from src import _native as Nn
    c = torch.empty(M, N, dtype=out_dtype, device=a.device)
    Nn.check(Nn.lib().ldm_gemm(Nn.dtype_code(a.dtype), Nn.dtype_code(out_dtype), ta, tb, M, N, K, 1.0,
                               a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), N,
                               Nn.ptr(bias), Nn.ptr(res), N if res is not None else 0, Nn.stream()), "ldm_gemm")
    return c


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(30, 96, 32), (200, 130, 70), (128, 128, 64)])
def test_gemm_f32(dev, ta, tb, M, N, K):
    torch.manual_seed(0)
    a = torch.randn((K, M) if ta else (M, K), device=dev)
    b = torch.randn((N, K) if tb else (K, N), device=dev)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    c = _gemm_call(a, b, ta, tb, M, N, K, torch.float32, bias, res)
    A = a.double().t() if ta else a.double()
    Bm = b.double().t() if tb else b.double()
    assert rel_err(c, A @ Bm + bias.double() + res.double()) < FP32_TOL


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 768), (200, 136, 128), (1000, 30528, 64),
                                   (776, 72, 1024)])
def test_gemm_bf16_mfma(dev, ta, tb, M, N, K):
    """bf16 MFMA kernel (K % 64 == 0): exact bf16 inputs, fp32 accumulation, asymmetric data."""
    torch.manual_seed(1)
    a = torch.randn((K, M) if ta else (M, K), device=dev).bfloat16()
    b = torch.randn((N, K) if tb else (K, N), device=dev).bfloat16()
    bias = torch.randn(N, device=dev)
    A = a.double().t() if ta else a.double()
    Bm = b.double().t() if tb else b.double()
    ref = A @ Bm
    c32 = _gemm_call(a, b, ta, tb, M, N, K, torch.float32)
    assert rel_err(c32, ref) < 1e-5
    res = torch.randn(M, N, device=dev).bfloat16()
    c16 = _gemm_call(a, b, ta, tb, M, N, K, torch.bfloat16, bias, res)
    assert rel_err(c16.float(), ref + bias.double() + res.double()) < 4e-3


@pytest.mark.parametrize("ta,tb", [(1, 0), (0, 1)])
@pytest.mark.parametrize("M,N,K", [(768, 768, 4096), (2304, 768, 32768), (200, 136, 640), (64, 72, 50)])
def test_gemm_acc_split_k(dev, ta, tb, M, N, K):
    """C += A*B (weight-gradient form): K is split over workgroups with float atomics."""
    from src import _native as Nn
    torch.manual_seed(3)
    a = torch.randn((K, M) if ta else (M, K), device=dev).bfloat16()
    b = torch.randn((N, K) if tb else (K, N), device=dev).bfloat16()
    c0 = torch.randn(M, N, device=dev)
    c = c0.clone()
    Nn.check(Nn.lib().ldm_gemm_acc(1, ta, tb, M, N, K, 1.0, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0),
                                   c.data_ptr(), N, Nn.stream()), "ldm_gemm_acc")
    A = a.double().t() if ta else a.double()
    Bm = b.double().t() if tb else b.double()
    assert rel_err(c, c0.double() + A @ Bm) < 1e-5


@pytest.mark.parametrize("ta,tb", [(1, 0), (0, 1)])
@pytest.mark.parametrize("M,N,K", [(768, 768, 4096), (2304, 768, 32768), (2048, 768, 65536), (30528, 768, 2048),
                                   (200, 136, 640)])
def test_gemm_acc_rowsum(dev, ta, tb, M, N, K):
    """C += A*B and rowsum += row sums of op(A) from one kernel (weight + bias gradient)."""
    from src import _native as Nn
    torch.manual_seed(4)
    a = torch.randn((K, M) if ta else (M, K), device=dev).bfloat16()
    b = torch.randn((N, K) if tb else (K, N), device=dev).bfloat16()
    c0 = torch.randn(M, N, device=dev)
    r0 = torch.randn(M, device=dev)
    c, r = c0.clone(), r0.clone()
    Nn.check(Nn.lib().ldm_gemm_acc_rowsum(ta, tb, M, N, K, 1.0, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0),
                                          c.data_ptr(), N, r.data_ptr(), Nn.stream()), "ldm_gemm_acc_rowsum")
    A = a.double().t() if ta else a.double()
    Bm = b.double().t() if tb else b.double()
    assert rel_err(c, c0.double() + A @ Bm) < 1e-5
    assert rel_err(r, r0.double() + A.sum(dim=1)) < 1e-5


@pytest.mark.parametrize("M,N,K", [(256, 256, 512), (256, 512, 576), (768, 2304, 8192), (2304, 768, 16384),
                                   (2048, 768, 65536), (1024, 768, 3200)])
@pytest.mark.parametrize("rowsum", [False, True])
def test_gemm_acc_elementwise(dev, M, N, K, rowsum):
    """The weight-gradient product (A [K, M], B [K, N], fp32 accumulate) on full 256 x 256 tiles: one work item
    and many, uneven splits of the reduction, padded leading dimensions, row sums.  Every element is checked (a
    misplaced fragment must not hide in a norm) against the fp64 product of the bf16 inputs."""
    from src import _native as Nn
    torch.manual_seed(12)
    lda, ldb, ldc = M + 64, N + 8, N + 12
    a = torch.randn(K, lda, device=dev).bfloat16()
    b = torch.randn(K, ldb, device=dev).bfloat16()
    c0 = torch.randn(M, ldc, device=dev)
    r0 = torch.randn(M, device=dev)
    c, r = c0.clone(), r0.clone()
    if rowsum:
        Nn.check(Nn.lib().ldm_gemm_acc_rowsum(1, 0, M, N, K, 1.0, a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), ldc,
                                              r.data_ptr(), Nn.stream()), "ldm_gemm_acc_rowsum")
    else:
        Nn.check(Nn.lib().ldm_gemm_acc(1, 1, 0, M, N, K, 1.0, a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), ldc,
                                       Nn.stream()), "ldm_gemm_acc")
    A, Bm = a[:, :M].double().t(), b[:, :N].double()
    ref = c0[:, :N].double() + A @ Bm
    err = (c[:, :N].double() - ref).abs()
    bound = 1e-4 * math.sqrt(K) + 1e-5 * ref.abs()      # fp32 accumulation of K products of O(1) terms
    assert bool((err <= bound).all()), f"max excess {float((err - bound).max())}"
    assert torch.equal(c[:, N:], c0[:, N:])               # nothing written beyond the N columns
    if rowsum:
        rr = r0.double() + A.sum(dim=1)
        assert bool(((r.double() - rr).abs() <= 1e-4 * math.sqrt(K) + 1e-5 * rr.abs()).all())
    else:
        assert torch.equal(r, r0)


def test_gemm_bf16_fallback_shapes(dev):
    """Shapes the MFMA kernel does not take (K % 64 != 0) run on the FMA kernel."""
    torch.manual_seed(2)
    a = torch.randn(30, 32, device=dev).bfloat16()
    b = torch.randn(96, 32, device=dev).bfloat16()
    c = _gemm_call(a, b, 0, 1, 30, 96, 32, torch.float32)
    assert rel_err(c, a.double() @ b.double().t()) < 1e-5


def test_colsum(dev):
    from src import _native as Nn
    x = torch.randn(1000, 130, device=dev)
    out = torch.empty(130, device=dev)
    ws = torch.empty(Nn.lib().ldm_colsum_ws_floats(1000, 130), device=dev)
    Nn.check(Nn.lib().ldm_colsum(x.data_ptr(), 0, 1000, 130, 130, out.data_ptr(), ws.data_ptr(), Nn.stream()), "colsum")
    assert rel_err(out, x.double().sum(0)) < 1e-5


# ------------------------------------------------------------------ attention
def _attn_ref(qkv, key_valid, B, S, H, hd, keep=None, keep_scale=1.0, add=None):
    """Reference arithmetic of src/transformer.py:246-285 in fp64 (RoPE + scale included); ``keep``
    [B,H,S,S] bool with ``keep_scale`` = 1/(1-p) is the dropout of line 278 with a given mask; ``add``
    is the attn_mask of lines 267-268."""
    from oracle import step_oracle as so
    D = H * hd
    x = qkv.double().view(B, S, 3, H, hd)
    q, k, v = x[:, :, 0], x[:, :, 1], x[:, :, 2]
    cos, sin = so.rope_tables(S, hd)
    cos = cos.double().to(qkv.device)[None, :, None, :]
    sin = sin.double().to(qkv.device)[None, :, None, :]
    q = q * cos + so._rot_half(q) * sin
    k = k * cos + so._rot_half(k) * sin
    q, k, v = (t.transpose(1, 2) for t in (q, k, v))
    s = (q / math.sqrt(hd)) @ k.transpose(-2, -1)
    if add is not None:
        s = s + add.double()
    if key_valid is not None:
        s = s.masked_fill(~key_valid.bool()[:, None, None, :], float("-inf"))
    pr = torch.softmax(s, -1)
    if keep is not None:
        pr = torch.where(keep, pr * keep_scale, torch.zeros_like(pr))
    o = pr @ v
    return o.transpose(1, 2).reshape(B, S, D)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,S,H,hd,ragged", [(3, 10, 4, 8, True), (2, 48, 2, 96, True), (2, 130, 2, 64, True),
                                             (1, 256, 2, 128, False), (2, 200, 3, 32, True), (1, 512, 8, 96, False),
                                             (2, 200, 2, 128, True), (2, 130, 3, 96, True), (1, 3, 1, 96, False)])
def test_attention_fwd_bwd(dev, dtype, B, S, H, hd, ragged):
    from src import ops
    from src.transformer import RotaryPositionEmbedding
    torch.manual_seed(0)
    D = H * hd
    qkv = (torch.randn(B, S, 3 * D, device=dev) * 1.5).to(dtype).requires_grad_(True)
    key_valid = None
    if ragged:
        lens = torch.randint(max(2, S // 2), S + 1, (B,))
        lens[0] = S
        key_valid = (torch.arange(S)[None, :] < lens[:, None]).to(torch.uint8).to(dev)
    rot = RotaryPositionEmbedding(hd)
    cos, sin = rot.half_tables(qkv, S)
    go = torch.randn(B, S, D, device=dev).to(dtype)
    o = ops.rope_attention(qkv * 1.0, key_valid, B, S, H, hd, cos, sin)
    o.backward(go)
    qr = qkv.detach().clone().requires_grad_(True)
    orf = _attn_ref(qr, key_valid, B, S, H, hd)
    orf.backward(go.double())
    tol = FP32_TOL if dtype == torch.float32 else BF16_TOL
    assert rel_err(o, orf) < tol, "forward"
    gq, gr = qkv.grad.float().view(B, S, 3, D), qr.grad.view(B, S, 3, D)
    for i, nm in enumerate("qkv"):
        assert rel_err(gq[:, :, i], gr[:, :, i]) < tol, f"d{nm}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", ["SS", "B1SS", "BHSS", "1H1S"])
def test_attention_additive_mask(dev, dtype, shape):
    """scores + attn_mask (src/transformer.py:267-268) in every broadcast form, with the key mask on
    top, forward and backward; a causal -inf part and a finite bias part."""
    from src import ops
    from src.transformer import RotaryPositionEmbedding
    torch.manual_seed(2)
    B, S, H, hd = 2, 70, 3, 32
    D = H * hd
    qkv = (torch.randn(B, S, 3 * D, device=dev) * 1.5).to(dtype).requires_grad_(True)
    lens = torch.tensor([S, S - 9])
    key_valid = (torch.arange(S)[None, :] < lens[:, None]).to(torch.uint8).to(dev)
    causal = torch.full((S, S), float("-inf"), device=dev).triu(1)
    full = {"SS": (S, S), "B1SS": (B, 1, S, S), "BHSS": (B, H, S, S), "1H1S": (1, H, 1, S)}[shape]
    m = torch.randn(*full, device=dev)
    if shape != "1H1S":
        m = m + causal
    cos, sin = RotaryPositionEmbedding(hd).half_tables(qkv, S)
    go = torch.randn(B, S, D, device=dev).to(dtype)
    o = ops.rope_attention(qkv * 1.0, key_valid, B, S, H, hd, cos, sin, attn_mask=m)
    assert o.dtype == dtype
    o.backward(go)
    qr = qkv.detach().clone().requires_grad_(True)
    orf = _attn_ref(qr, key_valid, B, S, H, hd, add=m)
    orf.backward(go.double())
    tol = FP32_TOL if dtype == torch.float32 else BF16_TOL
    assert rel_err(o, orf) < tol, "forward"
    assert rel_err(qkv.grad.float(), qr.grad) < tol, "dqkv"
    with pytest.raises(NotImplementedError):
        ops.rope_attention(qkv * 1.0, key_valid, B, S, H, hd, cos, sin, attn_mask=m.clone().requires_grad_(True))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_module_causal_mask_hides_later_tokens(dev, dtype):
    """RotaryMultiheadAttention(x, x, x, attn_mask=causal): changing the last token leaves every
    earlier output row as it was (dropout on the probabilities active)."""
    from src.transformer import RotaryMultiheadAttention
    torch.manual_seed(4)
    B, S, d = 2, 96, 128
    att = RotaryMultiheadAttention(d, 2, dropout=0.1).to(dev)
    att.compute_dtype = dtype
    att.train()
    x = torch.randn(B, S, d, device=dev).to(dtype)
    x2 = x.clone()
    x2[:, -1] += 1.0
    causal = torch.full((S, S), float("-inf"), device=dev).triu(1)
    torch.manual_seed(9)
    y = att(x, x, x, attn_mask=causal)
    torch.manual_seed(9)
    y2 = att(x2, x2, x2, attn_mask=causal)
    assert torch.equal(y[:, :-1], y2[:, :-1]) and not torch.equal(y[:, -1], y2[:, -1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,S,H,hd,ragged,p", [(3, 10, 4, 8, True, 0.25), (2, 48, 2, 96, True, 0.1), (2, 130, 2, 64, True, 0.1),
                                               (1, 256, 2, 128, False, 0.3), (2, 200, 3, 32, True, 0.5),
                                               (1, 512, 8, 96, False, 0.1), (2, 200, 2, 128, True, 0.1),
                                               (2, 130, 3, 96, True, 0.9)])
def test_attention_dropout_fwd_bwd(dev, dtype, B, S, H, hd, ragged, p):
    """Dropout on the attention probabilities (src/transformer.py:278): output and gradients equal a
    plain attention that multiplies the softmax by the SAME mask, rebuilt on the host from the seed
    by oracle/dropout_oracle.py; the kept fraction is 1 - p.  (The reference's own masks come from
    torch's generator: parity with them is statistical only.)"""
    from oracle import dropout_oracle as do
    from src import ops
    from src.transformer import RotaryPositionEmbedding
    torch.manual_seed(0)
    D = H * hd
    seed = 0x1234_5678_9ABC_DEF0 ^ (S * 7919 + hd)
    qkv = (torch.randn(B, S, 3 * D, device=dev) * 1.5).to(dtype).requires_grad_(True)
    key_valid = None
    if ragged:
        lens = torch.randint(max(2, S // 2), S + 1, (B,))
        lens[0] = S
        key_valid = (torch.arange(S)[None, :] < lens[:, None]).to(torch.uint8).to(dev)
    cos, sin = RotaryPositionEmbedding(hd).half_tables(qkv, S)
    go = torch.randn(B, S, D, device=dev).to(dtype)
    for fused in ((True, False) if dtype == torch.bfloat16 and hd in (32, 64, 96, 128) else (True,)):
        ops.FUSED_UNROPE = fused
        try:
            qkv.grad = None
            o = ops.rope_attention(qkv * 1.0, key_valid, B, S, H, hd, cos, sin, dropout_p=p, seed=seed)
            o.backward(go)
        finally:
            ops.FUSED_UNROPE = True
        keep_np = do.keep_mask(p, seed, B * H, S, S)
        assert abs(float(keep_np.mean()) - (1.0 - p)) < 4.0 * math.sqrt(p * (1 - p) / keep_np.size) + 1e-3
        keep = torch.from_numpy(keep_np).to(dev).view(B, H, S, S)
        qr = qkv.detach().clone().requires_grad_(True)
        orf = _attn_ref(qr, key_valid, B, S, H, hd, keep=keep, keep_scale=do.scale(p))
        orf.backward(go.double())
        tol = FP32_TOL if dtype == torch.float32 else BF16_TOL
        assert rel_err(o, orf) < tol, "forward"
        gq, gr = qkv.grad.float().view(B, S, 3, D), qr.grad.view(B, S, 3, D)
        for i, nm in enumerate("qkv"):
            assert rel_err(gq[:, :, i], gr[:, :, i]) < tol, f"d{nm} fused={fused}"
    # a different seed gives a different mask, the same seed the same output
    o2 = ops.rope_attention(qkv.detach() * 1.0, key_valid, B, S, H, hd, cos, sin, dropout_p=p, seed=seed)
    o3 = ops.rope_attention(qkv.detach() * 1.0, key_valid, B, S, H, hd, cos, sin, dropout_p=p, seed=seed + 1)
    assert torch.equal(o2, o.detach()) and not torch.equal(o3, o2)


def test_attention_dropout_is_unbiased(dev):
    """nn.Dropout semantics (src/transformer.py:278): kept probabilities are scaled by 1/(1-p), so the
    mean output over many masks tends to the dropout-free output."""
    from src import ops
    from src.transformer import RotaryPositionEmbedding
    torch.manual_seed(0)
    B, S, H, hd, p, n = 1, 128, 2, 96, 0.2, 256
    qkv = torch.randn(B, S, 3 * H * hd, device=dev)
    cos, sin = RotaryPositionEmbedding(hd).half_tables(qkv, S)
    base = ops.rope_attention(qkv * 1.0, None, B, S, H, hd, cos, sin)
    acc = torch.zeros_like(base)
    for i in range(n):
        acc += ops.rope_attention(qkv * 1.0, None, B, S, H, hd, cos, sin, dropout_p=p, seed=1000 + i)
    one = ops.rope_attention(qkv * 1.0, None, B, S, H, hd, cos, sin, dropout_p=p, seed=1000)
    err_one, err_mean = rel_err(one, base), rel_err(acc / n, base)
    assert err_one > 0.1 and err_mean < 2.0 * err_one / math.sqrt(n), (err_one, err_mean)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,d,p", [(7, 768, 0.1), (1000, 32, 0.5), (33, 13, 0.3), (4096, 2048, 0.1), (5, 8, 0.0)])
def test_dropout_add(dev, dtype, rows, d, p):
    """res + keep*y/(1-p) and its backward against the same mask rebuilt by oracle/dropout_oracle.py
    (src/transformer.py:352,361: src + self.dropout(...)); keep rate 1 - p."""
    from oracle import dropout_oracle as do
    from src import ops
    torch.manual_seed(rows)
    seed = 0x0BAD_5EED_0000_0000 + rows * d
    y = torch.randn(rows, d, device=dev).to(dtype).requires_grad_(True)
    res = torch.randn(rows, d, device=dev).to(dtype).requires_grad_(True)
    g = torch.randn(rows, d, device=dev).to(dtype)
    if p == 0.0:
        out = ops.dropout_add(y, res, p)
        assert torch.equal(out, res + y)
        return
    out = ops._DropoutAdd.apply(y, res, p, seed)
    out.backward(g)
    keep_np = do.keep_mask_rows(p, seed, rows, d)
    assert abs(float(keep_np.mean()) - (1.0 - p)) < 4.0 * math.sqrt(p * (1 - p) / keep_np.size) + 1e-3
    keep = torch.from_numpy(keep_np).to(dev)
    sc = do.scale(p)
    want = res.detach().double() + torch.where(keep, y.detach().double() * sc, torch.zeros_like(y, dtype=torch.float64))
    tol = 1e-6 if dtype == torch.float32 else 4e-3
    assert rel_err(out, want) < tol
    assert rel_err(y.grad, torch.where(keep, g.double() * sc, torch.zeros_like(g, dtype=torch.float64))) < tol
    assert torch.equal(res.grad, g)
    assert float((out.detach()[~keep] - res.detach()[~keep]).abs().max()) == 0.0     # dropped: exactly the residual


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_res", [True, False])
@pytest.mark.parametrize("T,d,p", [(257, 768, 0.1), (64, 4096, 0.3), (1000, 64, 0.5), (4099, 2048, 0.1), (33, 8, 0.1)])
def test_rmsnorm_bwd_with_dropout_gradient(dev, dtype, with_res, T, d, p):
    """ldm_rmsnorm_bwd_dropgrad == ldm_rmsnorm_bwd followed by the masking pass ldm_dropout_add(dx, no residual),
    bit for bit: dx, dw, and the masked copy (the backward of `src + self.dropout(...)` feeding the next norm,
    src/transformer.py:352-361); widths the vector kernels do not take are declined."""
    from src import _native as Nn
    lib = Nn.lib()
    torch.manual_seed(T + d)
    seed = 0x0DD0_0000_0000_0000 + T * 131 + d
    code = Nn.dtype_code(dtype)
    x = torch.randn(T, d, device=dev).to(dtype)
    dy = torch.randn(T, d, device=dev).to(dtype)
    dres = torch.randn(T, d, device=dev).to(dtype) if with_res else None
    w = torch.randn(d, device=dev)
    rstd = torch.rsqrt(x.float().pow(2).mean(-1) + 1e-6)
    ws = torch.empty(lib.ldm_rmsnorm_bwd_ws_floats(T, d), device=dev)
    dx1, dw1 = torch.empty_like(x), torch.empty(d, device=dev)
    Nn.check(lib.ldm_rmsnorm_bwd(x.data_ptr(), w.data_ptr(), rstd.data_ptr(), dy.data_ptr(), Nn.ptr(dres), T, d, code,
                                 dx1.data_ptr(), dw1.data_ptr(), ws.data_ptr(), Nn.stream()), "ldm_rmsnorm_bwd")
    m1 = torch.empty_like(x)
    Nn.check(lib.ldm_dropout_add(dx1.data_ptr(), None, m1.data_ptr(), T, d, code, p, seed, Nn.stream()), "ldm_dropout_add")
    dx2, dw2, m2 = torch.full_like(x, 7.0), torch.empty(d, device=dev), torch.full_like(x, 7.0)
    Nn.check(lib.ldm_rmsnorm_bwd_dropgrad(x.data_ptr(), w.data_ptr(), rstd.data_ptr(), dy.data_ptr(), Nn.ptr(dres), T, d,
                                          code, dx2.data_ptr(), m2.data_ptr(), p, seed, dw2.data_ptr(), ws.data_ptr(),
                                          Nn.stream()), "ldm_rmsnorm_bwd_dropgrad")
    assert torch.equal(dx1, dx2) and torch.equal(dw1, dw2)
    assert torch.equal(m1, m2)
    kept = float((m2 != 0).float().mean())
    assert abs(kept - (1.0 - p)) < 4.0 * math.sqrt(p * (1 - p) / m2.numel()) + 2e-3
    # a width that is not a multiple of 8 goes to the two-call form
    rc = lib.ldm_rmsnorm_bwd_dropgrad(x.data_ptr(), w.data_ptr(), rstd.data_ptr(), dy.data_ptr(), None, 4, 12, code,
                                      dx2.data_ptr(), m2.data_ptr(), p, seed, dw2.data_ptr(), ws.data_ptr(), Nn.stream())
    assert rc in (-2, -3)


@pytest.mark.parametrize("with_res", [True, False])
@pytest.mark.parametrize("T,d", [(257, 768), (64, 4096), (1000, 64), (4099, 2048), (33, 8)])
def test_rmsnorm_fp32_stream_kernels(dev, with_res, T, d):
    """The norm between an fp32 residual stream and bf16 projections (what src/transformer.py:54-58 is under
    bf16 autocast): ldm_rmsnorm_fwd_f32_bf16 == the fp32 kernel's result rounded to bf16, and
    ldm_rmsnorm_bwd_f32_bf16 == the fp32 backward kernel on the up-cast gradient, bit for bit; the autograd
    wrapper uses them for fp32 input with a bf16 result."""
    from src import _native as Nn, ops
    lib = Nn.lib()
    torch.manual_seed(T * 3 + d)
    x = torch.randn(T, d, device=dev) * 3.0
    w = torch.randn(d, device=dev)
    dy = torch.randn(T, d, device=dev).bfloat16()
    dres = torch.randn(T, d, device=dev) if with_res else None
    y32, rstd32 = torch.empty_like(x), torch.empty(T, device=dev)
    Nn.check(lib.ldm_rmsnorm_fwd(x.data_ptr(), w.data_ptr(), T, d, 1e-6, Nn.F32, y32.data_ptr(), rstd32.data_ptr(), Nn.stream()), "fwd")
    y16, rstd = torch.empty(T, d, device=dev, dtype=torch.bfloat16), torch.empty(T, device=dev)
    Nn.check(lib.ldm_rmsnorm_fwd_f32_bf16(x.data_ptr(), w.data_ptr(), T, d, 1e-6, y16.data_ptr(), rstd.data_ptr(), Nn.stream()), "fwd mixed")
    assert torch.equal(rstd, rstd32) and torch.equal(y16, y32.bfloat16())
    ws = torch.empty(lib.ldm_rmsnorm_bwd_ws_floats(T, d), device=dev)
    dx32, dw32 = torch.empty_like(x), torch.empty(d, device=dev)
    g32 = dy.float()
    Nn.check(lib.ldm_rmsnorm_bwd(x.data_ptr(), w.data_ptr(), rstd.data_ptr(), g32.data_ptr(), Nn.ptr(dres), T, d, Nn.F32,
                                 dx32.data_ptr(), dw32.data_ptr(), ws.data_ptr(), Nn.stream()), "bwd")
    dx, dw = torch.full_like(x, 7.0), torch.empty(d, device=dev)
    Nn.check(lib.ldm_rmsnorm_bwd_f32_bf16(x.data_ptr(), w.data_ptr(), rstd.data_ptr(), dy.data_ptr(), Nn.ptr(dres), T, d,
                                          dx.data_ptr(), dw.data_ptr(), ws.data_ptr(), Nn.stream()), "bwd mixed")
    assert torch.equal(dx, dx32) and torch.equal(dw, dw32)
    # the autograd wrapper: fp32 in, bf16 out, fp32 gradient of the input
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    if with_res:
        y, stream = ops.rms_norm_with_residual(xr, wr, 1e-6, torch.bfloat16)
        (y.float() * dy.float()).sum().backward(retain_graph=True)
        assert stream.dtype == torch.float32
    else:
        y = ops.rms_norm(xr, wr, 1e-6, torch.bfloat16)
        y.backward(dy)
    assert y.dtype == torch.bfloat16 and torch.equal(y, y16) and xr.grad.dtype == torch.float32
    if not with_res:
        assert torch.equal(xr.grad, dx32) and torch.equal(wr.grad, dw32)
    # widths the 16-byte-vector kernels do not take are declined by the entry point and composed by the wrapper
    assert lib.ldm_rmsnorm_fwd_f32_bf16(x.data_ptr(), w.data_ptr(), 4, 12, 1e-6, y16.data_ptr(), rstd.data_ptr(), Nn.stream()) in (-2, -3)
    x12 = torch.randn(5, 12, device=dev)
    y12 = ops.rms_norm(x12, torch.ones(12, device=dev), 1e-6, torch.bfloat16)
    assert y12.dtype == torch.bfloat16 and rel_err(y12.float(), x12 * torch.rsqrt(x12.pow(2).mean(-1, keepdim=True) + 1e-6)) < 4e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layer_dropout_backward_handoff_changes_nothing(dev, dtype):
    """Two encoder layers + final norm in training mode with dropout 0.1: the masked gradient that the norm's
    backward leaves for the projection behind the residual dropout gives the gradients of the two-pass form,
    bit for bit, under the same torch seed -- and the hand-off is what ran (every residual dropout took it).
    A tensor with a second consumer gets its gradient summed by autograd, so the projection masks it itself."""
    from src import ops
    from src.transformer import TransformerEncoderLayer, RMSNorm
    torch.manual_seed(5)
    d = 256
    layers = torch.nn.ModuleList([TransformerEncoderLayer(d, 4, 512, dropout=0.1, layer_norm_eps=1e-6) for _ in range(2)]).to(dev)
    norm = RMSNorm(d, eps=1e-6).to(dev)
    for m in layers.modules():
        if hasattr(m, "compute_dtype"):
            m.compute_dtype = dtype
    layers.train()
    x0 = torch.randn(2, 128, d, device=dev).to(dtype)
    gout = torch.randn(2, 128, d, device=dev).to(dtype)

    def run(fused, tap=False):
        ops.FUSED_DROP_GRAD = fused
        for prm in list(layers.parameters()) + list(norm.parameters()):
            prm.grad = None
        torch.manual_seed(99)
        x = x0.clone().requires_grad_(True)
        h = x
        extra = 0.0
        for layer in layers:
            h = layer(h)
            if tap:
                extra = extra + h.float().sum() * 1e-3      # a second consumer of the layer's output
        y = norm(h)
        ((y * gout).float().sum() + extra).backward()
        return x.grad.clone(), [prm.grad.clone() for prm in list(layers.parameters()) + list(norm.parameters())]

    taken = []
    orig_take = ops._Drop.take

    def counting_take(self, gy, T, n):
        r = orig_take(self, gy, T, n)
        taken.append(r is not None)
        return r
    try:
        ops._Drop.take = counting_take
        gx_f, gp_f = run(True)
        assert taken == [True] * 4
        gx_t, gp_t = run(False)
        assert torch.equal(gx_f, gx_t)
        for a, b in zip(gp_f, gp_t):
            assert torch.equal(a, b)
        del taken[:]
        gx_f2, gp_f2 = run(True, tap=True)
        assert taken.count(False) == 2 and taken.count(True) == 2      # each layer's output has two consumers
        gx_t2, gp_t2 = run(False, tap=True)
        assert torch.equal(gx_f2, gx_t2)
        for a, b in zip(gp_f2, gp_t2):
            assert torch.equal(a, b)
    finally:
        ops._Drop.take = orig_take
        ops.FUSED_DROP_GRAD = True


@pytest.mark.parametrize("M,N,K,p,with_bias", [(512, 768, 768, 0.1, True), (8192, 2304, 256, 0.3, True),
                                               (1024, 256, 1024, 0.5, False)])
def test_gemm_dropout_residual(dev, M, N, K, p, with_bias):
    """ldm_gemm_dropout_residual: R + keep * (A W^T + b) / (1 - p) from one GEMM, with the mask of
    ldm_dropout_add for the same p and seed (src/transformer.py:352,361 with the projection in front);
    dropped elements are the residual exactly; shapes it does not take are declined."""
    from oracle import dropout_oracle as do
    from src import _native as Nn
    torch.manual_seed(M + N)
    seed = 0x0123_4567_0000_0000 + M * 31 + N
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device=dev) if with_bias else None
    r = torch.randn(M, N, device=dev).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    lib = Nn.lib()
    Nn.check(lib.ldm_gemm_dropout_residual(M, N, K, a.data_ptr(), K, w.data_ptr(), K, Nn.ptr(b), r.data_ptr(), N,
                                           c.data_ptr(), N, p, seed, Nn.stream()), "ldm_gemm_dropout_residual")
    keep = torch.from_numpy(do.keep_mask_rows(p, seed, M, N)).to(dev)
    y = a.double() @ w.double().t() + (b.double() if with_bias else 0.0)
    want = r.double() + torch.where(keep, y * do.scale(p), torch.zeros_like(y))
    assert rel_err(c.float(), want) < 4e-3
    assert float((c[~keep].float() - r[~keep].float()).abs().max()) == 0.0
    # the same mask as the two-call form
    y16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    Nn.check(lib.ldm_gemm(Nn.BF16, Nn.BF16, 0, 1, M, N, K, 1.0, a.data_ptr(), K, w.data_ptr(), K, y16.data_ptr(), N,
                          Nn.ptr(b), None, 0, Nn.stream()), "ldm_gemm")
    two = torch.empty_like(y16)
    Nn.check(lib.ldm_dropout_add(y16.data_ptr(), r.data_ptr(), two.data_ptr(), M, N, Nn.BF16, p, seed, Nn.stream()), "ldm_dropout_add")
    assert torch.equal(two == r, c == r) or float(((two == r) != (c == r)).float().mean()) < 1e-4   # (a kept value may round onto the residual)
    assert rel_err(c.float(), two.float()) < 4e-3
    # declined: rows not a multiple of 256
    rc = lib.ldm_gemm_dropout_residual(M - 8, N, K, a.data_ptr(), K, w.data_ptr(), K, Nn.ptr(b), r.data_ptr(), N,
                                       c.data_ptr(), N, p, seed, Nn.stream())
    assert rc in (-2, -3)


@pytest.mark.parametrize("B,S,H,hd", [(2, 130, 2, 64), (1, 512, 8, 96), (2, 200, 3, 32), (1, 256, 2, 128)])
def test_attention_bwd_fused_unrope_matches_two_pass(dev, B, S, H, hd):
    """The rotary backward fused into the attention-backward stores gives what the separate
    ldm_rope_qk(inverse) pass gives (the fused form rounds to bf16 once instead of twice)."""
    from src import ops
    from src.transformer import RotaryPositionEmbedding
    torch.manual_seed(1)
    D = H * hd
    base = (torch.randn(B, S, 3 * D, device=dev) * 1.5).bfloat16()
    cos, sin = RotaryPositionEmbedding(hd).half_tables(base, S)
    go = torch.randn(B, S, D, device=dev).bfloat16()
    grads = {}
    for fused in (True, False):
        ops.FUSED_UNROPE = fused
        try:
            x = base.clone().requires_grad_(True)
            ops.rope_attention(x * 1.0, None, B, S, H, hd, cos, sin).backward(go)
        finally:
            ops.FUSED_UNROPE = True
        grads[fused] = x.grad.float()
    assert rel_err(grads[True], grads[False]) < 6e-3
    assert torch.equal(grads[True].view(B, S, 3, D)[:, :, 2], grads[False].view(B, S, 3, D)[:, :, 2])   # dv untouched


@pytest.mark.parametrize("T", [1, 63, 1024, 1250, 65536 + 17, 262144, 300001])
def test_ce_compact_rows(dev, T):
    """rows[0..count) = ascending indices where masked id == mask token and attention == 1."""
    from src import _native as Nn
    torch.manual_seed(T)
    ids = torch.where(torch.rand(T, device=dev) < 0.4, 103, 2000).to(torch.int64)
    att = (torch.rand(T, device=dev) < 0.8).to(torch.uint8)
    rows = torch.full((T,), -1, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    Nn.check(Nn.lib().ldm_ce_compact_rows(T, 1, ids.data_ptr(), 103, None, att.data_ptr(), rows.data_ptr(),
                                          cnt.data_ptr(), Nn.stream()), "ldm_ce_compact_rows")
    want = torch.nonzero((ids == 103) & (att != 0)).flatten().to(torch.int32)
    assert int(cnt.item()) == want.numel()
    assert torch.equal(rows[:want.numel()], want)
    assert bool((rows[want.numel():] == -1).all())


@pytest.mark.parametrize("tb", [1, 0])
@pytest.mark.parametrize("M,N,K", [(8192, 2304, 256), (8192, 2560, 96), (8192, 2304, 128), (10240, 2176, 160),
                                   (16384 + 40, 1280 + 8, 192)])
def test_gemm_persistent_many_tiles(dev, tb, M, N, K):
    """More 256x256 tiles than CUs: every workgroup walks several tiles (next tile's operands fetched
    before the epilogue, store-aware waits, ragged last tiles), with and without bias / residual."""
    torch.manual_seed(5)
    a = torch.randn(M, K, device=dev).bfloat16()
    b = torch.randn((N, K) if tb else (K, N), device=dev).bfloat16()
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev).bfloat16()
    ref = a.double() @ (b.double().t() if tb else b.double())
    c = _gemm_call(a, b, 0, tb, M, N, K, torch.bfloat16)
    assert rel_err(c.float(), ref) < 4e-3
    c = _gemm_call(a, b, 0, tb, M, N, K, torch.bfloat16, bias)
    assert rel_err(c.float(), ref + bias.double()) < 4e-3
    c = _gemm_call(a, b, 0, tb, M, N, K, torch.bfloat16, bias, res)
    assert rel_err(c.float(), ref + bias.double() + res.double()) < 4e-3
    c = _gemm_call(a, b, 0, tb, M, N, K, torch.bfloat16, None, res)     # residual without bias
    assert rel_err(c.float(), ref + res.double()) < 4e-3
    c = _gemm_call(a, b, 0, tb, M, N, K, torch.float32, bias)
    assert rel_err(c, ref + bias.double()) < 1e-5


@pytest.mark.parametrize("M,N,K", [(512, 512, 320), (256, 256, 2304), (2048, 768, 1024), (8192, 2304, 768),
                                   (66 * 256, 512, 384)])
@pytest.mark.parametrize("bias,res", [(False, False), (True, False), (True, True), (False, True)])
def test_gemm_four_wave(dev, M, N, K, bias, res):
    """Shapes the four-wave kernel takes (K-contiguous operands, full 256 x 256 tiles, K % 64 == 0, K >= 320):
    one item per workgroup, several items per workgroup (deferred stores between items, the last item's flush),
    more tiles than CUs with an uneven last round, padded leading dimensions, bias and a bf16 residual with the
    output's row stride (added before the rounding).  Checked element by element (a wrong tile or a lost store
    must not hide in a norm) against the fp64 product of the bf16 inputs."""
    from src import _native as Nn
    torch.manual_seed(11)
    lda, ldb, ldc = K + 64, K + 8, N + 24
    a = torch.randn(M, lda, device=dev).bfloat16()
    b = torch.randn(N, ldb, device=dev).bfloat16()
    bv = torch.randn(N, device=dev) if bias else None
    c = torch.full((M, ldc), float("nan"), dtype=torch.bfloat16, device=dev)
    r = (torch.randn(M, ldc, device=dev) * 3).bfloat16() if res else None
    Nn.check(Nn.lib().ldm_gemm(Nn.BF16, Nn.BF16, 0, 1, M, N, K, 1.0, a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(),
                               ldc, Nn.ptr(bv), Nn.ptr(r), ldc if res else 0, Nn.stream()), "ldm_gemm")
    ref = a[:, :K].double() @ b[:, :K].double().t()
    if bias:
        ref = ref + bv.double()
    if res:
        ref = ref + r[:, :N].double()
    got = c[:, :N].double()
    err = (got - ref).abs()
    bound = ref.abs() * 2.0 ** -8 + 1e-2      # bf16 rounding of the result (2^-9 relative) with headroom
    assert bool((err <= bound).all()), f"max excess {float((err - bound).max())}"
    assert bool(torch.isnan(c[:, N:].float()).all())      # nothing written beyond the N columns


def test_gemm_rows_dyn_many_tiles(dev):
    """Device-side row count with more populated tiles than CUs, bias, identity and mapped rows."""
    from src import _native as Nn
    torch.manual_seed(6)
    M, N, K, cnt_v = 12288, 2304, 128, 9000
    a = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    bias = torch.randn(N, device=dev)
    cnt = torch.tensor([cnt_v], dtype=torch.int32, device=dev)
    c = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev)
    Nn.check(Nn.lib().ldm_gemm_rows_dyn(1, 1, M, N, K, a.data_ptr(), K, w.data_ptr(), K, c.data_ptr(), N,
                                        bias.data_ptr(), cnt.data_ptr(), None, Nn.stream()), "rows_dyn")
    ref = a[:cnt_v].double() @ w.double().t() + bias.double()
    assert rel_err(c[:cnt_v].float(), ref) < 4e-3
    assert bool((c[cnt_v:] == 7.0).all()), "rows past the device-side count must not be written"
    perm = torch.randperm(M, device=dev)[:cnt_v].to(torch.int32)
    c2 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    Nn.check(Nn.lib().ldm_gemm_rows_dyn(1, 1, M, N, K, a.data_ptr(), K, w.data_ptr(), K, c2.data_ptr(), N,
                                        None, cnt.data_ptr(), perm.data_ptr(), Nn.stream()), "rows_dyn map")
    want = torch.zeros(M, N, dtype=torch.float64, device=dev)
    want[perm.long()] = a[:cnt_v].double() @ w.double().t()
    assert rel_err(c2.float(), want) < 4e-3


@pytest.mark.parametrize("cnt_v", [9000, 0, 1, 12288, 8960, 12161])
def test_gemm_rows_dyn_four_wave_keeps_rows_past_the_count(dev, cnt_v):
    """The device-side row count on the four-wave kernel (K >= 320, whole tiles, no row map): whole tiles
    are computed, the rows of C at and beyond the count stay as they were (counts inside a wave's 128-row
    block, on its edge, zero, one, all)."""
    from src import _native as Nn
    torch.manual_seed(8)
    M, N, K = 12288, 768, 2048
    a = torch.randn(M, K, device=dev).bfloat16()
    a[cnt_v:] = float("nan")                       # rows past the count may be read, never used
    w = torch.randn(N, K, device=dev).bfloat16()
    bias = torch.randn(N, device=dev)
    cnt = torch.tensor([cnt_v], dtype=torch.int32, device=dev)
    c = torch.full((M + 256, N), 7.0, dtype=torch.bfloat16, device=dev)
    Nn.check(Nn.lib().ldm_gemm_rows_dyn(1, 1, M, N, K, a.data_ptr(), K, w.data_ptr(), K, c.data_ptr(), N,
                                        bias.data_ptr(), cnt.data_ptr(), None, Nn.stream()), "rows_dyn")
    if cnt_v:
        ref = a[:cnt_v].double() @ w.double().t() + bias.double()
        assert rel_err(c[:cnt_v].float(), ref) < 4e-3
    assert bool((c[cnt_v:] == 7.0).all()), "rows past the device-side count must not be written"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scatter_rows(dev, dtype):
    from src import _native as Nn
    torch.manual_seed(9)
    T, d, cnt_v = 5000, 768, 3111
    x = torch.randn(4096, d, device=dev).to(dtype)
    rows = torch.randperm(T, device=dev)[:4096].to(torch.int32)
    cnt = torch.tensor([cnt_v], dtype=torch.int32, device=dev)
    out = torch.full((T, d), 3.0, device=dev).to(dtype)
    Nn.check(Nn.lib().ldm_scatter_rows(x.data_ptr(), Nn.dtype_code(dtype), d, rows.data_ptr(), cnt.data_ptr(), 4096, T, d,
                                       out.data_ptr(), d, Nn.stream()), "ldm_scatter_rows")
    want = torch.full((T, d), 3.0, device=dev).to(dtype)
    want[rows[:cnt_v].long()] = x[:cnt_v]
    assert torch.equal(out, want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("rows,V,count", [(128, 30522, 100), (64, 2051, 64), (256, 32000, 0), (64, 40000, 37),
                                          (64, 126464, 3)])
def test_ce_fwd_grad_one_pass(dev, dtype, rows, V, count, weighted):
    """ldm_ce_fwd_grad (loss and gradient from one read of each compact row) against
    F.cross_entropy + autograd on the same rows (src/loss.py:22), incl. a vocabulary that is not a
    multiple of 8, zero participating rows, the zeroed rounding rows, and the 1024-thread form."""
    import torch.nn.functional as F
    from src import _native as N
    lib = N.lib()
    torch.manual_seed(rows + V)
    ld = (V + 63) // 64 * 64
    x = (torch.randn(rows, ld, device=dev) * 2).to(dtype)
    x[:, V:] = 0
    ref_in = x[:count, :V].float().clone().requires_grad_(True)
    tg = torch.randint(0, V, (rows,), device=dev, dtype=torch.int64)
    cnt = torch.tensor([count], dtype=torch.int32, device=dev)
    lse = torch.empty(rows, device=dev)
    tmp = torch.empty(rows, device=dev)
    out = torch.full((2,), -1.0, device=dev)
    sentinel = x[(count + 63) // 64 * 64:].clone()
    # weighted: per-row weights and the fixed normaliser of the paper-form objective (sum_i w_i CE_i / norm_T)
    wr = (0.5 + 4 * torch.rand(rows, device=dev)) if weighted else None
    norm_mode, norm_T = (1, 777.0) if weighted else (0, 0.0)
    N.check(lib.ldm_ce_fwd_grad(x.data_ptr(), N.dtype_code(dtype), ld, rows, V, tg.data_ptr(), cnt.data_ptr(), norm_mode,
                                norm_T, N.ptr(wr), lse.data_ptr(), tmp.data_ptr(), out[0:1].data_ptr(),
                                out[1:2].data_ptr(), N.stream()), "ldm_ce_fwd_grad")
    assert float(out[1]) == count
    if count == 0:
        assert float(out[0]) == 0.0
    else:
        if weighted:
            loss = (F.cross_entropy(ref_in, tg[:count], reduction="none") * wr[:count]).sum() / norm_T
        else:
            loss = F.cross_entropy(ref_in, tg[:count])
        loss.backward()
        assert abs(float(out[0]) - float(loss)) <= 1e-5 * abs(float(loss))
        tol = FP32_TOL if dtype == torch.float32 else BF16_TOL
        assert rel_err(x[:count, :V].float(), ref_in.grad) < tol
        assert torch.allclose(lse[:count], torch.logsumexp(ref_in.detach(), -1), rtol=1e-5, atol=1e-5)
    pad_rows = (count + 63) // 64 * 64
    assert not x[:count, V:].any() and not x[count:pad_rows].any()
    assert torch.equal(x[pad_rows:], sentinel)          # rows past the rounding are not touched
    # upstream gradient != 1: scaled in place; == 1: untouched
    if count:
        before = x.clone()
        one = torch.ones(1, device=dev)
        N.check(lib.ldm_scale_rows_dev(x.data_ptr(), N.dtype_code(dtype), ld, rows, ld, cnt.data_ptr(), one.data_ptr(),
                                       N.stream()), "ldm_scale_rows_dev")
        assert torch.equal(x, before)
        half = torch.full((1,), 0.5, device=dev)
        N.check(lib.ldm_scale_rows_dev(x.data_ptr(), N.dtype_code(dtype), ld, rows, ld, cnt.data_ptr(), half.data_ptr(),
                                       N.stream()), "ldm_scale_rows_dev")
        assert torch.equal(x[:count], (before[:count].float() * 0.5).to(dtype)) and torch.equal(x[count:], before[count:])


def test_fused_adamw_matches_oracle_and_torch(dev):
    """FusedAdamW (ldm_adamw_step) against the numpy restatement of torch.optim.AdamW and against
    torch.optim.AdamW on the device: odd sizes (scalar tail), >24 tensors (two launches), a bf16 and an
    fp32 shadow copy, and the gradient scale."""
    from oracle.adamw_oracle import adamw_step
    from src.optim import FusedAdamW
    torch.manual_seed(5)
    shapes = [(768, 768), (30522, 8), (5,), (1,), (4099,), (64, 1024)] + [(17 + i,) for i in range(24)]
    ps = [torch.randn(*s, device=dev).requires_grad_(True) for s in shapes]
    qs = [p.detach().clone().requires_grad_(True) for p in ps]
    opt = FusedAdamW(ps, lr=5e-4, weight_decay=0.1)
    ref = torch.optim.AdamW(qs, lr=5e-4, weight_decay=0.1)
    sh_bf = torch.zeros(768, 768, dtype=torch.bfloat16, device=dev)
    sh_f = torch.zeros(4099, device=dev)
    opt._shadows = {id(ps[0]): [(sh_bf, 1)], id(ps[4]): [(sh_f, 2)]}
    orc = [(p.detach().cpu().numpy().copy(), np.zeros(s, np.float32), np.zeros(s, np.float32)) for p, s in zip(ps, shapes)]
    for step in range(1, 6):
        gs = [torch.randn(*s, device=dev) * step for s in shapes]
        for p, q, g in zip(ps, qs, gs):
            p.grad, q.grad = g.clone(), g.clone()
        opt.step()
        ref.step()
        orc = [adamw_step(p, g.cpu().numpy(), m, v, step, lr=5e-4, weight_decay=0.1) for (p, m, v), g in zip(orc, gs)]
        for p, q, (o, _, _) in zip(ps, qs, orc):
            assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), step
            assert np.allclose(p.detach().cpu().numpy(), o, rtol=1e-5, atol=1e-7), step
        assert torch.equal(sh_bf, ps[0].detach().to(torch.bfloat16)) and torch.equal(sh_f, ps[4].detach())
    # gradient scale: the same update as scaling the gradient first
    a = torch.randn(1000, device=dev).requires_grad_(True)
    b = a.detach().clone().requires_grad_(True)
    oa, ob = FusedAdamW([a], lr=1e-3), FusedAdamW([b], lr=1e-3)
    oa.grad_scale = 0.125
    g = torch.randn(1000, device=dev)
    a.grad, b.grad = g.clone(), g * 0.125
    oa.step(); ob.step()
    assert torch.equal(a, b)
    # state_dict round trip with torch.optim.AdamW's layout
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}


@pytest.mark.parametrize("T,F,K", [(256, 256, 64), (512, 512, 768), (8192, 1024, 768), (16384, 768, 96)])
def test_gemm_swiglu_bwd_epilogue(dev, T, F, K):
    """ldm_gemm_swiglu_bwd (dag from the epilogue of dy @ W3, hidden gradient never written) against
    the fp32 composition of src/transformer.py:93 under autograd, and its shape refusals."""
    from src import _native as N
    lib = N.lib()
    torch.manual_seed(T + F)
    dy = torch.randn(T, K, device=dev).bfloat16()
    w = (torch.randn(K, F, device=dev) / math.sqrt(K)).bfloat16()
    ag = torch.randn(T, 2 * F, device=dev).bfloat16()
    dag = torch.full((T, 2 * F), float("nan"), device=dev, dtype=torch.bfloat16)
    N.check(lib.ldm_gemm_swiglu_bwd(T, F, K, dy.data_ptr(), K, w.data_ptr(), F, ag.data_ptr(), 2 * F, dag.data_ptr(),
                                    2 * F, N.stream()), "ldm_gemm_swiglu_bwd")
    agf = ag.float().requires_grad_(True)
    a, g = agf[:, :F], agf[:, F:]
    (a * torch.nn.functional.silu(g)).backward(dy.float() @ w.float())
    assert rel_err(dag.float(), agf.grad) < 6e-3
    # shapes the fused form does not take
    assert lib.ldm_gemm_swiglu_bwd(T + 8, F, K, dy.data_ptr(), K, w.data_ptr(), F, ag.data_ptr(), 2 * F, dag.data_ptr(),
                                   2 * F, N.stream()) == -3
    assert lib.ldm_gemm_swiglu_bwd(T, F - 128, K, dy.data_ptr(), K, w.data_ptr(), F, ag.data_ptr(), 2 * F,
                                   dag.data_ptr(), 2 * F, N.stream()) == -3


@pytest.mark.parametrize("T,F,K,bias", [(256, 128, 64, True), (512, 1024, 768, True), (1024, 5632, 2048, False),
                                        (768, 384, 96, True)])
def test_gemm_swiglu_fwd_epilogue(dev, T, F, K, bias):
    """ldm_gemm_swiglu_fwd: [a | g] = x W12^T + b and h = a * silu(g) (src/transformer.py:91-95) from one
    GEMM, against fp64; shapes it does not take are declined, not mis-computed."""
    from src import _native as Nn
    torch.manual_seed(T + F)
    x = torch.randn(T, K, device=dev).bfloat16()
    W = (torch.randn(2 * F, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(2 * F, device=dev) if bias else None
    ag = torch.full((T, 2 * F), float("nan"), device=dev, dtype=torch.bfloat16)
    h = torch.full((T, F), float("nan"), device=dev, dtype=torch.bfloat16)
    Nn.check(Nn.lib().ldm_gemm_swiglu_fwd(T, F, K, x.data_ptr(), K, W.data_ptr(), K, Nn.ptr(b), ag.data_ptr(), 2 * F,
                                          h.data_ptr(), F, Nn.stream()), "ldm_gemm_swiglu_fwd")
    ref = x.double() @ W.double().t() + (b.double() if bias else 0)
    a, g = ref[:, :F], ref[:, F:]
    assert rel_err(ag, ref) < 4e-3
    assert rel_err(h, a * g * torch.sigmoid(g)) < 4e-3
    # h is what the separate activation kernel makes of the stored (bf16-rounded) [a | g], up to one rounding
    h2 = torch.empty_like(h)
    Nn.check(Nn.lib().ldm_swiglu_fwd(ag.data_ptr(), T, F, 1, h2.data_ptr(), Nn.stream()), "ldm_swiglu_fwd")
    assert rel_err(h, h2) < 6e-3
    assert Nn.lib().ldm_gemm_swiglu_fwd(T - 8, F, K, x.data_ptr(), K, W.data_ptr(), K, Nn.ptr(b), ag.data_ptr(), 2 * F,
                                        h.data_ptr(), F, Nn.stream()) == -3
    assert Nn.lib().ldm_gemm_swiglu_fwd(T, F - 64, K, x.data_ptr(), K, W.data_ptr(), K, Nn.ptr(b), ag.data_ptr(), 2 * F,
                                        h.data_ptr(), F, Nn.stream()) == -3


def test_swiglu_module_fused_up_projection_matches_two_pass(dev):
    """SwiGLU.forward with the activation formed inside the gate/up GEMM and with the separate
    activation pass: same output and parameter gradients up to bf16 rounding of h."""
    from src import ops
    from src.transformer import SwiGLU
    torch.manual_seed(0)
    m = SwiGLU(768, 1024, 768).to(dev)
    m.compute_dtype = torch.bfloat16
    x = torch.randn(2, 512, 768, device=dev).bfloat16()
    go = torch.randn(2, 512, 768, device=dev).bfloat16()
    out = {}
    for fused in (True, False):
        ops.FUSED_SWIGLU_UP = fused
        try:
            m.zero_grad(set_to_none=True)
            xx = x.clone().requires_grad_(True)
            y = m(xx)
            y.backward(go)
        finally:
            ops.FUSED_SWIGLU_UP = True
        out[fused] = (y.detach().float(), xx.grad.float(), {n: p.grad.clone() for n, p in m.named_parameters()})
    assert rel_err(out[True][0], out[False][0]) < 6e-3
    assert rel_err(out[True][1], out[False][1]) < 6e-3
    for n in out[True][2]:
        assert rel_err(out[True][2][n], out[False][2][n]) < 6e-3, n


# ------------------------------------------------------------------ q/k/v projection with the rotary embedding in its epilogue
@pytest.mark.parametrize("T,S,H,hd,K", [(512, 512, 8, 96, 768), (1024, 128, 2, 128, 256), (512, 256, 4, 64, 320),
                                        (256, 128, 8, 32, 64), (4096, 4096, 16, 128, 512)])
def test_gemm_qkv_rope_epilogue(dev, T, S, H, hd, K):
    """ldm_gemm_qkv_rope against fp64 arithmetic (x W^T + b, then q*cos + rotate_half(q)*sin on the q and k
    thirds at position row % S, q / sqrt(hd): src/transformer.py:242-261) and against what ldm_gemm followed
    by ldm_rope_qk leaves; a row stride wider than 3D; shapes the kernel must decline."""
    from oracle import step_oracle as so
    from src import _native as N
    lib = N.lib()
    D = H * hd
    torch.manual_seed(T + hd)
    x = torch.randn(T, K, device=dev).bfloat16()
    W = (torch.randn(3 * D, K, device=dev) / math.sqrt(K)).bfloat16()
    b = torch.randn(3 * D, device=dev)
    cos, sin = so.rope_tables(S, hd)
    ch, sh = cos[:, :hd // 2].contiguous().to(dev), sin[:, :hd // 2].contiguous().to(dev)
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).double() / hd))
    step = torch.stack(((16 * inv_freq).cos(), (16 * inv_freq).sin())).float().contiguous().to(dev)
    scale = 1.0 / math.sqrt(hd)
    ld = 3 * D + 64
    out = torch.full((T, ld), 7.0, device=dev).bfloat16()
    rc = lib.ldm_gemm_qkv_rope(T, D, K, hd, S, x.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), ch.data_ptr(), sh.data_ptr(),
                               step.data_ptr(), scale, out.data_ptr(), ld, N.stream())
    assert rc == 0, rc
    # fp64 reference
    y = x.double() @ W.double().t() + b.double()
    pos = torch.arange(T, device=dev) % S
    c = cos.double().to(dev)[pos][:, None, :]
    s_ = sin.double().to(dev)[pos][:, None, :]

    def rot(v):
        v = v.reshape(T, H, hd)
        a, bb = v.chunk(2, dim=-1)
        return (v * c + torch.cat((-bb, a), dim=-1) * s_).reshape(T, D)
    ref = torch.cat((rot(y[:, :D]) * scale, rot(y[:, D:2 * D]), y[:, 2 * D:]), dim=1)
    for name, sl in (("q", slice(0, D)), ("k", slice(D, 2 * D)), ("v", slice(2 * D, 3 * D))):
        assert rel_err(out[:, sl].float(), ref[:, sl]) < 4e-3, name       # one bf16 rounding of the result
    assert torch.all(out[:, 3 * D:] == 7.0)                                # columns past 3D untouched
    # the two-call form (rounds q, k to bf16 before rotating): equal up to that extra rounding
    two = torch.empty(T, 3 * D, device=dev, dtype=torch.bfloat16)
    N.check(lib.ldm_gemm(N.BF16, N.BF16, 0, 1, T, 3 * D, K, 1.0, x.data_ptr(), K, W.data_ptr(), K, two.data_ptr(), 3 * D,
                         b.data_ptr(), None, 0, N.stream()), "ldm_gemm")
    N.check(lib.ldm_rope_qk(two.data_ptr(), N.BF16, T // S, S, H, hd, 3 * D, ch.data_ptr(), sh.data_ptr(), scale, 0,
                            N.stream()), "ldm_rope_qk")
    assert rel_err(out[:, :3 * D].float(), two.float()) < 8e-3
    assert rel_err(two.float(), ref) >= rel_err(out[:, :3 * D].float(), ref) * 0.9    # the fused form is not the less accurate one
    # declined shapes: ragged T, S that is not a multiple of 128, an unsupported head dim
    assert lib.ldm_gemm_qkv_rope(T - 8, D, K, hd, S, x.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), ch.data_ptr(),
                                 sh.data_ptr(), step.data_ptr(), scale, out.data_ptr(), ld, N.stream()) == -3
    assert lib.ldm_gemm_qkv_rope(T, D, K, hd, 100, x.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), ch.data_ptr(),
                                 sh.data_ptr(), step.data_ptr(), scale, out.data_ptr(), ld, N.stream()) == -3


def test_attention_module_fused_qkv_rope_matches_two_pass(dev):
    """RotaryMultiheadAttention with the rotary embedding in the projection epilogue and with the separate
    in-place pass: same output and the same gradients up to bf16 rounding."""
    from src import ops
    from src.transformer import RotaryMultiheadAttention
    torch.manual_seed(1)
    m = RotaryMultiheadAttention(768, 8, dropout=0.0).to(dev)
    m.compute_dtype = torch.bfloat16
    B, S = 2, 512
    x = torch.randn(B, S, 768, device=dev).bfloat16()
    go = torch.randn(B, S, 768, device=dev).bfloat16()
    kpm = torch.zeros(B, S, dtype=torch.bool, device=dev)
    kpm[1, 400:] = True
    out = {}
    for fused in (True, False):
        ops.FUSED_QKV_ROPE = fused
        try:
            m.zero_grad(set_to_none=True)
            xx = x.clone().requires_grad_(True)
            y = m(xx, xx, xx, key_padding_mask=kpm)
            y.backward(go)
        finally:
            ops.FUSED_QKV_ROPE = True
        out[fused] = (y.detach().float(), xx.grad.float(), {n: p.grad.clone() for n, p in m.named_parameters()})
    assert rel_err(out[True][0], out[False][0]) < 8e-3
    assert rel_err(out[True][1], out[False][1]) < 1e-2
    for n in out[True][2]:
        if n == "k_proj.bias":
            continue
        assert rel_err(out[True][2][n], out[False][2][n]) < 1e-2, n
