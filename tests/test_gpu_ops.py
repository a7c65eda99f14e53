"""GPU parity: norms, RoPE (+q/k norm), SwiGLU/GELU', layout helpers, embed/splice, CE, tap mix and
flash attention forward/backward — each against plain torch fp32 (autograd for backward)."""
import math

import pytest
import torch
import torch.nn.functional as F

import desta_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    assert torch.cuda.is_available()
    from desta import _hip
    return _hip


def bf(x):
    return x.to(torch.bfloat16)


def rel_err(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("rows,cols,f32in", [(37, 1280, True), (64, 384, False), (5, 128, True), (12, 2048, False), (9, 4096, True), (70, 2560, False)])
def test_layernorm_fwd_bwd(hip, rows, cols, f32in):
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols, generator=g) * 2 + 0.5
    if not f32in:
        x = bf(x).float()
    gam, bet = 1 + 0.1 * torch.randn(cols, generator=g), 0.1 * torch.randn(cols, generator=g)
    dy = torch.randn(rows, cols, generator=g)
    xr = x.clone().requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    y = F.layer_norm(xr, (cols,), gr, br, 1e-5)
    y.backward(dy)
    xd = (x if f32in else bf(x)).cuda()
    y16 = torch.empty(rows, cols, dtype=torch.bfloat16, device="cuda")
    y32 = torch.empty(rows, cols, dtype=torch.float32, device="cuda")
    st = torch.empty(rows, 2, device="cuda")
    hip.layernorm_fwd(xd, gam.cuda(), bet.cuda(), 1e-5, y16=y16, y32=y32, stats=st)
    torch.testing.assert_close(y32.cpu(), y.detach(), rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(y16.float().cpu(), y.detach(), rtol=1e-2, atol=1e-2)
    dx32 = torch.empty_like(y32)
    dx16 = torch.empty_like(y16)
    dg = torch.full((cols,), 7.0, device="cuda")
    db = torch.full((cols,), 7.0, device="cuda")
    hip.layernorm_bwd(dy.cuda(), xd, gam.cuda(), st, dx32=dx32, dx16=dx16, dgamma=dg, dbeta=db)
    torch.testing.assert_close(dx32.cpu(), xr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dg.cpu(), gr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu(), br.grad, rtol=1e-4, atol=1e-4)
    hip.layernorm_bwd(dy.cuda(), xd, gam.cuda(), st, dx32=dx32, dgamma=dg, dbeta=db, accumulate=True)
    torch.testing.assert_close(dg.cpu(), 2 * gr.grad, rtol=1e-4, atol=2e-4)
    assert rel_err(dx16.float().cpu(), xr.grad) < 1e-2


@pytest.mark.parametrize("rows,cols", [(33, 4096), (7, 256), (16, 5120)])
def test_rmsnorm_fwd_bwd(hip, rows, cols):
    g = torch.Generator().manual_seed(cols)
    x = bf(torch.randn(rows, cols, generator=g) * 1.5)
    w = 1 + 0.1 * torch.randn(cols, generator=g)
    dy = bf(torch.randn(rows, cols, generator=g))
    dres = bf(torch.randn(rows, cols, generator=g))
    xr = x.float().requires_grad_(True)
    var = xr.pow(2).mean(-1, keepdim=True)
    y = w * (xr * torch.rsqrt(var + 1e-5))
    y.backward(dy.float())
    yd = torch.empty(rows, cols, dtype=torch.bfloat16, device="cuda")
    rstd = torch.empty(rows, device="cuda")
    hip.rmsnorm_fwd(x.cuda(), w.cuda(), 1e-5, yd, rstd)
    assert rel_err(yd.float().cpu(), y.detach()) < 6e-3
    dx = torch.empty_like(yd)
    hip.rmsnorm_bwd(dy.cuda(), x.cuda(), w.cuda(), rstd, dx, dres=dres.cuda())
    assert rel_err(dx.float().cpu(), xr.grad + dres.float()) < 6e-3


def test_colsum_transpose_cast_add(hip):
    g = torch.Generator().manual_seed(2)
    x = bf(torch.randn(1000, 136, generator=g))
    out = torch.zeros(136, device="cuda")
    hip.colsum(x.cuda(), 1000, 136, 136, out)
    torch.testing.assert_close(out.cpu(), x.float().sum(0), rtol=1e-4, atol=1e-3)
    hip.colsum(x.cuda(), 1000, 136, 136, out, accumulate=True)
    torch.testing.assert_close(out.cpu(), 2 * x.float().sum(0), rtol=1e-4, atol=2e-3)
    # tall matrix (many row splits), a strided view (ld > cols), cols not a multiple of 8, tiny row counts; run-to-run identical
    for rows, cols, ld in ((48000, 2560, 2560), (2048, 3840, 3840), (4097, 132, 140), (5, 8, 8), (33, 1280, 2560)):
        xs = bf(torch.randn(rows, ld, generator=g)).cuda()
        o1, o2 = torch.zeros(cols, device="cuda"), torch.zeros(cols, device="cuda")
        hip.colsum(xs, rows, cols, ld, o1)
        hip.colsum(xs, rows, cols, ld, o2)
        ref = xs[:, :cols].double().sum(0).float()
        torch.testing.assert_close(o1, ref, rtol=2e-4, atol=2e-3 * math.sqrt(rows / 1000))
        assert torch.equal(o1, o2)
    # transpose with zero-padded K tail, bf16 and fp32 inputs
    t = torch.full((136, 1024), 9.0, dtype=torch.bfloat16, device="cuda")
    hip.transpose_to_bf16(x.cuda(), 1000, 136, t, 1024)
    assert torch.equal(t[:, :1000].cpu(), x.T.contiguous())
    assert float(t[:, 1000:].abs().max()) == 0.0
    xf = torch.randn(70, 200, generator=g)
    t2 = torch.empty(200, 128, dtype=torch.bfloat16, device="cuda")
    hip.transpose_to_bf16(xf.cuda(), 70, 200, t2, 128)
    assert torch.equal(t2[:, :70].cpu(), bf(xf).T.contiguous())
    # 16-byte path (cols, ld_in, ld_out multiples of 8): ragged rows, zero-padded tail, strided input, fp32 input
    for rows, cols, ld_in, ld_out, f32 in ((1000, 136, 136, 1024, False), (48000, 256, 512, 48000, False), (70, 200, 200, 128, True), (2048, 1280, 3840, 2048, False)):
        src = torch.randn(rows, ld_in, generator=g)
        src = src if f32 else bf(src)
        tv = torch.full((cols, ld_out), 9.0, dtype=torch.bfloat16, device="cuda")
        hip.transpose_to_bf16(src.cuda(), rows, cols, tv, ld_out, ld_in=ld_in)
        assert torch.equal(tv[:, :rows].cpu(), bf(src[:, :cols].float()).T.contiguous())
        assert ld_out == rows or float(tv[:, rows:].abs().max()) == 0.0
    y = torch.empty(70 * 200, dtype=torch.bfloat16, device="cuda")
    hip.cast_bf16(xf.cuda().reshape(-1), y)
    assert torch.equal(y.cpu(), bf(xf).reshape(-1))
    a, b = torch.randn(4096, generator=g), torch.randn(4096, generator=g)
    ad = a.cuda()
    hip.add_f32(ad, b.cuda())
    torch.testing.assert_close(ad.cpu(), a + b)


def test_mel_to_rows(hip):
    g = torch.Generator().manual_seed(3)
    mel = torch.randn(2, 80, 200, generator=g)
    out = torch.zeros(2, 202, 128, dtype=torch.bfloat16, device="cuda")
    hip.mel_to_rows(mel.cuda(), 128, out)
    ref = torch.zeros(2, 202, 128)
    ref[:, 1:201, :80] = mel.permute(0, 2, 1)
    assert torch.equal(out.cpu(), bf(ref))


@pytest.mark.parametrize("hd,qknorm", [(128, False), (64, False), (128, True), (64, True)])
def test_rope_fwd_bwd(hip, hd, qknorm):
    g = torch.Generator().manual_seed(hd)
    B, S, hq, hkv = 2, 9, 4, 2
    d = O.tiny_dims(qknorm)
    d.llm_hd = hd
    inv = O.rope_inv_freq(d)
    fr = torch.outer(torch.arange(S).float(), inv)
    cs = torch.stack([fr.cos(), fr.sin()], dim=1).contiguous()          # [S, 2, hd/2]
    ld = (hq + 2 * hkv) * hd
    buf = bf(torch.randn(B * S, ld, generator=g))
    wq, wk = 1 + 0.2 * torch.randn(hd, generator=g), 1 + 0.2 * torch.randn(hd, generator=g)

    def ref(xb):
        x = xb.view(B, S, hq + 2 * hkv, hd)
        qk = x[:, :, :hq + hkv]
        if qknorm:
            wcat = torch.cat([wq[None].expand(hq, -1), wk[None].expand(hkv, -1)])[None, None]
            qk = wcat * (qk * torch.rsqrt(qk.pow(2).mean(-1, keepdim=True) + 1e-6))
        cos = torch.cat([fr, fr], -1).cos()[None, :, None]
        sin = torch.cat([fr, fr], -1).sin()[None, :, None]
        qk = qk * cos + O._rot_half(qk) * sin
        return torch.cat([qk, x[:, :, hq + hkv:]], dim=2).reshape(B * S, ld)

    xr = buf.float().requires_grad_(True)
    y = ref(xr)
    dy = bf(torch.randn(B * S, ld, generator=g))
    y.backward(dy.float())
    out = buf.clone().cuda()
    hip.rope(out, ld, B * S, S, hq, hkv, hd, cs.cuda(), wq.cuda() if qknorm else None, wk.cuda() if qknorm else None, 1e-6)
    assert rel_err(out.float().cpu(), y.detach()) < 8e-3
    assert torch.equal(out[:, (hq + hkv) * hd:].cpu(), buf[:, (hq + hkv) * hd:])      # v untouched
    dbuf = dy.clone().cuda()
    hip.rope(dbuf, ld, B * S, S, hq, hkv, hd, cs.cuda(), wq.cuda() if qknorm else None, wk.cuda() if qknorm else None,
             1e-6, pre_norm=buf.cuda() if qknorm else None, ld_pre=ld, backward=True)
    assert rel_err(dbuf.float().cpu(), xr.grad) < 8e-3


def test_swiglu_gelu(hip):
    g = torch.Generator().manual_seed(4)
    rows, I = 19, 512
    gu = bf(torch.randn(rows, 2 * I, generator=g) * 2)
    dact = bf(torch.randn(rows, I, generator=g))
    gr = gu.float().requires_grad_(True)
    act = F.silu(gr[:, :I]) * gr[:, I:]
    act.backward(dact.float())
    a = torch.empty(rows, I, dtype=torch.bfloat16, device="cuda")
    hip.swiglu_fwd(gu.cuda(), a, rows, I)
    assert rel_err(a.float().cpu(), act.detach()) < 6e-3
    dgu = torch.empty(rows, 2 * I, dtype=torch.bfloat16, device="cuda")
    hip.swiglu_bwd(gu.cuda(), dact.cuda(), dgu, rows, I)
    assert rel_err(dgu.float().cpu(), gr.grad) < 6e-3
    pre = bf(torch.randn(rows, I, generator=g) * 2)
    pr = pre.float().requires_grad_(True)
    F.gelu(pr).backward(dact.float())
    dpre = torch.empty(rows, I, dtype=torch.bfloat16, device="cuda")
    hip.gelu_bwd(pre.cuda(), dact.cuda(), dpre, rows * I)
    assert rel_err(dpre.float().cpu(), pr.grad) < 6e-3


def test_embed_gather_and_rows(hip):
    g = torch.Generator().manual_seed(5)
    V, h = 50, 256
    table = bf(torch.randn(V, h, generator=g))
    audio = bf(torch.randn(6, h, generator=g))
    src = torch.tensor([3, 49, -1, -2, -6, 0, 7], dtype=torch.int32)
    out = torch.empty(7, h, dtype=torch.bfloat16, device="cuda")
    hip.embed_gather(table.cuda(), audio.cuda(), src.cuda(), 7, h, out)
    ref = torch.stack([table[3], table[49], audio[0], audio[1], audio[5], table[0], table[7]])
    assert torch.equal(out.cpu(), ref)
    idx = torch.tensor([6, 0, 2], dtype=torch.int32)
    o2 = torch.empty(3, h, dtype=torch.bfloat16, device="cuda")
    hip.gather_rows(out, idx.cuda(), 3, h, o2)
    assert torch.equal(o2.cpu(), ref[[6, 0, 2]])


@pytest.mark.parametrize("V", [512, 1000, 24000, 128256, 151936])
def test_causal_lm_loss(hip, V):
    g = torch.Generator().manual_seed(V)
    B, S = 2, 7
    logits = bf(torch.randn(B * S, V, generator=g) * 3)
    labels = torch.randint(0, V, (B, S), generator=g)
    labels[0, :3] = -100
    labels[1, 5] = -100
    lr = logits.float().view(B, S, V).requires_grad_(True)
    loss = O.causal_lm_loss(lr, labels)
    loss.backward()
    ld = (V + 7) // 8 * 8
    buf = torch.zeros(B * S, ld, dtype=torch.bfloat16, device="cuda")
    buf[:, :V] = logits.cuda()
    out = torch.zeros(1, device="cuda")
    hip.causal_lm_loss(buf, ld, labels.cuda(), B, S, V, out, write_grad=True)
    assert abs(float(out) - float(loss)) < 2e-5 * max(1.0, float(loss))
    got = buf[:, :V].float().cpu().view(B, S, V)
    assert rel_err(got, lr.grad) < 5e-3
    assert float(got[0, :2].abs().max()) == 0.0 and float(got[:, -1].abs().max()) == 0.0   # ignored rows
    # all labels ignored -> loss 0 (n_valid = 0 guarded), grads 0
    lab2 = torch.full((B, S), -100)
    buf[:, :V] = logits.cuda()
    hip.causal_lm_loss(buf, ld, lab2.cuda(), B, S, V, out, write_grad=True)
    assert float(out) == 0.0 and float(buf.float().abs().max()) == 0.0


def test_tap_mix(hip):
    g = torch.Generator().manual_seed(6)
    taps, B, K, d = 4, 3, 64, 128
    x = torch.randn(taps, B * K, d, generator=g)
    lw = torch.randn(K, taps, generator=g)
    dout = torch.randn(B * K, d, generator=g)
    xr, lr = x.clone().requires_grad_(True), lw.clone().requires_grad_(True)
    xx = xr.view(taps, B, K, d).permute(1, 2, 0, 3)
    y = (xx * torch.softmax(lr, -1).unsqueeze(-1)).sum(2).reshape(B * K, d)
    y.backward(dout)
    out = torch.empty(B * K, d, device="cuda")
    hip.tap_mix_fwd(x.cuda(), lw.cuda(), taps, B, K, d, out)
    torch.testing.assert_close(out.cpu(), y.detach(), rtol=1e-5, atol=1e-5)
    dx = torch.empty(taps, B * K, d, device="cuda")
    dlw = torch.empty(K, taps, device="cuda")
    hip.tap_mix_bwd(x.cuda(), lw.cuda(), dout.cuda(), taps, B, K, d, dx, dlw)
    torch.testing.assert_close(dx.cpu(), xr.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dlw.cpu(), lr.grad, rtol=1e-4, atol=1e-4)


def _attn_ref(q, k, v, scale, causal, kv_start):
    """q [B,Sq,Hq,D], k/v [B,Sk,Hkv,D] fp32 -> out [B,Sq,Hq,D]; fully masked rows -> 0."""
    B, Sq, Hq, D = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    rep = Hq // Hkv
    kk = k.repeat_interleave(rep, dim=2).permute(0, 2, 1, 3)
    vv = v.repeat_interleave(rep, dim=2).permute(0, 2, 1, 3)
    s = (q.permute(0, 2, 1, 3) @ kk.transpose(-1, -2)) * scale
    ok = torch.ones(B, 1, Sq, Sk, dtype=torch.bool)
    if causal:
        ok = ok & (torch.arange(Sk)[None, :] <= torch.arange(Sq)[:, None] + (Sk - Sq))[None, None]
    if kv_start is not None:
        ok = ok & (torch.arange(Sk)[None, None, None, :] >= kv_start[:, None, None, None])
    s = s.masked_fill(~ok, float("-inf"))
    p = torch.softmax(s, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)
    return (p @ vv).permute(0, 2, 1, 3)


ATTN_CASES = [
    # B, Hq, Hkv, Sq, Sk, D, causal, pad
    (2, 4, 4, 64, 64, 64, False, None),          # Q-Former self-attention
    (2, 2, 2, 64, 200, 64, False, None),         # Q-Former cross-attention (ragged Sk)
    (1, 3, 3, 333, 333, 64, False, None),        # Whisper-like, ragged
    (2, 4, 2, 160, 160, 128, True, [0, 37]),     # Llama GQA causal + left padding
    (2, 4, 1, 70, 70, 64, True, [5, 0]),         # tiny-LLM head dim 64
    (1, 8, 2, 300, 300, 128, True, None),
    # the 8-wave forward (seq_q >= 128): GQA groups of 4 / 3 / 2 / 1 heads, ragged ends, left padding, seq_k > seq_q
    (2, 8, 2, 640, 640, 128, True, [0, 100]),    # the LLM's shape per kv head: 4 heads x 64 rows per block
    (1, 6, 2, 200, 260, 128, True, [3]),         # group of 3 -> one head per block; causal with 60 cached keys in front
    (1, 4, 2, 257, 257, 128, True, None),        # group of 2 -> 2 heads x 128 rows; one row past a block boundary
    (2, 2, 1, 257, 257, 64, True, [0, 5]),       # head dim 64 causal, 2 heads per block
    (1, 2, 2, 1500, 1500, 64, False, None),      # Whisper: 1500 frames = 23.4 key tiles, 5.9 query blocks
    (1, 2, 2, 130, 700, 128, False, None),       # non-causal head dim 128, ragged both ways
    # one query tile over many keys (the Q-Former's cross-attention): dQ / dK / dV in one pass, the keys cut into 1 / 2 / 4 chunks
    (2, 3, 3, 64, 1500, 64, False, None),        # 12 key blocks -> 4 chunks of 3; last block ragged (1500 = 11 x 128 + 92)
    (1, 2, 2, 64, 600, 64, False, [77]),         # 5 key blocks -> 2 chunks (3 + 2), left padding inside the first block
    (2, 2, 2, 50, 300, 64, False, None),         # 3 key blocks -> 1 chunk; 50 query rows: the second 32-row slice is ragged
    (1, 2, 2, 20, 1100, 64, False, [0]),         # 9 key blocks -> 4 chunks of 3 / 3 / 3 / 0: an EMPTY chunk; only one query slice
]


@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention_fwd_bwd(hip, case):
    B, Hq, Hkv, Sq, Sk, D, causal, pad = case
    g = torch.Generator().manual_seed(Sq * 3 + Sk + D)
    # fused projection buffers: q | k | v in one row (self-attention) or separate q / kv buffers
    fused = Sq == Sk
    wq, wkv = Hq * D, Hkv * D
    if fused:
        qkv = bf(torch.randn(B * Sq, wq + 2 * wkv, generator=g))
        qb, kb_, vb = qkv, qkv, qkv
        q_off, k_off, v_off = 0, wq, wq + wkv
    else:
        qb = bf(torch.randn(B * Sq, wq, generator=g))
        kvb = bf(torch.randn(B * Sk, 2 * wkv, generator=g))
        kb_, vb = kvb, kvb
        q_off, k_off, v_off = 0, 0, wkv
    q = qb[:, q_off:q_off + wq].float().reshape(B, Sq, Hq, D).clone().requires_grad_(True)
    k = kb_[:, k_off:k_off + wkv].float().reshape(B, Sk, Hkv, D).clone().requires_grad_(True)
    v = vb[:, v_off:v_off + wkv].float().reshape(B, Sk, Hkv, D).clone().requires_grad_(True)
    scale = D ** -0.5
    kvs = torch.tensor(pad, dtype=torch.int32) if pad is not None else None
    ref = _attn_ref(q, k, v, scale, causal, kvs)
    do = bf(torch.randn(B * Sq, wq, generator=g))
    ref.backward(do.float().view(B, Sq, Hq, D))

    qd, kd, vd = qb.cuda(), (qb.cuda() if fused else kvb.cuda()), None
    if fused:
        kd = vd = qd
    else:
        vd = kd
    o = torch.zeros(B * Sq, wq, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, Hq, Sq, device="cuda")
    d = hip.attn_desc(qd, kd, vd, o, lse, batch=B, hq=Hq, hkv=Hkv, sq=Sq, sk=Sk, hd=D, scale=scale, causal=causal,
                      kv_start=kvs.cuda() if kvs is not None else None, q_off=q_off, k_off=k_off, v_off=v_off)
    hip.attention_fwd(d)
    got = o.float().cpu().view(B, Sq, Hq, D)
    assert rel_err(got, ref.detach()) < 8e-3, rel_err(got, ref.detach())
    if pad is not None and Sq == Sk:                     # left-pad QUERY rows see no key at all (with cached keys in front, seq_k > seq_q, they do)
        for b, pl in enumerate(pad):
            assert float(got[b, :pl].abs().max()) == 0.0 if pl else True

    if fused:
        dqkv = torch.zeros(B * Sq, wq + 2 * wkv, dtype=torch.bfloat16, device="cuda")
        hip.attention_bwd(d, do.cuda(), dqkv, dqkv, dqkv, dq_off=0, dk_off=wq, dv_off=wq + wkv)
        gq, gk, gv = dqkv[:, :wq], dqkv[:, wq:wq + wkv], dqkv[:, wq + wkv:]
    else:
        dq = torch.zeros(B * Sq, wq, dtype=torch.bfloat16, device="cuda")
        dkv = torch.zeros(B * Sk, 2 * wkv, dtype=torch.bfloat16, device="cuda")
        hip.attention_bwd(d, do.cuda(), dq, dkv, dkv, dk_off=0, dv_off=wkv)
        gq, gk, gv = dq, dkv[:, :wkv], dkv[:, wkv:]
    assert rel_err(gq.float().cpu().view(B, Sq, Hq, D), q.grad) < 1.5e-2
    assert rel_err(gk.float().cpu().view(B, Sk, Hkv, D), k.grad) < 1.5e-2
    assert rel_err(gv.float().cpu().view(B, Sk, Hkv, D), v.grad) < 1.5e-2
    # dQ-only variant (Whisper states carry no gradient)
    dq2 = torch.zeros(B * Sq, wq, dtype=torch.bfloat16, device="cuda")
    hip.attention_bwd(d, do.cuda(), dq2)
    if D == 64 and Sq <= 64 and Sk >= 256 and not causal:
        # the one-pass backward sums dQ over key chunks in its own (fixed) order: equal to the dQ-only kernel to rounding
        assert rel_err(dq2.float().cpu(), gq.float().cpu().contiguous()) < 1e-2
        dq3 = torch.zeros_like(dq2)
        dkv3 = torch.zeros(B * Sk, 2 * wkv, dtype=torch.bfloat16, device="cuda")
        hip.attention_bwd(d, do.cuda(), dq3, dkv3, dkv3, dk_off=0, dv_off=wkv)
        assert torch.equal(dq3, gq) and torch.equal(dkv3[:, :wkv], gk) and torch.equal(dkv3[:, wkv:], gv)   # bit-identical run to run
    else:
        assert torch.equal(dq2.cpu(), gq.cpu().contiguous())


def test_attention_forced_rescale(hip):
    """Spike one key late in the sequence so the running max jumps at a chosen tile (online-softmax
    rescale branch), plus an exact-integer check of the P·V operand order (V = one-hot rows)."""
    B, H, S, D = 1, 1, 256, 64
    g = torch.Generator().manual_seed(0)
    q = bf(torch.randn(B * S, D, generator=g))
    k = bf(torch.randn(B * S, D, generator=g))
    k[200] = (q[5].float() * 6).to(torch.bfloat16)          # huge score for query 5 at key 200 (4th tile)
    v = bf(torch.randn(B * S, D, generator=g))
    ref = _attn_ref(q.float().view(1, S, 1, D), k.float().view(1, S, 1, D), v.float().view(1, S, 1, D), D ** -0.5, False, None)
    o = torch.zeros(S, D, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(1, 1, S, device="cuda")
    d = hip.attn_desc(q.cuda(), k.cuda(), v.cuda(), o, lse, batch=1, hq=1, hkv=1, sq=S, sk=S, hd=D, scale=D ** -0.5)
    hip.attention_fwd(d)
    assert rel_err(o.float().cpu().view(1, S, 1, D), ref) < 8e-3
    # uniform attention (q = 0) over asymmetric integer V: out = column means, exact in bf16-friendly ints
    S2 = 64
    v2 = torch.zeros(S2, D)
    v2[torch.arange(S2), torch.arange(S2) % D] = 64.0
    v2[:, 0] += torch.arange(S2).float()
    q0 = torch.zeros(S2, D, dtype=torch.bfloat16)
    o2 = torch.zeros(S2, D, dtype=torch.bfloat16, device="cuda")
    lse2 = torch.zeros(1, 1, S2, device="cuda")
    d2 = hip.attn_desc(q0.cuda(), q0.cuda(), bf(v2).cuda(), o2, lse2, batch=1, hq=1, hkv=1, sq=S2, sk=S2, hd=D, scale=1.0)
    hip.attention_fwd(d2)
    torch.testing.assert_close(o2.float().cpu(), v2.mean(0, keepdim=True).expand(S2, D), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("spike", [3.0, 6.5, 40.0])
def test_attention_deferred_rescale_threshold(hip, D, spike):
    """The 8-wave forward moves its softmax reference only when a row's tile maximum exceeds it by more than 2^6 (guide T13).
    Three regimes of ONE late key whose score for ONE query sits `spike` (natural-log units) above everything before it:
    3.0 and 6.5 around ln 64 = 4.16 (deferred: probabilities up to e^3 against a stale reference / just rescaled), 40 (far
    beyond).  A wrong deferred path is silent (no NaN), so the whole tensor is compared with an fp64 reference, and the
    4-wave kernel (reference moves on every tile) must agree with the 8-wave one to rounding."""
    S, H = 384, 2
    g = torch.Generator().manual_seed(int(spike * 10) + D)
    q = bf(torch.randn(S, H * D, generator=g))
    k = bf(0.3 * torch.randn(S, H * D, generator=g))
    v = bf(torch.randn(S, H * D, generator=g))
    for h in range(H):                                   # key 300 (5th tile) for query 7 + h: score = spike + max of the earlier scores
        qq = q[7 + h, h * D:(h + 1) * D].float()
        base = float((q[7 + h, h * D:(h + 1) * D].float() @ k[:300, h * D:(h + 1) * D].float().T).max()) * D ** -0.5
        k[300, h * D:(h + 1) * D] = (qq * ((spike + base) * D ** 0.5 / float(qq @ qq))).to(torch.bfloat16)
    ref = _attn_ref(q.double().view(1, S, H, D), k.double().view(1, S, H, D), v.double().view(1, S, H, D), D ** -0.5, False, None)
    outs = []
    for eight in (1, 0):
        hip.attention_set_option(0, eight)
        try:
            o = torch.zeros(S, H * D, dtype=torch.bfloat16, device="cuda")
            lse = torch.zeros(1, H, S, device="cuda")
            d = hip.attn_desc(q.cuda(), k.cuda(), v.cuda(), o, lse, batch=1, hq=H, hkv=H, sq=S, sk=S, hd=D, scale=D ** -0.5)
            hip.attention_fwd(d)
            outs.append((o.float().cpu(), lse.cpu().clone()))
        finally:
            hip.attention_set_option(0, 1)
    for o, _ in outs:
        assert rel_err(o.view(1, S, H, D), ref.float()) < 8e-3
        assert float((o.view(1, S, H, D) - ref.float()).abs().max()) < 4e-2          # |V| ~ N(0,1): a mis-scaled row is off by O(1)
    assert float((outs[0][0] - outs[1][0]).abs().max()) < 4e-2
    torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-4, atol=1e-4)                 # log-sum-exp independent of the reference used


def test_attention_fp32_output_copy_feeds_delta(hip):
    """desta_attn_desc.O_f32: forward also writes the unrounded output, backward takes delta = rowsum(dO * O) from it.  Checked:
    bf16(O_f32) == O bit for bit, O_f32 is closer to the fp64 reference than O, and the backward with it is at least as
    accurate on dQ as without (flat softmax over many keys with a common component in K: the regime of the Q-Former's
    cross-attention, where the rounded-O delta error is coherent over the keys)."""
    B, H, Sq, Sk, D = 2, 2, 64, 1500, 64
    g = torch.Generator().manual_seed(5)
    q = bf(0.2 * torch.randn(B * Sq, H * D, generator=g))
    kv = bf(0.5 * torch.randn(B * Sk, 2 * H * D, generator=g) + 1.0)          # keys / values with a large common component
    do = bf(torch.randn(B * Sq, H * D, generator=g))
    qf_ = q.double().view(B, Sq, H, D).requires_grad_(True)
    kf = kv[:, :H * D].double().reshape(B, Sk, H, D).requires_grad_(True)
    vf = kv[:, H * D:].double().reshape(B, Sk, H, D).requires_grad_(True)
    ref = _attn_ref(qf_, kf, vf, D ** -0.5, False, None)
    ref.backward(do.double().view(B, Sq, H, D))
    errs = {}
    for use32 in (True, False):
        o = torch.zeros(B * Sq, H * D, dtype=torch.bfloat16, device="cuda")
        o32 = torch.zeros(B * Sq, H * D, dtype=torch.float32, device="cuda") if use32 else None
        lse = torch.zeros(B, H, Sq, device="cuda")
        d = hip.attn_desc(q.cuda(), kv.cuda(), kv.cuda(), o, lse, batch=B, hq=H, hkv=H, sq=Sq, sk=Sk, hd=D, scale=D ** -0.5,
                          q_off=0, k_off=0, v_off=H * D, o_f32=o32)
        hip.attention_fwd(d)
        if use32:
            assert torch.equal(o32.to(torch.bfloat16), o)
            assert rel_err(o32.cpu().view(B, Sq, H, D), ref.detach().float()) < rel_err(o.float().cpu().view(B, Sq, H, D), ref.detach().float())
        dq = torch.zeros(B * Sq, H * D, dtype=torch.bfloat16, device="cuda")
        dkv = torch.zeros(B * Sk, 2 * H * D, dtype=torch.bfloat16, device="cuda")
        hip.attention_bwd(d, do.cuda(), dq, dkv, dkv, dk_off=0, dv_off=H * D)
        errs[use32] = rel_err(dq.float().cpu().view(B, Sq, H, D), qf_.grad.float())
    print("dQ rel err with fp32 O:", errs[True], " with bf16 O:", errs[False])
    assert errs[True] < 1.5e-2 and errs[True] <= errs[False] * 1.05


@pytest.mark.parametrize("hd,smajor", [(128, True), (64, False)])
def test_rope_in_gemm_epilogue_and_attention_backward(hip, hd, smajor):
    """desta_gemm_desc.rope_*: the q|k|v projection with permuted q / k weight rows leaves q, k rotated (adjacent pairs) —
    equal, after undoing the permutation, to HF's apply_rotary_pos_emb on the plain projection.  desta_attn_desc.rope_cos_sin:
    the backward returns gradients of the projection OUTPUTS (checked against torch autograd through rotate + attention)."""
    B, S, Hq, Hkv, K = 2, 160, 4, 2, 256
    g = torch.Generator().manual_seed(hd)
    nqk, w = (Hq + Hkv) * hd, (Hq + 2 * Hkv) * hd
    x = bf(torch.randn(B * S, K, generator=g))
    W = bf(torch.randn(w, K, generator=g) / 16)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = torch.outer(torch.arange(S).float(), inv)                       # [S, hd/2]
    cs_il = torch.stack([fr.cos(), fr.sin()], dim=2).contiguous()
    pos = (torch.arange(B * S) // B if smajor else torch.arange(B * S) % S).to(torch.int32)
    Wil = W.clone()
    Wil[:nqk] = W[:nqk].view(Hq + Hkv, 2, hd // 2, K).transpose(1, 2).reshape(nqk, K)
    out = torch.empty(B * S, w, dtype=torch.bfloat16, device="cuda")
    hip.gemm(x.cuda(), Wil.cuda(), out, B * S, w, K, rope=(cs_il.cuda(), pos.cuda(), nqk, hd))
    # reference: plain projection (fp32), HF rotation, then the same permutation of the head dim
    y = (x.float() @ W.float().T).requires_grad_(True)
    yh = y[:, :nqk].view(B * S, Hq + Hkv, hd)
    c = torch.cat([fr, fr], -1).cos()[pos.long()][:, None, :]
    sn = torch.cat([fr, fr], -1).sin()[pos.long()][:, None, :]
    rot = torch.cat([-yh[..., hd // 2:], yh[..., :hd // 2]], -1)
    ye = yh * c + rot * sn                                                # [rows, heads, hd] HF layout
    ye_il = ye.view(B * S, Hq + Hkv, 2, hd // 2).transpose(2, 3).reshape(B * S, nqk)
    assert rel_err(out[:, :nqk].float().cpu(), ye_il.detach()) < 6e-3
    assert rel_err(out[:, nqk:].float().cpu(), y[:, nqk:].detach()) < 6e-3
    # attention on the rotated projections, backward to the projection outputs
    rs, bs = (B * w, w) if smajor else (w, S * w)
    o = torch.zeros(B * S, Hq * hd, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, Hq, S, device="cuda")
    d = hip.attn_desc(out, out, out, o, lse, batch=B, hq=Hq, hkv=Hkv, sq=S, sk=S, hd=hd, scale=hd ** -0.5, causal=True,
                      q_off=0, k_off=Hq * hd, v_off=nqk, q_rs=rs, k_rs=rs, v_rs=rs, o_rs=rs // w * Hq * hd,
                      q_bs=bs, k_bs=bs, v_bs=bs, o_bs=bs // w * Hq * hd)
    hip.attention_fwd(d)
    do = bf(torch.randn(B * S, Hq * hd, generator=g))
    dqkv = torch.zeros(B * S, w, dtype=torch.bfloat16, device="cuda")
    ors, obs = rs // w * Hq * hd, bs // w * Hq * hd
    hip.attention_bwd(d, do.cuda(), dqkv, dqkv, dqkv, dq_off=0, dk_off=Hq * hd, dv_off=nqk, do_rs=ors, dq_rs=rs, dk_rs=rs, dv_rs=rs,
                      do_bs=obs, dq_bs=bs, dk_bs=bs, dv_bs=bs, rope_cos_sin=cs_il.cuda())
    # torch: rows -> [B, S] order, rotate, attention, autograd back to y
    idx = (torch.arange(S)[None, :] * B + torch.arange(B)[:, None]).reshape(-1) if smajor else torch.arange(B * S)
    qe = ye[idx][:, :Hq].view(B, S, Hq, hd)
    ke = ye[idx][:, Hq:].view(B, S, Hkv, hd)
    v = y[idx][:, nqk:].view(B, S, Hkv, hd)
    ref = _attn_ref(qe, ke, v, hd ** -0.5, True, None)
    assert rel_err(o.float().cpu()[idx].view(B, S, Hq, hd), ref.detach()) < 8e-3
    ref.backward(do.float()[idx].view(B, S, Hq, hd))
    gy = y.grad                                                           # gradient of the PLAIN projection, plain layout
    gy_il = torch.cat([gy[:, :nqk].view(B * S, Hq + Hkv, 2, hd // 2).transpose(2, 3).reshape(B * S, nqk), gy[:, nqk:]], 1)
    got = dqkv.float().cpu()
    assert rel_err(got[:, :Hq * hd], gy_il[:, :Hq * hd]) < 1.5e-2
    assert rel_err(got[:, Hq * hd:nqk], gy_il[:, Hq * hd:nqk]) < 1.5e-2
    assert rel_err(got[:, nqk:], gy_il[:, nqk:]) < 1.5e-2


def test_attention_desc_keeps_tensors_alive_and_clamps_kv_start(hip):
    """The descriptor holds raw device pointers: it must keep temporaries alive (a freed kv_start buffer
    reused by a later allocation once turned into garbage pad lengths).  Out-of-range pad lengths are
    clamped to [0, Sk] inside the kernels instead of indexing out of bounds."""
    B, H, S, D = 2, 2, 96, 128
    g = torch.Generator().manual_seed(1)
    qkv = bf(torch.randn(B * S, 3 * H * D, generator=g))
    o = torch.zeros(B * S, H * D, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, H, S, device="cuda")
    d = hip.attn_desc(qkv.cuda(), qkv.cuda(), qkv.cuda(), o, lse, batch=B, hq=H, hkv=H, sq=S, sk=S, hd=D, scale=D ** -0.5,
                      causal=True, kv_start=torch.tensor([-5, 1000], dtype=torch.int32).cuda(), q_off=0, k_off=H * D, v_off=2 * H * D)
    junk = [torch.full((1024,), -1.0, device="cuda") for _ in range(8)]       # would recycle freed temporaries
    hip.attention_fwd(d)
    dq = torch.zeros(B * S, 3 * H * D, dtype=torch.bfloat16, device="cuda")
    hip.attention_bwd(d, bf(torch.randn(B * S, H * D, generator=g)).cuda(), dq, dq, dq, dk_off=H * D, dv_off=2 * H * D)
    torch.cuda.synchronize()
    out = o.float().cpu().view(B, S, H, D)
    q = qkv[:, :H * D].float().view(B, S, H, D)
    k = qkv[:, H * D:2 * H * D].float().view(B, S, H, D)
    v = qkv[:, 2 * H * D:].float().view(B, S, H, D)
    ref = _attn_ref(q, k, v, D ** -0.5, True, torch.tensor([0, S]))
    assert rel_err(out, ref) < 8e-3
    assert float(out[1].abs().max()) == 0.0 and torch.isfinite(dq.float()).all()
    del junk


# ----------------------------------------------------------------------------- dropout (Q-Former, hazard H3)
def test_dropout_mask_gemm_epilogue_and_backward_kernel(hip):
    g = torch.Generator().manual_seed(4)
    M, N, K, pdrop, seed = 320, 256, 128, 0.1, (7 << 40) | 12345
    mask = hip.dropout_mask(seed, M * N, pdrop).float().cpu().view(M, N)
    keep = float(mask.mean())
    assert abs(keep - 0.9) < 0.01                                         # Bernoulli(0.9) over 82k elements
    assert abs(float(mask[:, ::2].mean()) - float(mask[:, 1::2].mean())) < 0.01
    m2 = hip.dropout_mask(seed + 1, M * N, pdrop).float().cpu().view(M, N)
    assert abs(float((mask * m2).mean()) - 0.81) < 0.01                   # streams are independent
    A, B = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) / 8)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    for variant in (1, 3):
        try:
            hip.gemm_force_variant(variant)
            out = torch.empty(M, N, dtype=torch.float32, device="cuda")
            hip.gemm(A.cuda(), B.cuda(), out, M, N, K, bias=bias.cuda(), residual=res.cuda(), dropout_p=pdrop, dropout_seed=seed)
        finally:
            hip.gemm_force_variant(0)
        ref = (A.float() @ B.float().T + bias) * mask / (1 - pdrop) + res
        torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-4)
    x = bf(torch.randn(M, N, generator=g))
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    hip.dropout_bf16(x.cuda(), y, M, N, N, pdrop, seed)
    torch.testing.assert_close(y.float().cpu(), x.float() * mask / (1 - pdrop), rtol=8e-3, atol=1e-6)


@pytest.mark.parametrize("Sq,Sk", [(64, 64), (64, 200), (64, 1500), (40, 520)])
def test_attention_dropout_fwd_bwd(hip, Sq, Sk):
    B, H, D, pdrop, seed = 2, 3, 64, 0.1, (3 << 40) | 77
    g = torch.Generator().manual_seed(Sq + Sk)
    qb = bf(torch.randn(B * Sq, H * D, generator=g))
    kvb = bf(torch.randn(B * Sk, 2 * H * D, generator=g))
    do = bf(torch.randn(B * Sq, H * D, generator=g))
    mask = hip.dropout_mask(seed, B * H * Sq * Sk, pdrop).float().cpu().view(B, H, Sq, Sk)
    q = qb.float().view(B, Sq, H, D).clone().requires_grad_(True)
    k = kvb[:, :H * D].float().reshape(B, Sk, H, D).clone().requires_grad_(True)
    v = kvb[:, H * D:].float().reshape(B, Sk, H, D).clone().requires_grad_(True)
    p = torch.softmax((q.permute(0, 2, 1, 3) @ k.permute(0, 2, 3, 1)) * D ** -0.5, dim=-1)
    ref = ((p * mask / (1 - pdrop)) @ v.permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
    ref.backward(do.float().view(B, Sq, H, D))
    qd, kvd, dod = qb.cuda(), kvb.cuda(), do.cuda()
    o = torch.zeros(B * Sq, H * D, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, H, Sq, device="cuda")
    d = hip.attn_desc(qd, kvd, kvd, o, lse, batch=B, hq=H, hkv=H, sq=Sq, sk=Sk, hd=D, scale=D ** -0.5, k_off=0, v_off=H * D,
                      dropout_p=pdrop, dropout_seed=seed)
    hip.attention_fwd(d)
    assert rel_err(o.float().cpu().view(B, Sq, H, D), ref.detach()) < 8e-3
    dq = torch.zeros(B * Sq, H * D, dtype=torch.bfloat16, device="cuda")
    dkv = torch.zeros(B * Sk, 2 * H * D, dtype=torch.bfloat16, device="cuda")
    hip.attention_bwd(d, dod, dq, dkv, dkv, dk_off=0, dv_off=H * D)
    assert rel_err(dq.float().cpu().view(B, Sq, H, D), q.grad) < 1.5e-2
    assert rel_err(dkv[:, :H * D].float().cpu().view(B, Sk, H, D), k.grad) < 1.5e-2
    assert rel_err(dkv[:, H * D:].float().cpu().view(B, Sk, H, D), v.grad) < 1.5e-2
    # p = 0 through the same entry point is the plain kernel
    d0 = hip.attn_desc(qd, kvd, kvd, o, lse, batch=B, hq=H, hkv=H, sq=Sq, sk=Sk, hd=D, scale=D ** -0.5, k_off=0, v_off=H * D)
    hip.attention_fwd(d0)
    ref0 = (p @ v.permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
    assert rel_err(o.float().cpu().view(B, Sq, H, D), ref0.detach()) < 8e-3


def test_target_rows_and_scatter(hip):
    """desta_target_rows: rows whose SHIFTED label is a target, their labels in the layout the CE entry point consumes with
    (batch = 1, seq = n + 1), and the count; desta_scatter_rows_bf16 inverts desta_gather_rows_bf16."""
    g = torch.Generator().manual_seed(12)
    B, S = 5, 37
    labels = torch.randint(0, 1000, (B, S), generator=g)
    labels[:, :9] = -100
    labels[2, 20:25] = -100
    labels[4] = -100
    idx = torch.full((B * S,), -1, dtype=torch.int32, device="cuda")
    lab = torch.full((B * S + 2,), 7, dtype=torch.int64, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int32, device="cuda")            # (n, first position that carries a target in any sequence)
    hip.target_rows(labels.cuda(), B, S, idx, lab, cnt)
    ref = [(b * S + s, int(labels[b, s + 1])) for b in range(B) for s in range(S - 1) if labels[b, s + 1] != -100]
    n = int(cnt[0])
    assert int(cnt[1]) == min(r % S for r, _ in ref) == 8
    assert n == len(ref) and idx[:n].cpu().tolist() == [r for r, _ in ref]
    assert lab[:n + 2].cpu().tolist() == [-100] + [t for _, t in ref] + [-100]
    # CE on compact logits == CE on the full grid (loss and the gradient rows)
    V = 1000
    logits = bf(torch.randn(B * S, V, generator=g) * 2).cuda()
    full = logits.clone()
    loss_f, loss_c = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    hip.causal_lm_loss(full, V, labels.cuda(), B, S, V, loss_f, write_grad=True)
    comp = torch.zeros(n + 1, V, dtype=torch.bfloat16, device="cuda")
    hip.gather_rows(logits, idx, n, V, comp)
    hip.causal_lm_loss(comp, V, lab, 1, n + 1, V, loss_c, write_grad=True)
    assert abs(float(loss_f) - float(loss_c)) < 1e-6 * max(1.0, float(loss_f))
    back = torch.zeros_like(full)
    hip.scatter_rows(comp, idx, n, V, back)
    assert torch.equal(back, full)                                  # non-target rows of the full gradient are exactly zero
    # position-major row ids (s * B + b), same compact order
    idx_s = torch.full((B * S,), -1, dtype=torch.int32, device="cuda")
    hip.target_rows(labels.cuda(), B, S, idx_s, lab, cnt, s_major=True)
    assert idx_s[:n].cpu().tolist() == [(r % S) * B + r // S for r, _ in ref]
    # no targets at all -> count 0
    hip.target_rows(torch.full((B, S), -100).cuda(), B, S, idx, lab, cnt)
    assert cnt.cpu().tolist() == [0, S] and lab[:2].cpu().tolist() == [-100, -100]


@pytest.mark.gpu
def test_context_create_info_destroy_and_attention_still_runs(hip):
    """`desta_create` / `desta_handle_info` / `desta_handle_last_error` / `desta_destroy`: the context reports the
    gfx950 device, an invalid device is refused with a message, and after `destroy` released the internal fork stream the
    attention backward recreates it at first use (same result as before)."""
    B, H, S, D = 2, 2, 128, 128
    g = torch.Generator().manual_seed(3)
    qkv = bf(torch.randn(B * S, 3 * H * D, generator=g)).cuda()
    do = bf(torch.randn(B * S, H * D, generator=g)).cuda()

    def run():
        o = torch.zeros(B * S, H * D, dtype=torch.bfloat16, device="cuda")
        lse = torch.zeros(B, H, S, device="cuda")
        d = hip.attn_desc(qkv, qkv, qkv, o, lse, batch=B, hq=H, hkv=H, sq=S, sk=S, hd=D, scale=D ** -0.5, causal=True,
                          q_off=0, k_off=H * D, v_off=2 * H * D)
        hip.attention_fwd(d)
        dqkv = torch.zeros(B * S, 3 * H * D, dtype=torch.bfloat16, device="cuda")
        hip.attention_bwd(d, do, dqkv, dqkv, dqkv, dk_off=H * D, dv_off=2 * H * D)
        torch.cuda.synchronize()
        return o, dqkv

    before = run()
    ctx = hip.Context(0)
    info = ctx.info()
    assert info["device"] == 0 and info["arch"].startswith("gfx950") and info["compute_units"] == 256
    with pytest.raises(RuntimeError):
        hip.Context(99)
    assert "device 99" in ctx.last_error()
    ctx.close()
    after = run()
    for a, b in zip(before, after):
        assert torch.equal(a, b)
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("Sq,Sk,pdrop", [(64, 1500, 0.0), (64, 1500, 0.1), (40, 520, 0.1), (64, 300, 0.0)])
def test_attention_bwd_transposed_dkv_and_bias_sums(hip, Sq, Sk, pdrop):
    """`desta_attn_desc.dkv_transposed` (one-query-tile backward): dK / dV come out as [heads * 64][ld] matrices with column
    batch * seq_k + key — bit-identical to the row-major outputs of the same kernel —, `dkv_bias_grad` = their sums over batch
    and keys (fp32, from the unrounded accumulators), dQ unchanged; pad columns [batch * seq_k, ld) are not touched."""
    B, H, D, seed = 3, 2, 64, (5 << 40) | 9
    g = torch.Generator().manual_seed(Sq + Sk)
    qb = bf(torch.randn(B * Sq, H * D, generator=g)).cuda()
    kvb = bf(torch.randn(B * Sk, 2 * H * D, generator=g)).cuda()
    do = bf(torch.randn(B * Sq, H * D, generator=g)).cuda()
    o = torch.zeros(B * Sq, H * D, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, H, Sq, device="cuda")
    d = hip.attn_desc(qb, kvb, kvb, o, lse, batch=B, hq=H, hkv=H, sq=Sq, sk=Sk, hd=D, scale=D ** -0.5, k_off=0, v_off=H * D,
                      dropout_p=pdrop, dropout_seed=seed)
    hip.attention_fwd(d)
    dq = torch.zeros(B * Sq, H * D, dtype=torch.bfloat16, device="cuda")
    dkv = torch.zeros(B * Sk, 2 * H * D, dtype=torch.bfloat16, device="cuda")
    hip.attention_bwd(d, do, dq, dkv, dkv, dk_off=0, dv_off=H * D)
    ld = (B * Sk + 63) // 64 * 64 + 64
    t = torch.full((2 * H * D, ld), 7.0, dtype=torch.bfloat16, device="cuda")
    bias = torch.full((2 * H * D,), -1.0, device="cuda")
    dq2 = torch.zeros_like(dq)
    hip.attention_bwd(d, do, dq2, dkv_t=(t, ld, bias))
    torch.cuda.synchronize()
    assert torch.equal(dq2, dq)
    assert torch.equal(t[:, :B * Sk], dkv.t().contiguous())
    assert float((t[:, B * Sk:].float() - 7.0).abs().max()) == 0.0
    ref = dkv.float().sum(0)
    assert rel_err(bias, ref) < 4e-3, rel_err(bias, ref)                     # (the reference sum is over bf16-rounded values)
