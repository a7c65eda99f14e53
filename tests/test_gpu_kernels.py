"""GPU parity tests of the individual HIP kernels against the CPU oracle / plain torch fp32.
All calls go through the C ABI (desta._hip ctypes binding)."""
import math

import pytest
import torch

import desta_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from desta import _hip
    return _hip


def _bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (200, 136, 192), (1500, 1280, 384), (77, 4, 64)])
def test_gemm_plain(hip, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    A = _bf(torch.randn(M, K, generator=g)).cuda()
    B = _bf(torch.randn(N, K, generator=g)).cuda()
    ref = A.float() @ B.float().T
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    hip.gemm(A, B, out, M, N, K)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-3)
    outb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    hip.gemm(A, B, outb, M, N, K)
    torch.testing.assert_close(outb.float(), ref, rtol=1e-2, atol=1e-2 * math.sqrt(K))


@pytest.mark.parametrize("M,N,K", [(256, 256, 512), (512, 768, 1024), (300, 520, 576), (1500, 1280, 1280), (257, 260, 640),
                                   (2560, 1280, 4096), (5120, 4096, 512)])
def test_gemm_256_tile_variant(hip, M, N, K):
    """The 256x256 8-phase kernel (forced), incl. ragged M/N edges and K tails shorter than the prefetch depth."""
    g = torch.Generator().manual_seed(M + N + K)
    A = _bf(torch.randn(M, K, generator=g)).cuda()
    B = _bf(torch.randn(N, K, generator=g)).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = _bf(torch.randn(M, N, generator=g)).cuda()
    ref = A.float() @ B.float().T
    try:
        hip.gemm_force_variant(2)
        out = torch.empty(M, N, dtype=torch.float32, device="cuda")
        hip.gemm(A, B, out, M, N, K)
        torch.testing.assert_close(out, ref, rtol=1e-4, atol=2e-3)
        outb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        hip.gemm(A, B, outb, M, N, K, bias=bias, residual=res, act=1)
        hip.gemm_force_variant(1)
        outb1 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        hip.gemm(A, B, outb1, M, N, K, bias=bias, residual=res, act=1)
        # same fp32 accumulation order unless the tail tiles were K-split: allow one bf16 ulp
        torch.testing.assert_close(outb.float(), outb1.float(), rtol=8e-3, atol=8e-3)
    finally:
        hip.gemm_force_variant(0)


@pytest.mark.parametrize("M,N,K", [(512, 768, 1024), (300, 520, 576), (2560, 1280, 4096), (256, 256, 64), (5120, 4096, 1024),
                                   (12000, 1280, 512), (4352, 4352, 512)])
def test_gemm_256_staggered_variant(hip, M, N, K):
    """Variants 3 / 4: waves 4-7 run half a phase behind waves 0-3; 4 additionally walks several tiles per block
    with the LDS-DMA stream continuing across tile boundaries (same arithmetic, different schedules)."""
    g = torch.Generator().manual_seed(M * 3 + N + K)
    A = _bf(torch.randn(M, K, generator=g)).cuda()
    B = _bf(torch.randn(N, K, generator=g)).cuda()
    outs = []
    try:
        for v in (2, 3, 4, 6, 7, 8):
            hip.gemm_force_variant(v)
            o = torch.empty(M, N, dtype=torch.float32, device="cuda")
            for _ in range(3):                      # repeat: a schedule race would show up as run-to-run differences
                hip.gemm(A, B, o, M, N, K)
            outs.append(o)
    finally:
        hip.gemm_force_variant(0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(outs[0], outs[3]) and torch.equal(outs[0], outs[4])      # 6 / 7: two-phase schedules (staggered / lockstep)
    assert torch.equal(outs[0], outs[5])                                        # 8: two-phase, persistent tile walk
    torch.testing.assert_close(outs[1], A.float() @ B.float().T, rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("M,N,K", [(1, 16, 64), (3, 100, 192), (8, 4096, 4096), (16, 1028, 2048), (8, 6144, 14336), (5, 36, 64 * 37)])
def test_gemm_skinny(hip, M, N, K):
    """M <= 16 takes the weight-streaming kernel (decode projections): same results as the tile kernel / fp32 torch,
    incl. ragged N (multiple of 4 only), strided A rows, bias / GELU / residual epilogues and fp32 output."""
    g = torch.Generator().manual_seed(M * 11 + N + K)
    Abig = _bf(torch.randn(M, 3, K, generator=g)).cuda()              # rows strided by 3K, like hb[S-1::S]
    A = Abig[:, 1]
    B = _bf(torch.randn(N, K, generator=g) / math.sqrt(K)).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = _bf(torch.randn(M, N, generator=g)).cuda()
    ref = A.float() @ B.float().T
    out = torch.full((M + 1, N), 7.0, dtype=torch.float32, device="cuda")
    hip.gemm(A, B, out, M, N, K, lda=3 * K)
    torch.testing.assert_close(out[:M], ref, rtol=1e-4, atol=1e-4)
    assert bool((out[M] == 7.0).all())                                 # rows >= M untouched
    outb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    hip.gemm(A, B, outb, M, N, K, lda=3 * K, bias=bias, residual=res, act=1, alpha=0.5)
    ref2 = torch.nn.functional.gelu(0.5 * ref + bias) + res.float()
    torch.testing.assert_close(outb.float(), ref2, rtol=1e-2, atol=1e-2)
    try:                                                               # identical math on the 128x128 tile kernel
        hip.gemm_force_variant(1)
        outt = torch.empty(M, N, dtype=torch.float32, device="cuda")
        hip.gemm(A, B, outt, M, N, K, lda=3 * K)
    finally:
        hip.gemm_force_variant(0)
    torch.testing.assert_close(out[:M], outt, rtol=1e-5, atol=1e-5)
    o2 = torch.empty(M, N, dtype=torch.float32, device="cuda")
    hip.gemm(A, B, o2, M, N, K, lda=3 * K)
    assert torch.equal(o2, out[:M])                                    # fixed reduction order


@pytest.mark.parametrize("M,I,K", [(8, 512, 256), (3, 104, 192), (16, 14336, 4096)])
def test_gemm_skinny_fused_swiglu(hip, M, I, K):
    """act=4: concatenated gate|up rows, SwiGLU in the epilogue == skinny GEMM + swiglu kernel bit for bit."""
    g = torch.Generator().manual_seed(M + I)
    x = _bf(torch.randn(M, K, generator=g)).cuda()
    w = _bf(torch.randn(2 * I, K, generator=g) / math.sqrt(K) * 2).cuda()
    gu = torch.empty(M, 2 * I, dtype=torch.bfloat16, device="cuda")
    hip.gemm(x, w, gu, M, 2 * I, K)
    ref = torch.empty(M, I, dtype=torch.bfloat16, device="cuda")
    hip.swiglu_fwd(gu, ref, M, I)
    out = torch.empty(M, I, dtype=torch.bfloat16, device="cuda")
    hip.gemm(x, w, out, M, I, K, act=4)
    assert torch.equal(out, ref)
    try:                                                                # persistent grid: several tiles per block, same bits
        for blocks in (7, 64):
            hip.gemm_set_option(3, blocks)
            o2 = torch.empty_like(out)
            hip.gemm(x, w, o2, M, I, K, act=4)
            g2 = torch.empty_like(gu)
            hip.gemm(x, w, g2, M, 2 * I, K)
            assert torch.equal(o2, ref) and torch.equal(g2, gu)
    finally:
        hip.gemm_set_option(3, 512)


@pytest.mark.parametrize("M,N,K", [(8, 4096, 4096), (3, 100, 512), (16, 1028, 2048), (1, 16, 1024), (8, 28672, 4096), (5, 9000, 1536)])
def test_gemm_skinny_fused_rmsnorm(hip, M, N, K):
    """a_rms_weight: RMSNorm folded into the skinny GEMM == rmsnorm kernel followed by the GEMM, bit for bit
    (plain and fused-SwiGLU epilogues); shapes that do not fit the LDS budget are rejected loudly."""
    g = torch.Generator().manual_seed(M * 5 + N)
    assert hip.rms_fusable(M, K)
    x = _bf(torch.randn(M, K, generator=g) * 3).cuda()
    gamma = (1.0 + 0.1 * torch.randn(K, generator=g)).cuda()
    w = _bf(torch.randn(2 * N, K, generator=g) / math.sqrt(K) * 2).cuda()
    res = _bf(torch.randn(M, N, generator=g)).cuda()
    xn = torch.empty_like(x)
    hip.rmsnorm_fwd(x, gamma, 1e-5, xn, torch.empty(M, device="cuda"))
    ref = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    hip.gemm(xn, w, ref, M, N, K, residual=res)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    hip.gemm(x, w, out, M, N, K, residual=res, a_rms_weight=gamma, a_rms_eps=1e-5)
    assert torch.equal(out, ref)
    if N % 8 == 0:
        ref4 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        hip.gemm(xn, w, ref4, M, N, K, act=4)
        out4 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        hip.gemm(x, w, out4, M, N, K, act=4, a_rms_weight=gamma, a_rms_eps=1e-5)
        assert torch.equal(out4, ref4)
    assert not hip.rms_fusable(16, 4096)
    with pytest.raises(RuntimeError, match="a_rms_weight"):
        big = _bf(torch.randn(16, 4096, generator=g)).cuda()
        hip.gemm(big, w[:N, :K].contiguous() if K == 4096 else _bf(torch.randn(N, 4096, generator=g)).cuda(),
                 torch.empty(16, N, dtype=torch.bfloat16, device="cuda"), 16, N, 4096,
                 a_rms_weight=torch.ones(4096, device="cuda"), a_rms_eps=1e-5)


def test_rope_kv_append(hip):
    """rope + KV-cache append == rope in place followed by copying the K|V columns into the slab (prompt and decode forms)."""
    g = torch.Generator().manual_seed(4)
    B, S, hq, hkv, hd, Smax = 3, 5, 4, 2, 128, 9
    ld = (hq + 2 * hkv) * hd
    qkv = _bf(torch.randn(B * S, ld, generator=g)).cuda()
    fr = torch.outer(torch.arange(Smax, dtype=torch.float32), 1.0 / (10000 ** (torch.arange(0, hd, 2).float() / hd)))
    cs = torch.stack([fr.cos(), fr.sin()], 1).contiguous().cuda()
    shift = torch.tensor([-2, 0, -1], dtype=torch.int32).cuda()
    for qn in (None, torch.rand(hd, generator=g).cuda() + 0.5):
        ref = qkv.clone()
        hip.rope(ref, ld, B * S, S, hq, hkv, hd, cs, qn, qn, 1e-6, pos_shift=shift)
        cache = torch.zeros(B, Smax, 2 * hkv * hd, dtype=torch.bfloat16, device="cuda")
        got = qkv.clone()
        hip.rope_kv_append(got, ld, B * S, S, hq, hkv, hd, cs, qn, qn, 1e-6, shift, cache, cache.stride(0), cache.stride(1), 0)
        assert torch.equal(got, ref)
        assert torch.equal(cache[:, :S], ref.view(B, S, ld)[:, :, hq * hd:])
        assert bool((cache[:, S:] == 0).all())
        # decode form: one row per sequence appended at slot 7
        q1 = _bf(torch.randn(B, ld, generator=g)).cuda()
        r1 = q1.clone()
        sh1 = (7 + shift).contiguous()
        hip.rope(r1, ld, B, 1, hq, hkv, hd, cs, qn, qn, 1e-6, pos_shift=sh1)
        g1 = q1.clone()
        hip.rope_kv_append(g1, ld, B, 1, hq, hkv, hd, cs, qn, qn, 1e-6, sh1, cache, cache.stride(0), cache.stride(1), 7)
        assert torch.equal(g1, r1) and torch.equal(cache[:, 7], r1[:, hq * hd:])
        assert bool((cache[:, 8] == 0).all()) and torch.equal(cache[:, :S], ref.view(B, S, ld)[:, :, hq * hd:])


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (3840, 1280, 2048), (1280, 3072, 2048), (200, 136, 192), (200, 136, 320), (2048, 1280, 3840), (8, 8, 64), (8, 8, 256)])
def test_gemm_transposed_operands(hip, M, N, K):
    """trans_a / trans_b: operands stored [K,M] / [K,N] (autograd's dW = dY^T X, dX = dY W) == the NT kernel on
    materialised transposes, bit for bit (same products, same order), incl. ragged tile edges and padded leading dims."""
    g = torch.Generator().manual_seed(M + 3 * N + K)
    At = _bf(torch.randn(K, M + 8, generator=g)).cuda()                # [K, M] view with lda = M + 8
    Bt = _bf(torch.randn(K, N, generator=g)).cuda()
    A = At[:, :M].t().contiguous()
    B = Bt.t().contiguous()
    bias = torch.randn(N, generator=g).cuda()
    try:
        hip.gemm_force_variant(1)
        hip.gemm_set_option(6, 0)                                      # the double-buffered 128x128 kernel
        ref = torch.empty(M, N, dtype=torch.float32, device="cuda")
        hip.gemm(A, B, ref, M, N, K, bias=bias)
        torch.testing.assert_close(ref, A.float() @ B.float().T + bias, rtol=1e-4, atol=2e-3)
        # ring = 2: the four-slot software-pipelined form of the same tile (what small grids run by default): same
        # accumulation order, so every storage combination is bit-identical in both kernels
        for ring in (0, 2):
            hip.gemm_set_option(6, ring)
            for ta, tb in ((False, False), (True, True), (False, True), (True, False)):
                out = torch.empty(M, N, dtype=torch.float32, device="cuda")
                hip.gemm(At if ta else A, Bt if tb else B, out, M, N, K, bias=bias, trans_a=ta, trans_b=tb, lda=(M + 8) if ta else K)
                assert torch.equal(out, ref), (ring, ta, tb)
    finally:
        hip.gemm_force_variant(0)
        hip.gemm_set_option(6, 1)


@pytest.mark.parametrize("M,N,K", [(256, 512, 64), (256, 512, 128), (300, 520, 192), (512, 768, 576), (2304, 1280, 1280), (5120, 4096, 2048), (700, 4096, 4096)])
def test_gemm_256_tail_skip_is_bit_identical(hip, M, N, K):
    """Round 4: the two-phase 256x256 kernel stops its LDS-DMA half-tile stream at the last K-tile (counted `s_waitcnt vmcnt` shrinks
    8 -> 2 -> 0 over the last three slots) instead of re-loading dead slots to keep the count constant (option 10).  Same arithmetic in
    the same order: bit-identical to the dummy-load form and to the four-phase kernel, for one to many K-tiles, ragged edges, and a
    grid with a split-K tail (5120 x 4096: 64 tiles in 4 K-slices of 8 K-tiles); repeated runs agree (a mis-counted wait would race)."""
    g = torch.Generator().manual_seed(M + 7 * N + K)
    A = _bf(torch.randn(M, K, generator=g)).cuda()
    B = _bf(torch.randn(N, K, generator=g)).cuda()
    outs = {}
    try:
        for v, skip in ((2, 1), (6, 0), (6, 1), (7, 1)):
            hip.gemm_force_variant(v)
            hip.gemm_set_option(10, skip)
            o = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
            for _ in range(4):
                hip.gemm(A, B, o, M, N, K)
            outs[(v, skip)] = o.clone()
    finally:
        hip.gemm_force_variant(0)
        hip.gemm_set_option(10, 1)
    for k, o in outs.items():
        assert torch.equal(o, outs[(2, 1)]), k
    assert torch.isfinite(outs[(6, 1)]).all()


def test_gemm_256_identity_and_k64(hip):
    K = 64                                        # a single K-tile: the whole loop is prologue + dummy tail loads
    A = _bf(torch.eye(256)[:, :K]).cuda()
    B = _bf(torch.arange(512 * K).reshape(512, K).float() % 251 - 100).cuda()
    out = torch.empty(256, 512, dtype=torch.float32, device="cuda")
    try:
        hip.gemm_force_variant(2)
        hip.gemm(A, B, out, 256, 512, K)
    finally:
        hip.gemm_force_variant(0)
    ref = A.float() @ B.float().T
    torch.testing.assert_close(out, ref, rtol=0, atol=0)


def test_gemm_asymmetric_identity(hip):
    """A = I with an asymmetric B catches a swapped C-write or fragment map."""
    K = 128
    A = _bf(torch.eye(K)).cuda()
    B = _bf(torch.arange(256 * K).reshape(256, K).float() % 251 - 100).cuda()
    out = torch.empty(K, 256, dtype=torch.float32, device="cuda")
    hip.gemm(A, B, out, K, 256, K)
    torch.testing.assert_close(out, B.float().T, rtol=0, atol=0)


def test_gemm_epilogues_and_batch(hip):
    g = torch.Generator().manual_seed(5)
    M, N, K, nb = 192, 256, 128, 3
    A = _bf(torch.randn(nb, M, K, generator=g)).cuda()
    B = _bf(torch.randn(N, K, generator=g) / math.sqrt(K)).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res32 = torch.randn(nb, M, N, generator=g).cuda()
    resb = _bf(torch.randn(M, N, generator=g)).cuda()
    pre_ref = A.float() @ B.float().T * 0.5 + bias
    # gelu + fp32 residual (batched) + preact copy
    out = torch.empty(nb, M, N, dtype=torch.float32, device="cuda")
    pre = torch.empty(nb, M, N, dtype=torch.bfloat16, device="cuda")
    hip.gemm(A, B, out, M, N, K, bias=bias, residual=res32, act=1, preact=pre, alpha=0.5, batch=nb,
             stride_a=M * K, stride_c=M * N, stride_r=M * N, stride_p=M * N)
    ref = torch.nn.functional.gelu(pre_ref) + res32
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(pre.float(), pre_ref, rtol=1e-2, atol=1e-2)
    # bf16 residual shared across the batch (stride 0), bf16 out
    outb = torch.empty(nb, M, N, dtype=torch.bfloat16, device="cuda")
    hip.gemm(A, B, outb, M, N, K, bias=bias, residual=resb, alpha=0.5, batch=nb, stride_a=M * K, stride_c=M * N)
    torch.testing.assert_close(outb.float(), pre_ref + resb.float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("variant", [1, 2])
def test_gemm_gelu_epilogue_within_one_ulp(hip, variant):
    """bf16-output GELU(erf) epilogue (Whisper fc1 / conv stem, Q-Former FFN; TF:activations.py GELUActivation = erf form), the A&S 7.1.26
    form: within ONE bf16 ulp of gelu(pre-activation) evaluated in float64 and then rounded, and on the exactly rounded value for
    > 97 % of the elements; pre-activations cover |x| up to ~9.  (Round 4's second, packed-polynomial form measured equal in the step
    and was removed: its code in every epilogue fragment slowed the plain-store path of the same kernel.)"""
    g = torch.Generator().manual_seed(77)
    M, N, K = 512, 768, 256
    A = _bf(torch.randn(M, K, generator=g) * 0.5).cuda()
    B = _bf(torch.randn(N, K, generator=g) / 4).cuda()
    bias = torch.randn(N, generator=g).cuda()
    pre = (A.float() @ B.float().T + bias).double()
    ref = (0.5 * pre * (1 + torch.erf(pre / math.sqrt(2)))).float().to(torch.bfloat16)
    assert float(pre.abs().max()) > 6.0
    try:
        hip.gemm_force_variant(variant)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        hip.gemm(A, B, out, M, N, K, bias=bias, act=1)
        ulp = (out.view(torch.int16).int() - ref.view(torch.int16).int()).abs()
        big = ref.float().abs() >= 2.0 ** -8                                # (the negative tail, |gelu| < 2^-8: compared absolutely; one bf16 ulp there is <= 1.5e-5)
        assert int(ulp[big].max()) <= 1, int(ulp[big].max())
        assert float((out.float() - ref.float()).abs()[~big].max()) < 2e-5
        assert float((ulp[big] != 0).float().mean()) < 0.03, float((ulp[big] != 0).float().mean())
    finally:
        hip.gemm_force_variant(0)


def _gu_block_perm(I):
    b = torch.arange(I // 32).view(-1, 1) * 32
    r = torch.arange(32).view(1, -1)
    return torch.cat([b + r, I + b + r], dim=1).reshape(-1)


@pytest.mark.parametrize("variant,M", [(1, 384), (1, 200), (1, 8), (2, 384), (2, 777), (0, 1100)])
def test_gemm_fused_swiglu_epilogues(hip, variant, M):
    """act=2 / act=3 (ABI 6: 64-column gate|up BLOCKS, 32 gate + the 32 matching up columns) == unfused GEMM + swiglu kernels:
    same rounding points, <= 1 bf16 ulp (fma contraction may differ between two kernels); the saved projection is bit-identical to
    the plain GEMM's.  Both tile kernels (variant 1 = 128x128 incl. its ring form, 2 = 256x256), M not a multiple of the tile
    (rows >= M take part in the lane exchange and skip the stores), M <= 16 (must not fall to the weight-streaming kernel)."""
    g = torch.Generator().manual_seed(21 + M)
    I, h = 512, 256
    perm = _gu_block_perm(I)
    x = _bf(torch.randn(M, h, generator=g)).cuda()
    wg = _bf(torch.randn(I, h, generator=g) / 16).cuda()
    wu = _bf(torch.randn(I, h, generator=g) / 16).cuda()
    w_cat = torch.cat([wg, wu], 0).contiguous()
    w_blk = w_cat[perm.cuda()].contiguous()
    try:
        hip.gemm_force_variant(variant)
        gu_b = torch.full((M + 3, 2 * I), 7.0, dtype=torch.bfloat16, device="cuda")
        act = torch.full((M + 3, I), 7.0, dtype=torch.bfloat16, device="cuda")
        hip.gemm(x, w_blk, gu_b, M, 2 * I, h, act=2, aux=act, ld_aux=I)
        assert bool((gu_b[M:] == 7).all()) and bool((act[M:] == 7).all())                      # nothing written past row M
        hip.gemm_force_variant(0 if variant == 0 else variant)
        gu = torch.empty(M, 2 * I, dtype=torch.bfloat16, device="cuda")
        hip.gemm(x, w_cat, gu, M, 2 * I, h)
        act_ref = torch.empty(M, I, dtype=torch.bfloat16, device="cuda")
        hip.swiglu_fwd(gu, act_ref, M, I)
        assert torch.equal(gu_b[:M], gu[:, perm.cuda()])                                        # the projection itself: bit-identical
        torch.testing.assert_close(act[:M].float(), act_ref.float(), rtol=8e-3, atol=1e-3)      # <= 1 bf16 ulp
        assert float((act[:M] != act_ref).float().mean()) < 0.02                                # ... and almost everywhere equal
        # backward: d_act = dy @ Wd^T fused with d(gate|up)
        dy = _bf(torch.randn(M, h, generator=g)).cuda()
        wdT = _bf(torch.randn(I, h, generator=g) / 16).cuda()
        dgu_b = torch.full((M + 3, 2 * I), 7.0, dtype=torch.bfloat16, device="cuda")
        hip.gemm(dy, wdT, dgu_b, M, I, h, ldc=2 * I, act=3, aux=gu_b, ld_aux=2 * I)
        assert bool((dgu_b[M:] == 7).all())
        dact = torch.empty(M, I, dtype=torch.bfloat16, device="cuda")
        hip.gemm(dy, wdT, dact, M, I, h)
        dgu = torch.empty(M, 2 * I, dtype=torch.bfloat16, device="cuda")
        hip.swiglu_bwd(gu, dact, dgu, M, I)
        ref_b = dgu[:, perm.cuda()]
        torch.testing.assert_close(dgu_b[:M].float(), ref_b.float(), rtol=8e-3, atol=1e-3)
        assert float((dgu_b[:M] != ref_b).float().mean()) < 0.02
    finally:
        hip.gemm_force_variant(0)


def test_gemm_overlapping_rows_im2col(hip):
    """lda < K: rows of A overlap (zero-copy im2col of a k=3 stride-2 conv over channel-last input)."""
    g = torch.Generator().manual_seed(9)
    T, Cc, Co = 64, 64, 128
    x = _bf(torch.randn(2 * T + 1, Cc, generator=g)).cuda()          # padded row 0 .. 2T
    W = _bf(torch.randn(Co, 3 * Cc, generator=g) / math.sqrt(3 * Cc)).cuda()
    out = torch.empty(T, Co, dtype=torch.float32, device="cuda")
    hip.gemm(x, W, out, T, Co, 3 * Cc, lda=2 * Cc)
    cols = torch.stack([x[2 * t:2 * t + 3].reshape(-1) for t in range(T)]).float()
    torch.testing.assert_close(out, cols @ W.float().T, rtol=1e-4, atol=1e-4)


def test_gemm_rejects_bad_shapes(hip):
    A = torch.zeros(64, 96, dtype=torch.bfloat16, device="cuda")
    out = torch.zeros(64, 64, dtype=torch.float32, device="cuda")
    with pytest.raises(RuntimeError, match="multiple of 64"):
        hip.gemm(A, A, out, 64, 64, 96)


@pytest.mark.parametrize("n_mels", [80, 128])
def test_logmel_matches_oracle(hip, n_mels):
    g = torch.Generator().manual_seed(1234)
    wave = (0.1 * torch.randn(3, 480000, generator=g)).clamp(-1, 1)
    wave[1, 200000:] = 0.0                         # long silence: exercises the max-8 clamp
    wave[2] *= torch.linspace(0, 1, 480000)
    ref = O.logmel(wave, n_mels)
    out = hip.logmel(wave.cuda(), n_mels).cpu()
    assert out.shape == (3, n_mels, 3000)
    torch.testing.assert_close(out, ref, rtol=0, atol=2e-4)          # fp32, tolerance stated by the HF docstring: 1e-5..1e-4


def test_logmel_short_and_long_clips(hip):
    g = torch.Generator().manual_seed(7)
    short = 0.1 * torch.randn(2, 16000 * 3 + 17, generator=g)
    torch.testing.assert_close(hip.logmel(short.cuda(), 128).cpu(), O.logmel(short, 128), rtol=0, atol=2e-4)
    long = 0.1 * torch.randn(1, 480000 + 5000, generator=g)
    torch.testing.assert_close(hip.logmel(long.cuda(), 128).cpu(), O.logmel(long, 128), rtol=0, atol=2e-4)


def _run_adafactor_case(hip, shapes, steps, gscale):
    from desta.optim import ParamArena, FusedAdafactor
    g = torch.Generator().manual_seed(3)
    names = [f"t{i}.{'bias' if len(s) == 1 else 'weight'}" for i, s in enumerate(shapes)]
    arena = ParamArena(list(zip(names, shapes)), "cuda")
    p_o = []
    for n, s in zip(names, shapes):
        v = torch.randn(*s, generator=g)
        arena.param(n).copy_(v)
        p_o.append(v.clone())
    opt = FusedAdafactor(arena, weight_decay=0.01)
    wd = [0.01 if m else 0.0 for m in O.decay_mask(names)]
    st = O.adafactor_init(p_o)
    p_32 = [x.clone() for x in p_o]                 # second oracle copy driven by the plain fp32 norm
    st_32 = O.adafactor_init(p_32)
    for step in range(steps):
        grads = [torch.randn(*s, generator=g) * gscale[step % len(gscale)] for s in shapes]
        for n, gr in zip(names, grads):
            arena.grad(n).copy_(gr)
        lr = O.linear_warmup_lr(step + 1, 1e-2, 3, 20)
        opt.step(lr)
        go = [x.clone() for x in grads]
        O.clip_grad_norm(go, 1.0, f64=True)
        O.adafactor_step(p_o, go, st, lr, wd, f64_stats=True)     # exact-reduction oracle (tight pin)
        go32 = [x.clone() for x in grads]
        n_o = O.clip_grad_norm(go32, 1.0)
        O.adafactor_step(p_32, go32, st_32, lr, wd)               # all-fp32 oracle == HF arithmetic (loose: its own
                                                                  # fp32 reductions over 4 M elements are ~1e-4 off)
        # fp32 CPU norms of multi-million-element tensors carry ~1e-4 relative summation error themselves:
        # pin the kernel's norm against a float64 norm (tight) and the fp32 oracle norm (loose)
        n64 = math.sqrt(sum(float((x.double() ** 2).sum()) for x in grads))
        assert abs(float(opt.grad_norm()) - n64) <= 2e-6 * max(1.0, n64)
        assert abs(float(opt.grad_norm()) - float(n_o)) <= 5e-4 * max(1.0, float(n_o))
        for n, ref, ref32 in zip(names, p_o, p_32):
            torch.testing.assert_close(arena.param(n).cpu(), ref, rtol=2e-6, atol=3e-7)
            torch.testing.assert_close(arena.param(n).cpu(), ref32, rtol=0, atol=1e-5)
    return arena, opt


def test_adafactor_matches_oracle_small(hip):
    _run_adafactor_case(hip, [(1, 16, 24), (24, 40), (40,), (16, 4), (7,), (130, 260)], steps=5, gscale=[0.05, 10.0, 1e-3])


def test_adafactor_connector_shapes(hip):
    """Shapes of the real connector tensors (largest: 1280x3072, 4096x1280) incl. the 3-D prompts."""
    shapes = [(64, 4), (1, 64, 1280), (3072, 1280), (3072,), (1280, 3072), (1280,), (4096, 1280), (4096,)]
    _, opt = _run_adafactor_case(hip, shapes, steps=2, gscale=[0.02, 3.0])
    assert opt.plan.n_groups >= 1 and opt.plan.max_chunks_per_tensor == 320


def test_adafactor_ragged_rows_take_the_unit_kernels(hip):
    """cols % 4 != 0 (no 16-B row accesses): such tensors carry no chunks and are updated by the unit-based kernels over their own
    unit range (ABI 7: `ragged_units`), the others stay on the chunk kernels; same numbers.  Cases: a mix, a plan whose factored
    tensors are ALL ragged (no chunk launch at all, 1-D tensors on their own launch), and a Conv1d-like [out, in, 5] weight
    (ORCA's local_conv: batch = out, 5 columns, hundreds of units whose sums `k34_totals_range` adds up once per tensor)."""
    _, opt = _run_adafactor_case(hip, [(33, 7), (5, 3, 9), (11,), (40, 64)], steps=3, gscale=[0.5, 4.0])
    assert opt.plan.cols_multiple_of_4 == 1 and opt.plan.n_ragged == 2 and opt.plan.n_chunks == 1
    _, opt = _run_adafactor_case(hip, [(33, 7), (9,), (5, 3, 9)], steps=3, gscale=[0.5, 4.0])
    assert opt.plan.n_ragged == 2 and opt.plan.n_chunks == 0
    _, opt = _run_adafactor_case(hip, [(64, 4), (192, 130, 5), (192,), (256, 192)], steps=2, gscale=[0.05, 3.0])
    assert opt.plan.n_ragged == 1 and opt.plan.n_units == 1 + 192 + 4          # one unit per [130, 5] batch item of the ragged tensor
    # ragged AND wider than the one-thread-per-row form (<= 16 columns): the wave-per-row scalar form
    _, opt = _run_adafactor_case(hip, [(20, 30), (3, 70, 259), (5,), (8, 12)], steps=2, gscale=[0.05, 3.0])
    assert opt.plan.n_ragged == 2


def test_adafactor_many_chunk_tensor_and_full_arena_order(hip):
    """A tensor of 321 chunks between small ones, several launch groups in the reverse arena order; 2 steps against the
    exact-reduction oracle."""
    _, opt = _run_adafactor_case(hip, [(8, 64), (4100, 1280), (1280,), (2, 16, 1280), (257, 516)], steps=2, gscale=[0.03, 2.0])
    assert opt.plan.cols_multiple_of_4 == 1 and opt.plan.max_chunks_per_tensor == 321


def test_adafactor_is_deterministic(hip):
    a1, _ = _run_adafactor_case(hip, [(300, 520), (520,)], steps=3, gscale=[1.0])
    a2, _ = _run_adafactor_case(hip, [(300, 520), (520,)], steps=3, gscale=[1.0])
    assert torch.equal(a1.params, a2.params)
