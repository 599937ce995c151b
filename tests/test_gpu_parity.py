"""-m gpu: the HIP path (through the C ABI) against the CPU oracle and the golden
fixtures.  Bit-exact for every integer/byte/fp16 output; the two fp32 linear-algebra
stages (H accumulate, Cholesky chain) carry their tolerance in the test.
"""
import os

import numpy as np
import pytest

from conftest import load_golden, triu_unpack
from ggml_spec import unpack

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

TYPES = {"Q2_K": 10, "Q3_K": 11, "Q4_K": 12, "Q5_K": 13, "Q6_K": 14}


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from gptq_gguf_toolkit_amd import ops as _ops
    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def u16(t):
    return t.cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def npy(t):
    return t.cpu().numpy()


def f16t(bits):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.float16).cuda()


# ----------------------------------------------------------------- K4 scale search
@pytest.mark.parametrize("name", list(TYPES))
def test_scale_search_golden(ops, name):
    g = load_golden("g2_scale_search")
    d, s, dmin, m = ops.scale_search(dev(g[f"{name}_x"]), TYPES[name])
    assert np.array_equal(u16(d), g[f"{name}_ieee_d"]) and np.array_equal(u16(dmin), g[f"{name}_ieee_dmin"])
    assert np.array_equal(npy(s), g[f"{name}_ieee_s"]) and np.array_equal(npy(m), g[f"{name}_ieee_m"])


@pytest.mark.parametrize("name", list(TYPES))
@pytest.mark.parametrize("rows", [1, 5, 1003])
def test_scale_search_vs_oracle_ragged(ops, oracle, name, rows):
    rng = np.random.default_rng(rows * 31 + TYPES[name])
    x = (rng.standard_normal((rows, 256)) * 0.02).astype(np.float32)
    x[0, :40] = 0.0
    if rows > 4:
        x[3] = np.abs(x[3])
        x[4] *= 1e-3
    # strided view: panel inside a wider matrix
    wide = np.zeros((rows, 768), np.float32)
    wide[:, 256:512] = x
    xt = dev(wide)[:, 256:512]
    d, s, dmin, m = ops.scale_search(xt, TYPES[name])
    od, os_, odm, om = oracle.scale_search(x, TYPES[name])
    assert np.array_equal(u16(d), od) and np.array_equal(u16(dmin), odm)
    assert np.array_equal(npy(s), os_) and np.array_equal(npy(m), om)


def test_scale_search_params(ops, oracle):
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((64, 256)) * 0.05).astype(np.float32)
    for kw in (dict(rmin=-0.5, rdelta=0.05, nstep=10), dict(nstep=0), dict(rmin=-2.0, rdelta=0.2, nstep=23)):
        d, s, dmin, m = ops.scale_search(dev(x), 12, **kw)
        od, os_, odm, om = oracle.scale_search(x, 12, **kw)
        assert np.array_equal(u16(d), od) and np.array_equal(npy(s), os_)
        assert np.array_equal(u16(dmin), odm) and np.array_equal(npy(m), om)


# ------------------------------------------------------------- K5/K6 GPTQ step
def _g6_tags():
    g = load_golden("g6_g7_step_and_pack")
    return sorted({k.rsplit("_", 1)[0] for k in g.files if k.endswith("_q") and "mklsqrt" not in k})


@pytest.mark.parametrize("tag", _g6_tags())
def test_gptq_step_golden(ops, tag):
    """(W, U) -> 5-tuple, bit-identical to what the reference produced (G6) + packed bytes (G7)."""
    g = load_golden("g6_g7_step_and_pack")
    case, n1, n2, b, s_ = tag.split("_")
    name = f"{n1}_{n2}"
    block = None if b == "bNone" else int(b[1:])
    W0 = g[f"{case}_W0"]
    U = triu_unpack(g[f"{case}_U_triu"], W0.shape[1])
    W = dev(W0)
    q, d, s, dmin, m = ops.gptq_quantize(W, dev(U), TYPES[name], block_size=block, static_groups=(s_ == "s1"))
    assert np.array_equal(npy(q), g[f"{tag}_q"]), f"{(npy(q) != g[f'{tag}_q']).mean():.4%} ints differ"
    assert np.array_equal(u16(d), g[f"{tag}_d"]) and np.array_equal(u16(dmin), g[f"{tag}_dmin"])
    assert np.array_equal(npy(s), g[f"{tag}_s"]) and np.array_equal(npy(m), g[f"{tag}_m"])
    # gptq.py:266: W now holds the dequantized matrix
    deq = ops.dequantize(TYPES[name], q, d, s, dmin, m)
    assert torch.equal(W, deq) or np.array_equal(npy(W), npy(deq))
    if f"{tag}_packed" in g.files:
        assert np.array_equal(npy(ops.pack(TYPES[name], q, d, s, dmin, m)), g[f"{tag}_packed"])


@pytest.mark.parametrize("tag", ["Q4_K_b128", "Q2_K_b128", "Q6_K_b64", "Q5_K_b128"])
def test_gptq_step_act_order_golden(ops, tag):
    """act_order (gptq.py:208-216): permuted (W, U) + static scales of the original groups -> the reference's ints."""
    g = load_golden("g11_act_order")
    name, b = tag[:4], int(tag.split("_b")[1])
    t = TYPES[name]
    W0, perm = g["W0"], g["perm"]
    U = triu_unpack(g["U_triu"], W0.shape[1])
    _, d, s, dmin, m = ops.rtn_quantize(dev(W0), t)  # static scales (gptq.py:184-196) == fp32 RTN search
    assert np.array_equal(u16(d), g[f"{tag}_d"]) and np.array_equal(npy(s), g[f"{tag}_s"])
    Wp = dev(np.ascontiguousarray(W0[:, perm]))
    qp = ops.gptq_quantize_perm(Wp, dev(U), t, torch.from_numpy(perm).cuda(), d, s, dmin, m, block_size=b)
    q = npy(qp)[:, np.argsort(perm)]
    assert np.array_equal(q, g[f"{tag}_q"]), f"{(q != g[f'{tag}_q']).mean():.4%} ints differ"
    deq = npy(ops.dequantize(t, dev(q), d, s, dmin, m))
    assert np.array_equal(npy(Wp)[:, np.argsort(perm)], deq)


@pytest.mark.parametrize("name,R,C,block", [("Q4_K", 96, 1024, 128), ("Q2_K", 200, 512, 128), ("Q3_K", 64, 768, 64),
                                            ("Q5_K", 130, 512, 256), ("Q6_K", 64, 512, 96), ("Q4_K", 64, 768, 32),
                                            # C > 1024: the chained far update (whole-tile kernel at R = 128,
                                            # predicated kernel at R = 72)
                                            ("Q4_K", 128, 2304, 128), ("Q6_K", 72, 2048, 128)])
def test_gptq_step_vs_oracle(ops, oracle, name, R, C, block):
    rng = np.random.default_rng(R + C)
    W0 = (rng.standard_normal((R, C)) * 0.02).astype(np.float16).astype(np.float32)
    X = (rng.standard_normal((2 * C, C)) * np.exp(rng.standard_normal(C) * 0.5)).astype(np.float32)
    H = oracle.h_accumulate(np.zeros((C, C), np.float32), X, 0.0, 2.0 / 4)
    U, _, W1, bad = oracle.h_prepare(H, W0, 0.01)
    assert not bad
    W = dev(W1)
    q, d, s, dmin, m = ops.gptq_quantize(W, dev(U), TYPES[name], block_size=block)
    Wd, oq, od, os_, odm, om = oracle.gptq_step(W1, U, TYPES[name], block_size=block)
    assert np.array_equal(npy(q), oq), f"{(npy(q) != oq).mean():.4%} ints differ"
    assert np.array_equal(u16(d), od) and np.array_equal(u16(dmin), odm)
    assert np.array_equal(npy(s), os_) and np.array_equal(npy(m), om)
    assert np.array_equal(npy(W), Wd)


@pytest.mark.parametrize("R", [256, 200])
def test_gptq_lookahead_equals_per_block_updates(ops, R):
    """The look-ahead schedule (near updates inside a 1024-column super-block, ONE chained GEMM for all later
    columns) performs, per element, the same subtractions of the same k-ordered products in the same order as
    gptq.py:270 applied block by block (option no_lookahead): every output and the final W are bit-identical.
    R = 256 takes the whole-tile chained kernel, R = 200 the predicated one; C = 3328 = 3.25 super-blocks."""
    torch.manual_seed(R)
    C = 3328
    W0 = (torch.randn(R, C, device="cuda") * 0.02).half().float()
    X = (torch.randn(2 * C, C, device="cuda") * torch.exp(torch.randn(C, device="cuda") * 0.5)).half()
    H = torch.zeros(C, C, device="cuda")
    ops.h_accumulate(H, X, 0.0, 2.0 / 4)
    Wp = W0.clone()
    U, flag = ops.h_prepare(H, Wp, 0.01)
    assert int(flag.item()) == 0
    outs = []
    for off in (0, 1):
        with ops.options(no_lookahead=off):
            W = Wp.clone()
            outs.append((W,) + tuple(ops.gptq_quantize(W, U, 12, 128)))
    for a, b in zip(*outs):
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))


def test_h_prepare_never_reads_unwritten_scratch(ops):
    """gq_h_prepare does not clear its scratch: the inverse factor X is written block by block (diagonal blocks
    whole, with their zeros) and the k-range skips of the GEMMs never leave the written blocks.  With X filled with
    NaN patterns beforehand (option chol_poison; r03: A above its block diagonal too) the result must be the same, bit for
    bit, at a size that takes every level of the recursion below the image GEMMs (fp32 and split-bf16 GEMMs, 64- and
    128-tiles; the image levels: tests/test_gpu_round3.py)."""
    torch.manual_seed(4)
    C = 4096 + 896
    X = (torch.randn(2 * C, C, device="cuda") * torch.exp(torch.randn(C, device="cuda") * 0.5)).half()
    H = torch.zeros(C, C, device="cuda")
    ops.h_accumulate(H, X, 0.0, 2.0 / 4)
    del X
    W = torch.randn(64, C, device="cuda")
    U0, f0 = ops.h_prepare(H.clone(), W.clone(), 0.01)
    with ops.options(chol_poison=1):
        U1, f1 = ops.h_prepare(H.clone(), W.clone(), 0.01)
    assert int(f0.item()) == 0 and int(f1.item()) == 0
    assert bool(torch.isfinite(U1).all()) and torch.equal(U0, U1)


def test_far_update_next_to_the_loop_changes_nothing(ops):
    """Many super-blocks, few rows: the far update of a super-block is cut by 1024-column groups -- the next group on
    the caller's stream, the rest on the library's helper stream as persistent launches (gq_gptq.hip) -- and runs next
    to the column loop.  Per element the same subtractions in the same order: bit-identical to the one-stream
    schedule (option far_sync) and to gptq.py:270 block by block (option no_lookahead); C = 9472 = 9.25 super-blocks.
    Of two such calls enqueued back to back on two streams only the first holds the helper (one holder per device at
    a time; the other runs the one-stream schedule): neither disturbs the other."""
    torch.manual_seed(9)
    R, C = 384, 9472
    assert ops.uses_helper_stream(R, C, 128) and not ops.uses_helper_stream(R, 4096, 128)
    W0 = (torch.randn(R, C, device="cuda") * 0.02).half().float()
    X = (torch.randn(C + 512, C, device="cuda") * torch.exp(torch.randn(C, device="cuda") * 0.5)).half()
    H = torch.zeros(C, C, device="cuda")
    ops.h_accumulate(H, X, 0.0, 2.0 / 4)
    del X
    Wp = W0.clone()
    U, flag = ops.h_prepare(H, Wp, 0.01)
    assert int(flag.item()) == 0
    outs = []
    for kv in ({}, {"far_wgs": 24}, {"far_sync": 1}, {"no_lookahead": 1}):
        with ops.options(**kv):
            assert ops.uses_helper_stream(R, C, 128) == (not kv or "far_wgs" in kv)
            W = Wp.clone()
            outs.append((W,) + tuple(ops.gptq_quantize(W, U, 12, 128)))
    torch.cuda.synchronize()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))
    # two chains at once
    Wb = (Wp * 1.5).contiguous()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        Wa = Wp.clone()
        ra = ops.gptq_quantize(Wa, U, 12, 128)
    with torch.cuda.stream(s2):
        Wc = Wb.clone()
        rb = ops.gptq_quantize(Wc, U, 14, 128)
    torch.cuda.synchronize()
    with ops.options(far_sync=1):
        Wd = Wb.clone()
        rd = ops.gptq_quantize(Wd, U, 14, 128)
    assert torch.equal(Wa, outs[0][0]) and all(torch.equal(a, b) for a, b in zip(ra, outs[0][1:]))
    assert torch.equal(Wc, Wd) and all(torch.equal(a, b) for a, b in zip(rb, rd))


def test_gptq_bad_args(ops):
    from gptq_gguf_toolkit_amd import GQError
    W = torch.zeros(64, 300, device="cuda")
    with pytest.raises(GQError, match="256"):
        ops.gptq_quantize(W, torch.eye(300, device="cuda"), 12)
    W = torch.zeros(64, 512, device="cuda")
    with pytest.raises(GQError, match="multiple of 16"):
        ops.gptq_quantize(W, torch.eye(512, device="cuda"), 12, block_size=100)
    with pytest.raises(GQError):
        ops.gptq_quantize(W, torch.eye(512, device="cuda"), 99)


def test_trailing_update(ops):
    torch.manual_seed(0)
    for M, N, K in ((128, 384, 128), (200, 132, 64), (64, 4, 128), (33, 260, 96)):  # ld % 4 == 0 (ABI)
        A = torch.randn(M, K, device="cuda")
        B = torch.randn(K, N, device="cuda")
        C0 = torch.randn(M, N, device="cuda")
        C = C0.clone()
        ops.trailing_update(C, A, B)
        ref = C0.double() - A.double() @ B.double()
        assert (C.double() - ref).abs().max().item() < 1e-4 * K ** 0.5


# --------------------------------------------------------- K7/K8/K9-13 codecs
@pytest.mark.parametrize("name", list(TYPES))
def test_rtn_dequant_pack_golden(ops, name):
    g = load_golden("g8_g9_rtn_dequant")
    t = TYPES[name]
    q, d, s, dmin, m = ops.rtn_quantize(dev(g["W"]), t)
    assert np.array_equal(npy(q), g[f"{name}_q"]) and npy(q).dtype == g[f"{name}_q"].dtype
    assert np.array_equal(u16(d), g[f"{name}_d"]) and np.array_equal(u16(dmin), g[f"{name}_dmin"])
    assert np.array_equal(npy(s), g[f"{name}_s"]) and np.array_equal(npy(m), g[f"{name}_m"])
    deq = ops.dequantize(t, q, d, s, dmin, m)
    assert np.array_equal(npy(deq).view(np.uint32), g[f"{name}_deq"].view(np.uint32))
    for dt in (torch.float16, torch.bfloat16):  # the caller's cast (quantizer.py:264)
        assert torch.equal(ops.dequantize(t, q, d, s, dmin, m, out_dtype=dt), deq.to(dt))
    q0, s0 = q.clone(), s.clone()
    packed = ops.pack(t, q, d, s, dmin, m)
    assert np.array_equal(npy(packed), g[f"{name}_packed"])
    assert torch.equal(q, q0) and torch.equal(s, s0)  # inputs untouched (unlike pack_Q3K/pack_Q6K)


@pytest.mark.parametrize("name", list(TYPES))
def test_pack_ragged_and_roundtrip(ops, oracle, name):
    """block counts that are not multiples of the 8-block staging unit + ggml-spec round trip."""
    t = TYPES[name]
    ti = oracle.type_info(t)
    rng = np.random.default_rng(t)
    for R, C in ((1, 256), (3, 768), (37, 512)):
        q = rng.integers(ti["qmin"], ti["qmax"] + 1, (R, C)).astype(np.int8 if ti["is_signed"] else np.uint8)
        sdt = np.int8 if ti["is_signed"] else np.uint8
        s = rng.integers(0, ti["scale_maxq"] + 1, (R, C // ti["group"])).astype(sdt)
        m = (rng.integers(0, ti["scale_maxq"] + 1, (R, C // ti["group"])) * ti["k_search"]).astype(sdt)
        d = rng.integers(0, 0x7BFF, (R, C // 256)).astype(np.uint16)
        dmin = (rng.integers(0, 0x7BFF, (R, C // 256)) * ti["k_search"]).astype(np.uint16)
        packed = npy(ops.pack(t, dev(q), f16t(d), dev(s), f16t(dmin), dev(m)))
        assert np.array_equal(packed, oracle.pack(t, q, d, s, dmin, m))
        codes, d2, sc2, dm2, mn2 = unpack(t, packed)
        assert np.array_equal(codes, q.astype(np.int32)) and np.array_equal(sc2, s.astype(np.int32))
        assert np.array_equal(d2, d) and np.array_equal(dm2, dmin) and np.array_equal(mn2, m.astype(np.int32))


@pytest.mark.parametrize("tag,dt", [("f16", torch.float16), ("bf16", torch.bfloat16)])
def test_rtn_model_dtype_golden(ops, oracle, tag, dt):
    """embed/lm_head RTN with the weight in fp16/bf16: the reference runs make_*quants in the model dtype
    (quantizer.py:109,195); G9 holds its outputs, the kernel rounds after every op like ATen does."""
    g = load_golden("g8_g9_rtn_dequant")
    W = dev(g[f"W_{tag}"]).to(dt)  # values are exactly representable
    names = sorted({k.split("_", 1)[1].rsplit("_", 1)[0] for k in g.files if k.startswith(f"{tag}_") and k.endswith("_q")})
    assert "Q4_K" in names and "Q6_K" in names
    for name in names:
        t = TYPES[name]
        q, d, s, dmin, m = ops.rtn_quantize(W, t)
        assert np.array_equal(npy(q), g[f"{tag}_{name}_q"]), f"{tag} {name}: {(npy(q) != g[f'{tag}_{name}_q']).mean():.4%}"
        assert np.array_equal(u16(d), g[f"{tag}_{name}_d"]) and np.array_equal(u16(dmin), g[f"{tag}_{name}_dmin"])
        assert np.array_equal(npy(s), g[f"{tag}_{name}_s"]) and np.array_equal(npy(m), g[f"{tag}_{name}_m"])
        oq, od, os_, odm, om = oracle.rtn_quantize_lp(g[f"W_{tag}"], 1 if tag == "f16" else 2, t)
        assert np.array_equal(oq, npy(q)) and np.array_equal(od, u16(d)) and np.array_equal(os_, npy(s))


@pytest.mark.parametrize("dt,rmode", [(torch.bfloat16, 2), (torch.float16, 1)])
def test_rtn_embedding_size(ops, oracle, dt, rmode):
    """Llama-3 embed_tokens / lm_head shape (128256 x 4096) in the model dtype, Q6_K (README mixed map) and Q4_K:
    row blocks sampled across the matrix (first, last, odd offsets) equal the per-op-rounding oracle bit for bit;
    the dequantized matrix is within the type's relative error of the weights."""
    torch.manual_seed(21)
    R, C = 128256, 4096
    W = (torch.randn(R, C, device="cuda") * 0.02).to(dt)
    rows = [0, 1, 63, 64, 4097, 65535, 100001, R - 65, R - 1]
    for name in ("Q6_K", "Q4_K"):
        t = TYPES[name]
        q, d, s, dmin, m = ops.rtn_quantize(W, t)
        Wn = W[rows].float().cpu().numpy()
        oq, od, os_, odm, om = oracle.rtn_quantize_lp(Wn, rmode, t)
        assert np.array_equal(oq, npy(q[rows])) and np.array_equal(od, u16(d[rows])) and np.array_equal(os_, npy(s[rows]))
        assert np.array_equal(odm, u16(dmin[rows])) and np.array_equal(om, npy(m[rows]))
        deq = ops.dequantize(t, q[:4096], d[:4096], s[:4096], dmin[:4096], m[:4096])
        rel = ((deq - W[:4096].float()).norm() / W[:4096].float().norm()).item()
        assert rel < (0.10 if name == "Q4_K" else 0.04), (name, rel)


# ------------------------------------------------------------------ K1 Hessian
def test_h_accumulate_golden(ops):
    g = load_golden("g4_g5_hessian")
    for tag, dt in (("f32", torch.float32), ("f16", torch.float16), ("bf16", torch.bfloat16)):
        X = g[f"X_{tag}"]
        C = X.shape[-1]
        H = torch.zeros(C, C, device="cuda")
        n = 0
        for xb in X:  # b = 1 per 3-D update (gptq.py:88,106-112)
            ops.h_accumulate(H, dev(xb).to(dt), n / (n + 1), 2.0 / (n + 1))
            n += 1
        ref = g[f"H_{tag}"]
        err = np.abs(npy(H) - ref).max()
        assert err <= 3e-6 * np.abs(ref).max(), (tag, err)  # fp32 accumulation-order tolerance
        assert np.array_equal(npy(H), npy(H).T)  # exactly symmetric


def test_h_accumulate_shapes(ops):
    torch.manual_seed(1)
    for T, C in ((100, 256), (2048, 1024), (37, 384)):
        X = torch.randn(T, C, device="cuda").half()
        H = torch.zeros(C, C, device="cuda")
        ops.h_accumulate(H, X, 0.0, 2.0)
        ref = 2.0 * (X.double().T @ X.double())
        assert (H.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item() * max(1, T / 512)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_h_accumulate_natural_layout_equals_relayout(ops, dt):
    """T % 128 == 0 takes the kernel that reads X in place through ds_read_b64_tr_b16; it feeds the MFMAs the
    same operands in the same order as the re-layout kernel, so the two agree BIT FOR BIT (and with fp64 to
    accumulation-order tolerance).  Random, non-symmetric data: a transposed or permuted operand cannot pass."""
    torch.manual_seed(12)
    T, C = 1536, 1280
    X = (torch.randn(T, C, device="cuda") * torch.exp(torch.randn(C, device="cuda") * 0.5)).to(dt)
    H0 = torch.randn(C, C, device="cuda")
    H0 = H0 + H0.T
    outs = []
    for image in (0, 1):
        with ops.options(syrk_image=image):
            H = H0.clone()
            ops.h_accumulate(H, X[:1024], 0.25, 2.0 / 3)
            ops.h_accumulate(H, X[1024:], 0.5, 0.125)   # T = 512: a single turn of the ring
            outs.append(H)
    assert torch.equal(outs[0], outs[1])
    Xd = X.double()
    ref = 0.5 * (0.25 * H0.double() + (2.0 / 3) * (Xd[:1024].T @ Xd[:1024])) + 0.125 * (Xd[1024:].T @ Xd[1024:])
    assert (outs[0].double() - ref).abs().max().item() <= 5e-6 * ref.abs().max().item()
    assert torch.equal(outs[0], outs[0].T)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("q_type", [10, 11, 12, 13, 14])
def test_group_search_lane_kernel_equals_wide_kernel(ops, q_type, dt):
    """The lane-per-group search kernel (option ss_wide = 0; default for large panels), the lane-pair kernel (2; mid-size panels) and the 8-lanes-per-group kernel (1, also the
    fallback for rows that are not 16-B aligned) add in the same order: every output is bit-identical, on ragged
    row counts, constant groups, all-zero rows and a misaligned view (which takes the wide kernel by itself)."""
    torch.manual_seed(31 + q_type)
    rows = 1003
    x = torch.randn(rows, 256, device="cuda") * torch.exp(torch.randn(rows, 1, device="cuda"))
    x[5] = 0.0
    x[6, :32] = 0.75
    x[7] = -1.5
    x[8, 40:72] = x[8, 40:72].abs()
    x = x.to(dt)
    outs = []
    for wide in (0, 1, 2):   # one lane / eight lanes / a lane pair per group
        with ops.options(ss_wide=wide):
            outs.append(ops.group_search(x, q_type))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a.view(torch.uint8) if a.dtype != torch.float32 else a.view(torch.int32),
                               b.view(torch.uint8) if b.dtype != torch.float32 else b.view(torch.int32))
    # misaligned rows: a view that starts one element into a wider buffer
    buf = torch.zeros(rows, 264, device="cuda", dtype=dt)
    buf[:, 1:257] = x
    with ops.options(ss_wide=0):
        mis = ops.group_search(buf[:, 1:257], q_type)
    for a, b in zip(outs[0], mis):
        assert torch.equal(a.view(torch.uint8) if a.dtype != torch.float32 else a.view(torch.int32),
                           b.view(torch.uint8) if b.dtype != torch.float32 else b.view(torch.int32))


@pytest.mark.parametrize("Cs", [(2048,), (4096, 2304, 2048)])
def test_h_accumulate_k_split_of_the_last_round(ops, Cs):
    """Long token ranges (T >= 8192): the tiles of the last, partial round of 256 CUs are cut into token ranges
    whose raw sums are combined by syrk_reduce_kernel in fixed order.  Checked against fp64 on every element of
    the smallest Hessian and on sampled elements of the others, against the unsplit schedule (option syrk_nosplit,
    fp32 summation-order tolerance), for exact symmetry, the beta/alpha telescoping and run-to-run determinism."""
    torch.manual_seed(77)
    T = 8192 + 384
    Xs = [(torch.randn(T, C, device="cuda") * torch.exp(torch.randn(C, device="cuda") * 0.5)).half() for C in Cs]
    H0s = [torch.randn(C, C, device="cuda") for C in Cs]
    H0s = [h + h.T for h in H0s]
    outs = []
    for nosplit in (0, 0, 1):
        with ops.options(syrk_nosplit=nosplit):
            Hs = [h.clone() for h in H0s]
            ops.h_accumulate_grouped(Hs, Xs, [0.5] * len(Cs), [2.0 / T] * len(Cs))
            outs.append(Hs)
    for a, b, c_, X, H0, C in zip(outs[0], outs[1], outs[2], Xs, H0s, Cs):
        assert torch.equal(a, b)                      # deterministic
        assert torch.equal(a, a.T)
        assert (a - c_).abs().max().item() <= 2e-6 * a.abs().max().item()
        if C == min(Cs):
            ref = 0.5 * H0.double() + (2.0 / T) * (X.double().T @ X.double())
            assert (a.double() - ref).abs().max().item() <= 3e-6 * ref.abs().max().item()
        else:
            i = torch.randint(0, C, (4096,), device="cuda")
            j = torch.randint(0, C, (4096,), device="cuda")
            ref = 0.5 * H0[i, j].double() + (2.0 / T) * (X[:, i].double() * X[:, j].double()).sum(0)
            assert (a[i, j].double() - ref).abs().max().item() <= 3e-6 * a.abs().max().item()
    if len(Cs) == 1:
        assert not torch.equal(outs[0][0], outs[2][0])  # the split schedule really ran (different rounding)


@pytest.mark.parametrize("C", [128, 1280, 4096])
def test_h_pack_unpack_upper(ops, C):
    """The all-reduce payload: pack(H) holds every 128x128 tile on or above the diagonal exactly once; unpack
    restores a symmetric H bit for bit and mirrors whatever the (reduced) buffer holds below the diagonal."""
    torch.manual_seed(C)
    A = torch.randn(C, C, device="cuda")
    H = A + A.T
    buf = ops.h_pack_upper(H)
    nt = C // 128
    assert buf.shape == (nt * (nt + 1) // 2, 128, 128)
    tiles = H.view(nt, 128, nt, 128).permute(0, 2, 1, 3)
    got = sorted(float(t.double().sum()) for t in buf)
    want = sorted(float(tiles[i, j].double().sum()) for i in range(nt) for j in range(i, nt))
    assert got == want
    H2 = torch.full_like(H, float("nan"))
    ops.h_unpack_upper(buf, H2)
    assert torch.equal(H2, H)
    ops.h_unpack_upper(buf * 0.5, H2)   # what a collective would hand back
    assert torch.equal(H2, 0.5 * H) and torch.equal(H2, H2.T)


def test_h_accumulate_grouped_full_size(ops):
    """BASELINE shapes (C = 14336 and 4096 in one grouped launch, ragged token counts): several rounds of the
    balanced tile table, the partial last round, the mirrored epilogue and the beta/alpha telescoping, checked on
    sampled 256-blocks against fp64 products."""
    torch.manual_seed(5)
    Ts, Cs = (4096 + 37, 2048), (14336, 4096)
    Xs = [torch.randn(T, C, device="cuda").half() for T, C in zip(Ts, Cs)]
    Hs = [torch.zeros(C, C, device="cuda") for C in Cs]
    ops.h_accumulate_grouped(Hs, [x[: T // 2] for x, T in zip(Xs, Ts)], [0.0, 0.0], [2.0, 2.0])
    ops.h_accumulate_grouped(Hs, [x[T // 2:] for x, T in zip(Xs, Ts)], [0.5, 0.5], [1.0, 1.0])  # beta*H + alpha*X^T X
    for H, X, C in zip(Hs, Xs, Cs):
        T = X.shape[0]
        h = T // 2
        for (r0, c0) in ((0, 0), (C - 256, C - 256), (256, C - 512), (C // 2, C // 2 + 256), (C - 512, 0)):
            a, b = X[:, r0:r0 + 256].double(), X[:, c0:c0 + 256].double()
            ref = 0.5 * 2.0 * (a[:h].T @ b[:h]) + 1.0 * (a[h:].T @ b[h:])
            got = H[r0:r0 + 256, c0:c0 + 256].double()
            assert (got - ref).abs().max().item() <= 3e-6 * ref.abs().max().item() * max(1, T / 512), (C, r0, c0)
        i = torch.randint(0, C, (4096,), device="cuda")
        j = torch.randint(0, C, (4096,), device="cuda")
        assert torch.equal(H[i, j], H[j, i])  # mirrored bit-for-bit


def test_h_prepare_full_size(ops):
    """C = 14336 (down_proj): split-bf16 GEMMs at every level of the recursion; U^T U H_damped == I on sampled
    columns in fp64, U upper triangular."""
    torch.manual_seed(6)
    C = 14336
    X = (torch.randn(2 * C, C, device="cuda") * torch.exp(torch.randn(C, device="cuda") * 0.5)).half()
    H = torch.zeros(C, C, device="cuda")
    ops.h_accumulate(H, X, 0.0, 2.0 / 8)
    del X
    W = torch.randn(64, C, device="cuda")
    U, flag = ops.h_prepare(H, W, 0.01)  # H now holds the damped matrix
    assert int(flag.item()) == 0
    cols = torch.tensor([0, 1, 127, 128, 4095, 7167, 7168, 9000, C - 129, C - 1], device="cuda")
    Ud = U.double()
    Y = Ud.T @ (Ud @ H[:, cols].double())  # (U^T U) H e_c
    E = torch.zeros(C, cols.numel(), device="cuda", dtype=torch.float64)
    E[cols, torch.arange(cols.numel(), device="cuda")] = 1.0
    assert (Y - E).abs().max().item() < 2e-2
    r = torch.randint(1, C, (4096,), device="cuda")
    c = (torch.rand(4096, device="cuda") * r).long()  # c < r: strictly lower
    assert bool((U[r, c] == 0).all())


@pytest.mark.parametrize("name", list(TYPES))
def test_full_width_linear_every_type_vs_oracle_rows(ops, oracle, name):
    """BASELINE config 3 (every K-quant encoder on Llama-3-8B Linears) at FULL width C = 4096, R = 1024 (k_proj):
    the whole GPU path (scale search at every 256-column boundary, 32 column-loop blocks, near and chained far
    trailing updates) against the oracle on a 48-row slice given the same U -- rows are independent, so the
    slice's integers, scales and final weights must be bit-identical.  Plus the size-independent properties on
    all rows: W_out == dequantize(outputs), packed bytes decode (independent ggml-spec decoder) to the same fields."""
    torch.manual_seed(40 + TYPES[name])
    R, C, T = 1024, 4096, 8192
    X = (torch.randn(T, C, device="cuda") * torch.exp(torch.randn(C, device="cuda") * 0.5)).half()
    X[:, torch.randperm(C, device="cuda")[:4]] *= 20.0  # outlier channels
    H = torch.zeros(C, C, device="cuda")
    ops.h_accumulate(H, X, 0.0, 2.0 / 4)
    del X
    W0 = (torch.randn(R, C, device="cuda") * 0.02).half().float()
    Wp = W0.clone()
    U, flag = ops.h_prepare(H, Wp, 0.01)
    assert int(flag.item()) == 0
    t = TYPES[name]
    W = Wp.clone()
    q, d, s, dmin, m = ops.gptq_quantize(W, U, t, block_size=128)
    rows = slice(517, 565)
    Wd, oq, od, os_, odm, om = oracle.gptq_step(npy(Wp[rows]), npy(U), t, block_size=128)
    assert np.array_equal(npy(q[rows]), oq), f"{(npy(q[rows]) != oq).mean():.4%} ints differ"
    assert np.array_equal(u16(d[rows]), od) and np.array_equal(u16(dmin[rows]), odm)
    assert np.array_equal(npy(s[rows]), os_) and np.array_equal(npy(m[rows]), om)
    assert np.array_equal(npy(W[rows]), Wd)
    deq = ops.dequantize(t, q, d, s, dmin, m)
    assert torch.equal(W, deq)
    codes, d2, sc2, dm2, mn2 = unpack(t, npy(ops.pack(t, q, d, s, dmin, m)))
    assert np.array_equal(codes, npy(q).astype(np.int32)) and np.array_equal(sc2, npy(s).astype(np.int32))
    assert np.array_equal(d2, u16(d)) and np.array_equal(dm2, u16(dmin)) and np.array_equal(mn2, npy(m).astype(np.int32))


def test_llama70b_down_proj_shape(ops):
    """Largest BASELINE shape (Llama-3-70B down_proj: C = 28672): accumulate -> prepare -> column loop on a row slice;
    exercises 112 x 112 tile tables, 224 diagonal blocks and 32-bit-safe indexing (C*C = 8.2e8 elements)."""
    torch.manual_seed(8)
    C, T, R = 28672, 4096, 256
    X = (torch.randn(T, C, device="cuda") * torch.exp(torch.randn(C, device="cuda") * 0.5)).half()
    H = torch.zeros(C, C, device="cuda")
    ops.h_accumulate(H, X, 0.0, 2.0 / 2)
    i = torch.randint(0, C, (2048,), device="cuda")
    j = torch.randint(0, C, (2048,), device="cuda")
    ref = 1.0 * (X[:, i].double() * X[:, j].double()).sum(0)
    assert ((H[i, j].double() - ref).abs() <= 1e-5 * ref.abs().max()).all()
    del X
    W0 = (torch.randn(R, C, device="cuda") * 0.02).half().float()
    W = W0.clone()
    U, flag = ops.h_prepare(H, W, 0.01)  # T < C: H is rank-deficient, the damping makes it definite
    assert int(flag.item()) == 0
    t = TYPES["Q4_K"]
    q, d, s, dmin, m = ops.gptq_quantize(W, U, t, block_size=128)
    deq = ops.dequantize(t, q, d, s, dmin, m)
    assert bool((W == deq).all()) and int(q.min()) >= 0 and int(q.max()) <= 15
    rel = ((deq - W0).norm() / W0.norm()).item()
    assert rel < 0.2, rel


def test_mixtral_expert_shapes(ops):
    """BASELINE config 5 (Mixtral-8x7B: experts Q3_K, attention Q6_K): an expert's w2 is 4096 x 14336 and sees
    only the tokens routed to it -- ragged [tokens, C] batches with batch = tokens (gptq.py:86), T % 128 != 0
    (the 128x128 SYRK path, C = 14336), sample-weighted telescoping; then prepare and the Q3_K column loop
    (signed ints, 16-wide groups, no mins) on a row slice, and Q6_K on an attention-shaped 256 x 4096 slice."""
    torch.manual_seed(55)
    C, R = 14336, 256
    sig = torch.exp(torch.randn(C, device="cuda") * 0.5)
    batches = [1237, 611, 2053]
    H = torch.zeros(C, C, device="cuda")
    n = 0
    i = torch.randint(0, C, (1024,), device="cuda")
    j = torch.randint(0, C, (1024,), device="cuda")
    ref = torch.zeros(1024, device="cuda", dtype=torch.float64)
    for b in batches:
        X = (torch.randn(b, C, device="cuda") * sig).half()
        ops.h_accumulate(H, X, n / (n + b), 2.0 / (n + b))
        ref += (X[:, i].double() * X[:, j].double()).sum(0)
        n += b
    ref *= 2.0 / n
    assert ((H[i, j].double() - ref).abs() <= 2e-5 * ref.abs().max()).all()
    assert torch.equal(H, H.T)
    W0 = (torch.randn(R, C, device="cuda") * 0.02).half().float()
    W = W0.clone()
    U, flag = ops.h_prepare(H, W, 0.01)  # 3901 tokens < C: rank-deficient, made definite by the damping
    assert int(flag.item()) == 0
    t = TYPES["Q3_K"]
    q, d, s, dmin, m = ops.gptq_quantize(W, U, t, block_size=128)
    assert q.dtype == torch.int8 and int(q.min()) >= -4 and int(q.max()) <= 3
    assert int(s.min()) >= 0 and int(s.max()) <= 31 and not bool(dmin.any()) and not bool(m.any())
    deq = ops.dequantize(t, q, d, s, dmin, m)
    assert torch.equal(W, deq)
    codes, d2, sc2, dm2, mn2 = unpack(t, npy(ops.pack(t, q, d, s, dmin, m)))
    assert np.array_equal(codes, npy(q).astype(np.int32)) and np.array_equal(sc2, npy(s).astype(np.int32))
    assert ((deq - W0).norm() / W0.norm()).item() < 0.35
    # attention side: Q6_K, 4096 wide
    Ca = 4096
    Xa = (torch.randn(3000, Ca, device="cuda") * sig[:Ca]).half()
    Ha = torch.zeros(Ca, Ca, device="cuda")
    ops.h_accumulate(Ha, Xa, 0.0, 2.0 / 3000)
    Wa0 = (torch.randn(R, Ca, device="cuda") * 0.02).half().float()
    Wa = Wa0.clone()
    Ua, flag = ops.h_prepare(Ha, Wa, 0.01)
    assert int(flag.item()) == 0
    t = TYPES["Q6_K"]
    q, d, s, dmin, m = ops.gptq_quantize(Wa, Ua, t, block_size=128)
    assert int(q.min()) >= -32 and int(q.max()) <= 31 and torch.equal(Wa, ops.dequantize(t, q, d, s, dmin, m))
    assert ((Wa - Wa0).norm() / Wa0.norm()).item() < 0.05


# --------------------------------------------------------------- K2/K3 prepare
def test_h_prepare_golden(ops):
    g = load_golden("g4_g5_hessian")
    H, W = dev(g["prep_H_in"]), dev(g["prep_W_in"])
    C = H.shape[0]
    U, flag = ops.h_prepare(H, W, 0.01)
    assert int(flag.item()) == 0
    assert np.array_equal(npy(W), g["prep_W_after_prestep"])
    assert np.allclose(np.diag(npy(H)), g["prep_H_after_diag"], rtol=1e-6)
    for r in (5, 17):
        assert np.allclose(npy(H)[r], g[f"prep_H_after_row{r}"], rtol=1e-6, atol=0)
    Uref = triu_unpack(g["prep_U_triu"], C)
    Un = npy(U)
    assert np.all(np.tril(Un, -1) == 0)
    assert np.abs(Un - Uref).max() <= 2e-4 * np.abs(Uref).max()  # fp32 factorisation tolerance
    # definition check in fp64: U^T U == inv(H_damped)
    Hd = npy(H).astype(np.float64)
    assert np.abs(Un.astype(np.float64).T @ Un.astype(np.float64) @ Hd - np.eye(C)).max() < 5e-3


def test_h_prepare_singular_falls_back_to_identity(ops):
    g = load_golden("g4_g5_hessian")
    X = dev(g["sing_X"])
    H = torch.zeros(256, 256, device="cuda")
    ops.h_accumulate(H, X, 0.0, 2.0)
    U, flag = ops.h_prepare(H, torch.ones(8, 256, device="cuda"), 0.0)
    assert int(flag.item()) == 1 and torch.equal(U, torch.eye(256, device="cuda"))


def test_h_prepare_larger(ops):
    torch.manual_seed(3)
    C = 1536
    X = (torch.randn(4096, C, device="cuda") * torch.exp(torch.randn(C, device="cuda") * 0.5)).half()
    H = torch.zeros(C, C, device="cuda")
    ops.h_accumulate(H, X, 0.0, 2.0 / 4096)
    W = torch.randn(64, C, device="cuda")
    U, flag = ops.h_prepare(H, W, 0.01)
    assert int(flag.item()) == 0
    Ud = U.double()
    resid = (Ud.T @ Ud @ H.double() - torch.eye(C, device="cuda", dtype=torch.float64)).abs().max().item()
    assert resid < 2e-2, resid
    ref = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H.double())), upper=True)
    assert (Ud - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()


# ------------------------------------------- full-size properties (BASELINE shapes)
@pytest.mark.parametrize("name", ["Q4_K", "Q3_K"])
def test_full_size_properties(ops, name):
    """4096 x 4096 (Llama-3-8B q/o_proj shape): size-independent invariants."""
    t = TYPES[name]
    torch.manual_seed(11)
    R = C = 4096
    W0 = (torch.randn(R, C, device="cuda") * 0.02).half().float()
    X = (torch.randn(8192, C, device="cuda") * torch.exp(torch.randn(C, device="cuda") * 0.5)).half()
    H = torch.zeros(C, C, device="cuda")
    for i in range(4):
        ops.h_accumulate(H, X[i * 2048:(i + 1) * 2048], i / (i + 1), 2.0 / (i + 1))
    W = W0.clone()
    U, flag = ops.h_prepare(H, W, 0.01)
    assert int(flag.item()) == 0
    q, d, s, dmin, m = ops.gptq_quantize(W, U, t, block_size=128)
    deq = ops.dequantize(t, q, d, s, dmin, m)
    assert bool((W == deq).all()), "W after step must be the dequantized matrix (gptq.py:266)"
    ti_min, ti_max = {"Q4_K": (0, 15), "Q3_K": (-4, 3)}[name]
    assert int(q.min()) >= ti_min and int(q.max()) <= ti_max
    # GPTQ must beat RTN on the Hessian-weighted error it minimises
    q2, d2, s2, dm2, m2 = ops.rtn_quantize(W0, t)
    rtn = ops.dequantize(t, q2, d2, s2, dm2, m2)
    def herr(Wq):
        E = (Wq - W0)[:256].double()
        return float(((E @ H.double()) * E).sum())
    assert herr(deq) < herr(rtn)
    # packed bytes decode (independent ggml-layout decoder) to exactly the stored tensors
    packed = ops.pack(t, q, d, s, dmin, m)
    codes, dd, sc, dmn, mn = unpack(t, npy(packed[:64]))
    assert np.array_equal(codes, npy(q[:64]).astype(np.int32)) and np.array_equal(dd, u16(d[:64]))
    assert np.array_equal(sc, npy(s[:64]).astype(np.int32))
    # idempotence: quantizing the dequantized matrix with static groups of itself is stable in the ints' range
    assert packed.shape == (R, C // 256 * {"Q4_K": 144, "Q3_K": 110}[name])
