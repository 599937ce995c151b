"""CPU oracle (oracle/gq_oracle.c) pinned against outputs of the reference itself.

Fixtures: tests/golden/*.npz, produced by tests/golden/make_golden.py which imports
/root/reference in the build container.  Bit-exact everywhere except the two
fp32 linear-algebra stages (H accumulate, Cholesky chain) which are tolerance-class
(SURVEY 7 "hard parts": the reference is not bit-stable there between its own CPU
and CUDA runs either).
"""
import numpy as np
import pytest

from conftest import load_golden, triu_unpack

TYPES = {"Q2_K": 10, "Q3_K": 11, "Q4_K": 12, "Q5_K": 13, "Q6_K": 14}
K_TYPES = ("Q2_K", "Q4_K", "Q5_K")


def bits_eq(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32),
                          np.ascontiguousarray(b, np.float32).view(np.uint32))


def test_type_table(oracle):
    # reference quant_utils.py:19-26
    exp = {10: (2, 0, 3, 15, 16, 0, 84), 11: (3, -4, 3, 31, 16, 1, 110), 12: (4, 0, 15, 63, 32, 0, 144),
           13: (5, 0, 31, 63, 32, 0, 176), 14: (6, -32, 31, 63, 16, 1, 210)}
    for t, e in exp.items():
        ti = oracle.type_info(t)
        assert (ti["bits"], ti["qmin"], ti["qmax"], ti["scale_maxq"], ti["group"], ti["is_signed"],
                ti["type_size"]) == e
    with pytest.raises(ValueError):
        oracle.type_info(2)


def test_f16_roundtrip(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    f = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1, 1e3, 1e5)]
                       + [np.array([0.0, -0.0, 65504.0, 65520.0, 65519.99, 5.96e-8, 2.98e-8, 2.9802322e-8,
                                    6.1e-5, np.inf, -np.inf], np.float32)])
    with np.errstate(over="ignore"):
        want = f.astype(np.float16).view(np.uint16)
    got = np.array([L.gqo_f32_to_f16(float(v)) for v in f], np.uint16)
    assert np.array_equal(want, got)


@pytest.mark.parametrize("name", list(TYPES))
def test_g1_make_quants(oracle, name):
    g = load_golden("g1_make_quants")
    x = g[f"{name}_x"]
    bits = oracle.type_info(TYPES[name])["bits"]
    sc, ze = (oracle.make_k_quants if name in K_TYPES else oracle.make_quants)(x, bits)
    assert bits_eq(sc, g[f"{name}_ieee_scale"]), "group scales differ from the reference"
    assert bits_eq(ze, g[f"{name}_ieee_zero"]), "group zeros differ from the reference (incl. -0.0)"
    # for the record: the stock-MKL-sqrt reference run differs only through 1-ulp av_x
    flips = (sc.view(np.uint32) != g[f"{name}_mkl_scale"].view(np.uint32)).mean()
    assert flips < 0.02, f"{name}: {flips:.3%} of group scales differ vs the MKL-sqrt run"


def test_g1_search_params(oracle):
    g = load_golden("g1_make_quants")
    sc, ze = oracle.make_k_quants(g["Q4_K_alt_x"], 4, rmin=-0.5, rdelta=0.05, nstep=10)
    assert bits_eq(sc, g["Q4_K_alt_scale"]) and bits_eq(ze, g["Q4_K_alt_zero"])
    sc, ze = oracle.make_k_quants(g["Q4_K_alt_x"], 4, nstep=0)
    assert bits_eq(sc, g["Q4_K_nstep0_scale"]) and bits_eq(ze, g["Q4_K_nstep0_zero"])


@pytest.mark.parametrize("name", ["Q3_K", "Q6_K"])
def test_g12_mse_branch(oracle, name):
    """--quant_scale mse (quant_utils.py:164-191), the reference's own outputs in both modes.  The oracle follows the
    branch verbatim (the .round() of :180 lands on the scale, q_int stays un-rounded): for weight-like panels (every
    value below zero = (maxq+1)/2) the search returns the absmax result bit for bit, for the wide panel it does not --
    all three panels must match the reference in both modes."""
    g = load_golden("g12_mse_scale")
    try:
        for tag in ("w", "mid", "wide"):
            for mode in ("absmax", "mse"):
                oracle.set_quant_scale(mode)
                d, s, dmin, m = oracle.scale_search(g[f"{name}_{tag}_x"], TYPES[name])
                assert np.array_equal(d, g[f"{name}_{tag}_{mode}_d"]) and np.array_equal(s, g[f"{name}_{tag}_{mode}_s"]), (tag, mode)
            if tag != "wide":
                assert np.array_equal(g[f"{name}_{tag}_mse_d"], g[f"{name}_{tag}_absmax_d"])
                assert np.array_equal(g[f"{name}_{tag}_mse_s"], g[f"{name}_{tag}_absmax_s"])
        assert not np.array_equal(g[f"{name}_wide_mse_s"], g[f"{name}_wide_absmax_s"])
    finally:
        oracle.set_quant_scale("absmax")


@pytest.mark.parametrize("name", list(TYPES))
def test_g2_scale_search(oracle, name):
    g = load_golden("g2_scale_search")
    d, s, dmin, m = oracle.scale_search(g[f"{name}_x"], TYPES[name])
    assert np.array_equal(d, g[f"{name}_ieee_d"])
    assert np.array_equal(dmin, g[f"{name}_ieee_dmin"])
    assert np.array_equal(s, g[f"{name}_ieee_s"]) and s.dtype == g[f"{name}_ieee_s"].dtype
    assert np.array_equal(m, g[f"{name}_ieee_m"])


@pytest.mark.parametrize("name", list(TYPES))
def test_g3_elementwise(oracle, name):
    g = load_golden("g3_elementwise")
    L = oracle.lib()
    ti = oracle.type_info(TYPES[name])
    x, d, dmin, s, m = (g[f"{name}_{k}"] for k in ("x", "d", "dmin", "s", "m"))
    L.gqo_quantize1.restype = L.gqo_dequantize1.restype = __import__("ctypes").c_float
    import ctypes
    L.gqo_quantize1.argtypes = [ctypes.c_float, ctypes.c_uint16, ctypes.c_int, ctypes.c_uint16, ctypes.c_int,
                                ctypes.c_int, ctypes.c_int]
    L.gqo_dequantize1.argtypes = [ctypes.c_float, ctypes.c_uint16, ctypes.c_int, ctypes.c_uint16, ctypes.c_int]
    q = np.array([L.gqo_quantize1(float(x[i]), int(d[i]), int(s[i]), int(dmin[i]), int(m[i]), ti["qmin"], ti["qmax"])
                  for i in range(len(x))], np.float32)
    assert np.array_equal(q, g[f"{name}_q"])
    w = np.array([L.gqo_dequantize1(float(q[i]), int(d[i]), int(s[i]), int(dmin[i]), int(m[i]))
                  for i in range(len(x))], np.float32)
    assert bits_eq(w, g[f"{name}_w"])


def test_g4_h_accumulate(oracle):
    g = load_golden("g4_g5_hessian")
    for tag in ("f32", "f16", "bf16"):
        X = g[f"X_{tag}"]  # [3, L, C], b = 1 per update (3-D input: gptq.py:88)
        C = X.shape[-1]
        H = np.zeros((C, C), np.float32)
        n = 0
        for xb in X:
            H = oracle.h_accumulate(H, xb, n / (n + 1), 2.0 / (n + 1))
            n += 1
        ref = g[f"H_{tag}"]
        assert np.abs(H - ref).max() <= 2e-6 * np.abs(ref).max()
    X = g["X_2d"]
    H = np.zeros((X.shape[-1],) * 2, np.float32)
    n = 0
    for xb in X:  # 2-D input: b = #tokens
        b = xb.shape[0]
        H = oracle.h_accumulate(H, xb, n / (n + b), 2.0 / (n + b))
        n += b
    assert n == int(g["n_2d"])
    assert np.abs(H - g["H_2d"]).max() <= 2e-6 * np.abs(g["H_2d"]).max()


def test_g5_h_prepare(oracle):
    g = load_golden("g4_g5_hessian")
    H, W = g["prep_H_in"], g["prep_W_in"]
    C = H.shape[0]
    U, H2, W2, bad = oracle.h_prepare(H, W, 0.01)
    assert not bad
    assert np.array_equal(W2, g["prep_W_after_prestep"])  # dead channel 5 zeroed
    assert np.allclose(np.diag(H2), g["prep_H_after_diag"], rtol=1e-6)
    for r in (5, 17):  # dead channel / zero weight column: row zeroed, diag 1 + damp
        assert np.allclose(H2[r], g[f"prep_H_after_row{r}"], rtol=1e-6, atol=0)
    Uref = triu_unpack(g["prep_U_triu"], C)
    assert np.all(np.tril(U, -1) == 0)
    assert np.abs(U - Uref).max() <= 1e-4 * np.abs(Uref).max()
    # singular H (rank 16 of 256, no damping) -> identity fallback, like the reference
    X = g["sing_X"]
    Hs = oracle.h_accumulate(np.zeros((256, 256), np.float32), X, 0.0, 2.0)
    U, _, _, bad = oracle.h_prepare(Hs, np.ones((8, 256), np.float32), 0.0)
    assert bool(g["sing_U_is_identity"]) and bool(g["sing_flag"])
    assert bad and np.array_equal(U, np.eye(256, dtype=np.float32))


def _g6_cases():
    g = load_golden("g6_g7_step_and_pack")
    tags = sorted({k.rsplit("_", 1)[0] for k in g.files if k.endswith("_q") and "mklsqrt" not in k})
    return tags


@pytest.mark.parametrize("tag", _g6_cases())
def test_g6_step(oracle, tag):
    g = load_golden("g6_g7_step_and_pack")
    case, name1, name2, b, s = tag.split("_")
    name = f"{name1}_{name2}"
    block = None if b == "bNone" else int(b[1:])
    static = s == "s1"
    W0 = g[f"{case}_W0"]
    U = triu_unpack(g[f"{case}_U_triu"], W0.shape[1])
    Wd, q, d, sc, dmin, m = oracle.gptq_step(W0, U, TYPES[name], block_size=block, static_groups=static)
    assert np.array_equal(q, g[f"{tag}_q"]), f"{(q != g[f'{tag}_q']).mean():.4%} ints differ"
    assert np.array_equal(d, g[f"{tag}_d"]) and np.array_equal(dmin, g[f"{tag}_dmin"])
    assert np.array_equal(sc, g[f"{tag}_s"]) and np.array_equal(m, g[f"{tag}_m"])
    if f"{tag}_Wdeq_head" in g.files:
        assert bits_eq(Wd[:4, :64], g[f"{tag}_Wdeq_head"])
        assert float(Wd.astype(np.float64).sum()) == float(g[f"{tag}_Wdeq_sum"])
    # gptq.py:266: the final working copy IS the dequantized matrix (up to the sign of
    # zero: q = rint(tiny negative) = -0.0 dequantizes to -0.0, the stored uint8 0 to +0.0)
    assert np.array_equal(Wd, oracle.dequantize(TYPES[name], q, d, sc, dmin, m))
    if f"{tag}_packed" in g.files:  # G7
        assert np.array_equal(oracle.pack(TYPES[name], q, d, sc, dmin, m), g[f"{tag}_packed"])
    if f"{tag}_mklsqrt_q" in g.files:
        flips = (q != g[f"{tag}_mklsqrt_q"]).mean()
        assert flips < 0.05, f"{flips:.3%} ints differ vs the stock-MKL-sqrt reference run"


@pytest.mark.parametrize("tag", ["Q4_K_b128", "Q2_K_b128", "Q6_K_b64", "Q5_K_b128"])
def test_g11_act_order(oracle, tag):
    """GPTQ.step with act_order=True (gptq.py:208-216, 233-235, 272-276): static scales of the original column
    groups, columns walked in descending-diag(H) order, qweight un-permuted at the end."""
    g = load_golden("g11_act_order")
    name, b = tag[:4], int(tag.split("_b")[1])
    t = TYPES[name]
    W0, perm = g["W0"], g["perm"]
    U = triu_unpack(g["U_triu"], W0.shape[1])
    # gptq.py:184-196: static scales from the unpermuted W == the fp32 RTN scale search
    _, d, sc, dmin, m = oracle.rtn_quantize(W0, t)
    assert np.array_equal(d, g[f"{tag}_d"]) and np.array_equal(dmin, g[f"{tag}_dmin"])
    assert np.array_equal(sc, g[f"{tag}_s"]) and np.array_equal(m, g[f"{tag}_m"])
    Wd, qp = oracle.gptq_step_perm(W0[:, perm], U, t, perm, d, sc, dmin, m, block_size=b)
    q = qp[:, np.argsort(perm)]
    assert np.array_equal(q, g[f"{tag}_q"]), f"{(q != g[f'{tag}_q']).mean():.4%} ints differ"
    # the working copy ends as the dequantized matrix, still permuted (the reference never un-permutes W)
    assert np.array_equal(Wd[:, np.argsort(perm)], oracle.dequantize(t, q, d, sc, dmin, m))


@pytest.mark.parametrize("name", list(TYPES))
def test_g8_g9_rtn_dequant_pack(oracle, name):
    g = load_golden("g8_g9_rtn_dequant")
    t = TYPES[name]
    q, d, s, dmin, m = oracle.rtn_quantize(g["W"], t)
    assert np.array_equal(q, g[f"{name}_q"]) and q.dtype == g[f"{name}_q"].dtype
    assert np.array_equal(d, g[f"{name}_d"]) and np.array_equal(dmin, g[f"{name}_dmin"])
    assert np.array_equal(s, g[f"{name}_s"]) and np.array_equal(m, g[f"{name}_m"])
    assert bits_eq(oracle.dequantize(t, q, d, s, dmin, m), g[f"{name}_deq"])
    packed = oracle.pack(t, q, d, s, dmin, m)
    assert packed.shape == g[f"{name}_packed"].shape and np.array_equal(packed, g[f"{name}_packed"])


def test_pack_does_not_mutate_inputs(oracle):
    # reference pack_Q3K / pack_Q6K offset their inputs in place (packing_utils.py:94-95, 279)
    g = load_golden("g8_g9_rtn_dequant")
    for name in ("Q3_K", "Q6_K"):
        q, d, s = g[f"{name}_q"].copy(), g[f"{name}_d"], g[f"{name}_s"].copy()
        q0, s0 = q.copy(), s.copy()
        oracle.pack(TYPES[name], q, d, s)
        assert np.array_equal(q, q0) and np.array_equal(s, s0)


@pytest.mark.parametrize("tag,rmode", [("f16", 1), ("bf16", 2)])
@pytest.mark.parametrize("name", list(TYPES))
def test_g9_rtn_model_dtype(oracle, tag, rmode, name):
    """quantizer.py:109,195: RTN of embed/lm_head runs make_*quants in the model dtype; the oracle rounds
    after every op the way ATen's CPU fp16/bf16 kernels do (fp32 compute, round to the tensor dtype)."""
    g = load_golden("g8_g9_rtn_dequant")
    q, d, s, dmin, m = oracle.rtn_quantize_lp(g[f"W_{tag}"], rmode, TYPES[name])
    assert np.array_equal(q, g[f"{tag}_{name}_q"]), f"{(q != g[f'{tag}_{name}_q']).mean():.4%} ints differ"
    assert np.array_equal(d, g[f"{tag}_{name}_d"]) and np.array_equal(dmin, g[f"{tag}_{name}_dmin"])
    assert np.array_equal(s, g[f"{tag}_{name}_s"]) and np.array_equal(m, g[f"{tag}_{name}_m"])
    # and it is NOT what an fp32 search gives (7-10 % of Q4_K ints differ)
    if name == "Q4_K":
        q32, *_ = oracle.rtn_quantize(g[f"W_{tag}"], TYPES[name])
        assert (q32 != q).mean() > 0.01


# ---------------------------------------------------------------- G14: EvoPress FastOBQ (uniform grids)
G14_TAGS = ("g128", "g64sym", "g128b64")


@pytest.mark.parametrize("tag", G14_TAGS)
def test_g14_fast_obq_step(oracle, tag):
    """evopress/src/fast_obq.py:146-200 given (W, U) as the reference's own _prepare produced them: ints, scales and
    zero points of every bit width are bit-exact."""
    g = load_golden("g14_fast_obq")
    R, C, gs, sym, block = (int(v) for v in g[f"{tag}_cfg"])
    U = triu_unpack(g[f"{tag}_U_triu"], C)
    for b in (2, 3, 4, 8):
        Wd, q, sc, ze = oracle.obq_step(g[f"{tag}_W0"], U, b, gs, bool(sym), block)
        assert np.array_equal(q, g[f"{tag}_b{b}_q"])
        assert np.array_equal(sc.view(np.uint32), g[f"{tag}_b{b}_scale"].view(np.uint32))
        assert np.array_equal(ze, g[f"{tag}_b{b}_zero"])
        grp = np.repeat(np.arange(C // gs), gs)
        assert np.array_equal(Wd, sc[:, grp] * (q.astype(np.float32) - ze[:, grp]))  # quant_utils.py:28-29


@pytest.mark.parametrize("tag", ("g64sym", "g128b64"))
def test_g14_obq_prepare(oracle, tag):
    """fast_obq.py:133-141 + 219-232: dead diagonal -> 1, damping, THEN the zero-column mask with an undamped 1."""
    g = load_golden("g14_fast_obq")
    R, C, *_ = (int(v) for v in g[f"{tag}_cfg"])
    U, H2, W2, bad = oracle.h_prepare(g[f"{tag}_H_in"], g[f"{tag}_W_in"], 0.01, obq_order=True)
    assert not bad and np.array_equal(W2, g[f"{tag}_W0"])  # dead channel 3 zeroed
    assert np.allclose(np.diag(H2), g[f"{tag}_H_after_diag"], rtol=1e-6)
    assert H2[3, 3] == 1.0 and H2[7, 7] == 1.0  # masked AFTER the damping
    for r in (3, 7):
        assert np.array_equal(H2[r], g[f"{tag}_H_after_row{r}"])
    Uref = triu_unpack(g[f"{tag}_U_triu"], C)
    assert np.abs(U - Uref).max() <= 1e-4 * np.abs(Uref).max()


def test_obq_step_per_row_grid(oracle):
    """group_size None (fast_obq.py:153-154): one grid per row from the original W; the first block checked against a direct
    numpy restatement of the column loop (the reference leaves scale / zero outputs uninitialised here)."""
    rng = np.random.default_rng(5)
    R, C = 8, 256
    W = (rng.standard_normal((R, C)) * 0.05).astype(np.float32)
    X = rng.standard_normal((512, C)).astype(np.float32)
    U, *_ = oracle.h_prepare((2.0 / 512 * X.T @ X).astype(np.float32), W, 0.01, obq_order=True)
    Wd, q, sc, ze = oracle.obq_step(W, U, 3, 0, False, 128)
    mn, mx = W.min(1), W.max(1)
    s_ref = ((mx - mn) / np.float32(7)).astype(np.float32)
    z_ref = np.round(-mn / s_ref)
    assert np.array_equal(sc[:, 0], s_ref) and np.array_equal(ze[:, 0], z_ref)
    blk = W[:, :128].copy()  # the first block: no trailing update has touched it, so numpy restates it bit for bit
    for i in range(128):
        qq = np.clip(np.round(blk[:, i] / np.maximum(s_ref, np.float32(1e-9)) + z_ref), 0, 7).astype(np.float32)
        wq = s_ref * (qq - z_ref)
        assert np.array_equal(q[:, i], qq.astype(np.uint8)) and np.array_equal(Wd[:, i], wq)
        err = ((blk[:, i] - wq) / U[i, i]).astype(np.float32)
        blk[:, i:] = blk[:, i:] + (-err)[:, None] * U[i, i:128][None]


@pytest.mark.parametrize("name", ["Q3_K", "Q6_K", "Q4_K"])
def test_g15_rtn_with_quant_scale_mse(oracle, name):
    """_quant_non_block_module with quant_scale="mse" (quantizer.py:293-295) on an embed-like fp32 weight with entries
    beyond +-32: the reference's run (G15), bit for bit.  With fp16 / bf16 weights the reference raises for the two
    make_quants types (recorded in the fixture; the driver mirrors that, tests/test_host_logic_cpu.py)."""
    g = load_golden("g15_rtn_mse")
    t = TYPES[name]
    oracle.set_quant_scale("mse")
    try:
        q, d, s, dmin, m = oracle.rtn_quantize(g["W_f32"], t)
    finally:
        oracle.set_quant_scale("absmax")
    assert np.array_equal(q, g[f"f32_{name}_q"])
    assert np.array_equal(d, g[f"f32_{name}_d"]) and np.array_equal(s, g[f"f32_{name}_s"])
    assert np.array_equal(dmin, g[f"f32_{name}_dmin"]) and np.array_equal(m, g[f"f32_{name}_m"])
    if name != "Q4_K":
        assert float(g[f"f32_{name}_differs_from_absmax"]) > 0.0  # the branch matters on this weight
        for tag in ("f16", "bf16"):
            assert "Index put requires" in str(g[f"{tag}_{name}_raises"])
    else:
        assert float(g["f32_Q4_K_differs_from_absmax"]) == 0.0 and "bf16_Q4_K_q" in g.files
