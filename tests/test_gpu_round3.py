"""Round 3, on the GPU: the image GEMMs of the Cholesky chain (gq_gemm3p.hpp) and what came with them."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gptq_gguf_toolkit_amd import ops as _ops
    return _ops


def _poison_upper_blocks(T):
    """128x128 blocks beyond the block diagonal of a lower-triangular operand are never written by the chain."""
    n = T.shape[0] // 128
    for i in range(n):
        T[128 * i:128 * i + 128, 128 * (i + 1):] = float("nan")
    return T


# (trans_b, mode, k_range, lower): the four products of a recursion node + the two plain forms
CASES = [(True, 1, 1, False), (True, 0, 0, True), (False, 1, 2, False), (False, 2, 3, False), (True, 1, 0, False),
         (False, 0, 0, False)]


@pytest.mark.parametrize("planes", [2, 3])
@pytest.mark.parametrize("case", CASES)
def test_chol_gemm_against_fp64(ops, planes, case):
    """gq_chol_gemm (linalg_utils.py:8-12's level-3 work as this build does it): every mode / k-range against fp64.
    Tolerance class, stated: |C - C64| <= 2e-6 (|A||B| + |C0|) componentwise (an fp32 sgemm of these sizes: ~1e-6);
    rows of A span e^+-4 in scale, triangular operands carry NaN in the blocks the chain never writes."""
    trans_b, mode, kr, lower = case
    dev = "cuda"
    torch.manual_seed(10 * planes + kr)
    for n in (512, 1024):
        M = N = K = n
        if kr == 0 and not lower:
            M, N, K = n, 768, 640
        A = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, 1, device=dev) * 2)
        B = (torch.randn(N, K, device=dev) if trans_b else torch.randn(K, N, device=dev)) * 0.1
        if kr == 3:
            A = _poison_upper_blocks(torch.tril(A))
        if kr in (1, 2):
            B = _poison_upper_blocks(torch.tril(B))
        if lower:
            B = A
        C0 = torch.randn(M, N, device=dev)
        Cm = C0.clone()
        ops.chol_gemm(Cm, A, B, trans_b, mode, kr, lower, planes)
        A64, B64 = torch.nan_to_num(A).double(), torch.nan_to_num(B).double()
        P = A64 @ (B64.T if trans_b else B64)
        ref = C0.double() - P if mode == 0 else (P if mode == 1 else -P)
        bound = A64.abs() @ (B64.abs().T if trans_b else B64.abs()) + C0.abs().double()
        msk = torch.ones(M, N, device=dev, dtype=torch.bool)
        if lower:  # only 256-tiles with tile row >= tile col are computed
            msk = torch.ones(M // 256, N // 256, device=dev).tril().bool().repeat_interleave(256, 0).repeat_interleave(256, 1)
            assert torch.equal(Cm[~msk], C0[~msk])
        assert bool(torch.isfinite(Cm[msk]).all())
        err = ((Cm.double() - ref).abs() / bound)[msk].max().item()
        assert err < 2e-6, (case, planes, n, err)
        Cm2 = C0.clone()  # deterministic: the K-split partial sums are added in fixed order
        ops.chol_gemm(Cm2, A, B, trans_b, mode, kr, lower, planes)
        assert torch.equal(Cm, Cm2)


def _hessian(C, seed, spread=0.5, outlier=20.0):
    torch.manual_seed(seed)
    sig = torch.exp(torch.randn(C, device="cuda") * spread)
    sig[torch.randperm(C, device="cuda")[:8]] *= outlier
    X = (torch.randn(2 * C, C, device="cuda") * sig).half()
    H = torch.zeros(C, C, device="cuda")
    from gptq_gguf_toolkit_amd import ops as _ops
    _ops.h_accumulate(H, X, 0.0, 2.0 / 4)
    return H


def test_h_prepare_image_levels_never_read_unwritten_scratch(ops):
    """As test_h_prepare_never_reads_unwritten_scratch, at a width whose two top recursion levels run on the image
    GEMMs (7168 = 3584 | 3584 = (1792 | 1792) x 2): A above its block diagonal and X are NaN-filled first."""
    C = 7168
    H = _hessian(C, 4)
    W = torch.randn(64, C, device="cuda")
    U0, f0 = ops.h_prepare(H.clone(), W.clone(), 0.01)
    with ops.options(chol_poison=1):
        U1, f1 = ops.h_prepare(H.clone(), W.clone(), 0.01)
    assert int(f0.item()) == 0 and int(f1.item()) == 0
    assert bool(torch.isfinite(U1).all()) and torch.equal(U0, U1)


def test_equilibration_is_exact_for_the_scale_invariant_kernels(ops):
    """gq_h_prepare factorises S H S with S = powers of two (H_jj S_jj^2 in [0.5, 2)) and returns U_hat S.  Scaling by
    powers of two commutes with fp32 arithmetic, so with the image GEMMs off (fp32 + exact-split bf16 kernels only) U
    is the same bit for bit with and without it -- the chain below the image levels is what r02 shipped."""
    C = 4096
    H = _hessian(C, 5)
    W = torch.randn(64, C, device="cuda")
    with ops.options(chol_3p_min=0):
        U0, _ = ops.h_prepare(H.clone(), W.clone(), 0.01)
        with ops.options(chol_no_equil=1):
            U1, _ = ops.h_prepare(H.clone(), W.clone(), 0.01)
    assert torch.equal(U0, U1)


@pytest.mark.parametrize("mode", ["fp16x2", "bf16x3"])
def test_image_chain_accuracy_wide_channel_scales(ops, mode):
    """U = chol_upper((H + damp)^-1) against the fp64 chain on Hessians whose channel scales span five decades plus
    x1000 outliers (H = S Z^T Z S built in fp32): the row-scaled fp16 images need the equilibration for their error
    bound; asserted rowwise -- max_j |U_ij - U64_ij| <= 4e-6 max_j |U64_ij| for every row -- and on the diagonal."""
    C = 3584
    torch.manual_seed(7)
    Z = torch.randn(2 * C, C, device="cuda").half()
    H = torch.zeros(C, C, device="cuda")
    ops.h_accumulate(H, Z, 0.0, 2.0 / 4)
    sig = torch.exp(torch.randn(C, device="cuda") * 2.5)
    sig[torch.randperm(C, device="cuda")[:8]] *= 1000.0
    H = H * sig[:, None] * sig[None, :]
    H = (H + H.T) * 0.5
    W = torch.randn(64, C, device="cuda")
    with ops.options(chol_planes=3 if mode == "bf16x3" else 2):
        Hc = H.clone()
        U, flag = ops.h_prepare(Hc, W.clone(), 0.01)
    assert int(flag.item()) == 0
    Hd = Hc.double()  # damped in place by the call (gptq.py:315-316)
    ref = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(Hd)), upper=True)
    d = (U.double() - ref).abs()
    assert (d.max(dim=1).values / ref.abs().max(dim=1).values).max().item() < 4e-6
    assert (d.diagonal() / ref.diagonal()).max().item() < 4e-6


def test_paired_node_launches_change_nothing(ops):
    """Below the image levels a recursion node issues its SYRK update and L21 X11 as ONE launch of 64-tiles
    (gemm32_pair_kernel); every output element is the same k-ordered fp32 chain as in the two separate launches
    (option chol_no_pair), so U is identical bit for bit."""
    C = 4096 + 896
    H = _hessian(C, 6)
    W = torch.randn(64, C, device="cuda")
    U0, f0 = ops.h_prepare(H.clone(), W.clone(), 0.01)
    with ops.options(chol_no_pair=1):
        U1, f1 = ops.h_prepare(H.clone(), W.clone(), 0.01)
    assert int(f0.item()) == 0 and torch.equal(U0, U1)


@pytest.mark.parametrize("name", ["Q3_K", "Q6_K", "Q4_K"])
def test_rtn_quant_scale_mse_golden(ops, name):
    """gq_rtn_quantize with gq_search_t.quant_scale = 1 on the fp32 embed-like weight of G15 (entries beyond +-32, where
    the grid search of quant_utils.py:164-191 really differs from absmax): the reference's own
    _quant_non_block_module(quant_scale="mse") run, bit for bit; Q4_K (make_k_quants) ignores the switch."""
    import numpy as np
    from conftest import load_golden
    g = load_golden("g15_rtn_mse")
    t = {"Q3_K": 11, "Q6_K": 14, "Q4_K": 12}[name]
    W = torch.from_numpy(g["W_f32"].copy()).cuda()
    q, d, s, dmin, m = ops.rtn_quantize(W, t, quant_scale="mse")
    assert np.array_equal(q.cpu().numpy(), g[f"f32_{name}_q"])
    assert np.array_equal(d.cpu().view(torch.int16).numpy().view(np.uint16), g[f"f32_{name}_d"])
    assert np.array_equal(s.cpu().numpy(), g[f"f32_{name}_s"])
    assert np.array_equal(dmin.cpu().view(torch.int16).numpy().view(np.uint16), g[f"f32_{name}_dmin"])
    assert np.array_equal(m.cpu().numpy(), g[f"f32_{name}_m"])
    qa, *_ = ops.rtn_quantize(W, t)  # absmax
    assert bool((qa != q).any()) == (name != "Q4_K")


@pytest.mark.parametrize("name", ["Q4_K", "Q6_K"])
def test_conv_handle_golden(ops, name):
    """GPTQ(nn.Conv2d).update / quantize on the GPU against the reference's run G16 (gptq.py:76, 96-104, 138-139): the
    unfolded patches through the SYRK (H to 3e-6 of its maximum), the flattened working copy exactly, the 5-tuple
    bit-exact GIVEN the reference's (W, U) and by rate (< 1 % of the ints) end to end."""
    import numpy as np
    from conftest import load_golden, triu_unpack
    from gptq_gguf_toolkit_amd.gptq import GPTQ
    g = load_golden("g16_conv_handle")
    t = {"Q4_K": 12, "Q6_K": 14}[name]
    cin, cout, k, st, pad = (int(v) for v in g["conv"])
    conv = torch.nn.Conv2d(cin, cout, kernel_size=k, stride=st, padding=pad, bias=False)
    conv.weight.data = torch.from_numpy(g["weight"].copy())
    conv = conv.cuda()
    h = GPTQ(conv, rel_damp=0.01, block_size=128)
    for x in g["x"]:
        h.update(torch.from_numpy(x.copy()).cuda())
    h.flush()
    assert h.num_samples == int(g["num_samples"])
    assert float((h.H.cpu() - torch.from_numpy(g["H_updated"])).abs().max()) <= 3e-6 * float(np.abs(g["H_updated"]).max())
    from gptq_gguf_toolkit_amd.quant_utils import GGMLQuantizationType
    q, d, s, dmin, m = h.quantize(GGMLQuantizationType(t))
    assert float((q.cpu().numpy() != g[f"{name}_q"]).mean()) < 0.01
    W = torch.from_numpy(g["W0"].copy()).cuda()
    U = torch.from_numpy(triu_unpack(g["U_triu"], W.shape[1])).cuda()
    q2, d2, s2, dmin2, m2 = ops.gptq_quantize(W, U, t, 128)
    assert np.array_equal(q2.cpu().numpy(), g[f"{name}_q"])
    assert np.array_equal(d2.cpu().view(torch.int16).numpy().view(np.uint16), g[f"{name}_d"])
    assert np.array_equal(s2.cpu().numpy(), g[f"{name}_s"]) and np.array_equal(m2.cpu().numpy(), g[f"{name}_m"])


@pytest.mark.parametrize("workload", ["tinyllama-block-q4k", "llama3-8b-block-q4k", "mixtral-block"])
def test_bench_eight_ranks_one_allgather_per_block(workload):
    """the N > 1 plumbing at the world size the driver will use, on the BASELINE blocks.  bench.py
    --gpus 8 through torch.distributed.run (RCCL when the box has 8 GPUs, else eight gloo ranks sharing the one GPU): every
    rank holds identical results, the line proves 8 ranks took part, and a block costs one collective per DISTINCT Hessian
    (the reference: one per Linear) -- a REDUCE to the owner when all Linears fed by that input belong to one rank, an
    all-reduce when several ranks need it (different owners, or a row-split matrix every rank factorises) -- + ONE all-gather
    of all results (the reference: 5 broadcasts per Linear) + no broadcast.  The expected kinds and BYTES ON THE WIRE follow
    from the owner map the line reports."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    n = 8
    backend = "nccl" if torch.cuda.device_count() >= n else "gloo"
    env = dict(os.environ, GQ_BENCH_VERIFY="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "gloo":
        # eight ranks SHARE one 288 GB GPU here: two chain lanes per rank instead of four (a Mixtral block holds 30 GB per
        # rank with four 14336-wide factorisations in flight -- fine on a GPU of its own, too much eight times over)
        env["GQ_CHAIN_STREAMS"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(27000 + os.getpid() % 2000), os.path.join(ROOT, "bench.py"), "--gpus",
           str(n), "--steps", "1", "--warmup", "1" if backend == "nccl" else "0", "--workload", workload, "--backend", backend,
           "--no-cpu-baseline", "--no-whole-model", "--no-side-legs"]
    if workload != "tinyllama-block-q4k":
        cmd += ["--calib-seqs", "32"]  # the plumbing, not the speed: a quarter of the 128 sequences (4 per rank)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=2400, cwd=ROOT)
    assert p.returncode == 0, "\n".join([ln[:300] for ln in p.stderr.splitlines() if "Error" in ln][:6]) + p.stderr[-1500:]
    assert "verify: all ranks hold identical results" in p.stderr
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-2])  # the full dict; the compact line is last
    assert line["n_gpus"] == n and line["ranks_seen"] == n and line["collective_backend"].startswith(backend)
    owners = line["config"]["owners"]
    shapes = bench.WORKLOADS[workload]["shapes"]
    groups = {}
    for name, (R, C, inp) in shapes.items():  # Linears fed by the same input share one Hessian
        groups.setdefault(inp, []).append(name)
    from gptq_gguf_toolkit_amd.dist_utils import hessian_payload_bytes
    want = {"all_reduce": 0, "reduce": 0}
    want_bytes = {"all_reduce": 0, "reduce": 0}
    for inp, names in groups.items():
        ranks = set()
        for nm in names:
            ranks |= set(range(n)) if str(owners[nm]).startswith("rows/") else {owners[nm]}
        kind = "reduce" if len(ranks) == 1 else "all_reduce"
        want[kind] += 1
        want_bytes[kind] += hessian_payload_bytes(shapes[names[0]][1])
    got, got_b = line["collectives_per_step"], line["collective_bytes_per_step"]
    assert (got["all_reduce"], got["reduce"], got["all_gather"], got["broadcast"]) == (want["all_reduce"], want["reduce"], 1, 0), got
    assert (got_b["all_reduce"], got_b["reduce"]) == (want_bytes["all_reduce"], want_bytes["reduce"]), (got_b, want_bytes)
    assert want["reduce"] >= 1  # o_proj's Hessian always has one owner
    if workload == "mixtral-block":
        assert want["reduce"] >= 8 and got["small_all_reduce"] >= 16  # every w2 has one owner; per-expert sample counts
        # 8 x 418 MB of w2 Hessians travel as reduces: half the link time of the all-reduces they replace
        assert want_bytes["reduce"] >= 8 * hessian_payload_bytes(14336)
    else:
        assert owners["down_proj"] == f"rows/{n}" and len({v for k, v in owners.items() if k != "down_proj"}) == 6, owners
        assert got["small_all_reduce"] == 0
    # ONE all-gather carries every rank's results: at least the block's packed ints + scales / 8
    assert got_b["all_gather"] >= sum(R * C for R, C, _ in shapes.values()) / n


_LOOP_HASH = r"""
import hashlib, sys, torch
sys.path.insert(0, {root!r})
from gptq_gguf_toolkit_amd import ops
g = torch.Generator(device="cuda").manual_seed(3)
R, C = 256, 9216  # >= 8 super-blocks of 1024 columns: the far updates take the helper-stream (persistent) form
X = torch.randn(10240, C, device="cuda", generator=g).half()
H = torch.zeros(C, C, device="cuda")
ops.h_accumulate(H, X, 0.0, 0.5)
W = torch.randn(R, C, device="cuda", generator=g) * 0.02
U, _ = ops.h_prepare(H, W.clone(), 0.01)
Wf = W.clone()
res = ops.gptq_quantize(Wf, U, 12, 128)
torch.cuda.synchronize()
h = hashlib.sha256()
for t in (Wf,) + tuple(res):
    h.update(t.cpu().contiguous().view(torch.uint8).numpy().tobytes())
print("HASH", h.hexdigest())
"""


def test_trailing_update_kernel_choices_are_bit_identical():
    """The column loop's trailing updates have several kernels: the 64-tile near kernel with whole K = 256 panels in LDS
    (default) vs the 128-tile chained kernel (option near64_maxn = 0), the pair form of the near updates vs a launch after
    every block (near_classic), the dedicated far kernel vs the generic chained one (chain_generic), set through the
    process's GQ_OPTIONS variable.  Every output element is the same k-ordered chain in all of them: W and the quantized
    tensors hash identically."""
    import subprocess
    import sys
    from conftest import ROOT
    hashes = {}
    for tag, opts in (("default", ""), ("near128", "near64_maxn=0"), ("classic", "near_classic"), ("generic_far", "chain_generic=1"),
                      ("quad", "near_quad=1"), ("far_b_by_dma", "far_bdma=1"), ("far_b_by_registers", "far_bdma=0"),
                      ("one_launch_per_block", "seg_pair=0")):  # r05: default = one column-loop launch per 256-column pair
        env = dict(os.environ, GQ_OPTIONS=opts)
        p = subprocess.run([sys.executable, "-c", _LOOP_HASH.format(root=ROOT)], env=env, capture_output=True, text=True,
                           timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        hashes[tag] = [ln for ln in p.stdout.splitlines() if ln.startswith("HASH")][-1]
    assert len(set(hashes.values())) == 1, hashes


def test_stage_to_host_copies_through_the_slot_mapping():
    """gq_stage_to_host (the saver's staging copy): device tensor -> pinned host memory by a kernel; unpinned memory is
    refused with an error instead of a fault."""
    from gptq_gguf_toolkit_amd import _cabi, ops
    st = torch.cuda.current_stream()
    for n, dt in ((1 << 20, torch.uint8), (12345, torch.uint8), (4096 * 33, torch.float16)):
        src = (torch.arange(n, device="cuda") % 251).to(dt)
        dst = torch.empty(n, dtype=dt).pin_memory()
        ops.stage_to_host(dst, src, st)
        torch.cuda.synchronize()
        assert torch.equal(dst, src.cpu())
    shm = torch.zeros(4096, dtype=torch.uint8).share_memory_()
    assert int(torch.cuda.cudart().cudaHostRegister(shm.data_ptr(), shm.numel(), 0)) == 0
    try:
        ops.stage_to_host(shm, torch.full((4096,), 7, dtype=torch.uint8, device="cuda"), st)
        torch.cuda.synchronize()
        assert int(shm.sum()) == 7 * 4096
    finally:
        torch.cuda.cudart().cudaHostUnregister(shm.data_ptr())
    with pytest.raises(_cabi.GQError):
        ops.stage_to_host(torch.empty(4096, dtype=torch.uint8), torch.zeros(4096, dtype=torch.uint8, device="cuda"), st)
