"""Host-side logic on CPU: CLI config rules, sharding / owner assignment, GGUF container + Llama
permute, and the driver (walk order, Hessian sharing, data.pth schema) with the compute backend
replaced by tests/fake_ops.py (oracle-backed).  World-size-2 gloo runs cover the N>1 path."""
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, load_golden

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def test_quant_config_rules(tmp_path):
    from gptq_gguf_toolkit_amd.quant import build_quant_config, parse_args
    from gptq_gguf_toolkit_amd.quant_utils import GGMLQuantizationType as T
    qc = build_quant_config("Q4_K", None)
    assert set(qc) == {"q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "down_proj", "up_proj", "embed_tokens",
                       "lm_head"} and all(v == T.Q4_K for v in qc.values())
    p = tmp_path / "config.json"
    p.write_text(json.dumps({"q_proj": "Q3_K", "down_proj": "Q6_K"}))
    qc = build_quant_config("Q4_K", str(p))  # the file REPLACES the default map (quant.py:203-217)
    assert qc == {"q_proj": T.Q3_K, "down_proj": T.Q6_K}
    with pytest.raises(ValueError):
        build_quant_config("Q8_0", None)
    with pytest.raises(ValueError):
        build_quant_config("Q4_K", str(tmp_path / "missing.json"))
    p.write_text(json.dumps({"q_proj": "Q4_0"}))
    with pytest.raises(ValueError):
        build_quant_config(None, str(p))
    a = parse_args(["--model_name_or_path", "m", "--quantizable_modules", "x", "--pre_block_modules", "e",
                    "--block_modules", "b", "--calibration_data", "c.pt", "--save_dir", "s"])
    assert (a.rel_damp, a.block_size, a.default_bit_width, a.rmin, a.rdelta, a.nstep, a.quant_scale) == \
        (1e-2, 128, "Q4_K", -1.0, 0.1, 20, "absmax")


def test_quant_scale_mse_through_the_quantizer_class(monkeypatch):
    """quant_utils.Quantizer(quant_scale="mse").get_scale_and_zero against the reference's outputs (G12): equal to absmax
    for weight-like panels, different -- and still the reference's -- for the wide one; make_k_quants ignores it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fake_ops
    fake_ops.install()
    from gptq_gguf_toolkit_amd.quant_utils import GGML_QUANT_SIZES, GGMLQuantizationType, Quantizer
    g = load_golden("g12_mse_scale")
    for qt in (GGMLQuantizationType.Q3_K, GGMLQuantizationType.Q6_K):
        bits, _, smq, G, SG, sdt, _ = GGML_QUANT_SIZES[qt]
        for tag in ("w", "wide"):
            for mode in ("absmax", "mse"):
                q = Quantizer()
                q.configure(bits, smq, G, sdt, SG, quant_scale=mode)
                d, s, dmin, m = q.get_scale_and_zero(torch.from_numpy(g[f"{qt.name}_{tag}_x"].copy()), qt)
                assert np.array_equal(d.view(torch.int16).numpy().view(np.uint16), g[f"{qt.name}_{tag}_{mode}_d"])
                assert np.array_equal(s.numpy(), g[f"{qt.name}_{tag}_{mode}_s"])
    torch.manual_seed(3)
    x = torch.randn(32, 256) * 30.0
    bits, _, smq, G, SG, sdt, _ = GGML_QUANT_SIZES[GGMLQuantizationType.Q4_K]
    outs = []
    for mode in ("absmax", "mse"):
        q = Quantizer()
        q.configure(bits, smq, G, sdt, SG, quant_scale=mode)
        outs.append(q.get_scale_and_zero(x.clone(), GGMLQuantizationType.Q4_K))
    assert all(torch.equal(a, b) for a, b in zip(*outs))


def test_rtn_forwards_quant_scale_and_mirrors_the_reference_error():
    """Quantizer._quant_non_block_module forwards quantizer_kwargs["quant_scale"] into the RTN of embed / lm_head
    (reference quantizer.py:293-295): on an fp32 weight the result is the reference's own (G15); on fp16 / bf16 weights
    with a make_quants type the reference raises (quant_utils.py:187) -- so does the driver, before any work."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fake_ops
    fake_ops.install()
    from gptq_gguf_toolkit_amd.quant_utils import GGMLQuantizationType
    from gptq_gguf_toolkit_amd.quantizer import Quantizer
    g = load_golden("g15_rtn_mse")
    drv = Quantizer.__new__(Quantizer)
    drv.non_block_fp32 = False
    for mode in ("mse", "absmax"):
        drv.quantizer_kwargs = {"quant_scale": mode}
        for qt in (GGMLQuantizationType.Q3_K, GGMLQuantizationType.Q6_K):
            q, d, s, dmin, m = drv._quant_non_block_module(torch.from_numpy(g["W_f32"].copy()), qt)
            same = np.array_equal(q.numpy(), g[f"f32_{qt.name}_q"]) and np.array_equal(s.numpy(), g[f"f32_{qt.name}_s"])
            assert same == (mode == "mse")
    drv.quantizer_kwargs = {"quant_scale": "mse"}
    for dt in (torch.float16, torch.bfloat16):
        w = torch.from_numpy(g["W_f32"].copy()).to(dt)
        for qt in (GGMLQuantizationType.Q3_K, GGMLQuantizationType.Q6_K):
            with pytest.raises(RuntimeError, match="quant_utils.py:187"):
                drv._quant_non_block_module(w, qt)
        drv._quant_non_block_module(w, GGMLQuantizationType.Q4_K)  # make_k_quants ignores quant_scale: runs
    drv.non_block_fp32 = True  # this build's opt-in fp32 search runs where the reference cannot
    q, *_ = drv._quant_non_block_module(torch.from_numpy(g["W_bf16"].copy()).to(torch.bfloat16), GGMLQuantizationType.Q6_K)
    assert q.shape == g["W_bf16"].shape


def test_sharding_and_owner_assignment():
    from gptq_gguf_toolkit_amd.dist_utils import assign_owners, shard_calibration
    data = list(range(11))
    assert shard_calibration(data, 0, 4) == [0, 1] and shard_calibration(data, 3, 4) == [6, 7]  # remainder dropped
    shapes = {"q": (4096, 4096), "k": (1024, 4096), "v": (1024, 4096), "o": (4096, 4096), "gate": (14336, 4096),
              "up": (14336, 4096), "down": (4096, 14336)}
    costs = {n: float(r) * c * (c + 128) for n, (r, c) in shapes.items()}
    for w in (1, 2, 4, 8):
        own = assign_owners(costs, w)
        assert set(own) == set(costs) and all(0 <= r < w for r in own.values())
        assert own == assign_owners(dict(reversed(list(costs.items()))), w)  # order independent => same on all ranks
    own = assign_owners(costs, 2)
    loads = [sum(costs[n] for n in own if own[n] == r) for r in range(2)]
    assert max(loads) == costs["down"]  # the largest matrix alone on one rank is the LPT optimum here


def test_llama_permute_and_names():
    from gptq_gguf_toolkit_amd.pack_gptq_into_gguf import map_tensor_name, permute
    # by hand from pack_gptq_into_gguf.py:2177-2183: rows [h, two, r] -> [h, r, two]
    n_head, hd = 2, 4
    w = torch.arange(n_head * hd * 3).reshape(n_head * hd, 3)
    out = permute(w, n_head, n_head)
    rows = [h * hd + two * (hd // 2) + r for h in range(n_head) for r in range(hd // 2) for two in range(2)]
    assert torch.equal(out, w[rows])
    assert torch.equal(permute(w, 4, 2), permute(w, 2, 2))  # GQA: k_proj uses n_kv_head
    v = torch.arange(8)  # 1-D (e.g. a [R] scale vector) permutes the same way
    assert torch.equal(permute(v, 2, 2), v[[0, 2, 1, 3, 4, 6, 5, 7]])
    assert map_tensor_name("model.layers.3.self_attn.k_proj.weight") == "blk.3.attn_k.weight"
    assert map_tensor_name("model.layers.0.mlp.down_proj.weight") == "blk.0.ffn_down.weight"
    assert map_tensor_name("lm_head.weight") == "output.weight"
    with pytest.raises(ValueError):
        map_tensor_name("model.layers.0.block_sparse_moe.experts.0.w1.weight")


def test_gguf_container_roundtrip(tmp_path):
    from gptq_gguf_toolkit_amd.gguf_writer import GGMLType, GGUFValueType, GGUFWriter, read_gguf
    w = GGUFWriter(str(tmp_path / "t.gguf"), "llama")
    w.add_uint32("llama.block_count", 2)
    w.add_float32("llama.rope.freq_base", 500000.0)
    w.add_array("tokenizer.ggml.tokens", ["a", "bc", "déf"], GGUFValueType.STRING)
    w.add_array("tokenizer.ggml.token_type", [1, 1, 3], GGUFValueType.INT32)
    f32 = np.arange(12, dtype=np.float32).reshape(3, 4)
    packed = (np.arange(5 * 2 * 144) % 251).astype(np.uint8).reshape(5, 288)  # Q4_K [5, 512]
    w.add_tensor("output_norm.weight", f32[0])
    w.add_tensor("a.weight", f32.astype(np.float16))
    w.add_tensor("blk.0.attn_q.weight", packed, raw_dtype=GGMLType.Q4_K)
    w.write()
    raw = open(tmp_path / "t.gguf", "rb").read()
    assert raw[:4] == b"GGUF" and int.from_bytes(raw[4:8], "little") == 3
    kv, ts = read_gguf(str(tmp_path / "t.gguf"))
    assert kv["general.architecture"] == "llama" and kv["llama.block_count"] == 2
    assert kv["tokenizer.ggml.tokens"] == ["a", "bc", "déf"] and kv["tokenizer.ggml.token_type"] == [1, 1, 3]
    assert ts["blk.0.attn_q.weight"][0] == (5, 512) and ts["blk.0.attn_q.weight"][1] == GGMLType.Q4_K
    assert np.array_equal(ts["blk.0.attn_q.weight"][2], packed.ravel())
    assert np.array_equal(ts["a.weight"][2].view(np.float16), f32.astype(np.float16).ravel())
    assert np.array_equal(ts["output_norm.weight"][2].view(np.float32), f32[0])


# --------------------------------------------------------------------------- driver (fake backend)
def _run_driver(save_dir, data, world=1):
    import fake_ops
    from make_golden_shim import MIXED, tiny_llama
    from gptq_gguf_toolkit_amd.quant_utils import GGMLQuantizationType as T
    from gptq_gguf_toolkit_amd.quantizer import Quantizer
    fake_ops.install()
    model = tiny_llama()
    drv = Quantizer(model, data_loader=data, quantizable_modules=r".*layers.*((q|k|v|o|gate|up|down)_proj)$",
                    quantizer_kwargs=dict(rel_damp=0.01, block_size=128, act_order=False, quant_scale="absmax",
                                          static_groups=False, rmin=-1.0, rdelta=0.1, nstep=20, verbose=False),
                    pre_block_modules=["model.embed_tokens"], block_modules="model.layers",
                    post_block_modules=["lm_head"], quant_non_block_modules=True, device="cpu", save_dir=save_dir)
    drv.quantize({k: T[v] for k, v in MIXED.items()})
    return model, fake_ops


def test_driver_walk_schema_and_hessian_sharing(tmp_path):
    """Same tree, keys, dtypes and shapes as the reference driver (G10); q/k/v and gate/up share one
    Hessian accumulation; results close to the reference's (H here comes from the fp64 oracle, the
    reference's from MKL fp32, so near-ties may flip: mismatch RATES are asserted)."""
    from make_golden_shim import tiny_calib
    g = load_golden("g10_driver")
    data = [([], {"input_ids": ids}) for ids in tiny_calib()]
    model, fake = _run_driver(str(tmp_path), data)
    names = sorted(os.listdir(tmp_path))
    assert names == list(g["names"])
    # 2 blocks x 4 distinct inputs, flushed once each (activations are buffered)
    assert fake.calls["h_accumulate"] == 2 * 4
    assert fake.calls["h_prepare"] == 2 * 4 and fake.calls["w_prepare"] == 2 * 3  # q/k/v + gate/up share U
    worst = 0.0
    for n in names:
        d = torch.load(os.path.join(tmp_path, n, "data.pth"), weights_only=True)
        assert set(d) == {"q_type", "qweight", "super_group_scale", "super_group_zero", "group_scale_quant",
                          "group_zero_quant"}
        assert d["q_type"] == int(g[f"{n}|q_type"])
        q = d["qweight"].numpy()
        assert q.dtype == g[f"{n}|qweight"].dtype and q.shape == g[f"{n}|qweight"].shape
        assert d["super_group_scale"].dtype == torch.float16 and d["group_scale_quant"].numpy().dtype == g[f"{n}|s"].dtype
        mism = float((q != g[f"{n}|qweight"]).mean())
        worst = max(worst, mism)
        if n in ("model.embed_tokens", "lm_head"):
            assert mism == 0.0  # RTN has no Hessian: bit-exact
    assert worst < 0.05, f"worst per-module int mismatch rate {worst:.3%}"


def test_handle_act_order_matches_reference():
    """GPTQ(act_order=True, static_groups=True).compute: the handle's permutation logic (dead-channel fix before the
    permutation, static scales of the ORIGINAL groups, qweight un-permuted) against the reference run G11.  U here
    comes from the fp64 oracle, the reference's from fp32 LAPACK, so near-ties may flip: a RATE is asserted; the
    scales have no Hessian in them and must be exact."""
    import fake_ops
    from gptq_gguf_toolkit_amd.gptq import GPTQ
    from gptq_gguf_toolkit_amd.quant_utils import GGMLQuantizationType as T
    fake_ops.install()
    g = load_golden("g11_act_order")
    W0, H0 = g["W0"], g["H0"]
    R, C = W0.shape
    for tag, qt, b in (("Q4_K_b128", T.Q4_K, 128), ("Q6_K_b64", T.Q6_K, 64)):
        layer = torch.nn.Linear(C, R, bias=False)
        layer.weight.data = torch.from_numpy(W0.copy())
        h = GPTQ(layer, rel_damp=0.01, block_size=b, static_groups=True, act_order=True)
        h.H = torch.from_numpy(H0.copy())
        h.make_working_copy()
        q, d, s, dmin, m = h.compute(qt)
        assert np.array_equal(d.view(torch.int16).numpy().view(np.uint16), g[f"{tag}_d"])
        assert np.array_equal(s.numpy(), g[f"{tag}_s"]) and np.array_equal(m.numpy(), g[f"{tag}_m"])
        mism = float((q.numpy() != g[f"{tag}_q"]).mean())
        assert mism < 0.02, f"{tag}: {mism:.3%} ints differ from the reference's act_order run"
    with pytest.raises(AssertionError):
        GPTQ(torch.nn.Linear(C, R, bias=False), act_order=True, static_groups=False)  # gptq.py:45-46


def test_data_pth_writer_failure_never_blocks_the_copier(tmp_path):
    """when torch.save fails in the writer (disk full, unwritable save_dir) the staging slots must keep
    coming back and the error must be reported at once -- the copier thread used to block forever on `freeq.get()`.
    The writer body runs here in a thread on plain queues (same protocol as the spawned process)."""
    import queue
    import threading
    from gptq_gguf_toolkit_amd import quantizer as qz
    bad_dir = tmp_path / "file_not_dir"
    bad_dir.write_text("x")  # os.makedirs(<file>/<name>) fails for every module
    slots = [torch.zeros(4096, dtype=torch.uint8) for _ in range(2)]
    inbox, freeq, outbox = queue.Queue(), queue.Queue(), queue.Queue()
    layout = [(0, 256, torch.uint8, (256,)), (256, 32, torch.float16, (16,)), (512, 8, torch.uint8, (8,)),
              (768, 32, torch.float16, (16,)), (1024, 8, torch.uint8, (8,))]
    th = threading.Thread(target=qz._writer_process, args=(str(bad_dir), slots, inbox, freeq, outbox), daemon=True)
    th.start()
    for i in range(5):  # more items than slots: each slot must come back although every write fails
        inbox.put(("slots", i % 2, [(f"m{i}", 12, layout)]))
        assert freeq.get(timeout=20) == i % 2
    status, info = outbox.get(timeout=20)
    assert status == "error" and "m0" in info  # reported with the first failure, not at the end
    inbox.put(None)
    th.join(timeout=20)
    assert not th.is_alive()
    assert outbox.get(timeout=5)[0] == "error"  # the final message repeats it
    # and a healthy directory still works through the same body
    good = tmp_path / "ok"
    good.mkdir()
    th = threading.Thread(target=qz._writer_process, args=(str(good), slots, inbox, freeq, outbox), daemon=True)
    th.start()
    inbox.put(("slots", 0, [("m", 12, layout)]))
    assert freeq.get(timeout=20) == 0
    # one slot carrying two modules (a block's Linears travel together): both files appear
    layout2 = [(2048 + off, n, dt, shape) for off, n, dt, shape in layout]
    slots[1][:256] = 7
    slots[1][2048:2048 + 256] = 9
    inbox.put(("slots", 1, [("a", 12, layout), ("b", 12, layout2)]))
    assert freeq.get(timeout=20) == 1
    inbox.put(None)
    th.join(timeout=20)
    assert outbox.get(timeout=5)[0] == "ok" and (good / "m" / "data.pth").is_file()
    a, b = (torch.load(good / n / "data.pth") for n in ("a", "b"))
    assert int(a["qweight"][0]) == 7 and int(b["qweight"][0]) == 9 and a["q_type"] == 12


def test_missing_data_pth_is_reported(tmp_path):
    """the files are dealt to the ranks; rank 0 checks after the barrier that all of them are in save_dir."""
    from gptq_gguf_toolkit_amd.quantizer import Quantizer
    q = Quantizer.__new__(Quantizer)
    q.save_dir = str(tmp_path)
    q._saved_names = ["a.b", "c.d"]
    (tmp_path / "a.b").mkdir()
    (tmp_path / "a.b" / "data.pth").write_text("x")
    with pytest.raises(RuntimeError, match="save_dir must be shared"):
        q._check_saved_files()
    (tmp_path / "c.d").mkdir()
    (tmp_path / "c.d" / "data.pth").write_text("x")
    q._check_saved_files()


def _conv_from_golden(g, device="cpu"):
    cin, cout, k, st, pad = (int(v) for v in g["conv"])
    conv = torch.nn.Conv2d(cin, cout, kernel_size=k, stride=st, padding=pad, bias=False)
    conv.weight.data = torch.from_numpy(g["weight"].copy())
    return conv.to(device)


def test_conv_handle_matches_reference_run(oracle):
    """The _ConvNd branch of the handle (reference gptq.py:76, 96-104, 138-139) against the reference's own run on a
    small nn.Conv2d (G16): nn.Unfold patches as Hessian rows, the flattened [out, in*kh*kw] working copy, the 5-tuple.
    H to fp32 summation order; the ints exactly GIVEN the reference's (W, U) and by rate through the handle (U here
    comes from the fp64 oracle, the reference's from fp32 LAPACK)."""
    import fake_ops
    from conftest import triu_unpack
    from gptq_gguf_toolkit_amd.gptq import GPTQ
    from gptq_gguf_toolkit_amd.quant_utils import GGMLQuantizationType as T
    fake_ops.install()
    g = load_golden("g16_conv_handle")
    for qt in (T.Q4_K, T.Q6_K):
        conv = _conv_from_golden(g)
        h = GPTQ(conv, rel_damp=0.01, block_size=128)
        assert (h.d_row, h.d_col) == g["W0"].shape
        for x in g["x"]:
            h.update(torch.from_numpy(x.copy()))
        h.flush()
        assert h.num_samples == int(g["num_samples"])
        assert np.abs(h.H.numpy() - g["H_updated"]).max() <= 2e-6 * np.abs(g["H_updated"]).max()
        q, d, s, dmin, m = h.quantize(qt)
        assert q.shape == g[f"{qt.name}_q"].shape
        assert float((q.numpy() != g[f"{qt.name}_q"]).mean()) < 0.01
        U = triu_unpack(g["U_triu"], g["W0"].shape[1])
        _, oq, od, os_, odm, om = oracle.gptq_step(g["W0"], U, int(qt), block_size=128)
        assert np.array_equal(oq, g[f"{qt.name}_q"]) and np.array_equal(od, g[f"{qt.name}_d"])
        assert np.array_equal(os_, g[f"{qt.name}_s"]) and np.array_equal(om, g[f"{qt.name}_m"])


def _worker(rank, world, port, tmp, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden_shim import tiny_calib
    from gptq_gguf_toolkit_amd import dist_utils
    data = dist_utils.shard_calibration(tiny_calib(), rank, world)
    data = [([], {"input_ids": ids}) for ids in data]
    model, fake = _run_driver(tmp, data, world)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ret[rank] = (sd, dict(fake.calls, **{f"coll_{k}": v for k, v in dist_utils.collective_calls.items()}))
    dist.barrier()
    dist.destroy_process_group()


def _worker_moe(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fake_ops
    fake_ops.install()
    from gptq_gguf_toolkit_amd.gptq import GPTQ
    torch.manual_seed(0)
    C, R = 256, 32
    lin = torch.nn.Linear(C, R, bias=False)
    toks = torch.randn(8, C)                       # the expert's tokens of the whole calibration set
    mine = toks[:3] if rank == 0 else toks[3:]     # 3 tokens routed on rank 0, 5 on rank 1
    h = GPTQ(lin, allow_no_samples=True)
    h.update(mine[:2])                             # 2-D [tokens, C] inputs, as an expert Linear sees them
    h.update(mine[2:])
    h.sync_hessian()
    idle = GPTQ(torch.nn.Linear(C, R, bias=False), allow_no_samples=True)  # an expert no rank routed to
    idle.sync_hessian()
    is_eye = torch.equal(idle.H, torch.eye(C))     # before quantize(): the damping is applied to H in place
    q = idle.quantize(12)[0]
    ret[rank] = (h.H.clone(), idle.no_samples, is_eye, q.clone())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_moe_sample_weighted_hessian_and_idle_expert():
    """Expert Linears see [tokens, C] inputs (batch = tokens, reference gptq.py:86) and a different token count
    on every rank: the reduced H must be (2/N) sum x x^T over ALL tokens, i.e. the sample-weighted mean of the
    ranks' Hessians; an expert without any token falls back to H = I on every rank and still quantizes."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_moe, args=(world, 33000 + os.getpid() % 2000, ret), nprocs=world, join=True)
    torch.manual_seed(0)
    torch.nn.Linear(256, 32, bias=False)
    toks = torch.randn(8, 256)
    want = (2.0 / 8) * toks.double().T @ toks.double()
    for r in range(world):
        H, flag, is_eye, q = ret[r]
        assert (H.double() - want).abs().max() < 1e-5 * want.abs().max()
        assert flag and is_eye
    assert torch.equal(ret[0][3], ret[1][3])


def test_row_split_plan():
    from gptq_gguf_toolkit_amd.dist_utils import row_slice, row_split_names
    shapes = {"q": (4096, 4096), "k": (1024, 4096), "v": (1024, 4096), "o": (4096, 4096), "gate": (14336, 4096),
              "up": (14336, 4096), "down": (4096, 14336)}
    costs = {n: float(r) * c * (c + 128) for n, (r, c) in shapes.items()}
    os.environ.pop("GQ_ROW_SPLIT", None)
    assert row_split_names(costs, 1) == set() and row_split_names(costs, 2) == set()
    assert row_split_names(costs, 4) == {"down"} and row_split_names(costs, 8) == {"down"}  # 56 % of a Llama block
    for R, w in ((4096, 8), (4096, 4), (1000, 8), (130, 4)):
        sl = [row_slice(R, r, w) for r in range(w)]
        assert sl[0][0] == 0 and all(a[1] == b[0] for a, b in zip(sl, sl[1:])) and sl[-1][1] == R  # a partition
        assert all(c[2] % 128 == 0 and c[1] - c[0] <= c[2] for c in sl)


def _worker_split(rank, world, port, tmp, ret):
    os.environ["GQ_ROW_SPLIT"] = "all"
    _worker(rank, world, port, tmp, ret)


def test_two_rank_row_split_matches_owner_mode(tmp_path):
    """Row-split mode (every rank factorises, quantizes its own rows, all-gather) produces the same model and the
    same data.pth tree as owner mode (one rank per matrix, broadcast): rows are independent given U."""
    world = 2
    mgr = mp.Manager()
    outs = []
    for k, fn in enumerate((_worker, _worker_split)):
        ret = mgr.dict()
        d = str(tmp_path / f"m{k}")
        os.makedirs(d)
        mp.spawn(fn, args=(world, 31000 + k * 7 + os.getpid() % 2000, d, ret), nprocs=world, join=True)
        outs.append((d, ret[0], ret[1]))
    (d0, (sd0, c0), _), (d1, (sd1, c1), (sd1b, c1b)) = outs
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]) and torch.equal(sd1[k], sd1b[k]), f"{k} differs"
    # rank 0 walked (its rows of) every matrix, rank 1 those with more than one 128-row slice
    assert c1["gptq_quantize"] == 14 and 0 < c1b["gptq_quantize"] <= 14
    for n in sorted(os.listdir(d0)):
        a = torch.load(os.path.join(d0, n, "data.pth"), weights_only=True)
        b = torch.load(os.path.join(d1, n, "data.pth"), weights_only=True)
        for key in a:
            assert (a[key] == b[key]).all() if torch.is_tensor(a[key]) else a[key] == b[key], f"{n}/{key}"


def test_two_rank_gloo_matches_single_rank(tmp_path):
    """calib-sharded H (all-reduce AVG) + per-matrix owners + one all-gather per block: every rank ends with the same
    quantized model, equal to the 1-rank run on the full calibration set up to H rounding order."""
    from make_golden_shim import tiny_calib
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    d2 = str(tmp_path / "w2")
    os.makedirs(d2)
    mp.spawn(_worker, args=(world, 29000 + os.getpid() % 2000, d2, ret), nprocs=world, join=True)
    sd0, calls0 = ret[0]
    sd1, calls1 = ret[1]
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), f"ranks disagree on {k}"
    # owners split the work: each rank ran some, not all, of the 14 column loops
    assert 0 < calls0["gptq_quantize"] < 14 and calls0["gptq_quantize"] + calls1["gptq_quantize"] == 14
    # the data-path collectives of the two blocks: one all-reduce per DISTINCT Hessian (4 per block; the reference: 7),
    # ONE all-gather per block for all results (the reference: 5 broadcasts per Linear), no broadcast
    for c in (calls0, calls1):
        # (r04: a Hessian whose Linears all have ONE owner travels as a reduce to that rank, SURVEY section 5 iii)
        assert (c["coll_all_reduce"] + c["coll_reduce"], c["coll_all_gather"], c["coll_broadcast"]) == (8, 2, 0), c
        assert c["coll_reduce"] >= 2, c
    d1 = str(tmp_path / "w1")
    os.makedirs(d1)
    data = [([], {"input_ids": ids}) for ids in tiny_calib()]
    model1, _ = _run_driver(d1, data)
    assert sorted(os.listdir(d1)) == sorted(os.listdir(d2))  # rank 0 wrote the same tree
    tot = diff = 0
    for n in sorted(os.listdir(d1)):
        a = torch.load(os.path.join(d1, n, "data.pth"), weights_only=True)["qweight"]
        b = torch.load(os.path.join(d2, n, "data.pth"), weights_only=True)["qweight"]
        tot += a.numel()
        diff += int((a != b).sum())
    assert diff / tot < 0.02, f"{diff / tot:.3%} ints differ between 1-rank and 2-rank runs"


def _worker_g13(rank, world, port, ret, device="cpu", backend="gloo"):
    """The build's handle on one rank of a calibration-sharded run, fed the REFERENCE's per-rank Hessians (G13)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group(backend, rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    if device == "cpu":
        import fake_ops
        fake_ops.install()
    from gptq_gguf_toolkit_amd.gptq import GPTQ
    g = load_golden("g13_two_rank")
    R, C = g["W0"].shape
    out = {}
    for tag, qt in (("Q4_K", 12), ("Q6_K", 14)):
        layer = torch.nn.Linear(C, R, bias=False)
        layer.weight.data = torch.from_numpy(g["W0"].copy())
        layer = layer.to(device)
        h = GPTQ(layer, rel_damp=0.01, block_size=128)
        h.H = torch.from_numpy(g[f"H_local_rank{rank}"].copy()).to(device)
        h.num_samples = 2
        h.quantization_pre_step()  # all_reduce AVG over the ranks (gptq.py:131-132)
        out["H_reduced"] = h.H.cpu().numpy().copy()
        res = h.step(qt)           # the owner computes, everyone receives (gptq.py:286-293)
        out[tag] = [t.cpu() for t in res]
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def check_g13(ret):
    g = load_golden("g13_two_rank")
    Href = g["H_reduced"]
    dead = np.diag(Href) == 1.0
    for r in (0, 1):
        H = ret[r]["H_reduced"].copy()
        H[np.diag_indices_from(H)] = np.where(dead & (np.diag(H) == 0), 1.0, np.diag(H))  # fix of gptq.py:134
        assert np.array_equal(H, Href), f"rank {r}: reduced H differs from the reference's all_reduce(AVG)"
    rates = {}
    for tag in ("Q4_K", "Q6_K"):
        a, b = ret[0][tag], ret[1][tag]
        assert all(torch.equal(x, y) for x, y in zip(a, b)), f"{tag}: ranks hold different results"
        q, d, s, dmin, m = a
        rates[tag] = float((q.numpy() != g[f"{tag}_q"]).mean())
        assert rates[tag] < 0.02, f"{tag}: {rates[tag]:.3%} ints differ from the reference's 2-rank run"
        assert float((s.numpy() != g[f"{tag}_s"]).mean()) < 0.05
    return rates


def test_two_rank_run_vs_reference_two_rank_golden(oracle):
    """SURVEY 8c G11 (stored as G13): the REFERENCE run on 2 gloo ranks (per-rank shards -> all_reduce AVG ->
    rank 0 steps -> broadcast).  The build's 2-rank run from the same per-rank Hessians must hold the same reduced
    H bit for bit and the same results on both ranks (ints up to the tolerance-class Cholesky: a rate); and given
    the reference's own U the oracle reproduces the reference's ints exactly (dead channel included)."""
    g = load_golden("g13_two_rank")
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_g13, args=(2, 27000 + os.getpid() % 2000, ret), nprocs=2, join=True)
    check_g13(ret)
    W = g["W0"].copy()
    dead = np.diag(g["H_reduced"]) == 1.0
    W[:, dead] = 0.0
    for tag, qt in (("Q4_K", 12), ("Q6_K", 14)):
        C = W.shape[1]
        U = np.zeros((C, C), np.float32)
        U[np.triu_indices(C)] = g[f"{tag}_U_triu"]
        _, q, d, s, dmin, m = oracle.gptq_step(W, U, qt, block_size=128)
        assert np.array_equal(q, g[f"{tag}_q"]) and np.array_equal(d, g[f"{tag}_d"]) and np.array_equal(s, g[f"{tag}_s"])
        assert np.array_equal(dmin, g[f"{tag}_dmin"]) and np.array_equal(m, g[f"{tag}_m"])


# --------------------------------------------------------------------------- Mixtral-style MoE blocks (f3)
MOE_QCFG = {"q_proj": "Q6_K", "k_proj": "Q6_K", "v_proj": "Q6_K", "o_proj": "Q6_K", "w1": "Q3_K", "w2": "Q3_K",
            "w3": "Q3_K", "embed_tokens": "Q6_K", "lm_head": "Q6_K"}


def _run_moe_driver(save_dir, ids, device="cpu"):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tiny_moe import MOE_REGEX, TinyMoE
    from gptq_gguf_toolkit_amd.quant_utils import GGMLQuantizationType as T
    from gptq_gguf_toolkit_amd.quantizer import Quantizer
    model = TinyMoE().to(device)
    data = [([], {"input_ids": t}) for t in ids]
    drv = Quantizer(model, data_loader=data, quantizable_modules=MOE_REGEX,
                    quantizer_kwargs=dict(rel_damp=0.01, block_size=128, act_order=False, quant_scale="absmax",
                                          static_groups=False, rmin=-1.0, rdelta=0.1, nstep=20, verbose=False),
                    pre_block_modules=["model.embed_tokens"], block_modules="model.layers",
                    post_block_modules=["lm_head"], quant_non_block_modules=True, device=device, save_dir=save_dir)
    drv.quantize({k: T[v] for k, v in MOE_QCFG.items()})
    return model, drv


def _worker_moe_driver(rank, world, port, tmp, ret, device="cpu", backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group(backend, rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    if device == "cpu":
        import fake_ops
        fake_ops.install()
    from tiny_moe import moe_calib
    ids = moe_calib("A" if rank == 0 else "B")  # rank 0 routes to experts {0, 1} only, rank 1 to {2, 3} only
    model, drv = _run_moe_driver(tmp, ids, device)
    ret[rank] = {k: v.cpu().clone() for k, v in model.state_dict().items()}
    dist.barrier()
    dist.destroy_process_group()


def check_moe_tree(save_dir, n_layers=2, n_experts=5):
    names = sorted(os.listdir(save_dir))
    want = ["lm_head", "model.embed_tokens"]
    for i in range(n_layers):
        want += [f"model.layers.{i}.self_attn.{p}_proj" for p in "qkvo"]
        want += [f"model.layers.{i}.block_sparse_moe.experts.{e}.w{k}" for e in range(n_experts) for k in (1, 2, 3)]
    assert names == sorted(want)
    for n in names:
        d = torch.load(os.path.join(save_dir, n, "data.pth"), weights_only=True)
        is_expert = ".experts." in n
        assert d["q_type"] == (11 if is_expert else 14)
        assert d["qweight"].dtype == torch.int8 and int(d["qweight"].min()) >= (-4 if is_expert else -32)


def test_moe_driver_single_rank_with_idle_expert(tmp_path):
    """A Mixtral-layout block through the Quantizer driver: expert Linears see [tokens, C] inputs, w1/w3 of an
    expert share one Hessian, the expert no token is routed to gets H = I and is still quantized and saved."""
    import fake_ops
    from tiny_moe import moe_calib
    fake_ops.install()
    model, drv = _run_moe_driver(str(tmp_path), moe_calib("AB"))
    check_moe_tree(str(tmp_path))
    # per block: 4 routed experts x (w1/w3 shared + w2) + attention (q/k/v shared + o) = 10 Hessians accumulated;
    # the idle expert has none (H = I), its w1 / w3 / w2 are three independent identity "factorisations"
    assert fake_ops.calls["h_accumulate"] == 2 * 10
    assert drv.schedule_stats["reused_U"] == 4 + 2  # 4 x w3 + k, v (last block's stats)


def test_moe_two_ranks_with_rank_local_experts(tmp_path):
    """experts that receive tokens on ONE rank only.  The rank without tokens never fires the hooks, so it
    cannot know that w1 and w3 share an input; the sharing pattern (hence the number of collectives) is agreed on
    across ranks before the first all-reduce.  Both ranks must end with the same model, equal to the single-rank run
    on the union of the calibration data up to the tolerance-class stages."""
    from tiny_moe import moe_calib
    mgr = mp.Manager()
    ret = mgr.dict()
    d2 = str(tmp_path / "w2")
    os.makedirs(d2)
    mp.spawn(_worker_moe_driver, args=(2, 26000 + os.getpid() % 2000, d2, ret), nprocs=2, join=True)
    for k in ret[0]:
        assert torch.equal(ret[0][k], ret[1][k]), f"ranks disagree on {k}"
    check_moe_tree(d2)
    import fake_ops
    fake_ops.install()
    d1 = str(tmp_path / "w1")
    os.makedirs(d1)
    _run_moe_driver(d1, moe_calib("AB"))
    tot = diff = 0
    for n in sorted(os.listdir(d1)):
        a = torch.load(os.path.join(d1, n, "data.pth"), weights_only=True)["qweight"]
        b = torch.load(os.path.join(d2, n, "data.pth"), weights_only=True)["qweight"]
        tot += a.numel()
        diff += int((a != b).sum())
    assert diff / tot < 0.02, f"{diff / tot:.3%} ints differ between the 1-rank and the 2-rank MoE runs"


# --------------------------------------------------------------------------- GGUF container bytes (f1)
def test_gguf_container_bytes_vs_independent_spec_writer(tmp_path):
    """Byte-level golden from an independent serialiser written from the GGUF v3 spec (tests/gguf_spec_writer.py):
    header, value-type ids, string / array encoding, tensor-info order (ne[0] innermost), offsets and the 32-byte
    alignment of every tensor, for F32 / F16 / BF16 / K-quant payloads in 1, 2 and 3 dimensions."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gguf_spec_writer import serialise
    from gptq_gguf_toolkit_amd.gguf_writer import GGMLType, GGUFValueType, GGUFWriter
    rng = np.random.default_rng(0)
    f32 = rng.standard_normal(7).astype(np.float32)                  # 28 bytes: needs padding
    f16 = rng.standard_normal((3, 5)).astype(np.float16)             # 30 bytes
    q4 = rng.integers(0, 256, (5, 2 * 144), dtype=np.uint8)          # Q4_K [5, 512]
    q3e = rng.integers(0, 256, (4, 3, 110), dtype=np.uint8)          # stacked experts: Q3_K [4, 3, 256]
    bf = rng.integers(0, 65536, (2, 6), dtype=np.uint16)             # BF16 [2, 6] as bytes
    w = GGUFWriter(str(tmp_path / "a.gguf"), "llama")
    w.add_string("general.type", "model")
    w.add_uint32("llama.block_count", 2)
    w.add_float32("llama.rope.freq_base", 500000.0)
    w.add_bool("tokenizer.ggml.add_space_prefix", False)
    w.add_array("tokenizer.ggml.tokens", ["a", "bc", "déf", ""], GGUFValueType.STRING)
    w.add_array("tokenizer.ggml.token_type", [1, 1, 3, 5], GGUFValueType.INT32)
    w.add_tensor("rope_freqs.weight", f32)
    w.add_tensor("a.weight", f16)
    w.add_tensor("blk.0.attn_q.weight", q4, raw_dtype=GGMLType.Q4_K)
    w.add_tensor("blk.0.ffn_down_exps.weight", q3e, raw_dtype=GGMLType.Q3_K)
    w.add_tensor("b.weight", bf.view(np.uint8).reshape(2, 12), raw_dtype=GGMLType.BF16)
    w.write()
    want = serialise(
        [("general.architecture", "str", "llama"), ("general.type", "str", "model"), ("llama.block_count", "u32", 2),
         ("llama.rope.freq_base", "f32", 500000.0), ("tokenizer.ggml.add_space_prefix", "bool", False),
         ("tokenizer.ggml.tokens", ("arr", "str"), ["a", "bc", "déf", ""]),
         ("tokenizer.ggml.token_type", ("arr", "i32"), [1, 1, 3, 5])],
        [("rope_freqs.weight", (7,), "F32", f32.tobytes()), ("a.weight", (3, 5), "F16", f16.tobytes()),
         ("blk.0.attn_q.weight", (5, 512), "Q4_K", q4.tobytes()),
         ("blk.0.ffn_down_exps.weight", (4, 3, 256), "Q3_K", q3e.tobytes()), ("b.weight", (2, 6), "BF16", bf.tobytes())])
    got = open(tmp_path / "a.gguf", "rb").read()
    assert got == want, f"first differing byte at {next(i for i, (x, y) in enumerate(zip(got, want)) if x != y)}"


def test_packer_metadata_order_rope_freqs_and_experts(tmp_path, monkeypatch):
    """convert(): the Llama KV set in the reference's order (pack_gptq_into_gguf.py:412-441, :594-638, :2160-2175),
    rope_freqs.weight first for rope_type llama3 (:2259-2287), linear scaling keys, loud refusal of
    other scaling types, and Mixtral's router + stacked expert tensors."""
    import fake_ops
    from safetensors.torch import save_file
    from gptq_gguf_toolkit_amd import packing_utils
    from gptq_gguf_toolkit_amd.gguf_writer import read_gguf
    from gptq_gguf_toolkit_amd.pack_gptq_into_gguf import convert, rope_freqs_llama3
    monkeypatch.setattr(packing_utils, "_ops", fake_ops)  # no GPU here: the oracle's bit-packer stands in
    monkeypatch.setattr(packing_utils, "_dev", lambda t: t.contiguous())
    h, ffn, L, E = 256, 512, 1, 3
    cfg = {"architectures": ["MixtralForCausalLM"], "hidden_size": h, "intermediate_size": ffn, "num_hidden_layers": L,
           "num_attention_heads": 4, "num_key_value_heads": 2, "vocab_size": 64, "max_position_embeddings": 128,
           "rms_norm_eps": 1e-5, "rope_theta": 500000.0, "num_local_experts": E, "num_experts_per_tok": 2,
           "rope_scaling": {"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                            "original_max_position_embeddings": 64}}
    g = torch.Generator().manual_seed(0)
    sd = {"model.embed_tokens.weight": torch.randn(64, h, generator=g), "model.norm.weight": torch.ones(h),
          "lm_head.weight": torch.randn(64, h, generator=g)}
    p = "model.layers.0."
    for n, shape in (("self_attn.q_proj", (h, h)), ("self_attn.k_proj", (h // 2, h)), ("self_attn.v_proj", (h // 2, h)),
                     ("self_attn.o_proj", (h, h)), ("block_sparse_moe.gate", (E, h))):
        sd[p + n + ".weight"] = torch.randn(*shape, generator=g)
    sd[p + "input_layernorm.weight"] = torch.ones(h)
    sd[p + "post_attention_layernorm.weight"] = torch.ones(h)
    for e in range(E):
        for wid, shape in (("w1", (ffn, h)), ("w2", (h, ffn)), ("w3", (ffn, h))):
            sd[f"{p}block_sparse_moe.experts.{e}.{wid}.weight"] = torch.randn(*shape, generator=g)
    hf = tmp_path / "hf"
    hf.mkdir()
    save_file(sd, str(hf / "model.safetensors"))
    (hf / "config.json").write_text(json.dumps(cfg))
    # quantized results for the w2 experts only (Q3_K, signed ints) -- written as the driver writes them
    qdir = tmp_path / "q"
    for e in range(E):
        d = qdir / f"model.layers.0.block_sparse_moe.experts.{e}.w2"
        d.mkdir(parents=True)
        torch.save({"q_type": 11, "qweight": torch.randint(-4, 4, (h, ffn), generator=g).to(torch.int8),
                    "super_group_scale": torch.rand(h, ffn // 256, generator=g).half(),
                    "super_group_zero": torch.zeros(h, ffn // 256).half(),
                    "group_scale_quant": torch.randint(0, 32, (h, ffn // 16), generator=g).to(torch.int8),
                    "group_zero_quant": torch.zeros(h, ffn // 16).to(torch.int8)}, str(d / "data.pth"))
    out = convert(hf, qdir, tmp_path / "m.gguf", "f16", vocab=False)
    kv, ts = read_gguf(str(out))
    assert list(kv) == ["general.architecture", "general.type", "general.name", "general.size_label", "llama.block_count",
                        "llama.context_length", "llama.embedding_length", "llama.feed_forward_length",
                        "llama.attention.head_count", "llama.attention.head_count_kv", "llama.rope.freq_base",
                        "llama.attention.layer_norm_rms_epsilon", "llama.expert_count", "llama.expert_used_count",
                        "general.file_type", "llama.vocab_size", "llama.rope.dimension_count",
                        "general.quantization_version"]
    assert kv["llama.expert_count"] == E and kv["llama.rope.dimension_count"] == 64
    assert list(ts)[0] == "rope_freqs.weight"  # generate_extra_tensors() leads the chain (:290)
    assert np.array_equal(ts["rope_freqs.weight"][2].view(np.float32), rope_freqs_llama3(cfg).numpy())
    assert ts["blk.0.ffn_gate_inp.weight"][1] == 0  # the router stays F32 (:358-376)
    shape, gt, raw = ts["blk.0.ffn_down_exps.weight"]
    assert shape == (E, h, ffn) and gt == 11 and raw.size == E * h * ffn // 256 * 110
    d1 = torch.load(str(qdir / "model.layers.0.block_sparse_moe.experts.1.w2" / "data.pth"), weights_only=True)
    want = packing_utils.pack_tensor(11, d1["qweight"], d1["super_group_scale"], d1["group_scale_quant"],
                                     d1["super_group_zero"], d1["group_zero_quant"])
    assert np.array_equal(raw.reshape(E, h, -1)[1], want)  # expert 1's packed rows sit at index 1 of the stack
    assert ts["blk.0.ffn_gate_exps.weight"][0] == (E, ffn, h) and ts["blk.0.ffn_gate_exps.weight"][1] == 1  # F16
    cfg["rope_scaling"] = {"type": "linear", "factor": 4.0}
    (hf / "config.json").write_text(json.dumps(cfg))
    kv, ts = read_gguf(str(convert(hf, qdir, tmp_path / "m2.gguf", "f16", vocab=False)))
    assert kv["llama.rope.scaling.type"] == "linear" and kv["llama.rope.scaling.factor"] == 4.0 and "rope_freqs.weight" not in ts
    cfg["rope_scaling"] = {"rope_type": "yarn", "factor": 4.0}
    (hf / "config.json").write_text(json.dumps(cfg))
    with pytest.raises(NotImplementedError, match="rope_scaling"):
        convert(hf, qdir, tmp_path / "m3.gguf", "f16", vocab=False)
    cfg.pop("rope_scaling")
    (hf / "config.json").write_text(json.dumps(cfg))
    with pytest.raises(FileNotFoundError):
        convert(hf, qdir, tmp_path / "m5.gguf", "f16")


def test_packer_sentencepiece_vocabulary(tmp_path, monkeypatch):
    """f1 (VERDICT r02 missing #1): a checkpoint with a `tokenizer.model` (TinyLlama, Llama-2, Mistral, Mixtral) gets the
    `llama` tokenizer of the reference's _set_vocab_sentencepiece (pack_gptq_into_gguf.py:1018-1118, set_vocab
    :2126-2139) + gguf.SpecialVocab's keys.  Read back through an INDEPENDENT spec-level reader and compared with what
    the SentencePiece model itself says, piece by piece (llama.cpp token types: 1 normal, 2 unknown, 3 control,
    4 user-defined, 5 unused, 6 byte).  Whole-file identity with gguf-py stays unpinned (not installable here)."""
    import shutil
    import fake_ops
    from safetensors.torch import save_file
    from sentencepiece import SentencePieceProcessor
    from gguf_spec_reader import read_kv
    from gptq_gguf_toolkit_amd import packing_utils
    from gptq_gguf_toolkit_amd.pack_gptq_into_gguf import convert
    monkeypatch.setattr(packing_utils, "_ops", fake_ops)
    monkeypatch.setattr(packing_utils, "_dev", lambda t: t.contiguous())
    h, V = 256, 384  # the model's vocab_size exceeds the 320 pieces of the fixture: padded with [PAD<i>] / UNUSED
    cfg = {"architectures": ["LlamaForCausalLM"], "hidden_size": h, "intermediate_size": 512, "num_hidden_layers": 0,
           "num_attention_heads": 4, "num_key_value_heads": 4, "vocab_size": V, "max_position_embeddings": 128,
           "rms_norm_eps": 1e-5, "rope_theta": 10000.0, "bos_token_id": 1, "eos_token_id": 2, "pad_token_id": 400}
    g = torch.Generator().manual_seed(0)
    hf = tmp_path / "hf"
    hf.mkdir()
    save_file({"model.embed_tokens.weight": torch.randn(V, h, generator=g), "model.norm.weight": torch.ones(h),
               "lm_head.weight": torch.randn(V, h, generator=g)}, str(hf / "model.safetensors"))
    (hf / "config.json").write_text(json.dumps(cfg))
    shutil.copy(os.path.join(ROOT, "tests", "golden", "spm", "tokenizer.model"), hf / "tokenizer.model")
    (hf / "added_tokens.json").write_text(json.dumps({"<extra_a>": 330, "<too_far>": 9999}))
    (hf / "tokenizer_config.json").write_text(json.dumps({
        "bos_token": "<s>", "eos_token": {"content": "</s>"}, "unk_token": "<unk>", "add_bos_token": True,
        "add_eos_token": False, "chat_template": "{{ bos_token }}{% for m in messages %}{{ m.content }}{% endfor %}",
        "added_tokens_decoder": {"340": {"content": "<|tool|>", "special": False},
                                 "341": {"content": "\u2581user\u2581word", "special": False},
                                 "342": {"content": "<ctl>", "special": True}}}))
    (hf / "tokenizer.json").write_text(json.dumps({"added_tokens": [{"id": 0, "content": "<unk>"}, {"id": 1, "content": "<s>"},
                                                                    {"id": 2, "content": "</s>"}], "model": {"type": "BPE"}}))
    (tmp_path / "noquant").mkdir()
    out = convert(hf, tmp_path / "noquant", tmp_path / "spm.gguf", "f16")
    kv, types, _ = read_kv(str(out))
    assert kv["tokenizer.ggml.model"] == "llama" and kv["tokenizer.ggml.pre"] == "default"
    toks, scores, tt = kv["tokenizer.ggml.tokens"], kv["tokenizer.ggml.scores"], kv["tokenizer.ggml.token_type"]
    assert len(toks) == len(scores) == len(tt) == V
    assert types["tokenizer.ggml.scores"] == ("array", 6) and types["tokenizer.ggml.token_type"] == ("array", 5)
    sp = SentencePieceProcessor()
    sp.LoadFromFile(str(hf / "tokenizer.model"))
    n_byte = 0
    for i in range(sp.vocab_size()):
        assert toks[i] == sp.IdToPiece(i) and scores[i] == np.float32(sp.GetScore(i))
        want = 2 if sp.IsUnknown(i) else 3 if sp.IsControl(i) else 5 if sp.IsUnused(i) else 6 if sp.IsByte(i) else 1
        assert tt[i] == want
        n_byte += want == 6
    assert n_byte == 256 and toks[4] == "<0x00>" and tt[0] == 2 and tt[1] == 3 and tt[2] == 3  # byte fallback, <unk>, <s>, </s>
    assert toks[3] == "<custom>" and tt[3] == 1  # a user-defined symbol INSIDE the model file: NORMAL, the reference
    # has no such branch (:1057-1065 test unknown / control / unused / byte only)
    assert (toks[330], scores[330], tt[330]) == ("<extra_a>", -1000.0, 4)        # added_tokens.json
    assert (toks[340], tt[340]) == ("<|tool|>", 3)                                # looks special -> CONTROL
    assert (toks[341], tt[341]) == (" user word", 4)                              # U+2581 pre-normalised, USER_DEFINED
    assert (toks[342], scores[342], tt[342]) == ("<ctl>", -1000.0, 3)
    assert (toks[325], scores[325], tt[325]) == ("[PAD325]", -10000.0, 5)         # beyond the model file
    # gguf.SpecialVocab: ids via tokenizer_config -> added_tokens, config.json fills the rest, ids >= n_vocab dropped
    assert kv["tokenizer.ggml.bos_token_id"] == 1 and kv["tokenizer.ggml.eos_token_id"] == 2
    assert kv["tokenizer.ggml.unknown_token_id"] == 0 and "tokenizer.ggml.padding_token_id" not in kv
    assert kv["tokenizer.ggml.add_bos_token"] is True and kv["tokenizer.ggml.add_eos_token"] is False
    assert kv["tokenizer.chat_template"].startswith("{{ bos_token }}")
    keys = [k for k in kv if k.startswith("tokenizer.")]
    assert keys == ["tokenizer.ggml.model", "tokenizer.ggml.pre", "tokenizer.ggml.tokens", "tokenizer.ggml.scores",
                    "tokenizer.ggml.token_type", "tokenizer.ggml.bos_token_id", "tokenizer.ggml.eos_token_id",
                    "tokenizer.ggml.unknown_token_id", "tokenizer.ggml.add_bos_token", "tokenizer.ggml.add_eos_token",
                    "tokenizer.chat_template"]  # the order of the reference's writer calls (:1021-1028)
    # a tokenizer.json of a SentencePiece-derived vocabulary WITHOUT tokenizer.model: refused, not mis-written
    (hf / "tokenizer.model").unlink()
    (hf / "tokenizer.json").write_text(json.dumps({"added_tokens": [], "model": {"type": "BPE", "byte_fallback": True,
                                                                                 "vocab": {}, "merges": []}}))
    with pytest.raises(NotImplementedError, match="_set_vocab_llama_hf"):
        convert(hf, tmp_path / "noquant", tmp_path / "x.gguf", "f16")


def test_q8_0_encoder_against_the_scalar_restatement():
    """f1: `quantize_q8_0` (gguf-py 0.17.1 `gguf.quants.quantize(.., Q8_0)`, which the reference calls for
    un-quantized tensors under --outtype q8_0: pack_gptq_into_gguf.py:404-405,419) against an independent element-by-element
    restatement of ggml-quants.c quantize_row_q8_0_ref (tests/ggml_spec.py): bytes identical, incl. all-zero blocks, exact
    .5 ties (away from zero), the +-127 ends, large and tiny scales (blocks whose 1 / d overflows are undefined behaviour in the C routine too); decode error <= d / 2."""
    from ggml_spec import q8_0_decode, q8_0_encode_scalar
    from gptq_gguf_toolkit_amd.gguf_writer import QuantError, quantize_q8_0
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((24, 96)) * 0.02).astype(np.float32)
    x[0, :32] = 0.0                                                    # d = 0 -> id = 0
    x[1, :32] = np.float32(1.5) * np.arange(32, dtype=np.float32)      # multiples of d / 2 ...: ties
    x[2, :8] = [0.5, -0.5, 1.5, -1.5, 2.5, -2.5, 63.5, -63.5]
    x[2, 8] = 127.0                                                    # d = 1: the values above are exact .5 ties
    x[3, :32] = rng.standard_normal(32).astype(np.float32) * 1e4       # large
    x[4, :32] = rng.standard_normal(32).astype(np.float32) * 1e-7      # d below the fp16 normal range
    x[5, :32] = -x[1, :32]
    x[6, 32:64] = np.float32(2.0e-36)                                   # tiny: d rounds to an fp16 zero, 1 / d stays finite
    got = quantize_q8_0(x)
    assert got.dtype == np.uint8 and got.shape == (24, 96 // 32 * 34)
    assert got.tobytes() == q8_0_encode_scalar(x)
    y = q8_0_decode(got.tobytes(), x.size).reshape(x.shape)
    d = got.reshape(-1, 34)[:, :2].copy().view(np.float16).astype(np.float32).reshape(24, 3)
    amax = np.abs(x.reshape(24, 3, 32)).max(-1)
    # |x - d16 q| <= d/2 (rounding of q) + 127 |d - d16| (fp16 rounding of the stored scale, 2^-11 relative or half a
    # subnormal step)
    bound = amax / 127 / 2 + 127 * np.abs(d - amax / 127) + 1e-30
    assert (np.abs(y - x).reshape(24, 3, 32).max(-1) <= bound * 1.0001).all()
    q = got.reshape(-1, 34)[:, 2:].view(np.int8)
    assert q.min() >= -127 and q.max() == 127
    xh = x.astype(np.float16)                                          # fp16 checkpoints are widened first
    assert quantize_q8_0(xh).tobytes() == q8_0_encode_scalar(xh.astype(np.float32))
    with pytest.raises(QuantError):
        quantize_q8_0(np.zeros((4, 48), np.float32))


@pytest.mark.parametrize("ckpt_dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("outtype", ["f32", "f16", "bf16", "q8_0", "auto"])
def test_pack_into_gguf_outtypes(tmp_path, monkeypatch, outtype, ckpt_dtype):
    """--outtype (reference :8804, :395-410, :144-152): the type of every tensor GPTQ did not quantize, `general.file_type`,
    1-D tensors and norms always F32, `auto` = f16 for an fp16 checkpoint and bf16 otherwise, Q8_0's fallback to F16 for rows
    that are no multiple of 32; the quantized tensor keeps its K-quant payload whatever the outtype."""
    import fake_ops
    from safetensors.torch import save_file
    from ggml_spec import q8_0_encode_scalar
    from gptq_gguf_toolkit_amd import packing_utils
    from gptq_gguf_toolkit_amd.gguf_writer import read_gguf
    from gptq_gguf_toolkit_amd.pack_gptq_into_gguf import convert, permute
    monkeypatch.setattr(packing_utils, "_ops", fake_ops)
    monkeypatch.setattr(packing_utils, "_dev", lambda t: t.contiguous())
    h, V = 256, 48
    cfg = {"architectures": ["LlamaForCausalLM"], "hidden_size": h, "intermediate_size": 336, "num_hidden_layers": 1,
           "num_attention_heads": 4, "num_key_value_heads": 2, "vocab_size": V, "max_position_embeddings": 128,
           "rms_norm_eps": 1e-5, "rope_theta": 10000.0}
    g = torch.Generator().manual_seed(1)
    p = "model.layers.0."
    sd = {"model.embed_tokens.weight": torch.randn(V, h, generator=g), "model.norm.weight": torch.rand(h, generator=g),
          p + "self_attn.q_proj.weight": torch.randn(h, h, generator=g), p + "self_attn.k_proj.weight": torch.randn(h // 2, h, generator=g),
          p + "mlp.down_proj.weight": torch.randn(h, 336, generator=g),  # 336 % 32 != 0: no Q8_0 rows
          p + "input_layernorm.weight": torch.rand(h, generator=g), "lm_head.weight": torch.randn(V, h, generator=g)}
    sd = {k: v.to(ckpt_dtype) for k, v in sd.items()}
    hf = tmp_path / "hf"
    hf.mkdir()
    save_file(sd, str(hf / "model.safetensors"))
    (hf / "config.json").write_text(json.dumps(cfg))
    qdir = tmp_path / "q" / "model.layers.0.self_attn.q_proj"
    qdir.mkdir(parents=True)
    qd = {"q_type": 12, "qweight": torch.randint(0, 16, (h, h), generator=g).to(torch.uint8),
          "super_group_scale": torch.rand(h, 1, generator=g).half(), "super_group_zero": torch.rand(h, 1, generator=g).half(),
          "group_scale_quant": torch.randint(0, 64, (h, 8), generator=g).to(torch.uint8),
          "group_zero_quant": torch.randint(0, 64, (h, 8), generator=g).to(torch.uint8)}
    torch.save(qd, str(qdir / "data.pth"))
    tm = {}
    kv, ts = read_gguf(str(convert(hf, tmp_path / "q", tmp_path / "m.gguf", outtype, vocab=False, timing=tm)))
    # r06: the pipelined converter (payloads made at write time from mmap'd data.pth, un-permute and pack on the device, the
    # replaced checkpoint tensors never read) writes the file of the tensor-by-tensor flow (reference :282-349), byte for byte
    convert(hf, tmp_path / "q", tmp_path / "m_flow.gguf", outtype, vocab=False, pipelined=False)
    assert (tmp_path / "m.gguf").read_bytes() == (tmp_path / "m_flow.gguf").read_bytes()
    assert {"load", "h2d", "permute_pack", "d2h", "write", "wait", "hf_read", "plain"} <= set(tm)
    eff = outtype if outtype != "auto" else ("f16" if ckpt_dtype == torch.float16 else "bf16")
    assert kv["general.file_type"] == {"f32": 0, "f16": 1, "bf16": 32, "q8_0": 7}[eff]
    gg = {"f32": 0, "f16": 1, "bf16": 30, "q8_0": 8}[eff]
    for name in ("output_norm.weight", "blk.0.attn_norm.weight"):
        assert ts[name][1] == 0  # F32 whatever the outtype (:355-356)
    shape, gt, raw = ts["blk.0.attn_q.weight"]
    assert gt == 12 and shape == (h, h)  # the GPTQ tensor: Q4_K bytes, rows un-permuted (:320-324)
    five = [permute(qd[k], 4, 4) for k in ("qweight", "super_group_scale", "group_scale_quant", "super_group_zero", "group_zero_quant")]
    assert np.array_equal(raw.reshape(h, -1), packing_utils.pack_tensor(12, *five))
    wide = lambda t: t if t.dtype in (torch.float16, torch.float32) else t.float()  # noqa: E731  (:296-297)
    for name, src in (("token_embd.weight", sd["model.embed_tokens.weight"]), ("output.weight", sd["lm_head.weight"]),
                      ("blk.0.attn_k.weight", permute(sd[p + "self_attn.k_proj.weight"], 4, 2)),
                      ("blk.0.ffn_down.weight", sd[p + "mlp.down_proj.weight"])):
        shape, gt, raw = ts[name]
        src = wide(src)
        assert shape == tuple(src.shape)
        if eff == "q8_0" and src.shape[-1] % 32:
            assert gt == 1 and raw.tobytes() == src.half().numpy().tobytes()  # QuantError -> F16 (:419-424)
            continue
        assert gt == gg, (name, gt)
        if eff == "f32":
            assert raw.tobytes() == src.float().numpy().tobytes()
        elif eff == "f16":
            assert raw.tobytes() == src.half().numpy().tobytes()
        elif eff == "bf16":
            assert raw.tobytes() == src.to(torch.bfloat16).view(torch.int16).numpy().tobytes()
        else:
            assert raw.tobytes() == q8_0_encode_scalar(src.float().numpy())
    for bad in ("tq1_0", "tq2_0"):
        with pytest.raises(NotImplementedError, match="ternary"):
            convert(hf, tmp_path / "q", tmp_path / "t.gguf", bad, vocab=False)


def test_calibration_batch_merges_block_inputs(tmp_path):
    """calibration_batch (beyond the reference): 4 samples per block forward give the same tree and -- the Hessians
    being the same sums -- the same integers up to the tolerance-class rate; inputs that differ in anything but the
    hidden states are never merged."""
    import fake_ops
    from make_golden_shim import MIXED, tiny_calib, tiny_llama
    from gptq_gguf_toolkit_amd.quant_utils import GGMLQuantizationType as T
    from gptq_gguf_toolkit_amd.quantizer import Quantizer, _batch_block_inputs
    fake_ops.install()
    outs = []
    for b in (1, 4):
        d = str(tmp_path / f"b{b}")
        os.makedirs(d)
        model = tiny_llama()
        data = [([], {"input_ids": ids}) for ids in tiny_calib()]
        drv = Quantizer(model, data_loader=data, quantizable_modules=r".*layers.*((q|k|v|o|gate|up|down)_proj)$",
                        quantizer_kwargs=dict(rel_damp=0.01, block_size=128), pre_block_modules=["model.embed_tokens"],
                        block_modules="model.layers", post_block_modules=["lm_head"], quant_non_block_modules=True,
                        device="cpu", save_dir=d, calibration_batch=b)
        fake_ops.calls["h_accumulate"] = 0
        drv.quantize({k: T[v] for k, v in MIXED.items()})
        outs.append(d)
    tot = diff = 0
    for n in sorted(os.listdir(outs[0])):
        a = torch.load(os.path.join(outs[0], n, "data.pth"), weights_only=True)["qweight"]
        b = torch.load(os.path.join(outs[1], n, "data.pth"), weights_only=True)["qweight"]
        tot += a.numel()
        diff += int((a != b).sum())
    assert sorted(os.listdir(outs[0])) == sorted(os.listdir(outs[1])) and diff / tot < 0.02, diff / tot
    h = [torch.randn(1, 8, 4) for _ in range(5)]
    pos = torch.arange(8).unsqueeze(0)
    args = [(x,) for x in h]
    kws = [{"position_ids": pos.clone(), "use_cache": False} for _ in h]
    kws[3]["position_ids"] = pos + 1  # sample 3 differs from both neighbours: runs [0, 1, 2], [3], [4]
    a2, k2 = _batch_block_inputs(args, kws, 4)
    assert [a[0].shape[0] for a in a2] == [3, 1, 1] and torch.equal(a2[0][0], torch.cat(h[:3]))
    assert torch.equal(k2[1]["position_ids"], pos + 1) and a2[2][0] is h[4]
    a3, _ = _batch_block_inputs(args[:3], kws[:3], 2)
    assert [a[0].shape[0] for a in a3] == [2, 1]


def test_gguf_splitter_database(tmp_path):
    """f4: the EvoPress database producer (mapper/gguf_splitter.py:291-446) on a GGUF written by this package: raw tensor
    bytes as "<bpw>-<type>.pth", per-tensor metadata, manifest with the file's KV metadata, layer database."""
    from gptq_gguf_toolkit_amd.gguf_splitter import GGUFSplitter, main
    from gptq_gguf_toolkit_amd.gguf_writer import GGMLType, GGUFWriter
    rng = np.random.default_rng(1)
    q4 = rng.integers(0, 256, (6, 2 * 144), dtype=np.uint8)
    q3 = rng.integers(0, 256, (4, 110), dtype=np.uint8)
    nrm = rng.standard_normal(8).astype(np.float32)
    w = GGUFWriter(str(tmp_path / "m.gguf"), "llama")
    w.add_uint32("llama.block_count", 1)
    w.add_array("tokenizer.ggml.token_type", [1, 3], 5)
    w.add_tensor("blk.0.attn_q.weight", q4, raw_dtype=GGMLType.Q4_K)
    w.add_tensor("blk.0.ffn_down.weight", q3, raw_dtype=GGMLType.Q3_K)
    w.add_tensor("output_norm.weight", nrm)
    w.write()
    main([str(tmp_path / "m.gguf"), str(tmp_path / "db"), "--exact"])
    db = tmp_path / "db"
    assert (db / "blk.0.attn_q.weight" / "4.5-Q4_K.pth").read_bytes() == q4.tobytes()
    assert (db / "blk.0.ffn_down.weight" / "3.4375-Q3_K.pth").read_bytes() == q3.tobytes()
    assert (db / "output_norm.weight" / "32-F32.pth").read_bytes() == nrm.tobytes()
    meta = json.loads((db / "blk.0.attn_q.weight" / "4.5-Q4_K-metadata.json").read_text())["tensor_info"]
    assert meta["shape"] == [512, 6] and meta["np_shape"] == [6, 288] and meta["np_dtype"] == "uint8" and meta["type"] == 12
    man = json.loads((db / "manifest.json").read_text())
    assert man["metadata"]["llama.block_count"] == {"types": [4], "value": 1}
    assert man["metadata"]["tokenizer.ggml.token_type"] == {"types": [9, 5], "value": [1, 3]}
    assert man["layers"]["blk.0.ffn_down.weight"]["bitwidths"]["3.4375"]["quantization"] == "Q3_K"
    ldb = json.loads((db / "gguf_layer_database.json").read_text())
    raw = open(tmp_path / "m.gguf", "rb").read()
    o = ldb["blk.0.attn_q.weight"]["data_offset"]
    assert raw[o:o + q4.nbytes] == q4.tobytes() and ldb["output_norm.weight"]["exact_bitwidth"] == 32.0
    s2 = GGUFSplitter(str(tmp_path / "m.gguf"), str(tmp_path / "db2"))
    s2.split_gguf_model()
    assert (tmp_path / "db2" / "blk.0.attn_q.weight" / "4.pth").exists() and (tmp_path / "db2" / "output_norm.weight" / "32.pth").exists()


def test_fast_obq_handle_against_reference_run():
    """f4: the FastOBQ handle (EvoPress' uniform-grid GPTQ, evopress/src/fast_obq.py) fed the seeded inputs of G14
    through update() -> quantize([2, 3, 4, 8]) against the reference's own run.  The first group's grid has no
    Hessian in it and must be exact; U comes from the fp64 oracle here and from fp32 LAPACK there, so later ints are
    held to a rate."""
    import fake_ops
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden_shim import OBQ_CASES, obq_inputs
    from gptq_gguf_toolkit_amd.fast_obq import FastOBQ
    fake_ops.install()
    g = load_golden("g14_fast_obq")
    for tag, R, C, gs, sym, block in OBQ_CASES:
        W, xs = obq_inputs(R, C)
        layer = torch.nn.Linear(C, R, bias=False)
        layer.weight.data = W.clone()
        h = FastOBQ(layer, bitwidth_options=[2, 3, 4, 8], group_size=gs, sym=sym, rel_damp=0.01, block_size=block)
        for x in xs:
            h.update(x)
        q, s, z, perm = h.quantize([2, 3, 4, 8])
        assert perm is None and torch.equal(layer.weight.data, W)  # the handle never writes the layer
        assert np.array_equal(h.W.numpy(), g[f"{tag}_W0"])
        for b in (2, 3, 4, 8):
            assert q[b].dtype == torch.uint8 and s[b].shape == (R, C // gs) and z[b].dtype == W.dtype
            assert np.array_equal(s[b][:, 0].numpy(), g[f"{tag}_b{b}_scale"][:, 0])
            assert np.array_equal(z[b][:, 0].numpy(), g[f"{tag}_b{b}_zero"][:, 0])
            assert np.array_equal(q[b][:, 0].numpy(), g[f"{tag}_b{b}_q"][:, 0])
            mism = float((q[b].numpy() != g[f"{tag}_b{b}_q"]).mean())
            assert mism < 0.01, f"{tag} {b} bits: {mism:.3%} ints differ from the reference run"
        h.reset()
        assert h.H is None and h.num_samples == 0
    with pytest.raises(ValueError):
        FastOBQ(torch.nn.Linear(256, 8, bias=False), bitwidth_options=[9])
    with pytest.raises(NotImplementedError):
        FastOBQ(torch.nn.Linear(256, 8, bias=False), bitwidth_options=[4], perchannel=False)


def test_fast_obq_act_order_and_layer_dtype():
    """act_order (fast_obq.py:146-149): columns walked in descending order of the damped diagonal, results in
    permuted positions with `perm` returned; scale / zero come back in the layer's dtype (:150-151)."""
    import fake_ops
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden_shim import obq_inputs
    from gptq_gguf_toolkit_amd.fast_obq import FastOBQ
    fake_ops.install()
    R, C = 16, 256
    W, xs = obq_inputs(R, C)
    layer = torch.nn.Linear(C, R, bias=False).half()
    layer.weight.data = W.half()
    h = FastOBQ(layer, bitwidth_options=[4], group_size=128, rel_damp=0.01, block_size=128, act_order=True)
    for x in xs:
        h.update(x)
    h.flush()
    H0 = h.H.numpy().copy()
    q, s, z, perm = h.quantize([4])
    d = np.diag(H0).copy()
    d[d == 0] = 1.0
    want = np.argsort(-(d + np.float32(0.01) * d.mean(dtype=np.float32)), kind="stable")
    assert np.array_equal(perm.numpy(), want) and s[4].dtype == torch.float16
    Wp = W.numpy()[:, want].copy()
    Wp[:, want == 3] = 0
    U, *_ = O.h_prepare(H0[want][:, want], Wp, 0.01, obq_order=True)
    _, q_ref, s_ref, z_ref = O.obq_step(h.W.numpy(), U, 4, 128, False, 128)
    assert np.array_equal(q[4].numpy(), q_ref) and np.array_equal(s[4].numpy(), s_ref.astype(np.float16))


def test_fused_forward_patch_targets_and_restore():
    """forward_fused.py: patches exactly the HF modules whose text equals Llama's, falls back to the original torch code
    for tensors the kernels do not take (here: CPU fp32 -> identical logits), and restores everything on exit."""
    import torch
    from transformers.models.llama import modeling_llama as M
    from transformers.models.gemma import modeling_gemma as G
    from gptq_gguf_toolkit_amd.forward_fused import fused_forward
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden_shim import tiny_calib, tiny_llama
    model = tiny_llama()
    ids = tiny_calib(n=1)[0]
    orig = (M.LlamaRMSNorm.forward, M.apply_rotary_pos_emb, M.LlamaMLP.forward, G.GemmaRMSNorm.forward)
    with torch.no_grad():
        want = model(input_ids=ids).logits
        with fused_forward() as patched:
            assert {"LlamaRMSNorm.forward", "LlamaMLP.forward"} <= set(patched)
            assert any(p.endswith("modeling_llama.apply_rotary_pos_emb") for p in patched)
            assert "GemmaRMSNorm.forward" not in patched  # (1 + weight) scaling: a different module text
            assert M.LlamaRMSNorm.forward is not orig[0] and M.apply_rotary_pos_emb is not orig[1]
            got = model(input_ids=ids).logits
    assert torch.equal(want, got)
    assert (M.LlamaRMSNorm.forward, M.apply_rotary_pos_emb, M.LlamaMLP.forward, G.GemmaRMSNorm.forward) == orig
    with pytest.raises(ZeroDivisionError):
        with fused_forward():
            1 / 0
    assert M.LlamaRMSNorm.forward is orig[0]
    # levels: "exact" (the Quantizer's default) leaves RMSNorm alone
    from gptq_gguf_toolkit_amd.forward_fused import level_of
    assert [level_of(v) for v in (None, False, "off", "0", True, "all", "exact")] == ["off"] * 4 + ["all"] * 2 + ["exact"]
    with pytest.raises(ValueError):
        level_of("fast")
    with fused_forward("exact") as patched:  # RMSNorm is installed too, but only ever runs its bit-verified kernel
        assert {"LlamaMLP.forward", "LlamaRMSNorm.forward"} <= set(patched) and "GemmaRMSNorm.forward" not in patched
        assert M.apply_rotary_pos_emb is not orig[1]
        assert torch.equal(model(input_ids=ids).logits, want)  # CPU fp32: the originals
    assert (M.LlamaRMSNorm.forward, M.apply_rotary_pos_emb, M.LlamaMLP.forward) == orig[:3]


def test_fused_forward_refuses_a_llama_whose_text_moved(monkeypatch):
    """the source check is against PINNED definitions, not against whatever the installed transformers
    ships -- a LlamaMLP.forward that grew a branch is left to the eager code (for every family), with a warning; unknown
    keywords of apply_rotary_pos_emb go to the original."""
    import warnings
    from transformers.models.llama import modeling_llama as M
    from gptq_gguf_toolkit_amd import forward_fused as ff

    def forward(self, x):  # what a `pretraining_tp`-style change would look like
        if getattr(self.config, "pretraining_tp", 1) > 1:
            return None
        return self.down_proj(self.act_fn(self.gate_proj(x)) * self.up_proj(x))
    monkeypatch.setattr(M.LlamaMLP, "forward", forward)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        names = [f"{getattr(o, '__name__', o)}.{a}" for o, a, _ in ff._targets()]
    assert any("mlp" in str(x.message) for x in w)
    assert not any(n.endswith("MLP.forward") for n in names)  # Mistral / Qwen copies equal the OLD text: not patched either
    assert "LlamaRMSNorm.forward" in names and any(n.endswith("apply_rotary_pos_emb") for n in names)
    calls = []
    wrapped = ff._rope(lambda q, k, cos, sin, *a, **kw: calls.append((a, kw)) or "orig")
    assert wrapped(1, 2, 3, 4, position_ids=5) == "orig" and calls == [((), {"position_ids": 5})]
    assert wrapped(1, 2, 3, 4, 1, 7) == "orig"


def test_forward1_stops_at_the_last_hooked_linear(tmp_path):
    """VERDICT r03 next #3a: forward #1 of a block only feeds the Hessian hooks (reference quantizer.py:150-151 discards its
    output), so after the block's first sample it ends at the last hooked Linear -- down_proj's GEMM and the residual add
    are not run.  Same tree, same bytes as with the full forwards; down_proj is entered 1 + n (forward #2) times per
    block fewer... counted."""
    import fake_ops
    from make_golden_shim import MIXED, tiny_calib, tiny_llama
    from gptq_gguf_toolkit_amd.quant_utils import GGMLQuantizationType as T
    from gptq_gguf_toolkit_amd.quantizer import Quantizer
    fake_ops.install()
    trees, counts = [], []
    ids = tiny_calib()
    for interrupt in (True, False):
        d = str(tmp_path / f"i{int(interrupt)}")
        os.makedirs(d)
        model = tiny_llama()
        n_calls = [0]
        model.model.layers[0].mlp.down_proj.register_forward_hook(lambda *a: n_calls.__setitem__(0, n_calls[0] + 1))
        drv = Quantizer(model, data_loader=[([], {"input_ids": t}) for t in ids],
                        quantizable_modules=r".*layers.*((q|k|v|o|gate|up|down)_proj)$",
                        quantizer_kwargs=dict(rel_damp=0.01, block_size=128), pre_block_modules=["model.embed_tokens"],
                        block_modules="model.layers", post_block_modules=["lm_head"], quant_non_block_modules=True,
                        device="cpu", save_dir=d, interrupt_forward1=interrupt)
        drv.quantize({k: T[v] for k, v in MIXED.items()})
        trees.append(d)
        counts.append(n_calls[0])
    n = len(ids)
    assert counts == [1 + n, 2 * n]  # forward #1: the first sample only | every sample; forward #2: every sample
    for name in sorted(os.listdir(trees[0])):
        a = torch.load(os.path.join(trees[0], name, "data.pth"), weights_only=True)
        b = torch.load(os.path.join(trees[1], name, "data.pth"), weights_only=True)
        for k in a:
            assert (a[k] == b[k]) if k == "q_type" else torch.equal(a[k], b[k]), (name, k)
    assert sorted(os.listdir(trees[0])) == sorted(os.listdir(trees[1]))


def test_forward1_runs_in_full_when_the_last_linear_is_not_hooked(tmp_path):
    """with a --quantizable_modules regex that leaves out the block's structurally last Linear (down_proj), the
    learned "last hooked Linear" (up_proj) is NOT where the forward ends: forward #1 must run in full, every sample."""
    import fake_ops
    from make_golden_shim import tiny_calib, tiny_llama
    from gptq_gguf_toolkit_amd.quant_utils import GGMLQuantizationType as T
    from gptq_gguf_toolkit_amd.quantizer import Quantizer
    fake_ops.install()
    ids = tiny_calib()
    model = tiny_llama()
    n_calls = [0]
    model.model.layers[0].mlp.down_proj.register_forward_hook(lambda *a: n_calls.__setitem__(0, n_calls[0] + 1))
    drv = Quantizer(model, data_loader=[([], {"input_ids": t}) for t in ids],
                    quantizable_modules=r".*layers.*((q|k|v|o|gate|up)_proj)$",
                    quantizer_kwargs=dict(rel_damp=0.01, block_size=128), pre_block_modules=["model.embed_tokens"],
                    block_modules="model.layers", post_block_modules=["lm_head"], quant_non_block_modules=False,
                    device="cpu", save_dir=str(tmp_path), interrupt_forward1=True)
    drv.quantize({k: T.Q4_K for k in ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj")})
    assert n_calls[0] == 2 * len(ids)  # forward #1 and forward #2 of block 0, every sample
    assert "forward1_interrupts" not in drv.schedule_stats


def test_vocab_tail_codellama_granite_and_pair_merges(tmp_path):
    """the rest of LlamaModel.set_vocab (reference pack_gptq_into_gguf.py:2138-2158) -- the CodeLlama
    fill-in-the-middle ids (vocab 32016), granite's add_bos_token = False (vocab 49152, a duplicate key raises like gguf-py) --
    and merges given as [a, b] pairs, whose inner spaces gguf-py's SpecialVocab encodes as chr(ord(' ') + 256)."""
    from gptq_gguf_toolkit_amd.gguf_writer import GGUFWriter
    from gptq_gguf_toolkit_amd.pack_gptq_into_gguf import add_tokenizer
    d = tmp_path / "hf"
    d.mkdir()
    vocab = {"a": 0, "b": 1, "Ġc": 2, "ab": 3}
    (d / "tokenizer.json").write_text(json.dumps({"added_tokens": [{"id": 4, "content": "<|x|>", "special": True}],
                                                  "model": {"type": "BPE", "vocab": vocab, "merges": [["a", "b"], ["a b", "c"], "b c"]}}))
    (d / "tokenizer_config.json").write_text(json.dumps({"add_prefix_space": False}))

    def kv_of(vs):
        w = GGUFWriter(str(tmp_path / "x.gguf"), "llama")
        add_tokenizer(w, d, vs)
        return {k: v for k, _, v, _ in w.kv}, [k for k, *_ in w.kv]
    kv, order = kv_of(32016)
    assert kv["tokenizer.ggml.merges"] == ["a b", "aĠb c", "b c"]
    # all four fill-in-the-middle ids of :2144-2147 are written (no key is dropped on an unverifiable premise about
    # one gguf-py release); the believed 0.17.1 behaviour (eot only, three warnings) is an explicit compat switch
    assert [kv[f"tokenizer.ggml.{t}_token_id"] for t in ("prefix", "suffix", "middle", "eot")] == [32007, 32008, 32009, 32010]
    assert order.index("tokenizer.ggml.eot_token_id") < order.index("tokenizer.ggml.add_space_prefix")  # :2138 before :2150
    import gptq_gguf_toolkit_amd.pack_gptq_into_gguf as P
    P.GGUF_PY_FIM_COMPAT = True
    try:
        kvc, _ = kv_of(32016)
    finally:
        P.GGUF_PY_FIM_COMPAT = False
    assert kvc["tokenizer.ggml.eot_token_id"] == 32010
    assert not any(f"tokenizer.ggml.{t}_token_id" in kvc for t in ("prefix", "suffix", "middle"))
    # a CodeLlama-Instruct checkpoint: tokenizer_config.json carries a chat template.  It is written ONCE (by the first
    # SpecialVocab); the CodeLlama tail does not add it again, so the conversion does not die on a duplicate key
    (d / "tokenizer_config.json").write_text(json.dumps({"add_prefix_space": False, "chat_template": "{{ messages }}"}))
    kvt, ordert = kv_of(32016)
    assert kvt["tokenizer.chat_template"] == "{{ messages }}" and ordert.count("tokenizer.chat_template") == 1
    assert kvt["tokenizer.ggml.prefix_token_id"] == 32007
    (d / "tokenizer_config.json").write_text(json.dumps({"add_prefix_space": False}))
    kv, order = kv_of(49152)
    assert kv["tokenizer.ggml.add_bos_token"] is False and order[-1] == "tokenizer.ggml.add_bos_token"
    assert "tokenizer.ggml.prefix_token_id" not in kv
    (d / "tokenizer_config.json").write_text(json.dumps({"add_bos_token": True}))
    with pytest.raises(ValueError, match="Duplicated key"):
        kv_of(49152)
    kv, _ = kv_of(5)
    assert "tokenizer.ggml.prefix_token_id" not in kv and kv["tokenizer.ggml.add_bos_token"] is True


# --------------------------------------------------------------------------- stacked column loops
def _run_uniform_driver(save_dir, types, stack):
    import fake_ops
    from make_golden_shim import tiny_calib, tiny_llama
    from gptq_gguf_toolkit_amd.block_schedule import BlockSchedule
    from gptq_gguf_toolkit_amd.quant_utils import GGMLQuantizationType as T
    from gptq_gguf_toolkit_amd.quantizer import Quantizer
    fake_ops.install()
    for k in fake_ops.calls:
        fake_ops.calls[k] = 0
    data = [([], {"input_ids": ids}) for ids in tiny_calib()]
    drv = Quantizer(tiny_llama(), data_loader=data, quantizable_modules=r".*layers.*((q|k|v|o|gate|up|down)_proj)$",
                    quantizer_kwargs=dict(rel_damp=0.01, block_size=128, act_order=False, quant_scale="absmax",
                                          static_groups=False, rmin=-1.0, rdelta=0.1, nstep=20, verbose=False),
                    pre_block_modules=["model.embed_tokens"], block_modules="model.layers",
                    post_block_modules=["lm_head"], quant_non_block_modules=False, device="cpu", save_dir=save_dir)
    init = BlockSchedule.__init__

    def patched(self, *a, **k):
        init(self, *a, **k)
        self.stack = stack
    BlockSchedule.__init__ = patched
    try:
        drv.quantize({k: T[v] for k, v in types.items()})
    finally:
        BlockSchedule.__init__ = init
    return dict(fake_ops.calls), drv.schedule_stats


def test_stacked_column_loops_save_the_same_bytes(tmp_path):
    """Linears that share a factorisation (q / k / v, gate / up: same input, same dead / zero-column sets) and a grid walk
    the columns together, stacked by rows (GPTQ.compute_stacked -> gq_gptq_quantize_stacked): 4 column loops per block
    instead of 7 when every Linear takes Q4_K, 6 when k_proj takes another grid -- and every data.pth holds the same
    bytes as with BlockSchedule.stack = False."""
    uniform = {k: "Q4_K" for k in ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")}
    mixed = dict(uniform, k_proj="Q6_K", up_proj="Q3_K")
    for tag, types, loops, stacked in (("u", uniform, 4, 5), ("m", mixed, 6, 2)):
        a, b = str(tmp_path / f"{tag}0"), str(tmp_path / f"{tag}1")
        calls0, _ = _run_uniform_driver(a, types, False)
        calls1, stats1 = _run_uniform_driver(b, types, True)
        assert calls0["gptq_quantize"] == 2 * 7 and calls0["gptq_quantize_stacked"] == 0
        assert calls1["gptq_quantize"] == 2 * loops and stats1["stacked"] == stacked  # (stats: the last block's)
        assert calls1["h_prepare"] == calls0["h_prepare"] == 2 * 4 and calls1["w_prepare"] == calls0["w_prepare"] == 2 * 3
        names = sorted(os.listdir(a))
        assert names == sorted(os.listdir(b)) and len(names) == 14
        for n in names:
            x = torch.load(os.path.join(a, n, "data.pth"), weights_only=True)
            y = torch.load(os.path.join(b, n, "data.pth"), weights_only=True)
            assert set(x) == set(y)
            for k in x:
                assert (x[k] == y[k]) if k == "q_type" else (x[k].dtype == y[k].dtype and torch.equal(x[k], y[k])), (n, k)


def test_stack_groups():
    """Who walks together: the leader and the followers KNOWN to share its sets, per grid; followers with sets of their own
    (own[n] True), followers whose sharing is not known (not in own), act_order handles and rows that are no multiple of
    64 walk alone; at most eight to a group; the leader's group comes first."""
    import fake_ops
    import torch.nn as nn
    from gptq_gguf_toolkit_amd.block_schedule import BlockSchedule
    from gptq_gguf_toolkit_amd.gptq import GPTQ
    from gptq_gguf_toolkit_amd.quant_utils import GGMLQuantizationType as T
    fake_ops.install()
    rows = {"a": 128, "b": 64, "c": 192, "d": 128, "e": 96, "f": 128, "g": 64}
    layers = {n: nn.Linear(256, r, bias=False) for n, r in rows.items()}
    sch = BlockSchedule(layers, lambda l, n: GPTQ(l, act_order=(n == "f"), static_groups=(n == "f")))
    h = sch.handles
    h["a"]._has_followers = True
    for n in "bcdefg":
        h[n].shared_H_with = h["a"]
    q = {n: T.Q4_K for n in rows}
    q["d"] = T.Q6_K
    own = {"b": False, "c": True, "d": False, "e": False, "f": False}   # g: unknown
    groups = sch._stack_groups(list("abcdefg"), q, own)
    assert groups == [["a", "b"], ["c"], ["d"], ["e"], ["f"], ["g"]]
    q["g"] = T.Q6_K
    own["g"] = False
    assert sch._stack_groups(list("abcdefg"), q, own) == [["a", "b"], ["c"], ["d", "g"], ["e"], ["f"]]
    sch.stack = False
    assert sch._stack_groups(list("abcdefg"), q, own) == [[n] for n in "abcdefg"]
    many = {f"x{i}": nn.Linear(256, 64, bias=False) for i in range(11)}
    sch = BlockSchedule(many, lambda l, n: GPTQ(l))
    sch.handles["x0"]._has_followers = True
    for i in range(1, 11):
        sch.handles[f"x{i}"].shared_H_with = sch.handles["x0"]
    g = sch._stack_groups(list(many), {n: T.Q4_K for n in many}, {f"x{i}": False for i in range(1, 11)})
    assert [len(x) for x in g] == [8, 3] and g[0][0] == "x0"


# --------------------------------------------------------------------------- row split: the panel-wide `continue`
def _worker_rowsplit_corner(rank, world, port, ret, mode, tiny_slice, via_schedule, device="cpu"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), GQ_ROW_SPLIT=mode)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fake_ops
    if device == "cpu":
        fake_ops.install()  # the GPU test runs the same worker on the HIP library (tests/test_gpu_round5.py)
    from gptq_gguf_toolkit_amd import dist_utils
    from gptq_gguf_toolkit_amd.block_schedule import BlockSchedule
    from gptq_gguf_toolkit_amd.gptq import GPTQ
    from gptq_gguf_toolkit_amd.quant_utils import GGMLQuantizationType as T
    torch.manual_seed(0)
    R, C = 512, 512
    lin = torch.nn.Linear(C, R, bias=False)
    W = torch.randn(R, C) * 0.02
    if tiny_slice:
        W[128:256] *= 1e-6  # rank 1's rows: ~2e-8, no group of that slice is ever `valid` (quant_utils.py:250)
    lin.weight.data = W
    X = torch.randn(1, 96, C)
    if device != "cpu":
        lin, X = lin.to(device), X.to(device)
    before = dict(dist_utils.collective_calls)
    if via_schedule:
        sched = BlockSchedule({"down": lin}, lambda l, n: GPTQ(l, rel_damp=0.01, block_size=128))
        sched.feed("down", X)
        sched.sample_done()
        res = sched.quantize({"down": T.Q4_K}, writeback=False)["down"]
        redone = sched.stats.get("row_split_redone", 0)
        owners = dict(sched.owners)
    else:
        h = GPTQ(lin, rel_damp=0.01, block_size=128)
        h.row_split = mode != "0"
        h.update(X)
        res = h.quantize(T.Q4_K)
        redone = int(h.row_split_redone)
        owners = {}
    coll = {k: v - before.get(k, 0) for k, v in dist_utils.collective_calls.items()}
    ret[rank] = (tuple(t.cpu().clone() for t in res), redone, coll, dict(fake_ops.calls), owners)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("via_schedule", [True, False])
def test_four_rank_row_split_equals_owner_mode_even_when_a_slice_has_no_valid_group(via_schedule):
    """VERDICT r04 next #2.  make_k_quants' `if not valid.any(): continue` (quant_utils.py:250-252) looks across ALL rows of the
    matrix; a row slice looks across its own.  4 gloo ranks, one Linear split 128 rows per rank, rank 1's rows ~2e-8 (no
    group of that slice is ever valid, groups of the other slices are): the slice's own verdict differs from the matrix's, the
    re-search counts (gq_gptq_quantize_slice) that ride in the block's all-gather say so, and every rank quantizes the whole
    matrix instead -- bytes identical to GQ_ROW_SPLIT=0 (owner mode = the N = 1 result).  With ordinary weights nothing is
    redone and no collective is added (BlockSchedule; the handle-level path carries the counts in its first row all-gather)."""
    world = 4
    mgr = mp.Manager()
    got = {}
    # (the handle-level path runs the corner case only: the CPU suite's budget)
    configs = (("0", True), ("all", True), ("all", False), ("0", False)) if via_schedule else (("0", True), ("all", True))
    for k, (mode, tiny) in enumerate(configs):
        ret = mgr.dict()
        mp.spawn(_worker_rowsplit_corner, args=(world, 34000 + 11 * k + os.getpid() % 2000, ret, mode, tiny, via_schedule),
                 nprocs=world, join=True)
        got[(mode, tiny)] = [ret[r] for r in range(world)]
    for tiny in (True, False) if via_schedule else (True,):
        ref = got[("0", tiny)][0][0]
        for r in range(world):
            for mode in ("0", "all"):
                res = got[(mode, tiny)][r][0]
                assert all(torch.equal(a, b) for a, b in zip(ref, res)), f"rank {r} mode {mode} tiny {tiny}"
    for r in range(world):
        res, redone, coll, calls, owners = got[("all", True)][r]
        assert redone == 1 and calls["gptq_quantize_slice"] == 1 and calls["gptq_quantize"] == 2  # the slice, then the whole matrix
        if via_schedule:
            res, redone, coll, calls, owners = got[("all", False)][r]
            assert redone == 0 and calls["gptq_quantize_slice"] == 1 and calls["gptq_quantize"] == 1
            assert owners == {"down": "rows/4"}
            assert (coll["all_gather"], coll["broadcast"], coll.get("small_all_reduce", 0)) == (1, 0, 0), coll
        else:
            # r06: the counts ride in the first row all-gather -- no 4-byte all-reduce, no host sync of its own;
            # redone: the other four tensors' slices are not gathered
            assert coll.get("small_all_reduce", 0) == 0 and coll["all_gather"] == 1, coll
    # the corner is real: without the fallback rank 1's slice result differs from the whole matrix's rows
    import fake_ops
    from oracle import oracle as O
    torch.manual_seed(0)
    torch.nn.Linear(512, 512, bias=False)
    W = torch.randn(512, 512) * 0.02
    W[128:256] *= 1e-6
    x = W[:, :256].contiguous().numpy()
    whole = O.scale_search(x, 12)
    part = O.scale_search(x[128:256].copy(), 12)
    assert any(not np.array_equal(np.asarray(a)[128:256], np.asarray(b)) for a, b in zip(whole, part))


# --------------------------------------------------------------------------- f1: the spec's open choices
def test_gguf_writer_makes_gguf_pys_choices_where_the_spec_is_open(tmp_path):
    """The GGUF v3 specification leaves open (or only implies) what this writer must do the way gguf-py 0.17.1's GGUFWriter does
    for a file to be byte-identical to the reference's (pack_gptq_into_gguf.py:155,348,437-475).  gguf-py is not installable
    here, so each rule is asserted against its PUBLISHED behaviour, by name (gguf_writer.py module docstring R1-R8; DESIGN.md
    section 0d): what stays unpinned is then a list of named keys, not the container."""
    import struct
    from gptq_gguf_toolkit_amd.gguf_writer import GGMLType, GGUFValueType as VT, GGUFWriter, parse_gguf, value_type_of
    p = str(tmp_path / "r.gguf")
    w = GGUFWriter(p, "llama")
    # R1: insertion order, general.architecture first
    w.add_uint32("llama.block_count", 2)
    w.add_string("general.name", "x")
    w.add_bool("tokenizer.ggml.add_bos_token", True)
    # R3: element type = Python type of the first element; homogeneous; empty arrays are not written
    w.add_array("tokenizer.ggml.tokens", ["a", "bc"])
    w.add_array("tokenizer.ggml.token_type", [1, 3])
    w.add_array("tokenizer.ggml.scores", [0.0, -1000.0])
    w.add_array("tokenizer.ggml.merges", [])
    with pytest.raises(ValueError, match="same type"):
        w.add_array("bad.mixed", [1, 2.0])
    with pytest.raises(ValueError, match="gguf-py would write"):
        w.add_array("bad.sub", [1, 2], VT.UINT32)  # python ints are INT32 in gguf-py unless the caller packs them itself
    assert (value_type_of("s"), value_type_of(1), value_type_of(1.0), value_type_of(True)) == (VT.STRING, VT.INT32, VT.FLOAT32, VT.BOOL)
    # R2: duplicates raise
    with pytest.raises(ValueError, match="Duplicated key name"):
        w.add_uint32("llama.block_count", 3)
    rng = np.random.default_rng(0)
    t0 = rng.standard_normal((3, 5)).astype(np.float32)             # 60 bytes: needs 4 pad bytes
    t1 = rng.integers(0, 256, (2, 144), dtype=np.uint8)             # one Q4_K block per row: logical [2, 256]
    t2 = rng.standard_normal(7).astype(np.float16)                  # 14 bytes: the LAST tensor is padded too
    w.add_tensor("z_last_in_name_order.weight", t0)
    w.add_tensor("a_first_in_name_order.weight", t1, raw_dtype=GGMLType.Q4_K)
    w.add_tensor("m.weight", t2)
    with pytest.raises(ValueError, match="Duplicated tensor name"):
        w.add_tensor("m.weight", t2)
    w.write()
    kv, tensors, buf = parse_gguf(p)
    assert list(kv) == ["general.architecture", "llama.block_count", "general.name", "tokenizer.ggml.add_bos_token",
                        "tokenizer.ggml.tokens", "tokenizer.ggml.token_type", "tokenizer.ggml.scores"]      # R1, R3 (no merges)
    assert "general.alignment" not in kv                                                                       # R7
    assert kv["tokenizer.ggml.tokens"][1] == [VT.ARRAY, VT.STRING] and kv["tokenizer.ggml.token_type"][1] == [VT.ARRAY, VT.INT32]
    assert kv["tokenizer.ggml.scores"][1] == [VT.ARRAY, VT.FLOAT32]
    # R4: u64 length + UTF-8, no terminator; BOOL one byte
    key = b"general.architecture"
    assert buf[24:24 + 8 + len(key) + 4 + 8 + 5] == struct.pack("<Q", len(key)) + key + struct.pack("<IQ", VT.STRING, 5) + b"llama"
    i = buf.index(b"tokenizer.ggml.add_bos_token") + len(b"tokenizer.ggml.add_bos_token")
    assert buf[i:i + 5] == struct.pack("<I", VT.BOOL) + b"\x01" and buf[i + 5:i + 13] == struct.pack("<Q", len("tokenizer.ggml.tokens"))
    # R5: infos and data in add_tensor order (NOT sorted by name), offsets relative to the data section, ggml_pad steps; R8
    assert [t[0] for t in tensors] == ["z_last_in_name_order.weight", "a_first_in_name_order.weight", "m.weight"]
    assert [t[1] for t in tensors] == [(3, 5), (2, 256), (7,)] and [t[2] for t in tensors] == [GGMLType.F32, GGMLType.Q4_K, GGMLType.F16]
    data0 = tensors[0][3]
    assert data0 % 32 == 0 and [t[3] - data0 for t in tensors] == [0, 64, 64 + 288]
    assert buf[tensors[0][3]:tensors[0][3] + 60] == t0.tobytes() and buf[tensors[1][3]:tensors[1][3] + 288] == t1.tobytes()
    # R6: every pad byte is zero -- before the data section, between tensors, after the last one; length a multiple of 32
    assert buf[data0 + 60:data0 + 64] == b"\x00" * 4
    end2 = tensors[2][3] + 14
    assert len(buf) == end2 + 18 and buf[end2:] == b"\x00" * 18 and len(buf) % 32 == 0
    info_end = buf.index(b"m.weight") + len(b"m.weight") + 4 + 8 + 4 + 8   # n_dims, one dim, type, offset
    assert set(buf[info_end:data0]) <= {0} and data0 - info_end < 32
    # dims innermost first (R5): the Q4_K tensor's info holds ne = [256, 2]
    j = buf.index(b"a_first_in_name_order.weight") + len(b"a_first_in_name_order.weight")
    assert struct.unpack_from("<IQQIQ", buf, j) == (2, 256, 2, GGMLType.Q4_K, 64)


def test_gguf_writer_lazy_tensors_are_written_in_order_and_errors_surface(tmp_path):
    """r06: GGUFWriter.add_tensor_lazy (gguf-py's add_tensor_info / write_tensor_data split): payloads are produced while the file
    is written -- several producers at a time, handed over in tensor order -- and the file equals the eager writer's byte for
    byte; a producer that returns the wrong number of bytes, or raises, fails write() instead of leaving a silent bad file."""
    import time
    from gptq_gguf_toolkit_amd.gguf_writer import GGMLType, GGUFWriter
    rng = np.random.default_rng(0)
    blobs = [rng.integers(0, 256, (4 + i, 144), dtype=np.uint8) for i in range(9)]
    plain = rng.standard_normal((3, 5)).astype(np.float32)

    def build(path, lazy):
        w = GGUFWriter(str(path), "llama")
        w.add_uint32("x.count", 9)
        w.add_tensor("first", plain)
        for i, b in enumerate(blobs):
            if lazy:
                # later tensors finish EARLIER: the hand-over order must still be the tensor order
                w.add_tensor_lazy(f"t{i}", (b.shape[0], 256), GGMLType.Q4_K, (lambda b=b, i=i: (time.sleep(0.02 * (9 - i)), b)[1]))
            else:
                w.add_tensor(f"t{i}", b, raw_dtype=GGMLType.Q4_K)
        w.add_tensor("last", plain.astype(np.float16))
        return w
    build(tmp_path / "eager.gguf", False).write()
    tm = {}
    build(tmp_path / "lazy.gguf", True).write(tm)
    assert (tmp_path / "eager.gguf").read_bytes() == (tmp_path / "lazy.gguf").read_bytes() and tm["write"] >= 0 and tm["wait"] > 0
    w = GGUFWriter(str(tmp_path / "bad.gguf"), "llama")
    w.add_tensor_lazy("t", (4, 256), GGMLType.Q4_K, lambda: blobs[1])  # 5 rows announced as 4
    with pytest.raises(ValueError, match="announced"):
        w.write()
    w = GGUFWriter(str(tmp_path / "bad2.gguf"), "llama")
    w.add_tensor_lazy("t", (4, 256), GGMLType.Q4_K, lambda: (_ for _ in ()).throw(RuntimeError("upload failed")))
    with pytest.raises(RuntimeError, match="upload failed"):
        w.write()
    with pytest.raises(ValueError, match="Duplicated tensor name"):
        w.add_tensor_lazy("t", (4, 256), GGMLType.Q4_K, lambda: blobs[0])
