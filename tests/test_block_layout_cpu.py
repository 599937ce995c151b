"""pack -> (independent llama.cpp-spec decoder) round trip on the oracle packers, and
C-ABI load/export checks.  No GPU needed."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden
from ggml_spec import unpack

TYPES = {"Q2_K": 10, "Q3_K": 11, "Q4_K": 12, "Q5_K": 13, "Q6_K": 14}


@pytest.mark.parametrize("name", list(TYPES))
def test_pack_roundtrip_against_ggml_layout(oracle, name):
    g = load_golden("g8_g9_rtn_dequant")
    t = TYPES[name]
    q, d, s, dmin, m = (g[f"{name}_{k}"] for k in ("q", "d", "s", "dmin", "m"))
    packed = oracle.pack(t, q, d, s, dmin, m)
    codes, d2, sc2, dmin2, mn2 = unpack(t, packed)
    assert np.array_equal(codes, q.astype(np.int32))
    assert np.array_equal(d2, d) and np.array_equal(sc2, s.astype(np.int32))
    if name in ("Q2_K", "Q4_K", "Q5_K"):
        assert np.array_equal(dmin2, dmin) and np.array_equal(mn2, m.astype(np.int32))
    # the reference's own packed bytes decode to the same thing
    codes_r, *_ = unpack(t, g[f"{name}_packed"])
    assert np.array_equal(codes_r, codes)


def test_cabi_exports_every_declared_symbol():
    """include/gptq_gguf.h <-> libgptqgguf_hip.so (no compute calls: there is no GPU here)."""
    hdr = open(os.path.join(ROOT, "include", "gptq_gguf.h")).read()
    declared = set(re.findall(r"\b(gq_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"gq_last_error"} - {"gq_last_error"}
    assert len(declared) >= 12
    from gptq_gguf_toolkit_amd import _cabi
    if not os.path.exists(_cabi.SO_PATH):
        _cabi.build()
    L = ctypes.CDLL(_cabi.SO_PATH)
    for sym in sorted(declared):
        assert hasattr(L, sym), f"{sym} declared in gptq_gguf.h but not exported"
    assert set(_cabi.EXPORTS) == declared
    lib = _cabi.lib()
    assert lib.gq_abi_version() == _cabi.ABI_VERSION == 6
    for t, ts in ((10, 84), (11, 110), (12, 144), (13, 176), (14, 210)):
        assert _cabi.type_info(t)["type_size"] == ts
    with pytest.raises(_cabi.GQError):
        _cabi.type_info(3)
    # host-side argument validation does not need a device
    assert lib.gq_workspace_bytes(_cabi.WS_GPTQ_QUANTIZE, 4096, 4096, 0, 128) >= 4096 * 128 * 4
    rc = lib.gq_pack(12, None, None, None, None, None, 16, 300, None, None)
    assert rc == -2 and b"256" in lib.gq_last_error()


def test_no_cpu_fallback():
    """Product ops refuse CPU tensors instead of silently computing elsewhere."""
    import torch
    from gptq_gguf_toolkit_amd import ops, GQError
    with pytest.raises(GQError):
        ops.pack(12, torch.zeros(4, 256, dtype=torch.uint8), torch.zeros(4, 1, dtype=torch.float16),
                 torch.zeros(4, 8, dtype=torch.uint8), torch.zeros(4, 1, dtype=torch.float16),
                 torch.zeros(4, 8, dtype=torch.uint8))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "gptq-gguf-toolkit_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("# oracle", "") or f == "README.md", f"{f} mentions the oracle"


def test_integration_doc_search_struct_matches_the_header(tmp_path):
    """INTEGRATION.md's binding snippet declares `_Search`; a maintainer who copies it passes that struct to
    gq_gptq_quantize, which reads all of gq_search_t.  The snippet's class is executed as written and compared, size and
    field offsets, with the header's struct as gcc lays it out (r02: the doc still showed the 3-field ABI-1 struct)."""
    import subprocess
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class _Search\(ctypes\.Structure\):.*?\n((?:    .*\n)+)", doc)
    assert m, "INTEGRATION.md no longer shows the _Search binding"
    ns = {"ctypes": ctypes}
    exec("class _Search(ctypes.Structure):\n" + m.group(1), ns)
    S = ns["_Search"]
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gptq_gguf.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(gq_search_t), offsetof(gq_search_t, rmin),'
                   ' offsetof(gq_search_t, rdelta), offsetof(gq_search_t, nstep), offsetof(gq_search_t, quant_scale),'
                   ' offsetof(gq_search_t, grid), offsetof(gq_search_t, maxshrink)); return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    size, *offs = map(int, subprocess.check_output([str(exe)]).split())
    assert ctypes.sizeof(S) == size
    assert [getattr(S, n).offset for n, _ in S._fields_] == offs
    from gptq_gguf_toolkit_amd import _cabi
    assert ctypes.sizeof(_cabi.Search) == size
    assert [n for n, _ in S._fields_] == [n for n, _ in _cabi.Search._fields_]
    # the snippet asserted ABI version 5 against a header at 6 -- it failed on its second line as printed.  The
    # version it asserts must be the header's and the binding's (the snippet runs verbatim in tests/test_gpu_round6.py)
    hdr = open(os.path.join(ROOT, "include", "gptq_gguf.h")).read()
    v_hdr = int(re.search(r"#define GQ_ABI_VERSION (\d+)", hdr).group(1))
    v_doc = int(re.search(r"assert _lib\.gq_abi_version\(\) == (\d+)", doc).group(1))
    assert v_doc == v_hdr == _cabi.ABI_VERSION
    snippet = re.search(r"```python\n(.*?)```", doc[doc.index("## Level 2"):], re.S).group(1)
    compile(snippet, "INTEGRATION.md", "exec")  # at least: it is Python


def test_environment_variables_are_few_and_documented():
    """the product reads at most 15 environment variables and include/gptq_gguf.h lists every one of
    them; the library's tuning / test switches live in ONE option table (csrc/gq_common.hpp) reachable through
    gq_option_set / GQ_OPTIONS, every option named in the header too; no timing-probe #ifdef is left in the shipped kernels."""
    import glob
    hdr = open(os.path.join(ROOT, "include", "gptq_gguf.h")).read()
    pkg = os.path.join(ROOT, "gptq-gguf-toolkit_amd")
    names = set()
    for f in glob.glob(os.path.join(pkg, "*.py")):
        src = open(f).read()
        names |= set(re.findall(r"os\.environ(?:\.get|\.setdefault)?\(\s*[\"'](GQ_[A-Z0-9_]+)", src))
        names |= set(re.findall(r"os\.environ\[\s*[\"'](GQ_[A-Z0-9_]+)", src))
    for f in glob.glob(os.path.join(pkg, "csrc", "*.h*")):
        names |= set(re.findall(r"getenv\(\s*\"(GQ_[A-Z0-9_]+)\"", open(f).read()))
    assert "GQ_OPTIONS" in names and len(names) <= 15, sorted(names)
    for n in sorted(names):
        assert n in hdr, f"{n} is read by the product but not documented in include/gptq_gguf.h"
    from gptq_gguf_toolkit_amd import _cabi
    opts = _cabi.option_names()
    assert len(opts) == len(set(opts)) >= 20
    for o in opts:
        assert re.search(rf"\b{o}\b", hdr), f"option {o} is not documented in include/gptq_gguf.h"
        assert _cabi.option_get(o) == _cabi.option_default(o)
    with _cabi.options(chol_fp32=1, la=4):
        assert (_cabi.option_get("chol_fp32"), _cabi.option_get("la")) == (1, 4)
    assert (_cabi.option_get("chol_fp32"), _cabi.option_get("la")) == (0, 8)
    with pytest.raises(_cabi.GQError):
        _cabi.option_set("no_such_option", 1)
    for f in glob.glob(os.path.join(pkg, "csrc", "*.h*")):
        src = open(f).read()
        assert not re.findall(r"^\s*#\s*(?:ifdef|ifndef|if)\b", src, re.M), f"{os.path.basename(f)}: preprocessor conditionals in a shipped kernel source"


def test_committed_r05_bench_lines():
    """The round's five collections (profiles/collect_r05.sh): the full dict and the compact headline of each BASELINE config that
    fits one GPU; the compact line carries the driver contract and stays under 2 KB; the traffic figure belongs to the kernel
    sources it names."""
    import glob
    import hashlib
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r05_bench_*_compact.json")))
    assert len(files) == 5
    for f in files:
        raw = open(f).read().strip()
        assert len(raw) <= 2048, (f, len(raw))
        c = json.loads(raw)
        full = json.loads(open(f.replace("_compact.json", ".json")).read())
        for k, t in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", (int, float)), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str)):
            assert isinstance(c[k], t) and c[k] == full[k], (f, k)
        assert c["vs_baseline"] is None and "workload" in c["config"] and "model" not in c["config"]
        r = c["roofline"]
        assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["trailing_far_alone_frac"] > 0.5
        assert all(not isinstance(v, (dict, list)) for v in r.values())
        assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] >= 1 and c["cpu_baseline"]["value"] > 0
        assert full["encoders"]["launches"]["gptq_segment"] <= full["encoders"]["launches"]["scale_search"]  # one launch per pair
    d = json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_llama3-8b-block-q4k_compact.json")).read())
    assert d["roofline"]["traffic"] > d["roofline"]["traffic_algorithmic"] > 0 and d["roofline"]["frac_on_model_forward_activations"] > 0.4
    assert d["whole_model_wall_s"] < d["whole_model_hf_eager_wall_s"]
    # (profiles/r05_syrk_traffic.json names r05's kernel sources; the pin on the CURRENT sources is round 6's: below)
    assert len(json.load(open(os.path.join(ROOT, "profiles", "r05_syrk_traffic.json")))["kernel_sources_sha256"]) == 16


def test_committed_r06_bench_lines():
    """Round 6's five collections (profiles/collect_r06.sh): full dict + compact headline per BASELINE config that fits one GPU; the
    compact line keeps the driver contract under 2 KB with room to spare and now carries the end-to-end figures (VERDICT r05 next
    #2: quantize -> pack_gptq_into_gguf -> .gguf); the traffic figure belongs to the kernel sources of THIS tree."""
    import glob
    import hashlib
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_bench_*_compact.json")))
    assert len(files) == 5
    for f in files:
        raw = open(f).read().strip()
        assert len(raw) <= 1980, (f, len(raw))
        c = json.loads(raw)
        full = json.loads(open(f.replace("_compact.json", ".json")).read())
        for k, t in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", (int, float)), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str)):
            assert isinstance(c[k], t) and c[k] == full[k], (f, k)
        assert c["vs_baseline"] is None and "workload" in c["config"] and "model" not in c["config"]
        r = c["roofline"]
        assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["trailing_far_alone_frac"] > 0.5
        assert all(not isinstance(v, (dict, list)) for v in r.values())
        assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] >= 1 and c["cpu_baseline"]["value"] > 0
    d = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_llama3-8b-block-q4k_compact.json")).read())
    assert d["roofline"]["traffic"] > d["roofline"]["traffic_algorithmic"] > 0 and d["roofline"]["frac_on_model_forward_activations"] > 0.4
    assert d["whole_model_wall_s"] < d["whole_model_hf_eager_wall_s"]
    # the target end to end: the packer costs less than the quantizer (VERDICT r05's bar) and the sum is what the line says
    assert 0 < d["gguf_pack_wall_s"] < d["whole_model_wall_s"]
    assert abs(d["end_to_end_gguf_wall_s"] - d["whole_model_wall_s"] - d["gguf_pack_wall_s"]) < 0.02 and "write" in d["gguf_pack_split_s"]
    full = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_llama3-8b-block-q4k.json")).read())
    e2e = full["whole_model"]["end_to_end_gguf"]
    assert 4.0 < e2e["gguf_GB"] < 5.5 and set(e2e["split_s"]) >= {"load", "h2d", "permute_pack", "d2h", "write", "wait"}
    tj = json.load(open(os.path.join(ROOT, "profiles", "r06_syrk_traffic.json")))
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(ROOT, "gptq-gguf-toolkit_amd", "csrc", "*.h*"))):
        h.update(open(fn, "rb").read())
    # (a later edit of a kernel source makes bench.py report traffic = null until collect_r06.sh is run again: say so here)
    assert tj["kernel_sources_sha256"] == h.hexdigest()[:16], "csrc changed after the traffic pass: re-run profiles/collect_r06.sh"


def test_committed_bench_lines_keep_the_driver_contract():
    """The five committed lines of round 4 (profiles/r04_bench_<workload>.json, written by bench.py on the GPU box) carry every
    key of the driver's contract with the right types, the two objects of the hot-path tier (roofline, cpu_baseline) and the
    section-8(d) figures added in r04 -- a guard against a bench.py edit that silently drops one."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r04_bench_*.json")))
    assert len(files) == 5
    for f in files:
        l = json.loads(open(f).read())
        for k, t in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", (int, float)), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str)):
            assert isinstance(l[k], t), (f, k)
        assert l["metric"].startswith("Mparams/s") and l["unit"] == "Mparams/s" and l["vs_baseline"] is None
        assert l["scaling"] in ("weak", "strong") and l["data"] == "synthetic" and "workload" in l["config"]
        assert "model" not in l["config"]
        r = l["roofline"]
        assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert "traffic" in r
        c = l["cpu_baseline"]
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == l["unit"] and c["sample"]
        assert l["encoders"]["bound"] == "hbm" and l["encoders"]["bytes_per_param"] > 5.5 and l["column_loop"]["steps_per_s"] > 0
        for leg in ("tolerance_parity", "tolerance_parity_widest"):
            assert "ints_differ" in l[leg] and "all_fp32_chain" in l[leg] and "ulp_noise_floor" in l[leg], (f, leg)
    d = json.loads(open(os.path.join(ROOT, "profiles", "r04_bench_llama3-8b-block-q4k.json")).read())
    assert abs(d["value"] - 218.1 / d["ms_per_step"] * 1e3) / d["value"] < 0.01  # 218.1 M params per step
    assert d["roofline"]["traffic"]["GB_per_launch"] > d["roofline"]["traffic"]["algorithmic_GB_per_launch"] > 0
    assert d["whole_model"]["wall_s_quantizer_region"] < d["whole_model_hf_eager"]["wall_s_quantizer_region"]


def test_bench_compact_line_and_self_launch_refusal():
    """VERDICT r04 next #1 / #6 (host side, no GPU): (a) bench.py's LAST stdout line is the compact headline: every key of the
    driver contract, `roofline` / `cpu_baseline` flat (scalars only: the driver's parser drops nested objects) with the traffic
    ratio and the trailing-update fractions, <= 2 KB; (b) `python bench.py --gpus 2` with no launcher and fewer than two
    GPUs refuses loudly instead of measuring one rank."""
    import importlib.util
    import json
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.loads(open(os.path.join(ROOT, "profiles", "r04_bench_llama3-8b-block-q4k.json")).read())
    full.update({"n_gpus": 8, "ranks_seen": 8, "collective_backend": "nccl (RCCL over xGMI)",
                 "collectives_per_step": {"all_reduce": 3.0, "reduce": 1.0, "all_gather": 1.0, "broadcast": 0.0, "small_all_reduce": 0.0},
                 "allreduce_probe": {"C": 14336, "payload_MB": 418.0, "ms": 3.21, "backend": "nccl"}})
    c = bench.compact_line(full)
    s = json.dumps(c)
    assert len(s) <= 2048, len(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in c, k
    assert all(not isinstance(v, (dict, list)) for v in c["roofline"].values())
    assert all(not isinstance(v, (dict, list)) for v in c["cpu_baseline"].values())
    r = c["roofline"]
    assert r["traffic"] > r["traffic_algorithmic"] > 0 and r["traffic_ratio"] > 1
    assert r["trailing_far_alone_frac"] > 0.7 and r["trailing_far_in_region_frac"] > 0 and r["trailing_loop_ms_as_run"] > 0
    assert c["whole_model_wall_s"] > 0 and c["cpu_baseline"]["stage_update_s"] > 0
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if not __import__("torch").cuda.is_available():
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
        assert p.returncode != 0 and "RCCL needs one GPU per rank" in p.stderr and "{" not in p.stdout


def test_gq_options_errors_fail_every_entry_point_without_abort():
    """GQ_OPTIONS with an unknown name, a non-integer value or a value outside the option's range must not silently
    become 0 / the default and must not abort() the host process: every entry point fails with GQ_E_UNSUPPORTED and a
    message (here: gq_option_get; the compute entries run the same check first); gq_option_set validates ranges too."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from gptq_gguf_toolkit_amd import _cabi\n"
            "try:\n    print('VALUE', _cabi.option_get('la'))\n"
            "except _cabi.GQError as e:\n    print('ERR', e)\n") % ROOT
    for env_val, want in (("la=abc", "not an integer"), ("far_wgs=", "not an integer"), ("la=3", "must be even"),
                          ("la=64", "outside"), ("no_such=1", "unknown option"), ("la=4,syrk_ck=128", "VALUE 4"),
                          # a checkpoint distance the launcher would silently read as "none" is a typo
                          ("syrk_ck=100", "power of two"), ("syrk_ck=8", "power of two"), ("syrk_gw=6", "power of two"),
                          ("syrk_ck=0,syrk_gw=8", "VALUE 8")):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GQ_OPTIONS=env_val), capture_output=True, text=True)
        assert p.returncode == 0 and want in p.stdout, (env_val, p.stdout, p.stderr[-500:])
    from gptq_gguf_toolkit_amd import _cabi
    with pytest.raises(_cabi.GQError, match="must be even"):
        _cabi.option_set("la", 5)
    with pytest.raises(_cabi.GQError, match="outside"):
        _cabi.option_set("chol_planes", 7)
    with pytest.raises(_cabi.GQError, match="power of two"):
        _cabi.option_set("syrk_ck", 48)
    # with a GQ_OPTIONS that did not parse, gq_option_set fails like every other entry point
    code2 = ("import sys; sys.path.insert(0, %r)\n"
             "from gptq_gguf_toolkit_amd import _cabi\n"
             "try:\n    _cabi.option_set('la', 4); print('SET')\n"
             "except _cabi.GQError as e:\n    print('ERR', e)\n") % ROOT
    p = subprocess.run([sys.executable, "-c", code2], env=dict(os.environ, GQ_OPTIONS="la=abc"), capture_output=True, text=True)
    assert p.returncode == 0 and "ERR" in p.stdout and "not an integer" in p.stdout, (p.stdout, p.stderr[-500:])
    assert _cabi.option_default("syrk_ck") == 256 and _cabi.option_default("seg_pair") == 1
