"""K3 on correlated, ill-conditioned Hessians: U of the three forms of the chain's level-3 work
against the fp64 chain, per (C, T, rank, eps).  Prints one line per case; run on the GPU box:
    python profiles/r04_chol_corr_probe.py [--cond] [--C 4096 14336]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from corr_hessian import CHAIN_MODES, correlated_x, env, equilibrated_cond, fp64_chain, u_errors  # noqa: E402
from gptq_gguf_toolkit_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--C", type=int, nargs="+", default=[4096, 14336])
ap.add_argument("--cond", action="store_true")
ap.add_argument("--damp", type=float, default=0.01)
args = ap.parse_args()

CASES = [(T, rank, eps, 6, 0.0) for T in (1, 4) for rank in (16, 64) for eps in (0.3, 0.03)]       # massive channels
CASES += [(T, rank, eps, 0, m) for T in (1, 4) for rank in (16, 64) for eps in (0.3, 0.03) for m in (0.0, 3.0)]  # none; + mean
for C in args.C:
    for Tq, rq, eps, massive, mshift in CASES:
        for _ in (0,):
            for _ in (0,):
                T, rank = Tq * C // 2, C // rq
                X = correlated_x(T, C, rank, eps, seed=C + rank + int(eps * 100), massive=massive, mean_shift=mshift)
                H = torch.zeros(C, C, device="cuda")
                ops.h_accumulate(H, X, 0.0, 2.0 / 8)
                del X
                W = torch.randn(64, C, device="cuda")
                res, flags, Hd = {}, {}, None
                for mode, kv in CHAIN_MODES.items():
                    with env(**kv):
                        Hc = H.clone()
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        U, flag = ops.h_prepare(Hc, W.clone(), args.damp)
                        torch.cuda.synchronize()
                        ms = (time.perf_counter() - t0) * 1e3
                    flags[mode] = int(flag.item())
                    if Hd is None:
                        Hd = Hc  # damped in place, the same in every mode
                        U64 = fp64_chain(Hd)
                    res[mode] = u_errors(U, U64) + (ms,)
                    del U
                cond = equilibrated_cond(Hd) if args.cond else float("nan")
                line = f"C={C} T={T} rank={rank} eps={eps} massive={massive} mean={mshift} cond_eq={cond:.2e} flags={flags}"
                for mode, (e_max, e_row, e_diag, ms) in res.items():
                    line += f" | {mode}: max {e_max:.2e} row {e_row:.2e} diag {e_diag:.2e} ({ms:.1f} ms)"
                print(line, flush=True)
                del H, Hd, U64
                torch.cuda.empty_cache()
