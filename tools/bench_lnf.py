#!/usr/bin/env python
"""Round 5: what do the epilogue extensions cost a launch?  One process per variant (the options are read from the environment by ug_bench_gemm), interleaved rounds.
usage: bench_lnf.py   -> table of us per launch: base (no bias), want_ext (tile choice only), lnf (LayerNorm-fold epilogue), rowpart (row partial sums)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from unigeo_amd._lib import Engine
    eng = Engine(0, workspace_bytes=24 << 30, persist_bytes=64 << 20)
    for (M, N, K) in [(19200, 1920, 640), (4800, 3840, 1280), (19200, 5120, 640), (4800, 10240, 1280), (19200, 640, 640), (19200, 640, 2560), (4800, 1280, 1280), (4800, 1280, 5120)]:
        best = 1e9
        for _ in range(3):
            ms, tf, c, s = eng.bench_gemm(M=M, N=N, K=K, cfg=int(os.environ.get("CFG", "-1")), split=0, iters=10)
            best = min(best, ms)
        print(f"{M}x{N}x{K} cfg{c}: {best * 1e3:.1f}")
    sys.exit(0)
VARS = {"base": {"UG_BENCH_NOBIAS": "1"}, "bias": {}, "want_ext": {"UG_BENCH_NOBIAS": "1", "UG_BENCH_WANTEXT": "1"}, "lnf": {"UG_BENCH_LNF": "1"},
        "rowpart": {"UG_BENCH_ROWPART": "1"}, "rowpart+R1": {"UG_BENCH_ROWPART": "1", "UG_BENCH_R1": "1"}, "bias+R1": {"UG_BENCH_R1": "1"}}
for geglu in (0, 1):
    print("GEGLU epilogue" if geglu else "plain epilogue")
    for name, env in VARS.items():
        if geglu and "rowpart" in name:
            continue
        e = dict(os.environ); e.update(env)
        if geglu:
            e["UG_BENCH_GEGLU"] = "1"
        out = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True).stdout.strip().split("\n")
        print(f"  {name:12s} " + "  ".join(out))
