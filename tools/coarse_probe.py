"""Coarse stage of the Deep1B shape (split-bf16 matrix-core filter, listed mode): time of the stage with the record epilogue as built, with no
epilogue at all and with a threshold-filter stand-in (debug build: IVFADC_COARSE_DBG = 1 / 2, wrong results by design -- the stale records of the
correct run before keep the stages behind it on their usual path).
usage (GPU box): python tools/coarse_probe.py [nq] [w]        env: COARSE_DBG=1 -> libivfadc_hip_dbg.so and the knock-outs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import ivfadc_jl_amd as pkg

dbg = bool(os.environ.get("COARSE_DBG"))
if dbg:
    import ivfadc_jl_amd._native as nat
    nat.SO_PATH = nat.SO_PATH.replace("libivfadc_hip.so", "libivfadc_hip_dbg.so")
cfg = dict(bench.CONFIGS["deep1b"])
nq = int(sys.argv[1]) if len(sys.argv) > 1 else cfg["nq"]
w = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["w"]
idx, _ = bench.build_synth(pkg, cfg, 0)
q = np.random.default_rng(11).standard_normal((nq, cfg["d"]), dtype=np.float32)
for flag in ([int(x) for x in os.environ.get("COARSE_FLAGS", "0,1,2,3,5,6,7,8,0").split(",")] if dbg else [0]):
    os.environ["IVFADC_COARSE_DBG"] = str(flag)
    idx.search_raw(q, 10, w)
    idx.set_profiling(True)
    idx.reset_stats()
    for _ in range(5):
        idx.search_raw(q, 10, w)
    st = idx.get_stats()
    idx.set_profiling(False)
    n = st["scan_launches"]
    print("coarse_dbg=%d nq=%d w=%d coarse_ms=%.4f scan_ms=%.4f fallbacks=%d" % (flag, nq, w, st["coarse_ms"] / n, st["scan_ms"] / n,
                                                                              st.get("coarse_fallbacks", -1)), flush=True)
