"""The per-rank step of the multi-GPU bench with a single-rank communicator (all this pool allows): search on a lane, ncclAllGather of the
packed block on the index's side stream -- one and two lanes, against the same steps without the collective.
usage (GPU box): python tools/collective_lanes_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import ivfadc_jl_amd as pkg

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = dict(bench.CONFIGS["sift1m"])
dev = torch.device("cuda:0")
K, w, nq = 10, cfg["w"], cfg["nq"]
idx, _ = bench.build_trained(pkg, cfg, dev, 0, None)
if os.environ.get("VIEW_FIRST"):      # stream creation order decides which hardware queues the lanes get
    lanes = [idx, idx.clone_view()]
    idx.comm_init(1, 0, pkg.comm_unique_id())
else:
    idx.comm_init(1, 0, pkg.comm_unique_id())
    lanes = [idx, idx.clone_view()]
q = bench.global_queries(cfg, nq, dev)
width = 2 * K + 1
NS = int(os.environ.get("SLOTS", "8"))
blocks = [torch.zeros(nq * width, dtype=torch.int32, device=dev) for _ in range(NS)]
gath = [torch.zeros(nq * width, dtype=torch.int32, device=dev) for _ in range(NS)]
torch.cuda.synchronize()


def run(name, step):
    for i in range(20):
        step(i)
    idx.comm_wait(); torch.cuda.synchronize()
    best, issue = 1e9, 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        t1 = time.perf_counter()
        idx.comm_wait(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
        issue = min(issue, (t1 - t0) / steps)
    print("%-52s %.2f us/step  %.2f M q/s   (host time to issue a step: %.2f us)" % (name, best * 1e6, nq / best / 1e6, issue * 1e6), flush=True)


def mk(nl, coll, hint):
    def f(i):
        ln = lanes[i % nl]
        if hint:
            ln.set_query_token(1)
            ln.set_next_queries(nq, q.data_ptr(), 1)
        s = i % NS
        if coll:
            idx.search_device_allgather_on(ln, nq, q.data_ptr(), K, w, blocks[s].data_ptr(), gath[s].data_ptr(), s)
        else:
            b = blocks[s].data_ptr()
            ln.search_device(nq, q.data_ptr(), K, w, b, b + nq * K * 4, b + 2 * nq * K * 4)
    return f


only = os.environ.get("PROBE_ONLY")          # e.g. "1c": one lane with the collective, nothing else (for a kernel trace)
if only:
    run("%s lane(s), %s" % (only[0], "with all-gather" if only.endswith("c") else "search only"), mk(int(only[0]), only.endswith("c"), True))
    sys.exit(0)
for nl in (1, 2):
    for coll in (False, True):
        run("%d lane(s), hinted, %s" % (nl, "search + all-gather (1 rank)" if coll else "search only"), mk(nl, coll, True))
print("gathered == block:", all(torch.equal(a, b) for a, b in zip(blocks, gath)))
