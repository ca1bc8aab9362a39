"""GPU box: does hipMemcpy from pageable (heap) memory pin the caller's pages?  torch H2D copies of several sizes under AMD_LOG_LEVEL=4; the
runtime's log lines that mention pinning / locking are counted per size."""
import os, sys, subprocess
if len(sys.argv) > 1:
    import numpy as np, torch
    n = int(sys.argv[1])
    a = np.ones(n // 4, np.float32)
    torch.cuda.init()
    d = torch.empty(n // 4, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    sys.stderr.write("==== COPY START %d\n" % n); sys.stderr.flush()
    d.copy_(torch.from_numpy(a))
    torch.cuda.synchronize()
    sys.stderr.write("==== COPY END\n"); sys.stderr.flush()
    sys.exit(0)
for n in (64 << 10, 256 << 10, 1 << 20, 4 << 20, 64 << 20, 256 << 20):
    env = dict(os.environ, AMD_LOG_LEVEL="4")
    p = subprocess.run([sys.executable, __file__, str(n)], env=env, capture_output=True, text=True)
    lines = p.stderr.splitlines()
    try:
        i0 = next(i for i, l in enumerate(lines) if "COPY START" in l); i1 = next(i for i, l in enumerate(lines) if "COPY END" in l)
    except StopIteration:
        print(n, "no markers", p.stderr[-300:]); continue
    seg = lines[i0:i1]
    hits = [l for l in seg if any(k in l.lower() for k in ("pin", "lock", "staging", "unpinned"))]
    print("%9d bytes: %d log lines in the copy, %d mention pin/lock/staging" % (n, len(seg), len(hits)))
    for l in hits[:6]:
        print("     ", l[:200])
