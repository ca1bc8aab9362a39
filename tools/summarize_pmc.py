#!/usr/bin/env python3
"""Per-launch HBM traffic of the scan kernel from separate rocprofv3 --pmc passes.

usage: tools/summarize_pmc.py <prof_dir> <name> <round_tag>
  <name> = one profiled run of tools/profile_all.sh (a configuration in ONE mode, e.g. sift1m_hinted, sift1b_w1)
  reads  <prof_dir>/<name>_fetch/**/**counter_collection.csv  (--pmc FETCH_SIZE)
         <prof_dir>/<name>_write/**/**counter_collection.csv  (--pmc WRITE_SIZE)
         <prof_dir>/<name>_trace.log                            (the bench line of the kernel-trace pass: roofline.traffic_key)
  writes profiles/<round_tag>_<name>_pmc.csv  (every ivf:: kernel: mean counter per dispatch)
         profiles/traffic.json[traffic_key]     (scan kernel: hbm_bytes_per_launch, read by bench.py)

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so reads are doubled;
WRITE_SIZE is exact for 16-B stores.  Other access widths are uncalibrated (the scan kernel's reads are
16-B-per-lane streaming loads; table/codebook reads are L2 hits and do not reach the fabric counters).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter and "ivf::" in r["Kernel_Name"]:
                    acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    prof, cfg, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    fetch = collect(os.path.join(prof, cfg + "_fetch"), "FETCH_SIZE")
    write = collect(os.path.join(prof, cfg + "_write"), "WRITE_SIZE")
    names = sorted(set(fetch) | set(write))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_csv = os.path.join(root, "profiles", "%s_%s_pmc.csv" % (tag, cfg))
    scan = None
    with open(out_csv, "w", newline="") as fh:
        fh.write("# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), bench.py --config %s; KiB per dispatch, mean\n" % cfg)
        fh.write("# hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half of wide coalesced reads)\n")
        w = csv.writer(fh)
        w.writerow(["Kernel", "dispatches", "FETCH_SIZE_KiB_mean", "WRITE_SIZE_KiB_mean", "hbm_bytes_per_launch_corrected"])
        for n in names:
            # the first dispatches of a bench run include warm-up of a cold cache: use the steady-state tail
            fv = fetch.get(n, [])
            wv = write.get(n, [])
            fm = sum(fv[len(fv) // 2:]) / max(1, len(fv[len(fv) // 2:])) if fv else 0.0
            wm = sum(wv[len(wv) // 2:]) / max(1, len(wv[len(wv) // 2:])) if wv else 0.0
            hbm = (2.0 * fm + wm) * 1024.0
            w.writerow([n, len(fv), "%.1f" % fm, "%.1f" % wm, "%.0f" % hbm])
            print("%-70s n=%-4d fetch=%12.1f KiB write=%10.1f KiB -> %.3f GB" % (n[:70], len(fv), fm, wm, hbm / 1e9))
            # scan_kernel, qscan_kernel, nf_scan_kernel, qscan_coarse_kernel: the one the run SETTLED on (most dispatches: a run whose plan
            # follows the pruned fraction launches a few two-probe kernels before it switches), the heavier one on a tie
            if ("scan_kernel" in n or "qscan_coarse_kernel" in n) and (scan is None or (len(fv), hbm) > (scan[4], scan[1])):
                scan = (n, hbm, fm, wm, len(fv))
    # SQ / LDS counter passes (optional)
    sq = {}
    for sub in ("_sq1", "_sq2"):
        for f in glob.glob(os.path.join(prof, cfg + sub, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if "ivf::" in r["Kernel_Name"]:
                        sq.setdefault(r["Kernel_Name"], defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if sq:
        with open(os.path.join(root, "profiles", "%s_%s_sq.csv" % (tag, cfg)), "w", newline="") as fh:
            fh.write("# rocprofv3 --pmc SQ_* (two passes), bench.py --config %s; mean per dispatch over the steady-state half\n" % cfg)
            w = csv.writer(fh)
            w.writerow(["Kernel", "Counter", "mean_per_dispatch", "dispatches"])
            for k in sorted(sq):
                for c in sorted(sq[k]):
                    v = sq[k][c]
                    v2 = v[len(v) // 2:]
                    w.writerow([k, c, "%.0f" % (sum(v2) / len(v2)), len(v)])
    if scan:
        # the workload / plan / kernel the passes were taken on: the bench line of the kernel-trace pass carries it (roofline.traffic_key);
        # bench.py replays the traffic only next to a run with the same key
        key = None
        for lg in glob.glob(os.path.join(prof, cfg + "_trace.log")):
            for ln in open(lg, errors="replace"):
                if ln.startswith("{") and "traffic_key" in ln:
                    try:
                        key = json.loads(ln)["roofline"]["traffic_key"]
                    except Exception:
                        pass
        tpath = os.path.join(root, "profiles", "traffic.json")
        try:
            allk = json.load(open(tpath))
        except Exception:
            allk = {}
        if key is None:
            print("no bench line with a traffic_key in %s_trace.log: traffic.json not updated" % cfg)
        else:
            allk[key] = {"kernel": scan[0], "hbm_bytes_per_launch": int(scan[1]), "FETCH_SIZE_KiB": scan[2], "WRITE_SIZE_KiB": scan[3],
                         "source": os.path.basename(out_csv), "correction": "(2*FETCH_SIZE + WRITE_SIZE)*1024"}
            # the same kernel's rocprofv3 --kernel-trace --stats average in the trace pass of this run (bench.py prints the fraction it
            # gives beside the one from its own HIP events)
            for f in glob.glob(os.path.join(prof, cfg + "_trace", "**", "*kernel_stats.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r["Name"] == scan[0]:
                            allk[key].update({"trace_avg_ms": float(r["AverageNs"]) / 1e6, "trace_calls": int(r["Calls"]),
                                              "trace_min_ms": float(r["MinNs"]) / 1e6, "trace_max_ms": float(r["MaxNs"]) / 1e6,
                                              "trace_source": "%s_%s_kernel_stats.csv" % (tag, cfg)})
            with open(tpath, "w") as fh:
                json.dump(allk, fh, indent=1, sort_keys=True)
            print("traffic.json[%s] = %.3f GB" % (key, scan[1] / 1e9))


if __name__ == "__main__":
    main()
