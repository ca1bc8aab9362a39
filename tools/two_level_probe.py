"""bench.measure_two_level alone (GPU box): python tools/two_level_probe.py [n_train] [n_index]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import ivfadc_jl_amd as pkg
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
ni = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
print(json.dumps(bench.measure_two_level(torch, pkg, 10, torch.device("cuda", 0), 0, nt, ni), indent=1))
