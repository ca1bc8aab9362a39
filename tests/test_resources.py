"""Register / LDS budgets of the occupancy-critical kernels, read from the code object inside the built library.

The m = 8 query-major kernel runs four workgroups per CU only while it needs <= 128 VGPRs (512 / 4 waves per SIMD); a
batch of 1024 queries is then resident at once.  One VGPR more silently costs a quarter of the throughput (measured:
15.8 M -> 8-11 M q/s), so the budget is part of the test suite, not of somebody's memory.  No GPU needed."""
import os
import re
import subprocess
import tempfile

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"

# kernel (mangled-name fragment) -> most VGPRs it may use
BUDGETS = {
    "qscan_kernelILi8ELi16ELi2ELb1ELb0E": 128,    # SIFT-like: 4 workgroups / CU
    "qscan_kernelILi8ELi16ELi1ELb1ELb0E": 128,
    "qscan_kernelILi16ELi6ELi2ELb1ELb0E": 128,    # Deep1B-like
    "qscan_kernelILi16ELi8ELi2ELb1ELb0E": 128,
    "qscan_kernelILi48ELi16ELi1ELb1ELb0E": 168,   # HD-like, exact tables: LDS allows three workgroups / CU, 512 / 3 = 170
    "qscan_kernelILi48ELi16ELi4ELb1ELb1E": 256,   # HD-like, lower-bound tables from the matrix cores: 70 KB of LDS, two workgroups / CU
    "11scan_kernelILi8ELi16ELi4ELb1ELb1E": 168,   # SIFT1B-like list-major, striped tables: three waves / SIMD
    "11scan_kernelILi8ELi16ELi4ELb1ELb0E": 168,   # ... and the reference-order form (table mode 1)
    "coarse_bf16_kernel": 168,                    # bf16 coarse filter: three workgroups / CU
    "qscan_coarse_kernelILi8ELi16ELi2E": 128,     # the SIFT-like scan with the next batch's coarse tiles behind it: still 4 workgroups / CU
    "qscan_coarse_kernelILi16ELi6ELi2E": 128,
    "nf_scan_kernelILi4E": 256,                   # narrow-field list-major kernel: 64 KB of tables, two workgroups / CU (no VGPR spills: the
                                                  # 32 lane-constant codeword addresses must not be hoisted out of the item loop)
}


def _kernel_resources(so_path):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    # amdhsa.kernels is a YAML list: an entry starts with "  - .agpr_count:", its own keys are indented by four spaces
    # (the keys of its .args entries sit deeper and are skipped)
    res, cur = {}, None
    for line in notes.splitlines():
        if re.match(r"^  - \.\w+:", line):
            cur = {}
            line = "    " + line[4:]
        if cur is None:
            continue
        m = re.match(r"^    \.(name|vgpr_count|vgpr_spill_count|group_segment_fixed_size):\s+(\S+)", line)
        if not m:
            continue
        if m.group(1) == "name":
            res[m.group(2)] = cur
        else:
            cur[m.group(1)] = int(m.group(2))
    return res


def _kernel_blocks(so_path, frag):
    """Basic blocks (lists of instruction strings) of the kernel whose name contains `frag`, from the disassembly."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        syms = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", co], text=True)
        name = [ln.split()[-1] for ln in syms.splitlines() if frag in ln and " FUNC " in ln][0]
        dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--disassemble-symbols=" + name, co], text=True)
    # a block ends at every branch; llvm-objdump prints no labels inside a function, which is enough here: the gathers of a step are
    # straight-line code between two branches
    blocks, cur = [], []
    for ln in dis.splitlines():
        ins = ln.split("//")[0].strip()
        if not ins or ins.endswith(":") or ins.startswith("Disassembly") or "file format" in ins:
            continue
        cur.append(ins)
        if ins.startswith("s_cbranch") or ins.startswith("s_branch") or ins.startswith("s_endpgm") or ins.startswith("s_setpc") or ins.startswith("s_swappc"):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    return blocks


def test_register_budgets(native):
    import ivfadc_jl_amd as pkg
    so = os.path.join(os.path.dirname(pkg._native.__file__), "csrc", "libivfadc_hip.so")
    if not (os.path.exists(os.path.join(LLVM, "llvm-readelf")) and os.path.exists(so)):
        pytest.skip("LLVM tools or the built library are not available")
    res = _kernel_resources(so)
    assert res, "no kernel metadata found"
    for frag, budget in BUDGETS.items():
        hits = {k: v for k, v in res.items() if frag in k and not k.endswith(".kd")}
        assert hits, "kernel %s not found in the code object" % frag
        for name, r in hits.items():
            assert r.get("vgpr_count", 0) <= budget, "%s uses %d VGPRs (budget %d)" % (name, r.get("vgpr_count", 0), budget)
            assert r.get("vgpr_spill_count", 0) == 0, "%s spills %d VGPRs" % (name, r["vgpr_spill_count"])
    # the eight-wave list-major kernel: sixteen waves per CU need <= 128 VGPRs.  Its cold paths (table build, candidate passes, merges) do
    # spill a few registers; what must stay clean is the scan loop itself -- the block with the sixteen-per-half table gathers
    w8 = {k: v for k, v in res.items() if "wg8_scan_kernel" in k and not k.endswith(".kd")}
    assert w8, "wg8_scan_kernel not found in the code object"
    for name, r in w8.items():
        assert r.get("vgpr_count", 0) <= 128, "%s uses %d VGPRs (budget 128)" % (name, r.get("vgpr_count", 0))
    w9 = {k: v for k, v in res.items() if "wg8q8_scan_kernel" in k and not k.endswith(".kd")}
    assert w9, "wg8q8_scan_kernel not found in the code object"
    for name, r in w9.items():
        assert r.get("vgpr_count", 0) <= 128, "%s uses %d VGPRs (budget 128)" % (name, r.get("vgpr_count", 0))
    hot9 = [b for b in _kernel_blocks(so, "wg8q8_scan_kernel") if sum("ds_read_b128" in x for x in b) >= 32]
    assert len(hot9) == 1 and not any("scratch_" in x for x in hot9[0]), "the scan loop of wg8q8_scan_kernel: one block, no scratch memory"
    blocks = _kernel_blocks(so, "wg8_scan_kernel")
    hot = [b for b in blocks if sum("ds_read_b64" in x for x in b) >= 32]
    assert len(hot) == 1, "expected ONE block with the step's 32 table gathers, found %d" % len(hot)
    assert not any("scratch_" in x for x in hot[0]), "the scan loop of wg8_scan_kernel touches scratch memory"
    assert sum(1 for x in hot[0] if x.startswith("v_")) <= 104, "the scan loop grew: %d vector instructions" % sum(1 for x in hot[0] if x.startswith("v_"))
    stream = [x for x in hot[0] if x.startswith("buffer_load_dwordx4") and "sc1" not in x]
    assert len(stream) == 2, "the step requests its two 16-byte code sets once each: %r" % stream
    loaded = set()
    for x in stream:
        a, b = re.search(r"v\[(\d+):(\d+)\]", x).groups()
        loaded.update("v%d" % r for r in range(int(a), int(b) + 1))
    copies = [x for x in hot[0] if x.startswith("v_mov") and x.replace(",", " ").split()[-1] in loaded]
    assert not copies, "a code set is copied after its load (the request has become synchronous): %r" % copies
    # the scan kernels address their tables by absolute LDS offsets: no static LDS allowed in them
    for name, r in res.items():
        if "scan_kernel" in name and "bucket" not in name:
            assert r.get("group_segment_fixed_size", 0) == 0, "%s carries static LDS" % name
