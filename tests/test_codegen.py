"""Code-generation guards (CPU only: hipcc cross-compiles gfx950 without a GPU).  The kernels' speed depends on properties
the compiler can silently take away: register budgets that allow the intended occupancy, no scratch in the hot loops, and
row rings that keep COUNTED s_waitcnt vmcnt(N) (a store next to the loads turns them into vmcnt(0) drains -- DESIGN.md 4.2)."""
import os
import re
import shutil
import subprocess
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "viterbidecodercpp_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")


def _sources_digest():
    """every file a kernel translation unit can see, and the compiler: the key of the compile cache below"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(f.encode() + b"\0" + open(f, "rb").read())
    h.update(subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.encode())
    return h.hexdigest()


_DIGEST = None


def _compile(src, defines, tmp_path):
    """hipcc -S of one translation unit, with its resource remarks.  A compile takes 40 - 100 s and the sources rarely change between
    two runs of the CPU suite: the assembly and the remarks are kept under csrc/build/codegen_cache/ (git-ignored), keyed by the
    command line and a digest of every source file and of the compiler's version -- any edit compiles afresh."""
    import hashlib
    global _DIGEST
    if _DIGEST is None:
        _DIGEST = _sources_digest()
    out = tmp_path / "k.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
           "-Rpass-analysis=kernel-resource-usage", "-o", str(out), src] + defines
    cache = os.path.join(CSRC, "build", "codegen_cache")
    key = hashlib.sha256((_DIGEST + "\0" + "\0".join(cmd[:-len(defines) - 3] + [src] + defines)).encode()).hexdigest()[:32]
    c_asm, c_err = os.path.join(cache, key + ".s"), os.path.join(cache, key + ".stderr")
    if os.path.exists(c_asm) and os.path.exists(c_err) and not os.environ.get("VIT_TEST_NO_CODEGEN_CACHE"):
        stderr = open(c_err).read()
        out.write_text(open(c_asm).read())
        os.utime(c_asm, None)                   # a live entry stays young: the sweep below drops what no run has used for two days
        os.utime(c_err, None)
    else:
        p = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        stderr = p.stderr
        os.makedirs(cache, exist_ok=True)
        shutil.copyfile(str(out), c_asm + ".tmp")
        os.replace(c_asm + ".tmp", c_asm)
        open(c_err, "w").write(stderr)
        for f in os.listdir(cache):             # entries of sources that no longer exist (the key holds their digest): 2 MB each
            path = os.path.join(cache, f)
            if time.time() - os.path.getmtime(path) > 2 * 86400:
                os.unlink(path)
    usage = {}
    name = None
    for line in stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and name:
            usage[name][m.group(1).split(" ")[0]] = int(m.group(2))
    asm = out.read_text()
    # What the wave launcher uses is the ALLOCATION in the kernel descriptor, not the count of registers the code touches (the
    # "VGPRs" remark, hipFuncGetAttributes().numRegs): hipcc pads the allocation of a kernel whose static LDS limits its occupancy
    # up to the count that enforces that occupancy -- round 3's K = 9 chainback: 22 used, 264 allocated.  Every co-residency
    # assertion below therefore reads .amdhsa_next_free_vgpr (unified file, granules of 8) from the -S output.
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", asm, re.S):
        d = usage.setdefault(m.group(1), {})
        for key, field in (("alloc", "next_free_vgpr"), ("accum_offset", "accum_offset"), ("lds_static", "group_segment_fixed_size")):
            f = re.search(r"\.amdhsa_" + field + r" (\d+)", m.group(2))
            if f:
                d[key] = int(f.group(1))
        d["alloc"] = -(-d["alloc"] // 8) * 8
    return asm, usage


def _kernel_body(asm, symbol_regex):
    m = re.search(r"^(" + symbol_regex + r"):", asm, re.M)
    assert m, symbol_regex
    return asm[m.end():asm.index(".Lfunc_end", m.end())]


def _inner_loops(body):
    """(start, end) line ranges of backward branches, innermost first"""
    lines = body.split("\n")
    labels = {m.group(1): n for n, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for n, l in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            loops.append((labels[m.group(1)], n))
    return lines, sorted(loops, key=lambda ab: ab[1] - ab[0])


def test_k7_kernels_register_budget_and_counted_waits(tmp_path):
    asm, usage = _compile("reg_inst.hip", ["-DVIT_REG_ID=0"], tmp_path)
    upd = [k for k in usage if "reg_update_kernel" in k]
    cb = [k for k in usage if "reg_chainback_kernel" in k]
    assert len(upd) == 2 and len(cb) == 1
    def alloc(u):   # the descriptor's allocation (granules of 8 of the SIMD's 512-entry unified file)
        return u["alloc"]

    assert usage[cb[0]]["ScratchSize"] == 0
    for k in upd:
        # two update waves (2048 tiles on 1024 SIMDs) + one chainback wave must fit a SIMD's 512 registers
        assert usage[k]["ScratchSize"] == 0, (k, usage[k])
        assert 2 * alloc(usage[k]) + alloc(usage[cb[0]]) <= 512, (k, usage[k], usage[cb[0]])
    # the other K = 7 chainback kernel (the LDS-ring body, dynamic LDS): small enough for THREE update waves beside it
    alt = [k for k in usage if "reg_chainback_alt_kernel" in k]
    assert len(alt) == 1 and usage[alt[0]]["lds_static"] == 0
    for k in upd:
        assert 3 * alloc(usage[k]) + alloc(usage[alt[0]]) <= 512, (k, usage[k], usage[alt[0]])
    # the update kernel's hot path falls through: the (rare) renormalisation bodies sit out of line behind not-taken
    # s_cbranch_vccnz, one per unrolled trellis step; a taken branch per step cost 1.4 - 1.9 % and made the speed depend on where
    # the linker put the kernel
    for sym in (r"_ZN3vit17reg_update_kernelINS_7RegSpecILi7ELi2ELj109ELj79ELj0ELj0ELi2ELj0ELj0EEELi0EEEvNS_13RegUpdateArgsE",):
        ulines, uloops = _inner_loops(_kernel_body(asm, sym))
        ua, ub = max(uloops, key=lambda ab: ab[1] - ab[0])
        labels = {m.group(1): n for n, l in enumerate(ulines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        cond = [(n, m.group(1)) for n, l in enumerate(ulines[ua:ub], ua) for m in [re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)] if m]
        fwd_in_loop = [ulines[n] for n, t in cond if n < labels.get(t, -1) <= ub]
        assert not fwd_in_loop, fwd_in_loop[:3]                     # no branch of the hot path jumps over code inside the loop
        assert len(cond) >= 24                                      # one (not-taken) exit to the out-of-line body per unrolled step
        # ... and its condition is ready long before: the compare is issued right behind the butterfly that makes state 0, into an
        # SGPR pair, dozens of instructions in front of the scalar test (a v_cmp + s_cbranch_vccnz pair at the end of the step made
        # the wave sit out the VALU -> VCC -> branch latency every step: -1.3 % K7, -2.5 % hard8)
        cmps = [n for n, l in enumerate(ulines[ua:ub], ua) if "v_cmp_ne_u32_e64" in l]
        assert len(cmps) >= 24
        for n, _ in cond[1:25]:
            prev = max(c for c in cmps if c < n)
            assert sum(1 for l in ulines[prev:n] if l.strip().startswith("v_")) >= 40, (prev, n)
    # the chainback ring: the innermost loop that refills rows must wait with counted vmcnt, never vmcnt(0), and hold no store
    lines, loops = _inner_loops(_kernel_body(asm, r"_ZN3vit20reg_chainback_kernel\S*"))
    ring = [(a, b) for a, b in loops if sum("global_load_dwordx4" in l for l in lines[a:b]) >= 16]
    assert ring, "row-ring loop not found"
    a, b = ring[0]
    text = "\n".join(lines[a:b])
    assert "global_store" not in text and "scratch_" not in text
    waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", text)
    assert waits and all(int(w) > 0 for w in waits), waits
    # the chase keeps the survivor's slot (one bit replaced per step) and the bit's position on the scalar unit: 8.5 vector
    # instructions per frame and step -- 543 for the loop's 32 steps of two frames (the rotate-the-state form: ~960)
    valu = sum(1 for l in lines[a:b] if l.strip().startswith("v_"))
    assert valu <= 600 and "v_readfirstlane" not in text, valu


@pytest.mark.parametrize("reg_id,cb_alloc,cb_lds,upd_lds", [(3, 24, 40960, 4096), (4, 24, 40960, 8192), (1, 32, 24576, 12288), (2, 32, 12288, 16384)])
def test_update_kernels_leave_room_for_a_chainback_wave(tmp_path, reg_id, cb_alloc, cb_lds, upd_lds):
    """CDMA IS-95A, CDMA 2000 (K = 9: 64 metric registers per lane; the chunked decision gather and -- at R = 4 -- the sub-chunk
    branch-metric fetch make room), LTE and DAB: two update waves per SIMD without scratch, at most 240 registers each (capped:
    amdgpu_num_vgpr(120)), so that they leave the chainback kernel of the overlapped schedule its registers; and eight update
    waves plus two chainback workgroups fit a CU's 160 KiB of LDS (vit_hip_pipeline_create, rule 3; round 4: CDMA 2000 allocated
    368 registers, LTE 248, DAB 16 + 24 KiB)."""
    text, usage = _compile("reg_inst.hip", [f"-DVIT_REG_ID={reg_id}"], tmp_path)
    upd = [k for k in usage if "reg_update_kernel" in k]
    assert len(upd) == 2
    for k in upd:
        assert usage[k]["ScratchSize"] == 0, (k, usage[k])
        assert usage[k]["alloc"] <= 240 and usage[k].get("AGPRs", 0) == 0, (k, usage[k])
        assert 2 * usage[k]["alloc"] + cb_alloc <= 512
        assert usage[k]["lds_static"] == upd_lds, (k, usage[k])
        assert 8 * usage[k]["lds_static"] + 2 * cb_lds <= 160 * 1024
    cb = [k for k in usage if ("reg_chainback_kernel" if reg_id >= 3 else "reg_chainback_alt_kernel") in k]
    assert len(cb) == 1 and usage[cb[0]]["alloc"] <= cb_alloc and usage[cb[0]]["lds_static"] == 0, usage[cb[0]]
    if reg_id == 2:
        # DAB repeats a polynomial (109, 79, 83, 109): 8 of the 16 patterns occur, and the producer writes only those rows of the
        # ring (RegSpec::pat_used) -- half the entry stores the full set took (192 per kernel)
        for k in upd:
            body = _kernel_body(text, re.escape(k))
            assert body.count("ds_write_b64") + 2 * body.count("ds_write2_b64") <= 110, k


def test_k9_chainback_streams_rows_through_lds_and_fits_beside_two_update_waves(tmp_path):
    """the K = 9 chainback must fit the 32 registers two 240-register update waves leave on a SIMD, its rows must arrive by
    direct-to-LDS loads (no register ring), its main loop must keep counted vmcnt waits, and three of its one-wave workgroups
    plus eight update waves must fit a CU's 160 KiB of LDS."""
    asm, usage = _compile("reg_inst.hip", ["-DVIT_REG_ID=3"], tmp_path)
    cb = [k for k in usage if "reg_chainback_kernel" in k]
    upd = [k for k in usage if "reg_update_kernel" in k]
    assert len(cb) == 1 and len(upd) == 2
    u = usage[cb[0]]
    # ALLOCATED, not used: the ring lives in dynamic LDS precisely so that hipcc does not pad the allocation (static: 264)
    assert u["alloc"] <= 32 and u["ScratchSize"] == 0 and u["lds_static"] == 0, u
    for k in upd:
        assert 2 * usage[k]["alloc"] + u["alloc"] <= 512, (k, usage[k], u)
    # a 65536-frame batch is 512 of these workgroups on 256 CUs: three of them (40 KiB of dynamic LDS each,
    # reg_cb_ring_lds_bytes) must still leave a CU's eight update waves their LDS
    ring_lds = 8 * 4096 + 8192                                  # reg_cb_ring_lds_bytes(64): eight slots of four rows + the parked output
    assert 3 * ring_lds + 8 * max(usage[k]["lds_static"] for k in upd) <= 160 * 1024, (u, [usage[k]["lds_static"] for k in upd])
    body = _kernel_body(asm, r"_ZN3vit20reg_chainback_kernel\w+")
    lines, loops = _inner_loops(body)
    main = max(loops, key=lambda ab: sum("global_load_lds_dwordx4" in l for l in lines[ab[0]:ab[1] + 1]))
    text = "\n".join(lines[main[0]:main[1] + 1])
    assert text.count("global_load_lds_dwordx4") >= 128 and "global_load_dwordx4" not in text      # 32 steps x 4 tiles
    waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", text)
    # one counted wait per step (hipcc drops the few that an earlier wait already implies); a drain only where the flush is
    assert waits.count("28") >= 24 and waits.count("0") <= 1 and set(waits) <= {"0", "28"}, waits
    # the alternative (cooperative) kernel keeps its own budget
    coop = [k for k in usage if "reg_chainback_alt_kernel" in k]
    assert len(coop) == 1 and usage[coop[0]]["ScratchSize"] == 0


def test_k15_kernel_fits_two_workgroups_per_cu(tmp_path):
    asm, usage = _compile("vit_hip.hip", [], tmp_path)
    k15 = [k for k in usage if "lds2_update_kernel_c120ILi15ELi0ELi6" in k]  # the Cassini instantiation (compile-time rate 6)
    assert len(k15) == 1
    u = usage[k15[0]]
    cb = [k for k in usage if "lds2_chainback_kernel" in k]
    assert len(cb) == 1
    for k in usage:                                             # EVERY large-K instantiation stays out of scratch
        if "lds2_update_kernel" in k:
            assert usage[k]["ScratchSize"] == 0, (k, usage[k])
        if "lds2_update_kernel_c120" in k:
            # 512 threads per workgroup, two workgroups per CU = 4 waves per SIMD of 120 registers, and the chainback kernel's
            # allocation (granules of 8) beside them: 512 per SIMD
            assert 4 * usage[k]["alloc"] + usage[cb[0]]["alloc"] <= 512 and usage[k]["alloc"] <= 120, (k, usage[k], usage[cb[0]])
    # (round 3: the 8-bit Cassini instantiation paid the cap with 16 bytes of scratch; since the table build keeps ONE sum per lane
    # and the symbols are paired on the scalar unit, none does)
    # two radix-16 groups per thread sit right at that budget, and since the block loop's control flow is scalar (step range
    # pinned uniform, the careful flag carried as a dword through v_readfirstlane) nothing is left in scratch; the fast
    # block (the code between two workgroup barriers that holds the 64 table reads and the eight 16-byte metric stores; ONE
    # copy serves both table sets) neither spills nor reloads
    assert u["ScratchSize"] == 0, u
    body = _kernel_body(asm, r"_ZN3vit23lds2_update_kernel_c120ILi15ELi0ELi6EEEvNS_14Lds2UpdateArgsE")
    seg, fast = [], []
    for l in body.split("\n") + ["s_barrier"]:
        if "s_barrier" in l:
            if sum("ds_read_b64" in x for x in seg) >= 60 and sum("ds_write_b128" in x for x in seg) == 8:
                fast.append(seg)
            seg = []
        else:
            seg.append(l)
    assert len(fast) == 1, len(fast)
    for seg in fast:
        assert not any("scratch_load" in x for x in seg), "the fast path reloads spilled registers"
        assert not any("scratch_store" in x for x in seg), "the fast path spills"
        # exactly one hardware barrier per block: the first one is split into an LDS arrive / await pair
        assert sum("ds_add_u32" in x for x in seg) >= 1
        # the instruction budget of a block (eight group-steps: 632 of it are add-compare-select and the decision gather, 32 the
        # unpacking of the table offsets, 70 the table build that four of the eight wavefronts run).  Round 2 stood at 834
        # vector instructions; the table-set toggle through one base register, loop-carried LDS offsets, scalar row bases for
        # the decision stores and unconditional symbol loads took out 40
        # round 4: 753 -- the table build makes ONE sum per lane and stores it into both tables (table B is a permutation of
        # table A), selects by mask instead of v_cmp + v_cndmask, and leaves the pairing of the two frames' symbols to the scalar
        # unit of the four building wavefronts (K15 4096 x 8192: 49.3 -> 48.2 ms)
        # round 5: 716 -- the table build picks the lane's expected symbol first (six instructions per symbol where nine were), the
        # metric loads / stores and the decision stores are spelled out (address register + instruction offset: the ten register
        # copies of the loop-carried addresses are gone), the arrive is one LDS add under a one-lane exec mask
        valu = sum(1 for x in seg if x.strip().startswith("v_"))
        assert valu <= 720, valu
        assert sum(x.strip().startswith("v_mov_b32") for x in seg) <= 4
        assert sum("flat_store" in x or "flat_load" in x for x in seg) == 0
    # the block loop branches on scalar conditions: its header compares the step counter in SGPRs
    hdr = body[body.index("This Loop Header: Depth=1"):]
    hdr = hdr[:hdr.index("s_cbranch")]
    assert "s_cmp_lt_u32" in hdr and "saveexec" not in hdr, hdr


def test_library_descriptor_table_matches_the_compiler(tmp_path):
    """vit_hip_list_kernels (csrc/kernel_desc.hpp: the descriptors read from the library's own .hip_fatbin, which the pipeline's
    residency rules consult) must say what hipcc's -S output says for the same source -- checked on the quickest unit (K = 3)
    for every kernel in it, and on the allocation that round 3 got wrong (K = 9 chainback) by name."""
    import ctypes as C
    import sys
    sys.path.insert(0, ROOT)
    from viterbidecodercpp_amd import _lib
    lib = _lib.load()
    table = _lib.list_kernels()
    assert len(table) >= 90
    _, usage = _compile("reg_inst.hip", ["-DVIT_REG_ID=5"], tmp_path)
    checked = 0
    for name, u in usage.items():
        if "alloc" not in u:
            continue
        got = table[name]
        assert (got["vgpr_alloc"], got["lds_static_bytes"]) == (u["alloc"], u["lds_static"]), (name, got, u)
        assert got["scratch_bytes"] == u.get("ScratchSize", 0)
        checked += 1
    assert checked == 7          # update x 2, resume x 2, chainback, alt chainback, export
    k9 = [v for k, v in table.items() if "20reg_chainback_kernelINS_7RegSpecILi9ELi2E" in k]
    assert len(k9) == 1 and k9[0]["vgpr_alloc"] <= 32 and k9[0]["lds_static_bytes"] == 0, k9
