"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol include/vit_hip.h declares,
argument checking happens before any device is touched, the host mirror of the reference interface builds the same branch
table as the reference, and there is NO CPU decode path (no GPU => loud failure)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from viterbidecodercpp_amd import (COMMON_CODES, ViterbiBranchTable, ViterbiDecoder_Config, ViterbiDecoder_Core, _lib,
                                   get_decoding_config, pack_blob, synth)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "vit_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(vit_hip_[a-z0-9_]+)\s*\(", header)))
    assert declared == sorted(_lib.EXPORTS)
    lib = _lib.load()
    for name in declared:
        assert getattr(lib, name) is not None


def test_branch_table_matches_reference_fixture():
    manifest = json.load(open(os.path.join(GOLD, "MANIFEST.json")))
    for name, meta in manifest.items():
        g = np.load(os.path.join(GOLD, name + ".npz"))
        pc = get_decoding_config(meta["decode_type"], meta["R"])
        t = ViterbiBranchTable(meta["K"], meta["R"], meta["G"], pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
        assert np.array_equal(t.data().astype(np.int16), g["table"]), name
        assert np.array_equal(t[0], t.data()[0])


def test_blob_round_trip_layout():
    code = COMMON_CODES[2]
    pc = get_decoding_config("SOFT16", code.R)
    t = ViterbiBranchTable(code.K, code.R, code.G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
    cfg = ViterbiDecoder_Config.from_decoder_config(pc)
    blob = pack_blob(t, cfg)
    lib = _lib.load()
    assert len(blob) == lib.vit_hip_blob_bytes(7, 2, 2, 2) == 20 + 2 * 32 * 2 + 8
    assert blob[20:20 + 128] == t.data().tobytes()
    assert blob[-8:] == cfg.as_array().tobytes()


def test_argument_errors_come_before_device_access():
    lib = _lib.load()
    h = C.c_void_p()
    tbl = np.zeros((2, 32), dtype=np.int16)
    cfg = np.zeros(4, dtype=np.uint16)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    assert lib.vit_hip_create(1, 2, 2, 2, p(tbl), p(cfg), 0, C.byref(h)) == _lib.ERR_UNSUPPORTED
    assert lib.vit_hip_create(17, 2, 2, 2, p(tbl), p(cfg), 0, C.byref(h)) == _lib.ERR_UNSUPPORTED
    assert lib.vit_hip_create(7, 9, 2, 2, p(tbl), p(cfg), 0, C.byref(h)) == _lib.ERR_UNSUPPORTED
    assert lib.vit_hip_create(7, 2, 2, 1, p(tbl), p(cfg), 0, C.byref(h)) == _lib.ERR_UNSUPPORTED
    assert lib.vit_hip_create(7, 2, 2, 2, None, p(cfg), 0, C.byref(h)) == _lib.ERR_INVALID_ARG
    assert b"NULL" in lib.vit_hip_last_error()
    assert lib.vit_hip_update_batch(None, None, 1, 1, 1, None, 0, None, None, None, None) == _lib.ERR_INVALID_ARG
    assert lib.vit_hip_create_from_blob(p(np.zeros(4, np.uint8)), 4, 0, C.byref(h)) == _lib.ERR_INVALID_ARG


def test_kernel_table_is_readable_without_a_gpu_and_argument_errors_are_reported():
    """vit_hip_list_kernels reads the library's own code objects from disk (no device needed): every stock update / chainback kernel
    is in it with a sane allocation; the handle-bound query and the new pipeline entry point reject NULL arguments."""
    lib = _lib.load()
    table = _lib.list_kernels()
    assert len(table) >= 90
    for frag in ("17reg_update_kernelINS_7RegSpecILi7ELi2ELj109ELj79E", "20reg_chainback_kernelINS_7RegSpecILi9ELi2ELj491ELj369E",
                 "24reg_chainback_alt_kernelINS_7RegSpecILi7ELi2E", "23lds2_update_kernel_c120ILi15ELi0ELi6E", "21lds2_chainback_kernelE"):
        hits = [v for k, v in table.items() if frag in k]
        assert hits and all(8 <= v["vgpr_alloc"] <= 512 and v["vgpr_alloc"] % 8 == 0 for v in hits), frag
    r = _lib.VitHipKernelResources()
    name = C.create_string_buffer(8)
    assert lib.vit_hip_list_kernels(0, name, len(name), C.byref(r)) == _lib.OK and len(name.value) == 7     # truncated, terminated
    assert lib.vit_hip_list_kernels(10 ** 6, name, len(name), C.byref(r)) == _lib.ERR_INVALID_ARG
    assert lib.vit_hip_get_kernel_resources(None, 0, C.byref(r)) == _lib.ERR_INVALID_ARG
    assert lib.vit_hip_pipeline_wait_event(None, None) == _lib.ERR_INVALID_ARG


def test_no_cpu_fallback_without_gpu():
    """on a machine without a GPU the product path must fail loudly, never decode on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    code = COMMON_CODES[2]
    pc = get_decoding_config("SOFT16", code.R)
    t = ViterbiBranchTable(code.K, code.R, code.G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
    with pytest.raises(_lib.VitHipError) as e:
        ViterbiDecoder_Core(t, ViterbiDecoder_Config.from_decoder_config(pc))
    assert e.value.code == _lib.ERR_NO_DEVICE


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "viterbidecodercpp_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "pyoracle" not in src and "viterbi_oracle" not in src and "libvitref" not in src, fn
    for fn in os.listdir(os.path.join(ROOT, "include")):
        pass


def test_synth_numpy_and_torch_encoders_agree():
    import torch

    code = COMMON_CODES[5]
    pc = get_decoding_config("SOFT16", code.R)
    tx, sym = synth.make_frames_torch(code, pc, 5, 256, None, seed=3, device="cpu")
    coded = synth.encode_bits_numpy(code.K, code.R, code.G, tx.numpy())
    want = np.where(coded != 0, pc.soft_decision_high, pc.soft_decision_low).astype(np.int16)
    assert np.array_equal(sym.numpy(), want)
    # noisy symbols stay inside the soft-decision range
    _, noisy = synth.make_frames_torch(code, pc, 3, 256, 1.0, seed=4, device="cpu")
    assert int(noisy.max()) <= pc.soft_decision_high and int(noisy.min()) >= pc.soft_decision_low


def test_kernel_descriptor_reader_on_a_run_time_compiled_code_object(tmp_path):
    """csrc/kernel_desc.hpp must read what `hipcc --genco` writes (an offload BUNDLE of code objects, not a bare ELF: the first
    version of the reader saw nothing in it and every run-time compiled code silently got the conservative schedule) and agree
    with the assembler's view of the same source: register allocation with LDS-occupancy padding, static LDS, scratch."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    kd_dump = os.path.join(ROOT, "tests", "cpp", "kd_dump")
    if not os.path.exists(kd_dump):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "kd_dump"], check=True, capture_output=True)
    src = tmp_path / "k.hip"
    src.write_text('#include <hip/hip_runtime.h>\n'
                   'extern "C" __global__ void __launch_bounds__(64) padded(int* p) { __shared__ int big[10240]; big[threadIdx.x] = p[threadIdx.x]; '
                   '__syncthreads(); p[threadIdx.x] = big[63 - threadIdx.x]; }\n'                       # 40 KiB of STATIC LDS: one wave per SIMD
                   'extern "C" __global__ void __launch_bounds__(64) lean(int* p) { extern __shared__ int dyn[]; dyn[threadIdx.x] = p[threadIdx.x]; '
                   '__syncthreads(); p[threadIdx.x] = dyn[63 - threadIdx.x]; }\n')                      # the same with dynamic LDS
    hsaco, asm = tmp_path / "k.hsaco", tmp_path / "k.s"
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "--genco", "-o", str(hsaco), str(src)], check=True, capture_output=True)
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", str(asm), str(src)], check=True, capture_output=True)
    assert open(hsaco, "rb").read(24) == b"__CLANG_OFFLOAD_BUNDLE__"
    got = {}
    for line in subprocess.run([kd_dump, str(hsaco)], check=True, capture_output=True, text=True).stdout.splitlines():
        v, a, l, s, name = line.split()
        got[name] = (int(v), int(l), int(s))
    text = asm.read_text()
    for name in ("padded", "lean"):
        m = re.search(r"\.amdhsa_kernel " + name + r"\n(.*?)\.end_amdhsa_kernel", text, re.S).group(1)
        f = lambda k: int(re.search(r"\.amdhsa_" + k + r" (\d+)", m).group(1))  # noqa: E731
        assert got[name] == (-(-f("next_free_vgpr") // 8) * 8, f("group_segment_fixed_size"), f("private_segment_fixed_size")), (name, got[name])
    # the finding itself, kept as a regression: static LDS that admits one wave per SIMD pads the allocation past 256 registers
    assert got["padded"][0] >= 264 and got["padded"][1] == 40960 and got["lean"][0] <= 32 and got["lean"][1] == 0, got


def test_precompile_writes_a_code_object_without_a_gpu(tmp_path):
    """vit_hip_precompile (include/vit_hip.h): install-time instantiation of the register plan for a polynomial set -- hipcc only,
    no GPU call, so it runs on this build host; the object's name carries code, width, target and source hash; an existing object is
    kept (second call: no compiler needed); what the register plan does not serve is an error code."""
    import ctypes as C
    import shutil

    from viterbidecodercpp_amd import _lib
    from viterbidecodercpp_amd.tools import precompile

    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    lib = _lib.load()
    paths = precompile.precompile([(2, 2, (0o3, 0o1), 2)], str(tmp_path), verbose=False)
    name = os.path.basename(paths[0])
    assert name.startswith("reg_K2R2_3_3_0_0_0_0_s16_gfx950_") and name.endswith(".hsaco"), name      # 0o1 -> 0o3: bit 0 and bit K-1 are implied
    blob = open(paths[0], "rb").read()
    assert b"vit_jit_update_16" in blob and b"vit_jit_chainback" in blob and b"gfx950" in blob
    assert sorted(os.listdir(tmp_path)) == [name]                                 # no source, log or temporary left behind
    os.environ["VIT_HIP_HIPCC"] = "/nonexistent/hipcc"
    try:
        assert precompile.precompile([(2, 2, (0o3, 0o1), 2)], str(tmp_path), verbose=False) == paths
        G = (C.c_uint32 * 6)(0o5, 0o7)
        assert lib.vit_hip_precompile(3, 2, G, 2, str(tmp_path).encode(), None, 0) == _lib.ERR_RUNTIME and b"hipcc" in lib.vit_hip_last_error()
    finally:
        del os.environ["VIT_HIP_HIPCC"]
    G = (C.c_uint32 * 6)(0o1167, 0o1545)
    assert lib.vit_hip_precompile(10, 2, G, 2, str(tmp_path).encode(), None, 0) == _lib.ERR_UNSUPPORTED
    assert lib.vit_hip_precompile(7, 2, G, 4, str(tmp_path).encode(), None, 0) == _lib.ERR_UNSUPPORTED
    # all polynomials zero names the GENERIC kernels of (K, R) (polynomials read at run time): K = 3..9 with R = 1..4 (not K = 6 at an odd rate)
    Z = (C.c_uint32 * 6)()
    assert lib.vit_hip_precompile(6, 3, Z, 2, str(tmp_path).encode(), None, 0) == _lib.ERR_UNSUPPORTED
    assert lib.vit_hip_precompile(6, 1, Z, 1, str(tmp_path).encode(), None, 0) == _lib.ERR_UNSUPPORTED
    assert lib.vit_hip_precompile(2, 2, Z, 2, str(tmp_path).encode(), None, 0) == _lib.ERR_UNSUPPORTED
    assert lib.vit_hip_precompile(7, 5, Z, 2, str(tmp_path).encode(), None, 0) == _lib.ERR_UNSUPPORTED
    # the package cache build() fills: every common set, both widths, compiled from the CURRENT kernel sources
    pkg = os.path.join(ROOT, "viterbidecodercpp_amd", "precompiled")
    have = set(os.listdir(pkg)) if os.path.isdir(pkg) else set()
    suffix = name.rsplit("_", 1)[1]
    for _, K, R, Gs in precompile.COMMON_SETS:
        for w in (8, 16):
            g = [x | 1 | (1 << (K - 1)) if any(Gs) else 0 for x in Gs] + [0] * (6 - R)     # (all zero: the GENERIC kernels of (K, R))
            want = f"reg_K{K}R{R}_" + "_".join(str(x) for x in g) + f"_s{w}_gfx950_{suffix}"
            assert want in have, f"{want} missing from {pkg}: run build()"


def test_traffic_json_follows_from_its_sources():
    """profiles/traffic.json (bench.py's roofline.traffic and roofline_valu instruction counts) is derived data: every entry must be
    what scripts/make_traffic.py computes from the rocprofv3 summary its `source` names."""
    import subprocess
    import sys

    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_traffic.py"), "--check"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    for key, e in json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).items():
        assert os.path.exists(os.path.join(ROOT, e["source"])), (key, e["source"])


def test_reference_parse_benchmark_reads_the_tools_records(capsys, monkeypatch):
    """INTEGRATION.md says the reference's examples/parse_benchmark.py digests viterbidecodercpp_amd.tools.run_benchmark's output
    once `SIMD_HIP = 4` joins its SimdType enum.  Checked with the reference's own script, imported from where it lies (build container
    only: nothing of it travels), that one entry added, and a committed record of the tool (sample lists cut to 64 entries)."""
    import glob
    import importlib.util
    from enum import Enum

    script = "/root/reference/examples/parse_benchmark.py"
    if not os.path.exists(script):
        pytest.skip("reference tree absent")
    records = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_run_benchmark_hip_raw.json")))
    assert records, "no committed run_benchmark record"
    spec = importlib.util.spec_from_file_location("ref_parse_benchmark", script)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # the maintainer's one-line edit: one more entry in the enum
    mod.SimdType = Enum("SimdType", {**{e.name: e.value for e in mod.SimdType}, "SIMD_HIP": 4})
    monkeypatch.setattr("sys.argv", ["parse_benchmark.py", records[-1]])
    mod.main()
    out = capsys.readouterr().out
    recs = json.load(open(records[-1]))
    assert len(recs) == 24 and out.count("simd=simd_hip") == 24
    assert "name='Voyager',K=7,R=2,decode=SOFT16" in out and "name='Cassini',K=15,R=6,decode=HARD8" in out
    import re as _re
    rates = [float(x) for x in _re.findall(r"update    = ([0-9.]+) ± [0-9.]+ gigasymbols/s", out)]
    assert len(rates) >= 20 and max(rates) > 100      # K = 3: hundreds of Gsym/s on one card
    # unmodified, the script refuses the strategy name -- that IS the documented edit
    mod2 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod2)
    with pytest.raises(Exception, match="invalid simd type"):
        mod2.Sample(recs[0])
