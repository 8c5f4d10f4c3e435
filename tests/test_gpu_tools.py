"""The two harness tools as a user runs them: their JSON must carry the reference's record layout
(examples/run_benchmark.cpp:297-327, examples/run_snr_ber.cpp:419-441) so that the reference's parse_benchmark.py /
plot_snr_ber.py read it with one more SIMD_Type entry."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(module, *args):
    p = subprocess.run([sys.executable, "-m", module, *args], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return json.loads(p.stdout), p.stderr


def test_run_benchmark_records():
    recs, _ = _run("viterbidecodercpp_amd.tools.run_benchmark", "--codes", "2", "7", "--decode-types", "SOFT16", "HARD8", "-T", "0.05")
    assert len(recs) == 4
    for r in recs:
        for key in ("name", "decode_type", "simd_type", "K", "R", "G", "total_input_bits", "total_symbols", "update_symbols_ns",
                    "chainback_bits_ns"):
            assert key in r, key
        assert r["simd_type"] == "SIMD_HIP" and len(r["G"]) == r["R"]
        assert r["total_symbols"] == r["frames"] * (2048 + r["K"] - 1) * r["R"] and r["total_input_bits"] == r["frames"] * 2048
        assert len(r["update_symbols_ns"]) == len(r["chainback_bits_ns"]) >= 1
        # what parse_benchmark.py computes from a record (symbols/s, bits/s): finite and positive
        upd = r["total_symbols"] / (sum(r["update_symbols_ns"]) / len(r["update_symbols_ns"]) * 1e-9)
        cb = r["total_input_bits"] / (sum(r["chainback_bits_ns"]) / len(r["chainback_bits_ns"]) * 1e-9)
        assert upd > 1e8 and cb > 1e8


def test_run_snr_ber_records():
    recs, log = _run("viterbidecodercpp_amd.tools.run_snr_ber", "--codes", "2", "--decode-types", "SOFT16", "--bits-scale", "0.02",
                     "--max-points", "30", "--ebn0-initial", "2.0", "--ebn0-step", "1.0")
    assert len(recs) == 1
    r = recs[0]
    for key in ("name", "decode_type", "simd_type", "K", "R", "G", "EbNo_dB", "ber"):
        assert key in r, key
    assert r["name"] == "Voyager" and r["simd_type"] == "SIMD_HIP" and len(r["EbNo_dB"]) == len(r["ber"]) >= 2
    assert r["EbNo_dB"][0] == 2.0 and r["EbNo_dB"][1] == 3.0
    # the curve falls, and the sweep stops at its first error-free point (run_snr_ber.cpp:396)
    assert all(a >= b for a, b in zip(r["ber"], r["ber"][1:])) and r["ber"][-1] == 0.0 and all(b > 0 for b in r["ber"][:-1])
    assert 1e-3 < r["ber"][0] < 2e-2      # Voyager soft16 at 2 dB: 6e-3
    assert "simd=SIMD_HIP" in log
