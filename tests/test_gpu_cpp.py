"""The C++ host layer (include/viterbi_hip/*.h) exercised by C++ programs that mirror the reference's own executables:
run_simple (examples/run_simple.cpp) and the run_tests matrix (examples/run_tests.cpp), the latter extended with noisy
frames checked bit for bit against the oracle.  Built by __graft_entry__.build() / `make -C tests/cpp`."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _ensure_built():
    if not all(os.path.exists(os.path.join(CPP, n)) for n in ("run_simple_hip", "run_tests_hip", "run_batch_hip", "run_multi_gpu_hip", "run_stream_hip")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True, capture_output=True)
        subprocess.run(["make", "-C", CPP], check=True, capture_output=True)


def test_cpp_programs_build():
    _ensure_built()
    subprocess.run(["make", "-q", "-C", CPP], check=False)
    assert os.access(os.path.join(CPP, "run_simple_hip"), os.X_OK)
    assert os.access(os.path.join(CPP, "run_tests_hip"), os.X_OK)
    assert os.access(os.path.join(CPP, "run_batch_hip"), os.X_OK)
    assert os.access(os.path.join(CPP, "run_multi_gpu_hip"), os.X_OK)


@pytest.mark.gpu
def test_run_simple_hip():
    _ensure_built()
    p = subprocess.run([os.path.join(CPP, "run_simple_hip")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "error_metric=0" in p.stdout and "0/8192 incorrect bits" in p.stdout


@pytest.mark.gpu
def test_run_stream_hip():
    """the reference's streaming call pattern (one update() per trellis step) through the header drop-in: queued on the host,
    run on the GPU in one launch per 2048 steps, every observable equal to one call over the whole stream"""
    _ensure_built()
    p = subprocess.run([os.path.join(CPP, "run_stream_hip")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.strip().endswith("PASS")
    us = float(p.stdout.split(" us per call")[0].split()[-1])
    assert us < 5.0, p.stdout            # 25 us per call before the queue; about 0.5 us with it


@pytest.mark.gpu
def test_run_tests_hip_matrix():
    _ensure_built()
    p = subprocess.run([os.path.join(CPP, "run_tests_hip")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr
    assert "FAIL" not in p.stdout
    last = p.stdout.strip().splitlines()[-1]
    assert last.startswith("PASSED ") and last.split()[1].split("/")[0] == last.split()[1].split("/")[1]


@pytest.mark.gpu
def test_run_batch_hip():
    """batched route from C++ (ViterbiDecoder_HIP_Batch) vs the oracle and vs the single-frame drop-in."""
    _ensure_built()
    p = subprocess.run([os.path.join(CPP, "run_batch_hip")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "mismatching frames=0" in p.stdout and p.stdout.strip().endswith("PASS")


@pytest.mark.gpu
def test_run_multi_gpu_hip():
    """C++ host, no Python: RCCL communicator over every visible GPU, vit_hip_broadcast_table, one decoder and one shard of
    the global batch per GPU (one rank on the one-GPU box; the same binary uses all eight on a full node)."""
    _ensure_built()
    p = subprocess.run([os.path.join(CPP, "run_multi_gpu_hip"), "4096", "2048"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stdout + p.stderr
    assert "table ok" in p.stdout and "BAD" not in p.stdout and p.stdout.strip().endswith("PASS")
    # the line before PASS is one record in bench.py's schema (a node run gives a scaling point from C++ too)
    import json
    rec = json.loads(p.stdout.strip().splitlines()[-2])
    assert rec["unit"] == "Mbit/s" and rec["n_gpus"] >= 1 and rec["scaling"] == "weak" and rec["value"] > 0
    assert len(rec["per_rank_Mbit_s"]) == rec["n_gpus"] and abs(sum(rec["per_rank_Mbit_s"]) - rec["value"]) < 0.35 * rec["value"]
