"""N > 1 path on CPU: two gloo ranks shard a batch, rank 0 broadcasts the decoder blob, results are gathered and must equal
the single-process answer.  The decode itself is stood in for by the oracle here (no GPU in this container); what is under
test is the sharding, the blob broadcast and the gather -- the code bench.py runs over RCCL on the GPU box."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, frames, L, tmpdir):
    import torch
    import torch.distributed as dist

    from oracle import pyoracle
    from viterbidecodercpp_amd import (COMMON_CODES, ViterbiBranchTable, ViterbiDecoder_Config, dist as vdist,
                                       get_decoding_config, pack_blob, synth)

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    code = COMMON_CODES[2]
    pc = get_decoding_config("SOFT16", code.R)
    blob = None
    if rank == 0:
        table = ViterbiBranchTable(code.K, code.R, code.G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
        blob = pack_blob(table, ViterbiDecoder_Config.from_decoder_config(pc))
    blob = vdist.broadcast_blob(blob, src=0)
    # every rank must now hold rank 0's table and config byte for byte
    H = 32
    table_bytes = np.frombuffer(blob[20:20 + 2 * H * 2], dtype=np.int16).reshape(2, H)
    cfg4 = np.frombuffer(blob[-8:], dtype=np.uint16)
    assert tuple(int(x) for x in cfg4) == pc.cfg4()

    _, sym = synth.make_frames_numpy(code, pc, frames, L, 3.0, seed=77)   # same seed everywhere: the "global" batch
    b, e = vdist.shard_range(frames, rank, world)
    oracle = pyoracle.Oracle()
    ocfg = pyoracle.stock_config(pyoracle.SOFT16, code.R)
    assert np.array_equal(oracle.branch_table(code.K, code.R, code.G, ocfg.high, ocfg.low), table_bytes)
    local, _, _ = oracle.decode_frames(code.K, code.R, code.G, ocfg, sym[b:e], L)
    parts = vdist.gather_frames(torch.from_numpy(local), dst=0)
    if rank == 0:
        got = np.concatenate([p.numpy() for p in parts], axis=0)
        np.save(os.path.join(tmpdir, "gathered.npy"), got)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    from viterbidecodercpp_amd.dist import shard_range

    for total in (0, 1, 7, 8, 65536, 262144 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_shard_broadcast_gather(tmp_path, oracle):
    import torch.multiprocessing as mp

    from oracle import pyoracle
    from viterbidecodercpp_amd import COMMON_CODES, get_decoding_config, synth

    frames, L = 13, 256          # odd count => ragged shards
    port = _free_port()
    mp.spawn(_worker, args=(2, port, frames, L, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "gathered.npy"))
    code = COMMON_CODES[2]
    pc = get_decoding_config("SOFT16", code.R)
    _, sym = synth.make_frames_numpy(code, pc, frames, L, 3.0, seed=77)
    want, _, _ = oracle.decode_frames(code.K, code.R, code.G, pyoracle.stock_config(pyoracle.SOFT16, code.R), sym, L)
    assert np.array_equal(got, want)


def test_bench_launcher_dry_run_with_eight_gloo_ranks():
    """bench.py's own launch_ranks (the `python bench.py --gpus N` form) with EIGHT ranks on the CPU: rendezvous on 127.0.0.1, the
    build marker, the blob broadcast, the all-reduce that proves eight ranks hold the same decoder, the per-rank gather -- and the
    skeleton of the 8-rank record (BASELINE configs[3]: 262144 hard-decision frames, 32768 per rank, contiguous shards)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--dry-run", "--config", "3"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                       # rank 0 only
    r = json.loads(lines[0])
    assert r["dry_run"] and r["value"] is None and r["n_gpus"] == 8 and r["scaling"] == "weak"
    assert r["ranks"]["ranks_in_blob_allreduce"] == 8 and r["ranks"]["blob_checksum_identical_on_all_ranks"]
    assert [x["rank"] for x in r["ranks"]["per_rank"]] == list(range(8))
    assert r["global_frames"] == 262144 and r["frame_ranges"][0] == [0, 32768] and r["frame_ranges"][7] == [229376, 262144]
    assert all(a[1] == b[0] for a, b in zip(r["frame_ranges"], r["frame_ranges"][1:]))     # contiguous, disjoint shards
