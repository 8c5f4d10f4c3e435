"""Parity at BASELINE.json's FULL sizes: encode -> AWGN -> decode round trips on frames generated in HBM
(vit_hip_synth_batch), two kernel plans agreeing byte for byte, and the oracle decoding EVERY frame of the K = 7 and K = 9
batches on the host cores (SURVEY 8 d, "parity at scale"): all chainback bytes, all final metrics, all renormalisation sums,
and a 64-bit digest per frame over every decision word (oracle/viterbi_oracle.h: vo_decode_frames_hashed).  K = 15, where
the scalar oracle needs 90 us per bit, is checked the same way on a spread subset of frames with the words themselves."""
import os

import numpy as np
import pytest

from viterbidecodercpp_amd import COMMON_CODES, BatchDecoder, _lib
from tests.helpers import make_table_config, oracle_cfg

pytestmark = pytest.mark.gpu


def _host_threads():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU box shows 256 logical CPUs
    and grants 16)"""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return n


def _need_hbm(torch, nbytes, what):
    free, total = torch.cuda.mem_get_info()
    if free < nbytes:
        # never shrink silently: a box that cannot hold the configuration fails the test with the numbers
        pytest.fail(f"{what} needs {nbytes / 2**30:.1f} GiB of HBM, {free / 2**30:.1f} of {total / 2**30:.1f} GiB free")


def _oracle_subset(oracle, dec, code, decode_type, pc, sym, out, met, rs, frames, L, n_pick, seed, ws=None):
    """`n_pick` slab-aligned groups spread over the batch (first, last, random): everything the decoder produced for one frame
    of each against the C restatement."""
    import torch

    dec._handle.refresh()
    tile = dec._handle.info.workspace_tile_frames
    rng = np.random.default_rng(seed)
    slabs = sorted(set([0, (frames - 1) // tile] + [int(x) for x in rng.integers(0, (frames - 1) // tile + 1, n_pick)]))[:n_pick]
    for k, sl in enumerate(slabs):
        f0 = sl * tile
        n = min(tile, frames - f0)
        f = f0 + (k * 7) % n                       # a different lane / half of the tile each time
        got_dec = dec.export_decisions(n, L, workspace=ws, first_frame=f0)[f - f0].cpu().numpy().view(np.uint64)
        want = oracle.decode(code.K, code.R, code.G, oracle_cfg(decode_type, code.R), sym[f].cpu().numpy(), L)
        assert np.array_equal(got_dec, want["decisions"]), f"decision words of frame {f} differ"
        assert np.array_equal(out[f].cpu().numpy(), want["bytes"]), f"chainback bytes of frame {f} differ"
        m = met[f].cpu().numpy()
        m = m.view(np.uint16) if pc.error_bytes == 2 else m
        assert np.array_equal(m.astype(np.uint32), want["metrics"]), f"final metrics of frame {f} differ"
        assert int(rs[f].item()) == want["renorm_sum"]
    torch.cuda.synchronize()


def _every_frame_exact(oracle, dec, code, decode_type, pc, sym, out, met, rs, frames, L):
    """The whole batch against the C restatement: bytes, metrics, renormalisation sums, decision-word digests."""
    import torch

    S, W = L + code.K - 1, dec.W
    threads = _host_threads()
    want_out, want_met, want_rs, want_hash = oracle.decode_frames(
        code.K, code.R, code.G, oracle_cfg(decode_type, code.R), sym.cpu().numpy(), L, threads=threads, want_metrics=True,
        want_hash=True)
    # the digest of the GPU's decision rows, exported slab by slab in the reference's [S][W] layout
    with np.errstate(over="ignore"):
        mul = (2 * np.arange(S * W, dtype=np.uint64) + 1) * np.uint64(oracle.HASH_MUL)
    mul = torch.from_numpy(mul.view(np.int64)).to(sym.device)
    dec._handle.refresh()
    tile = dec._handle.info.workspace_tile_frames
    chunk = max(tile, (2048 // tile) * tile)
    got_hash = torch.empty(frames, dtype=torch.int64, device=sym.device)
    for f0 in range(0, frames, chunk):
        n = min(chunk, frames - f0)
        words = dec.export_decisions(n, L, first_frame=f0).view(n, S * W)
        got_hash[f0:f0 + n] = (words * mul).sum(dim=1)      # int64 arithmetic wraps: the same sum mod 2^64
        del words
    bad = np.nonzero(got_hash.cpu().numpy().view(np.uint64) != want_hash)[0]
    assert bad.size == 0, f"decision words differ in {bad.size} frames, first {bad[:8]}"
    got = out.cpu().numpy()
    bad = np.nonzero((got != want_out).any(axis=1))[0]
    assert bad.size == 0, f"chainback bytes differ in {bad.size} frames, first {bad[:8]}"
    m = met.cpu().numpy()
    m = m.view(np.uint16) if pc.error_bytes == 2 else m
    assert np.array_equal(m, want_met.astype(m.dtype)) and int(want_met.max()) < (1 << (8 * pc.error_bytes))
    assert np.array_equal(rs.cpu().numpy().astype(np.uint64), want_rs)


def _slabs_exact_by_digest(oracle, dec, code, decode_type, pc, sym, out, met, rs, frames, L, n_slabs, budget_note):
    """`n_slabs` whole workspace slabs spread evenly over the batch, every frame of each against the C restatement run on all
    host cores: bytes, metrics, renormalisation sums and the 64-bit digest over every decision word (the K = 15 form of
    _every_frame_exact: the scalar oracle needs ~0.75 s per 8192-bit Cassini frame)."""
    import torch

    S, W = L + code.K - 1, dec.W
    dec._handle.refresh()
    tile = dec._handle.info.workspace_tile_frames
    n_tiles = (frames + tile - 1) // tile
    slabs = sorted(set(int(x) for x in np.linspace(0, n_tiles - 1, n_slabs).round()))
    ids = np.asarray([f for sl in slabs for f in range(sl * tile, min((sl + 1) * tile, frames))])
    threads = _host_threads()
    d_ids = torch.from_numpy(ids).to(sym.device)
    want_out, want_met, want_rs, want_hash = oracle.decode_frames(
        code.K, code.R, code.G, oracle_cfg(decode_type, code.R), sym[d_ids].cpu().numpy(), L, threads=threads,
        want_metrics=True, want_hash=True)
    with np.errstate(over="ignore"):
        mul = (2 * np.arange(S * W, dtype=np.uint64) + 1) * np.uint64(oracle.HASH_MUL)
    mul = torch.from_numpy(mul.view(np.int64)).to(sym.device)
    got_hash = []
    for sl in slabs:
        f0 = sl * tile
        n = min(tile, frames - f0)
        words = dec.export_decisions(n, L, first_frame=f0).view(n, S * W)
        got_hash.append((words * mul).sum(dim=1))
        del words
    got_hash = torch.cat(got_hash).cpu().numpy().view(np.uint64)
    bad = np.nonzero(got_hash != want_hash)[0]
    assert bad.size == 0, f"{budget_note}: decision words differ in frames {ids[bad][:8]}"
    assert np.array_equal(out[d_ids].cpu().numpy(), want_out), budget_note
    m = met[d_ids].cpu().numpy()
    m = m.view(np.uint16) if pc.error_bytes == 2 else m
    assert np.array_equal(m, want_met.astype(m.dtype)), budget_note
    assert np.array_equal(rs[d_ids].cpu().numpy().astype(np.uint64), want_rs), budget_note
    return len(ids)


@pytest.mark.parametrize("code_id,decode_type,frames,L,ebn0,ber_max", [
    (2, "SOFT16", 65536, 8192, 3.0, 2e-3),    # BASELINE configs[1]: K=7 R=1/2 u16, 64k frames x 8192 bits
    (5, "SOFT16", 65536, 8192, 3.0, 1e-3),    # configs[2]: K=9 R=1/2 u16, 64k frames (17.2 GB of decision rows)
    (2, "HARD8", 32768, 8192, 5.0, 1e-3),     # configs[3]: one GPU's share (32768 frames) of the 8-GPU hard-decision run
])
def test_full_size_round_trip_and_every_frame_exact(oracle, code_id, decode_type, frames, L, ebn0, ber_max):
    import torch

    code = COMMON_CODES[code_id]
    pc, table, config = make_table_config(code, decode_type)
    dec = BatchDecoder(table, config)
    S = L + code.K - 1
    _need_hbm(torch, dec.workspace_bytes(frames, L) + 2 * frames * S * code.R * pc.soft_bytes + (1 << 30),
              f"{code.name} {decode_type} {frames} x {L}")
    # noise-free: every frame must decode to exactly what was sent (the reference's own test property, at full size)
    tx, sym = dec.synth(frames, L, None, seed=5)
    out = dec.decode(sym, L)
    assert torch.equal(out, tx)
    # noisy: BER in the expected range, and a spread subset of frames bit-exact against the oracle
    tx, sym = dec.synth(frames, L, ebn0, seed=6, tx_out=tx, symbols_out=sym)
    out, met, rs = dec.decode(sym, L, want_metrics=True)
    errors = int(dec.count_bit_errors(out, tx).item())
    ber = errors / float(frames * L)
    assert 0 < ber < ber_max, ber
    _oracle_subset(oracle, dec, code, decode_type, pc, sym, out, met, rs, frames, L, n_pick=3, seed=1)
    _every_frame_exact(oracle, dec, code, decode_type, pc, sym, out, met, rs, frames, L)
    # the device's error counter against numpy on the same bytes (identical BER for GPU and oracle follows from the above)
    assert errors == int(np.unpackbits(out.cpu().numpy() ^ tx.cpu().numpy()).sum())
    # the LDS plan (different kernels, different decision layout) must give the same bytes on a slice
    n = 2048
    dec._ws = None                             # release the big workspace before the LDS decoder allocates its own
    lds = BatchDecoder(table, config, plan=_lib.PLAN_LDS)
    assert torch.equal(lds.decode(sym[:n].contiguous(), L), out[:n])


def test_cassini_k15_full_size(oracle):
    """configs[4]: K=15 R=1/6 u16, 16384 states in LDS, 4096 frames x 8192 bits: 68.8 GB of decision rows."""
    import torch

    code = COMMON_CODES[7]
    pc, table, config = make_table_config(code, "SOFT16")
    frames, L = 4096, 8192
    dec = BatchDecoder(table, config)
    S = L + code.K - 1
    _need_hbm(torch, dec.workspace_bytes(frames, L) + frames * S * code.R * 2 + (1 << 30), "Cassini 4096 x 8192")
    tx, sym = dec.synth(frames, L, 3.0, seed=9)
    out, met, rs = dec.decode(sym, L, want_metrics=True)
    ber = int(dec.count_bit_errors(out, tx).item()) / float(frames * L)
    assert ber < 1e-4, ber                    # Cassini at Eb/N0 = 3 dB decodes clean (the oracle's BER at 2 dB is already 0)
    _oracle_subset(oracle, dec, code, "SOFT16", pc, sym, out, met, rs, frames, L, n_pick=8, seed=2)
    # a noisier batch in the same buffers: errors, ties and renormalisation all occur
    tx, sym = dec.synth(frames, L, 1.0, seed=10, tx_out=tx, symbols_out=sym)
    out, met, rs = dec.decode(sym, L, want_metrics=True, out=out)
    ber = int(dec.count_bit_errors(out, tx).item()) / float(frames * L)
    assert 1e-4 < ber < 0.1, ber              # ~1e-2 at 1 dB
    _oracle_subset(oracle, dec, code, "SOFT16", pc, sym, out, met, rs, frames, L, n_pick=4, seed=3)
    # ... and 16 whole frames per usable host CPU through the threaded oracle (256 of the 4096 frames on the box's 16-CPU quota,
    # 12 s of wall clock), compared by digest over all 2.1 M decision words of each; VIT_TEST_K15_ALL=1 checks all 4096 (three
    # minutes; the first two full-suite runs of round 3 did, by accident of counting 256 logical CPUs: all equal)
    n = _slabs_exact_by_digest(oracle, dec, code, "SOFT16", pc, sym, out, met, rs, frames, L,
                               n_slabs=(frames + 1) // 2 if os.environ.get("VIT_TEST_K15_ALL") else 8 * _host_threads(),
                               budget_note="K15 full size, noisy batch")
    assert n >= 16
