"""Parity at BASELINE.json's full sizes through size-independent properties: encode -> AWGN -> decode round trips, the two
kernel plans agreeing byte for byte, and the oracle checking a random subset of frames exactly."""
import numpy as np
import pytest

from viterbidecodercpp_amd import COMMON_CODES, BatchDecoder, _lib, get_decoding_config, synth
from tests.helpers import make_table_config, oracle_cfg

pytestmark = pytest.mark.gpu


def _bit_errors(torch, a, b):
    lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=a.device)
    return int(lut[torch.bitwise_xor(a, b).long()].sum().item())


@pytest.mark.parametrize("code_id,decode_type,frames,L,ebn0,ber_max", [
    (2, "SOFT16", 65536, 8192, 3.0, 2e-3),    # BASELINE configs[1]: K=7 R=1/2 u16, 64k frames x 8192 bits
    (5, "SOFT16", 16384, 8192, 3.0, 1e-3),    # configs[2] (K=9 R=1/2 u16), quarter batch to bound test time
    (2, "HARD8", 32768, 8192, 5.0, 1e-3),     # configs[3]: one GPU's share (32768 frames) of the 8-GPU hard-decision run
])
def test_full_size_round_trip_and_subset_exact(oracle, code_id, decode_type, frames, L, ebn0, ber_max):
    import torch

    code = COMMON_CODES[code_id]
    pc, table, config = make_table_config(code, decode_type)
    dec = BatchDecoder(table, config)
    # noise-free: every frame must decode to exactly what was sent (the reference's own test property, at full size)
    tx, sym = synth.make_frames_torch(code, pc, frames, L, None, seed=5, device="cuda")
    out = dec.decode(sym, L)
    assert torch.equal(out, tx)
    del sym
    # noisy: BER in the expected range, and a random subset of frames bit-exact against the oracle
    tx, sym = synth.make_frames_torch(code, pc, frames, L, ebn0, seed=6, device="cuda")
    out, met, rs = dec.decode(sym, L, want_metrics=True)
    ber = _bit_errors(torch, out, tx) / float(frames * L)
    assert 0 < ber < ber_max, ber
    rng = np.random.default_rng(1)
    pick = np.sort(rng.choice(frames, size=24, replace=False))
    pick[0], pick[-1] = 0, frames - 1
    idx = torch.from_numpy(pick).cuda()
    sub = sym[idx].cpu().numpy()
    want, want_met, want_rs = oracle.decode_frames(code.K, code.R, code.G, oracle_cfg(decode_type, code.R), sub, L,
                                                   threads=8, want_metrics=True)
    assert np.array_equal(out[idx].cpu().numpy(), want)
    got_met = met[idx].cpu().numpy()
    got_met = got_met.view(np.uint16) if pc.error_bytes == 2 else got_met
    assert np.array_equal(got_met.astype(np.uint32), want_met)
    assert np.array_equal(rs[idx].cpu().numpy().astype(np.uint64), want_rs)
    # the LDS plan (different kernels, different decision layout) must give the same bytes on a slice
    n = 2048
    lds = BatchDecoder(table, config, plan=_lib.PLAN_LDS)
    assert torch.equal(lds.decode(sym[:n].contiguous(), L), out[:n])


def test_cassini_k15_batch(oracle):
    """configs[4]: K=15 R=1/6 u16, 16384 states in LDS (2 x 32 KiB metric buffers per workgroup)."""
    import torch

    code = COMMON_CODES[7]
    pc, table, config = make_table_config(code, "SOFT16")
    frames, L = 256, 1024
    dec = BatchDecoder(table, config)
    tx, sym = synth.make_frames_torch(code, pc, frames, L, 1.0, seed=9, device="cuda")
    out, met, rs = dec.decode(sym, L, want_metrics=True)
    ber = _bit_errors(torch, out, tx) / float(frames * L)
    assert ber < 5e-2
    pick = [0, 100, 255]
    sub = sym[pick].cpu().numpy()
    want, want_met, want_rs = oracle.decode_frames(code.K, code.R, code.G, oracle_cfg("SOFT16", code.R), sub, L, threads=3,
                                                   want_metrics=True)
    assert np.array_equal(out[pick].cpu().numpy(), want)
    assert np.array_equal(rs[pick].cpu().numpy().astype(np.uint64), want_rs)
