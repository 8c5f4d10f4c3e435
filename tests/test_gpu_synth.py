"""SURVEY 8 f-1 / f-2 on the device: the frame generator (vit_hip_synth_batch), the bit-error counter and the BER sweep driver,
checked against the reference's encoder (oracle/_ref), the C restatement, and the numpy mirror of the generator."""
import numpy as np
import pytest

from viterbidecodercpp_amd import COMMON_CODES, BatchDecoder, synth
from viterbidecodercpp_amd.tools import run_snr_ber
from tests.helpers import make_table_config, oracle_cfg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("code_id", range(8))
@pytest.mark.parametrize("decode_type", ["SOFT16", "HARD8"])
def test_noise_free_frames_equal_the_reference_encoder(oracle, code_id, decode_type):
    code = COMMON_CODES[code_id]
    pc, table, config = make_table_config(code, decode_type)
    dec = BatchDecoder(table, config)
    F, L = 7, 8 * 37
    tx, sym = dec.synth(F, L, None, seed=11, first_frame=3)
    tx, sym = tx.cpu().numpy(), sym.cpu().numpy()
    want_tx, want_sym = synth.make_frames_philox_numpy(code, pc, F, L, None, seed=11, first_frame=3)
    assert np.array_equal(tx, want_tx) and np.array_equal(sym, want_sym)
    # ... and the symbols are what the reference's encoders make of those bytes (encode_data, test_helpers.h:17-64)
    from oracle import pyoracle
    ref = pyoracle.RefLib() if pyoracle.RefLib.available() else None
    for f in range(F):
        bits = oracle.encode(code.K, code.R, code.G, tx[f]).reshape(-1, code.R)
        assert np.array_equal(sym[f], np.where(bits != 0, pc.soft_decision_high, pc.soft_decision_low))
        if ref is not None:
            for which in (0, 1):        # ConvolutionalEncoder_ShiftRegister, ConvolutionalEncoder_Lookup
                assert np.array_equal(ref.encode(code_id, tx[f], which).reshape(-1, code.R), bits)


def test_batch_does_not_depend_on_how_it_is_split():
    code = COMMON_CODES[2]
    pc, table, config = make_table_config(code, "SOFT16")
    dec = BatchDecoder(table, config)
    import torch
    tx, sym = dec.synth(16, 1024, 2.0, seed=5)
    tx_a, sym_a = dec.synth(9, 1024, 2.0, seed=5, first_frame=0)
    tx_b, sym_b = dec.synth(7, 1024, 2.0, seed=5, first_frame=9)
    assert torch.equal(tx, torch.cat([tx_a, tx_b])) and torch.equal(sym, torch.cat([sym_a, sym_b]))
    tx2, sym2 = dec.synth(16, 1024, 2.0, seed=6)
    assert not torch.equal(tx, tx2)


@pytest.mark.parametrize("code_id,decode_type,ebn0", [(2, "SOFT16", 3.0), (2, "HARD8", 5.0), (3, "SOFT8", 2.0), (7, "SOFT16", -4.0),
                                                      (6, "SOFT16", 0.0)])
def test_noisy_frames_follow_the_reference_channel_and_quantiser(code_id, decode_type, ebn0):
    """run_snr_ber.cpp:311-359.  Against the numpy mirror: identical except where the device's float32 log/sincos differ in
    the last place AND the value sits on a rounding boundary; and the first two moments of the noise are the channel's."""
    code = COMMON_CODES[code_id]
    pc, table, config = make_table_config(code, decode_type)
    dec = BatchDecoder(table, config)
    F, L = 64, 2048
    tx, sym = dec.synth(F, L, ebn0, seed=2024)
    tx, sym = tx.cpu().numpy(), sym.cpu().numpy().astype(np.int32)
    want_tx, want_sym = synth.make_frames_philox_numpy(code, pc, F, L, ebn0, seed=2024)
    assert np.array_equal(tx, want_tx)
    diff = np.abs(sym - want_sym.astype(np.int32))
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-3, (diff.max(), (diff != 0).mean())
    # unclamped SOFT16 values: (sym - mean)/scale = +-1 + sigma z.  Estimate sigma from the symbols that are not clamped.
    if decode_type == "SOFT16":
        var = synth.noise_variance(ebn0, code.R)
        scale = 127.0 / np.sqrt(1.0 + var)
        coded = synth.encode_bits_numpy(code.K, code.R, code.G, tx)
        x = sym / scale - (2.0 * coded - 1.0)
        inside = np.abs(sym) < 127
        if inside.mean() > 0.9:      # little clipping: the sample moments are the channel's
            assert abs(x[inside].mean()) < 0.01
            assert abs(x[inside].std() / np.sqrt(var) - 1.0) < 0.03
    assert sym.max() <= pc.soft_decision_high and sym.min() >= pc.soft_decision_low


def test_bit_error_counter():
    import torch
    code = COMMON_CODES[2]
    pc, table, config = make_table_config(code, "SOFT16")
    dec = BatchDecoder(table, config)
    rng = np.random.default_rng(3)
    for n in (1, 15, 16, 17, 4099, 1 << 20):
        a = rng.integers(0, 256, n, dtype=np.uint8)
        b = rng.integers(0, 256, n, dtype=np.uint8)
        cnt = dec.count_bit_errors(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
        cnt = dec.count_bit_errors(torch.from_numpy(a).cuda(), torch.from_numpy(a).cuda(), cnt)     # accumulates
        assert int(cnt.item()) == int(np.unpackbits(a ^ b).sum())
    # unaligned views
    a = torch.from_numpy(rng.integers(0, 256, 1000, dtype=np.uint8)).cuda()
    b = torch.from_numpy(rng.integers(0, 256, 1000, dtype=np.uint8)).cuda()
    want = int(np.unpackbits(a[3:].cpu().numpy() ^ b[3:].cpu().numpy()).sum())
    assert int(dec.count_bit_errors(a[3:], b[3:]).item()) == want


@pytest.mark.parametrize("code_id,decode_type,points", [(2, "SOFT16", (0.0, 2.0)), (2, "HARD8", (2.0, 4.0)), (5, "SOFT16", (0.0, 1.0))])
def test_ber_point_equals_the_oracle_decoding_the_same_frames(oracle, code_id, decode_type, points):
    """tools/run_snr_ber.py: the BER the sweep reports at a point is exactly the BER of the CPU checker on the same frames."""
    code = COMMON_CODES[code_id]
    pc, table, config = make_table_config(code, decode_type)
    dec = BatchDecoder(table, config)
    L, frames = 512 * 8, 96
    for k, ebn0 in enumerate(points):
        seed = 40 + k
        errors, bits, batches = run_snr_ber.ber_point(dec, L, frames, ebn0, seed, max_bits=2 * frames * L, max_error_bits=1 << 40)
        assert batches == 2 and bits == 2 * frames * L
        want_errors = 0
        for b in range(batches):
            tx, sym = dec.synth(frames, L, ebn0, seed=seed, first_frame=b * frames)
            got, _, _ = oracle.decode_frames(code.K, code.R, code.G, oracle_cfg(decode_type, code.R), sym.cpu().numpy(), L, threads=8)
            want_errors += int(np.unpackbits(got ^ tx.cpu().numpy()).sum())
        assert errors == want_errors
        if ebn0 <= 2.0:
            assert errors > 0          # the points are noisy enough to mean something
    # the early-stop rule (run_snr_ber.cpp:372)
    errors, bits, batches = run_snr_ber.ber_point(dec, L, frames, points[0], 99, max_bits=10 * frames * L, max_error_bits=1)
    assert batches == 1 and errors >= 1
