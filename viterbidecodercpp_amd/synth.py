"""Synthetic AWGN frames: encoder + noise + soft-decision quantiser.

Follows the reference's BER harness so inputs have the same statistics:
  bits     : MSB-first input bits, K-1 zero tail bits, symbols step-major / polynomial-minor
             (examples/helpers/test_helpers.h:17-64, include/viterbi/convolutional_encoder_shift_register.h:42-62)
  channel  : +-1.0 BPSK + N(0, sigma^2), sigma^2 = 10^(-(EbN0_dB - 10 log10 R + 3)/10)   (examples/run_snr_ber.cpp:319-325)
  quantise : clamp(round(x * (high-low)/2 / sqrt(1+sigma^2) + (high+low)/2), low, high)     (examples/run_snr_ber.cpp:352-359)

Two back ends with the same signature: numpy (deterministic, for tests) and torch (any device; used by bench.py to build the
full-size batch directly in HBM).  This is data synthesis, not part of the decode path.
"""
from __future__ import annotations

import math

import numpy as np


def noise_variance(ebn0_db: float, R: int) -> float:
    esn0_db = ebn0_db - 10.0 * math.log10(float(R))
    return float(10.0 ** (-(esn0_db + 3.0) / 10.0))


def encode_bits_numpy(K: int, R: int, G, data: np.ndarray) -> np.ndarray:
    """data [F][n_bytes] uint8 -> coded bits [F][S][R] uint8 (0/1), S = 8*n_bytes + K-1."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    if data.ndim == 1:
        data = data[None, :]
    F = data.shape[0]
    bits = np.unpackbits(data, axis=1)                       # MSB first
    L = bits.shape[1]
    S = L + K - 1
    x = np.zeros((F, S + K - 1), dtype=np.uint8)             # K-1 zeros of history + data + K-1 tail zeros
    x[:, K - 1:K - 1 + L] = bits
    out = np.zeros((F, S, R), dtype=np.uint8)
    for i in range(R):
        acc = np.zeros((F, S), dtype=np.uint8)
        for k in range(K):                                   # register bit k holds the input bit from k steps ago
            if (int(G[i]) >> k) & 1:
                acc ^= x[:, K - 1 - k:K - 1 - k + S]
        out[:, :, i] = acc
    return out


def _round_half_away(x):
    return np.sign(x) * np.floor(np.abs(x) + np.float32(0.5))


def quantise_numpy(coded_bits: np.ndarray, high: int, low: int, ebn0_db, R: int, rng, dtype) -> np.ndarray:
    """coded bits (0/1) -> noisy soft symbols.  ebn0_db=None gives noise-free symbols at exactly high/low."""
    if ebn0_db is None:
        return np.where(coded_bits != 0, high, low).astype(dtype)
    var = np.float32(noise_variance(ebn0_db, R))
    x = coded_bits.astype(np.float32) * np.float32(2.0) - np.float32(1.0)
    x = x + rng.standard_normal(size=x.shape, dtype=np.float32) * np.float32(math.sqrt(var))
    mean = np.float32((high + low) / 2.0)
    mag = np.float32((high - low) / 2.0) / np.float32(math.sqrt(1.0 + var))
    q = _round_half_away(x * mag + mean)
    return np.clip(q, low, high).astype(dtype)


def make_frames_numpy(code, cfg, frames: int, L: int, ebn0_db, seed: int = 1):
    """returns (tx_bytes [F][L/8] uint8, symbols [F][S][R] soft dtype)."""
    assert L % 8 == 0
    rng = np.random.default_rng(seed)
    data = rng.integers(0, 256, size=(frames, L // 8), dtype=np.uint8)
    coded = encode_bits_numpy(code.K, code.R, code.G, data)
    sym = quantise_numpy(coded, cfg.soft_decision_high, cfg.soft_decision_low, ebn0_db, code.R, rng, cfg.soft_dtype)
    return data, sym


def make_frames_torch(code, cfg, frames: int, L: int, ebn0_db, seed: int = 1, device="cuda", chunk: int = 4096):
    """torch back end: builds the batch on `device` in chunks of `chunk` frames.
    returns (tx_bytes [F][L/8] uint8, symbols [F][S][R] int16|int8), both on `device`."""
    import torch

    assert L % 8 == 0
    K, R, G = code.K, code.R, code.G
    S = L + K - 1
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    tdtype = torch.int16 if cfg.soft_bytes == 2 else torch.int8
    high, low = cfg.soft_decision_high, cfg.soft_decision_low
    tx = torch.empty((frames, L // 8), dtype=torch.uint8, device=device)
    sym = torch.empty((frames, S, R), dtype=tdtype, device=device)
    weights = (2 ** torch.arange(7, -1, -1, device=device)).to(torch.int32)
    for f0 in range(0, frames, chunk):
        f1 = min(frames, f0 + chunk)
        n = f1 - f0
        bits = torch.randint(0, 2, (n, L), dtype=torch.uint8, device=device, generator=gen)
        tx[f0:f1] = (bits.view(n, L // 8, 8).to(torch.int32) * weights).sum(dim=2).to(torch.uint8)
        x = torch.zeros((n, S + K - 1), dtype=torch.uint8, device=device)
        x[:, K - 1:K - 1 + L] = bits
        for i in range(R):
            acc = torch.zeros((n, S), dtype=torch.uint8, device=device)
            for k in range(K):
                if (int(G[i]) >> k) & 1:
                    acc ^= x[:, K - 1 - k:K - 1 - k + S]
            if ebn0_db is None:
                sym[f0:f1, :, i] = torch.where(acc != 0, high, low).to(tdtype)
            else:
                var = noise_variance(ebn0_db, R)
                v = acc.to(torch.float32) * 2.0 - 1.0
                v += torch.randn(v.shape, dtype=torch.float32, device=device, generator=gen) * math.sqrt(var)
                v = v * ((high - low) / 2.0 / math.sqrt(1.0 + var)) + (high + low) / 2.0
                v = torch.sign(v) * torch.floor(torch.abs(v) + 0.5)
                sym[f0:f1, :, i] = torch.clamp(v, low, high).to(tdtype)
    return tx, sym
