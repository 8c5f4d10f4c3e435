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


# ---- host mirror of the device generator (csrc/kernels_synth.hpp: vit_hip_synth_batch) -------------------------------------
# Same counter-based construction, so a test can regenerate any frame of a device batch: info bytes and noise-free symbols
# exactly; noisy symbols up to the last-ulp differences between the device's and numpy's float32 log / sin / cos.

def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11) on numpy uint32 arrays (broadcast); returns the 4 output words."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) for x in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    M32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = (p1 >> np.uint64(32)) ^ c1 ^ k0, p1 & M32, (p0 >> np.uint64(32)) ^ c3 ^ k1, p0 & M32
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    return tuple(x.astype(np.uint32) for x in (c0, c1, c2, c3))


def philox_info_bytes(frames: int, n_bytes: int, seed: int, first_frame: int = 0) -> np.ndarray:
    """info byte i of frame f = byte (i & 15) of Philox(ctr = {i >> 4, f_lo, 0, f_hi}, key = seed)."""
    f = (np.arange(frames, dtype=np.uint64) + np.uint64(first_frame))[:, None]
    blk = np.arange((n_bytes + 15) // 16, dtype=np.uint64)[None, :]
    x = philox4x32_10(blk, f & np.uint64(0xFFFFFFFF), 0, f >> np.uint64(32), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    words = np.stack(x, axis=-1)                                 # [F][blocks][4] little-endian words
    return np.ascontiguousarray(words).view(np.uint8).reshape(frames, -1)[:, :n_bytes].copy()


def philox_normals(frames: int, n: int, seed: int, first_frame: int = 0) -> np.ndarray:
    """deviate j of frame f: Box-Muller on Philox(ctr = {j >> 2, f_lo, 1, f_hi}); float32 [F][n]."""
    f = (np.arange(frames, dtype=np.uint64) + np.uint64(first_frame))[:, None]
    blk = np.arange((n + 3) // 4, dtype=np.uint64)[None, :]
    x = philox4x32_10(blk, f & np.uint64(0xFFFFFFFF), 1, f >> np.uint64(32), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    two24 = np.float32(5.9604644775390625e-08)
    z = []
    for h in range(2):
        u1 = ((x[2 * h] >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * two24
        u2 = (x[2 * h + 1] >> np.uint32(8)).astype(np.float32) * two24
        r = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
        ang = (np.float32(2.0) * u2).astype(np.float64) * np.pi   # sincospi(2 u2): exact argument, then one rounding
        z.append((r * np.cos(ang).astype(np.float32)).astype(np.float32))
        z.append((r * np.sin(ang).astype(np.float32)).astype(np.float32))
    return np.stack(z, axis=-1).reshape(frames, -1)[:, :n]


def make_frames_philox_numpy(code, cfg, frames: int, L: int, ebn0_db, seed: int = 1, first_frame: int = 0):
    """numpy mirror of BatchDecoder.synth / vit_hip_synth_batch: (tx_bytes [F][L/8], symbols [F][S][R])."""
    assert L % 8 == 0
    K, R = code.K, code.R
    S = L + K - 1
    tx = philox_info_bytes(frames, L // 8, seed, first_frame)
    coded = encode_bits_numpy(K, R, code.G, tx)
    high, low = cfg.soft_decision_high, cfg.soft_decision_low
    if ebn0_db is None:
        return tx, np.where(coded != 0, high, low).astype(cfg.soft_dtype)
    f32 = np.float32
    esn0 = f32(ebn0_db) - f32(10.0) * np.log10(f32(R)).astype(f32)
    var = np.power(f32(10.0), -(esn0 + f32(3.0)) / f32(10.0)).astype(f32)
    sigma = np.sqrt(var).astype(f32)
    mean = (f32(high) + f32(low)) / f32(2.0)
    scale = ((f32(high) - f32(low)) / f32(2.0)) * (f32(1.0) / np.sqrt(f32(1.0) + var).astype(f32))
    z = philox_normals(frames, S * R, seed, first_frame).reshape(frames, S, R)
    noisy = (np.where(coded != 0, f32(1.0), f32(-1.0)).astype(f32) + (sigma * z).astype(f32)).astype(f32)
    v = ((noisy * scale).astype(f32) + mean).astype(f32)
    q = _round_half_away(v)
    return tx, np.clip(q, low, high).astype(cfg.soft_dtype)
