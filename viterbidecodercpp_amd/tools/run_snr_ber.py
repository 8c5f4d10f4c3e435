#!/usr/bin/env python3
"""BER-versus-Eb/N0 sweep on the GPU: the counterpart of the reference's examples/run_snr_ber.cpp.

Same experiment, same stopping rules, same JSON record as the reference writes (run_snr_ber.cpp:419-441: name, decode_type,
simd_type, K, R, G, EbNo_dB[], ber[]) so examples/plot_snr_ber.py reads it unchanged; simd_type is "SIMD_HIP".
  * channel / quantiser : vit_hip_synth_batch, one HIP kernel (run_snr_ber.cpp:319-359), generated directly in HBM;
                          bit errors counted on the device (vit_hip_count_bit_errors)
  * sweep               : Eb/N0 from 0.0 dB in 0.5 dB steps (:229-230); a point stops at `max_error_bits` errors or at
                          1e9 / (R * 2^(K-1)) generated bits (:225-231); the sweep stops at the first error-free point
Every decoded bit comes from the HIP update()+chainback() kernels; the host only applies the stopping rules.

    python -m viterbidecodercpp_amd.tools.run_snr_ber --codes 2 5 --decode-types SOFT16 HARD8 > ber.json
"""
import argparse
import json
import math
import sys


def ber_point(dec, L, frames, ebn0, seed, max_bits, max_error_bits):
    """one Eb/N0 point (run_snr_ber.cpp:336-383): batches of `frames` blocks of L bits -- batch k is
    dec.synth(frames, L, ebn0, seed, first_frame=k*frames) -- are generated, decoded and compared on the device until
    `max_bits` bits or `max_error_bits` errors; returns (errors, bits, batches)."""
    import torch

    count = torch.zeros(1, dtype=torch.int64, device=dec.device)
    tx = sym = out = None
    bits = batches = 0
    while True:
        tx, sym = dec.synth(frames, L, ebn0, seed=seed, first_frame=batches * frames, tx_out=tx, symbols_out=sym)
        out = dec.decode(sym, L, out=out)
        dec.count_bit_errors(out, tx, count)
        bits += frames * L
        batches += 1
        errors = int(count.item())          # the stopping rule needs the count on the host, as in the reference
        if bits >= max_bits or errors >= max_error_bits:
            return errors, bits, batches


def sweep(code, decode_type, args, torch):
    from viterbidecodercpp_amd import BatchDecoder, ViterbiBranchTable, ViterbiDecoder_Config, get_decoding_config

    pc = get_decoding_config(decode_type, code.R)
    table = ViterbiBranchTable(code.K, code.R, code.G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
    dec = BatchDecoder(table, ViterbiDecoder_Config.from_decoder_config(pc), device=args.device)
    L = args.block_bytes * 8
    max_bits = int(math.ceil(args.bits_scale * (1e9 / (code.R * (1 << (code.K - 1))))))
    frames = max(32, min(args.max_frames, (max_bits + L - 1) // L))
    ebn0s, bers = [], []
    for point in range(args.max_points + 1):
        ebn0 = args.ebn0_initial + point * args.ebn0_step
        errors, bits, _ = ber_point(dec, L, frames, ebn0, args.seed + 1000 * point, max_bits, args.max_error_bits)
        ber = errors / float(bits)
        ebn0s.append(round(ebn0, 1))
        bers.append(ber)
        print(f"name='{code.name}',K={code.K},R={code.R},decode={decode_type},simd=SIMD_HIP,iter={point},"
              f"EbNo_dB={ebn0:.1f},BER={ber:.3e},bits={bits}", file=sys.stderr)
        if errors == 0:
            break
    return {"name": code.name, "decode_type": decode_type, "simd_type": "SIMD_HIP", "K": code.K, "R": code.R,
            "G": list(code.G), "EbNo_dB": ebn0s, "ber": [float(f"{b:.3e}") for b in bers]}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--codes", type=int, nargs="+", default=[2], help="indices into COMMON_CODES")
    ap.add_argument("--decode-types", nargs="+", default=["SOFT16", "SOFT8", "HARD8"])
    ap.add_argument("--block-bytes", type=int, default=512, help="traceback length in bytes (reference default 512)")
    ap.add_argument("--max-error-bits", type=int, default=1024)
    ap.add_argument("--max-points", type=int, default=30)
    ap.add_argument("--bits-scale", type=float, default=1.0, help="scale on the reference's 1e9/(R*2^(K-1)) bits per point")
    ap.add_argument("--max-frames", type=int, default=16384, help="frames per decode call")
    ap.add_argument("--ebn0-initial", type=float, default=0.0)
    ap.add_argument("--ebn0-step", type=float, default=0.5)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args()
    import torch
    from viterbidecodercpp_amd import COMMON_CODES

    records = [sweep(COMMON_CODES[c], dt, args, torch) for c in args.codes for dt in args.decode_types]
    json.dump(records, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
