#!/usr/bin/env python3
"""update()/chainback() timing per code and decode type: the counterpart of the reference's examples/run_benchmark.cpp
(SURVEY section 8 f-3).

Same protocol (run_benchmark.cpp:250-284: reset untimed, update() timed, chainback() timed, repeated for T seconds) and the
same JSON record (run_benchmark.cpp:297-327) so that examples/parse_benchmark.py reads the file once `SIMD_HIP = 4` is added
to its SimdType enum: name, decode_type, simd_type, K, R, G, total_input_bits, total_symbols, update_symbols_ns[],
chainback_bits_ns[].  One "sample" is one batch of `frames` noise-free frames of 2048 bits (the reference's frame size),
timed with HIP events around each kernel; total_* are per batch, so the parser's symbols/s and bits/s are whole-GPU rates.

    python -m viterbidecodercpp_amd.tools.run_benchmark -T 0.5 > bench_hip.json
"""
import argparse
import json
import sys
import time


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--codes", type=int, nargs="+", default=list(range(8)))
    ap.add_argument("--decode-types", nargs="+", default=["SOFT16", "SOFT8", "HARD8"])
    ap.add_argument("-T", "--seconds", type=float, default=0.5, help="sampling time per (code, decode type)")
    ap.add_argument("--bytes", type=int, default=256, help="frame size in bytes (reference: 256)")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args()
    import torch
    from viterbidecodercpp_amd import (COMMON_CODES, BatchDecoder, ViterbiBranchTable, ViterbiDecoder_Config,
                                       get_decoding_config, synth)

    L = args.bytes * 8
    records = []
    for c in args.codes:
        code = COMMON_CODES[c]
        frames = 65536 if code.K <= 7 else 16384 if code.K <= 9 else 512
        for dt in args.decode_types:
            pc = get_decoding_config(dt, code.R)
            table = ViterbiBranchTable(code.K, code.R, code.G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
            dec = BatchDecoder(table, ViterbiDecoder_Config.from_decoder_config(pc), device=args.device)
            tx, sym = synth.make_frames_torch(code, pc, frames, L, None, seed=c, device=dec.device)
            out = torch.empty((frames, L // 8), dtype=torch.uint8, device=dec.device)
            upd_ns, cb_ns = [], []
            t_end = time.perf_counter() + args.seconds
            first = True
            while first or time.perf_counter() < t_end:
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
                dec.update(sym, L, want_metrics=False)
                e1.record()
                dec.chainback(frames, L, out=out)
                e2.record()
                torch.cuda.synchronize()
                if not first:                      # first pass = warm-up
                    upd_ns.append(e0.elapsed_time(e1) * 1e6)
                    cb_ns.append(e1.elapsed_time(e2) * 1e6)
                first = False
            # scalar semantics with 8-bit soft metrics overflow at K=15 R=6 -- the reference skips that case too
            # (examples/run_tests.cpp:63-65); everything else must round-trip exactly
            if not (code.K == 15 and dt == "SOFT8"):
                assert torch.equal(out, tx), f"noise-free frames must decode exactly ({code.name} {dt})"
            S = L + code.K - 1
            records.append({"name": code.name, "decode_type": dt, "simd_type": "SIMD_HIP", "K": code.K, "R": code.R,
                            "G": list(code.G), "frames": frames, "total_input_bits": frames * L,
                            "total_symbols": frames * S * code.R, "update_symbols_ns": upd_ns, "chainback_bits_ns": cb_ns})
            print(f"{code.name:16s} {dt:6s} frames={frames:6d} update {frames*S*code.R/ (sum(upd_ns)/len(upd_ns)) :8.2f} Gsym/s "
                  f"chainback {frames*L/(sum(cb_ns)/len(cb_ns)):8.2f} Gbit/s", file=sys.stderr)
    json.dump(records, sys.stdout)
    print()


if __name__ == "__main__":
    main()
