#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference compiled in place: oracle/_ref/libvitref.so via oracle/ref_shim.cpp).
For every case it stores the inputs (soft symbols) and what the reference's ViterbiDecoder_Scalar + ViterbiDecoder_Core
produced: decision words, final metrics, update()'s return value, get_error(end_state), chainback bytes.  Each case is also
run through S incremental update() calls of R symbols (the streaming pattern of examples/helpers/puncture_code_helpers.h:51)
and must give identical results before it is written.  Where the AVX2 strategy is valid its decision words are stored as
well: they DIFFER from the scalar ones (tie rule) and serve as a negative control.

    python tests/golden/make_golden.py          # adds the cases that have no fixture yet
    python tests/golden/make_golden.py --force  # rewrites every tests/golden/*.npz and MANIFEST.json
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402
from viterbidecodercpp_amd import COMMON_CODES, get_decoding_config, synth  # noqa: E402

# (name, code id, decode type, L, Eb/N0 dB or None, seed, start_state, end_state)
CASES = [
    ("k3r2_soft16_clean", 0, "SOFT16", 256, None, 1, 0, 0),
    ("k5r2_soft16_3db", 1, "SOFT16", 512, 3.0, 2, 0, 0),
    ("k7r2_soft16_clean", 2, "SOFT16", 1024, None, 3, 0, 0),
    ("k7r2_soft16_2db", 2, "SOFT16", 2048, 2.0, 4, 0, 0),
    ("k7r2_soft16_0db", 2, "SOFT16", 2048, 0.0, 5, 0, 0),
    ("k7r2_soft16_states", 2, "SOFT16", 100, 3.0, 6, 37, 21),   # L % 8 != 0, non-zero start and end states
    ("k7r2_hard8_4db", 2, "HARD8", 2048, 4.0, 7, 0, 0),
    ("k7r2_soft8_3db", 2, "SOFT8", 1024, 3.0, 8, 0, 0),
    ("k7r3_soft8_3db", 3, "SOFT8", 512, 3.0, 9, 0, 0),
    ("k7r4_soft16_1db", 4, "SOFT16", 512, 1.0, 10, 0, 0),
    ("k9r2_soft16_2db", 5, "SOFT16", 1024, 2.0, 11, 0, 0),
    ("k9r4_hard8_3db", 6, "HARD8", 512, 3.0, 12, 0, 0),
    ("k15r6_soft16_m4db", 7, "SOFT16", 64, -4.0, 13, 0, 0),
    ("k15r6_soft8_clean_overflow", 7, "SOFT8", 64, None, 14, 0, 0),  # the reference's own skipped case (run_tests.cpp:63-65)
    # SURVEY 8(c) sizes: config 1 is "1 frame of 4096 info bits"
    ("k7r2_soft16_2db_l4096", 2, "SOFT16", 4096, 2.0, 21, 0, 0),
    ("k7r2_soft16_0db_l4096", 2, "SOFT16", 4096, 0.0, 22, 0, 0),
    ("k7r2_hard8_3db_l4096", 2, "HARD8", 4096, 3.0, 23, 0, 0),
    ("k9r2_soft16_2db_l4096", 5, "SOFT16", 4096, 2.0, 24, 0, 0),
    ("k15r6_soft16_0db_l256", 7, "SOFT16", 256, 0.0, 25, 0, 0),    # decision words (553 KB) kept as per-row SHA-256 prefixes
]
HASHED = {"k15r6_soft16_0db_l256"}


def row_hashes(decisions):
    """first 8 bytes of SHA-256 of every decision row: pins all words at 1/256 of the size, and still says WHICH step differs"""
    return np.asarray([np.frombuffer(hashlib.sha256(np.ascontiguousarray(r).tobytes()).digest()[:8], dtype=np.uint64)[0]
                       for r in decisions], dtype=np.uint64)
DT = {"SOFT16": pyoracle.SOFT16, "SOFT8": pyoracle.SOFT8, "HARD8": pyoracle.HARD8}


def main():
    pyoracle.ensure_built()
    assert pyoracle.RefLib.available(), "needs oracle/_ref/libvitref.so (the real reference)"
    ref = pyoracle.RefLib()
    mpath = os.path.join(HERE, "MANIFEST.json")
    manifest = json.load(open(mpath)) if os.path.exists(mpath) else {}
    force = "--force" in sys.argv
    for name, cid, dt, L, ebn0, seed, ss, es in CASES:
        if not force and name in manifest and os.path.exists(os.path.join(HERE, name + ".npz")):
            continue        # fixtures are committed data: only new cases are generated unless --force
        code = COMMON_CODES[cid]
        pc = get_decoding_config(dt, code.R)
        ocfg = ref.stock_config(DT[dt], code.R)
        tx, sym = synth.make_frames_numpy(code, pc, 1, ((L + 7) // 8) * 8, ebn0, seed=seed)
        S = L + code.K - 1
        sym = np.ascontiguousarray(sym[0, :S])
        one = ref.run(cid, ocfg, sym, L, simd=pyoracle.SCALAR, start_state=ss, end_state=es)
        inc = ref.run(cid, ocfg, sym, L, simd=pyoracle.SCALAR, start_state=ss, end_state=es, chunk_steps=1)
        for k in ("decisions", "metrics", "bytes"):
            assert np.array_equal(one[k], inc[k]), (name, k)
        assert one["renorm_sum"] == inc["renorm_sum"] and one["error"] == inc["error"]
        arrays = dict(symbols=sym, tx_bytes=tx[0], decisions=one["decisions"], metrics=one["metrics"],
                      renorm_sum=np.uint64(one["renorm_sum"]), error=np.uint32(one["error"]), bytes=one["bytes"],
                      table=ref.branch_table(cid, ocfg.soft_bytes, ocfg.high, ocfg.low), config=np.asarray(ocfg.cfg4, np.uint32))
        if ref.is_valid(cid, ocfg.soft_bytes, pyoracle.SIMD_AVX):
            avx = ref.run(cid, ocfg, sym, L, simd=pyoracle.SIMD_AVX, start_state=ss, end_state=es)
            arrays["avx_decisions"] = avx["decisions"]
            arrays["avx_bytes"] = avx["bytes"]
            arrays["avx_metrics"] = avx["metrics"]
        differing = int((arrays["avx_decisions"] != one["decisions"]).sum()) if "avx_decisions" in arrays else None
        if name in HASHED:
            arrays["decision_row_sha256_8"] = row_hashes(one["decisions"])
            arrays["decisions_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(one["decisions"]).tobytes()).digest(), dtype=np.uint8)
            del arrays["decisions"]
            arrays.pop("avx_decisions", None)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **arrays)
        manifest[name] = dict(code=cid, code_name=code.name, K=code.K, R=code.R, G=list(code.G), decode_type=dt, L=L,
                              ebn0_db=ebn0, seed=seed, start_state=ss, end_state=es, steps=S,
                              bit_errors=int(np.unpackbits(one["bytes"] ^ tx[0][:len(one["bytes"])])[:L].sum()),
                              renorm_sum=int(one["renorm_sum"]),
                              avx_decision_words_differing=differing,
                              sha256=hashlib.sha256(open(path, "rb").read()).hexdigest())
        print(name, manifest[name]["bit_errors"], manifest[name]["renorm_sum"], manifest[name]["avx_decision_words_differing"])
    json.dump(manifest, open(mpath, "w"), indent=1)


if __name__ == "__main__":
    main()
