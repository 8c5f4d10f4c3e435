"""Install-time instantiation of the register plan for polynomial sets outside the eight stock codes.

In the reference the generator polynomials are a run-time constructor argument (include/viterbi/viterbi_branch_table.h:34-55):
any set runs at the full speed of the compiled <K,R>.  The register plan's kernels are specialised per set, so a set that is not
built into libvit_hip.so is compiled AHEAD of use, on a host that has hipcc (no GPU needed), into the package cache
`viterbidecodercpp_amd/precompiled/`; `vit_hip_create` finds it there and runs PLAN_REG on hosts without any compiler.

    python -m viterbidecodercpp_amd.tools.precompile 7 2 0o171 0o133            # one set, both symbol widths
    python -m viterbidecodercpp_amd.tools.precompile --all-common               # the list below (what build() runs)
    python -m viterbidecodercpp_amd.tools.precompile --list codes.txt --dir D   # "K R G0 G1 ... [s8|s16]" per line, into D

A thin front end of the C ABI's vit_hip_precompile (include/vit_hip.h); objects that exist are kept, so a second run is free.
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

# widely deployed sets that are not among the reference's eight stock codes (examples/helpers/common_codes.h:20-30 holds
# Voyager's 0o155/0o117 orientation of the K = 7 pair, LTE's 0o133/0o171/0o165 order and IS-95A's 0o753/0o561 order)
COMMON_SETS = [
    ("802.11 / CCSDS / DVB-S K=7 R=1/2", 7, 2, (0o171, 0o133)),
    ("CCSDS K=7 R=1/2 (G1, G2 order)", 7, 2, (0o133, 0o171)),
    ("K=7 R=1/3 (0o171, 0o165, 0o133)", 7, 3, (0o171, 0o165, 0o133)),
    ("GSM K=5 R=1/2", 5, 2, (0o23, 0o33)),
    ("UMTS / LTE-CC K=9 R=1/2", 9, 2, (0o561, 0o753)),
    ("UMTS K=9 R=1/3", 9, 3, (0o557, 0o663, 0o711)),
    # the textbook maximum-free-distance sets below K = 7 (K >= 7: the GENERIC kernels serve whatever comes): seconds of hipcc each
    ("best K=3 R=1/2", 3, 2, (0o5, 0o7)), ("best K=3 R=1/3", 3, 3, (0o5, 0o7, 0o7)),
    ("best K=4 R=1/2", 4, 2, (0o15, 0o17)), ("best K=4 R=1/3", 4, 3, (0o13, 0o15, 0o17)),
    ("best K=5 R=1/2", 5, 2, (0o23, 0o35)), ("best K=5 R=1/3", 5, 3, (0o25, 0o33, 0o37)),
    ("best K=6 R=1/2", 6, 2, (0o53, 0o75)), ("IS-54 / IS-136 K=6 R=1/2", 6, 2, (0o65, 0o57)),
    # all polynomials zero: the GENERIC kernels of (K, R) -- polynomials read at run time, one code object for every set (what
    # vit_hip_create falls back to when neither the library nor this cache holds kernels specialised for a set)
]
COMMON_SETS += [(f"GENERIC K={K} R=1/{R} (any polynomials)", K, R, (0,) * R) for K in (3, 4, 5, 6, 7, 8, 9) for R in (1, 2, 3, 4) if not (K == 6 and R % 2)]


def parse_list(path):
    """one set per line: K R G0 .. G(R-1) [s8|s16]; numbers in any Python literal base; '#' starts a comment"""
    out = []
    for raw in open(path):
        line = raw.split("#", 1)[0].split()
        if not line:
            continue
        widths = (1, 2)
        if line[-1] in ("s8", "s16"):
            widths = (1,) if line[-1] == "s8" else (2,)
            line = line[:-1]
        K, R = int(line[0], 0), int(line[1], 0)
        G = tuple(int(x, 0) for x in line[2:])
        if len(G) != R:
            raise ValueError(f"{path}: {raw.strip()!r}: {R} polynomials expected")
        for w in widths:
            out.append((K, R, G, w))
    return out


def precompile(sets, directory=None, jobs=None, verbose=True):
    """sets: iterable of (K, R, G, soft_bytes).  Returns the list of code-object paths; raises on the first failure."""
    from viterbidecodercpp_amd import _lib

    lib = _lib.load()
    sets = list(dict.fromkeys(sets))
    jobs = jobs or max(1, min(len(sets), len(os.sched_getaffinity(0))))
    d = None if directory is None else os.fsencode(os.path.abspath(directory))
    if directory is not None:
        os.makedirs(directory, exist_ok=True)

    def one(item):
        K, R, G, w = item
        arr = (C.c_uint32 * 6)(*G)
        buf = C.create_string_buffer(4096)
        t0 = time.time()
        rc = lib.vit_hip_precompile(K, R, arr, w, d, buf, len(buf))
        if rc != _lib.OK:
            raise RuntimeError(f"vit_hip_precompile(K={K}, R={R}, G={[oct(g) for g in G]}, soft_bytes={w}) failed: "
                               f"{lib.vit_hip_last_error().decode()}")
        if verbose:
            print(f"  K={K} R={R} {'/'.join(oct(g) for g in G)} s{8 * w}: {os.path.basename(buf.value.decode())} ({time.time() - t0:.0f} s)",
                  flush=True)
        return buf.value.decode()

    with ThreadPoolExecutor(max_workers=jobs) as ex:        # the compiler runs in child processes; ctypes drops the GIL
        return list(ex.map(one, sets))


def common_sets():
    return [(K, R, G, w) for _, K, R, G in COMMON_SETS for w in (2, 1)]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("code", nargs="*", help="K R G0 G1 ... (octal as 0o171)")
    ap.add_argument("--all-common", action="store_true", help="the common non-stock sets: " + "; ".join(n for n, *_ in COMMON_SETS))
    ap.add_argument("--list", help="file with one set per line")
    ap.add_argument("--soft-bytes", type=int, choices=[1, 2], default=None, help="one symbol width only (default: both)")
    ap.add_argument("--dir", default=None, help="target directory (default: the package cache beside libvit_hip.so)")
    ap.add_argument("--jobs", type=int, default=None)
    args = ap.parse_args(argv)
    widths = (args.soft_bytes,) if args.soft_bytes else (2, 1)
    sets = []
    if args.code:
        K, R = int(args.code[0], 0), int(args.code[1], 0)
        G = tuple(int(x, 0) for x in args.code[2:])
        if len(G) != R:
            ap.error(f"{R} polynomials expected, got {len(G)}")
        sets += [(K, R, G, w) for w in widths]
    if args.all_common:
        sets += [s for s in common_sets() if s[3] in widths]
    if args.list:
        sets += [s for s in parse_list(args.list) if s[3] in widths]
    if not sets:
        ap.error("nothing to compile: give K R G..., --all-common or --list")
    paths = precompile(sets, args.dir, args.jobs)
    print(f"{len(paths)} code object(s) in {os.path.dirname(paths[0])}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
