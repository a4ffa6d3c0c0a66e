"""Development: random buffers (sizes 0 .. 200 KB; sparse-bitmap-like at several densities, noise, runs, periodic) through the stateless codec seam -
rc_compress on the device, rc_decompress on the device AND the stock decoder (liblz4 / libzstd; blosc: the oracle's from-spec decoder) - for
zstd (levels 0 / 1), LZ4 (levels 0 / 1) and blosc-lz4.  usage: fuzz_seam2.py [cases] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from pyrecode_amd import recode_compressors as rcmp
from oracle import oracle as orc


def run(cases, seed):
    rng = np.random.default_rng(seed)
    bad = 0
    for i in range(cases):
        n = int(rng.choice([0, 1, 7, 511, 512, 513, int(rng.integers(0, 5000)), int(rng.integers(0, 200000))]))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            a = np.packbits(rng.random(8 * n) < rng.choice([0.001, 0.01, 0.02, 0.05, 0.3]), bitorder="little")
        elif kind == 1:
            a = rng.integers(0, 256, n).astype(np.uint8)
        elif kind == 2:
            a = np.repeat(rng.integers(0, 256, n // 37 + 1).astype(np.uint8), 37)[:n]
        elif kind == 3:
            a = np.zeros(n, np.uint8)
            if n:
                a[::int(rng.integers(2, 40))] = rng.integers(1, 256)
        else:
            a = np.where(rng.random(n) < 0.1, rng.integers(1, 256, n), 0).astype(np.uint8)
        data = a.tobytes()
        for scheme, level in ((2, 1), (2, 0), (1, 1), (1, 0), (8, 1)):
            try:
                c = rcmp.compress(scheme, level, data, None)
                ok = rcmp.de_compress(scheme, c, None) == data
                if scheme == 2:
                    ok = ok and orc.lz4f_decode(c, len(data) + 8) == data
                elif scheme == 1:
                    ok = ok and rcmp._zstd_host_decompress(c) == data
                else:
                    ok = ok and orc.blosc1_decode(c) == data
            except Exception as e:   # noqa: BLE001
                ok = False
                print("case", i, "n", n, "kind", kind, "scheme", scheme, "level", level, "raised", repr(e))
            if not ok:
                bad += 1
                print("MISMATCH case", i, "n", n, "kind", kind, "scheme", scheme, "level", level)
    return bad


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    bad = run(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("fuzz done:", cases, "buffers x 5 codec settings,", bad, "failures")
    sys.exit(1 if bad else 0)
