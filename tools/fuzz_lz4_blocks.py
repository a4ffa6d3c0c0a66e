"""Development: random 512-byte blocks of every density and structure through the device LZ4 encoder (the stateless seam), compared block for
block with the serial parse model (tests/lz4_parse_model.py) and decoded by the oracle's decoder.  usage: fuzz_lz4_blocks.py [rounds] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import lz4_parse_model as model
from pyrecode_amd.recode_compressors import device_compress
from oracle import oracle as orc


def run(rounds, seed, quiet=False):
  rng = np.random.default_rng(seed)
  bad = tot = two = 0
  for r in range(rounds):
      blocks = []
      for _ in range(2000):
          kind = rng.integers(0, 6)
          b = np.zeros(512, np.uint8)
          if kind == 0:      # Bernoulli bits
              b = np.packbits(rng.random(4096) < rng.choice([0.002, 0.01, 0.015, 0.02, 0.03, 0.05]), bitorder="little")
          elif kind == 1:    # events of few values at random positions
              k = rng.integers(1, 127)
              b[rng.integers(0, 512, k)] = rng.choice([1, 2, 4, 8, 16, 32, 64, 128, 3, 0x81], k)
          elif kind == 2:    # periodic units with jitter
              p = rng.integers(3, 12)
              pos = np.arange(0, 512, p) + rng.integers(0, 2, len(np.arange(0, 512, p)))
              b[pos[pos < 512]] = 1 << rng.integers(0, 8)
          elif kind == 3:    # clusters: runs of set bits
              bits = np.zeros(4096, bool)
              for s in rng.integers(0, 4090, rng.integers(5, 120)):
                  bits[s:s + rng.integers(1, 5)] = True
              b = np.packbits(bits, bitorder="little")
          elif kind == 4:    # events near the block's end / start
              b[rng.integers(495, 512, 3)] = 0x10; b[rng.integers(0, 8, 2)] = 0x20; b[rng.integers(0, 512, rng.integers(0, 90))] = 1 << rng.integers(0, 8, 1)[0]
          else:              # growing / shrinking gaps of one value
              q, g = 0, rng.integers(1, 6)
              while q < 512:
                  b[q] = 0x40; q += g; g += rng.integers(0, 3)
          blocks.append(b.tobytes())
      buf = b"".join(blocks)
      for level in (1, 0):
          frame = device_compress(2, level, buf)
          got = model.frame_blocks(frame)
          for i, blk in enumerate(blocks):
              want = model.encode_block(blk, level)
              tot += 1
              if level:
                  two += model.EV_MAX < np.count_nonzero(np.frombuffer(blk, np.uint8)) <= model.EV_MAX2
              if got[i] != want:
                  bad += 1
                  if bad < 5:
                      print("MISMATCH round", r, "block", i, "level", level, "events", np.count_nonzero(np.frombuffer(blk, np.uint8)))
          assert orc.lz4f_decode(frame, len(buf) + 8) == buf
      if not quiet:
          print("round", r, "blocks", tot, "in the two-events-per-lane form", two, "mismatches", bad, flush=True)
  return tot, two, bad


if __name__ == "__main__":
    tot, two, bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("fuzz done:", tot, "block encodings,", two, "in the two-events-per-lane form,", bad, "mismatches")
    sys.exit(1 if bad else 0)
