"""Pixel / row pitches (bytes) of the bf16 plane images in LDS that keep the transposed operand reads
(ds_read_b64_tr_b16) of the bf16 x6 weight-gradient kernels free of bank conflicts.
A read: per 16-lane group g a block of 4 rows x 16 columns; lane 4 q + p4 supplies the address of row q, columns
4 p4 ..; banks = (addr / 4) % 64, conflicts counted per 32-lane half.  Contraction slot k = 8 g + 4 r + q of K step ks
holds pixel 32 ks + 16 r + 4 g + q (any permutation works as long as both operands use it): a half then reads 8
CONSECUTIVE pixels."""
import itertools


def conflicts(addrs):
  banks = {}
  for a in addrs:
    for b in ((a // 4) % 64, (a // 4 + 1) % 64):
      banks.setdefault(b, set()).add(a)
  return max(len(v) for v in banks.values())


def pixel(ks, g, r, q):
  return 32 * ks + 16 * r + 4 * g + q


def layer(name, S, IW, OW, OHW, KH, KW, IC, pitches, rowpads):
  steps = (OHW + 31) // 32
  best = []
  for PX, pad in itertools.product(pitches, rowpads):
    RP = IW * PX + pad
    worst, total, n = 0, 0, 0
    for ks, r, half in itertools.product(range(steps), range(2), range(2)):
      for kh, kw in itertools.product(range(KH), range(KW)):
        for c0 in range(0, IC, 16):
          addrs = []
          for g in (2 * half, 2 * half + 1):
            for q in range(4):
              p = min(pixel(ks, g, r, q), OHW - 1)
              py, px = divmod(p, OW)
              for p4 in range(4):
                addrs.append((S * py + kh) * RP + (S * px + kw) * PX + (c0 + 4 * p4) * 2)
          c = conflicts(addrs)
          worst = max(worst, c); total += c; n += 1
    best.append((total / n, worst, PX, pad))
  for mean, worst, PX, pad in sorted(best)[:6]:
    print(f"{name} X pixel pitch {PX} row pad {pad}: worst {worst}-way, mean {mean:.3f}")


def grows(name, OHW, pitches):
  steps = (OHW + 31) // 32
  for PG in pitches:
    worst, total, n = 0, 0, 0
    for ks, r, half, a in itertools.product(range(steps), range(2), range(2), range(4)):
      addrs = []
      for g in (2 * half, 2 * half + 1):
        for q in range(4):
          for p4 in range(4):
            addrs.append(pixel(ks, g, r, q) * PG + (16 * a + 4 * p4) * 2)
      c = conflicts(addrs)
      worst = max(worst, c); total += c; n += 1
    print(f"{name} G pitch {PG}: worst {worst}-way, mean {total / n:.3f}")


layer("conv1", 2, 20, 9, 81, 4, 4, 32, range(64, 113, 8), range(0, 257, 8))
grows("conv1/2", 81, range(128, 201, 8))
layer("conv2", 1, 9, 7, 49, 3, 3, 64, range(128, 201, 8), range(0, 257, 8))
