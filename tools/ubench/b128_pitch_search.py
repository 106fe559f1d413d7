"""Pixel / row pitches of a bf16 plane image in LDS for conflict-free ds_read_b128 operand reads: lane (n16 = pixel
16 mt + n16 of a raster of W-wide rows, kq = 16-byte k group).  The instruction serves four 16-lane groups, one LDS
cycle each when the 16 lanes hit 16 distinct 16-byte bank quads (MI355X_MICROARCH.md, LDS)."""
import itertools
import sys

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def cycles(addr):  # addr[lane] -> LDS cycles of one ds_read_b128
  total = 0
  for grp in GROUPS:
    quads = {}
    for l in grp:
      quads.setdefault((addr[l] // 16) % 16, set()).add(addr[l])
    total += max(len(v) for v in quads.values())
  return total


def search(name, W, NPIX, IMG_W, stride, pitches, pads):
  tiles = (NPIX + 15) // 16
  out = []
  for PX, pad in itertools.product(pitches, pads):
    RP = IMG_W * PX + pad
    tot = 0
    for mt in range(tiles):
      addr = []
      for lane in range(64):
        p = min(16 * mt + (lane & 15), NPIX - 1)
        y, x = divmod(p, W)
        addr.append(stride * y * RP + stride * x * PX + 16 * (lane >> 4))
      tot += cycles(addr)
    out.append((tot / tiles, PX, pad))
  for c, PX, pad in sorted(out)[:5]:
    print(f"{name}: pixel pitch {PX} row pad {pad}: {c:.2f} cycles per read (4 = conflict-free)")


# conv1 dgrad: 10 x 10 pixels of a parity class over the zero-bordered 11 x 11 gradient image (64 channels)
search("conv1_dgrad", 10, 100, 11, 1, range(128, 209, 16), range(0, 257, 16))
# conv2 dgrad: 9 x 9 input pixels over the zero-bordered 11 x 11 gradient image (64 channels)
search("conv2_dgrad", 9, 81, 11, 1, range(128, 209, 16), range(0, 257, 16))
