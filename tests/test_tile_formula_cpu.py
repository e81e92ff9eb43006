"""The two-part FILTER scan visits every P-th tile first and the other tiles afterwards (artmips.hip, shard
pipeline; art_amd/csrc/artmips_kernels.h scan_kernel: tile walker and tile_of).  The 8-wave kernels compute
the tile of a work index without a division: q = mulhi(w, magic) >> shift with a 32-bit magic number.  This
restates that arithmetic with Python integers and checks where it is exact (everywhere for / 3, below 2^34 / 5
for / 7; work indices are tile counts, below 2^27) and the coverage (every tile exactly once over the two
parts)."""
import numpy as np
import pytest

MAGIC = {4: (0xAAAAAAAB, 1), 8: (0x92492493, 2)}   # period P -> (magic, shift) for the division by P - 1


def q_of(w, period):
    magic, shift = MAGIC[period]
    return ((w * magic) >> 32) >> shift


@pytest.mark.parametrize("period", [4, 8])
def test_magic_division_is_exact_where_the_kernel_uses_it(period):
    d = period - 1
    magic, shift = MAGIC[period]
    e = d * magic - 2 ** (32 + shift)          # d * magic = 2^(32 + shift) + e: exact while w * e < 2^(32 + shift)
    assert 0 < e and magic < 2 ** 32
    limit = min(2 ** 32, -(-2 ** (32 + shift) // e))
    assert limit > 2 ** 27                     # tile counts: n_rows < 2^31, 16 rows per tile
    assert (limit == 2 ** 32) == (period == 4)
    rng = np.random.Generator(np.random.PCG64(7))
    ws = set(int(x) for x in rng.integers(0, 2 ** 32, size=200000, dtype=np.uint64))
    ws.update(range(0, 5000))
    ws.update(2 ** 32 - 1 - i for i in range(5000))
    for e in range(1, 33):   # around the powers of two and the multiples of d next to them
        for off in range(-d - 1, d + 2):
            v = 2 ** e + off
            if 0 <= v < 2 ** 32:
                ws.add(v)
                ws.add(v - v % d)
    for w in ws:
        if w < limit:
            assert q_of(w, period) == w // d, (w, period)


@pytest.mark.parametrize("period", [4, 8])
@pytest.mark.parametrize("n_tiles", [1, 2, 7, 8, 9, 64, 1000, 1313458])
def test_the_two_parts_visit_every_tile_once(period, n_tiles):
    part_a = (n_tiles + period - 1) // period
    seen = np.zeros(n_tiles, dtype=np.int32)
    seen[np.arange(part_a) * period] += 1                    # part A: work index w -> tile w * P
    w = np.arange(n_tiles - part_a, dtype=np.uint64)         # part B: w -> (w / (P-1)) * P + w % (P-1) + 1
    magic, shift = MAGIC[period]
    q = ((w * np.uint64(magic)) >> np.uint64(32)) >> np.uint64(shift)
    tile = q * np.uint64(period) + (w - q * np.uint64(period - 1)) + np.uint64(1)
    assert tile.size == 0 or int(tile.max()) < n_tiles
    np.add.at(seen, tile.astype(np.int64), 1)
    assert (seen == 1).all()
