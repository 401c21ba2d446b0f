"""A model of what the persistent ring GEMM fetches from beyond one XCD's L2 (no GPU needed): the 32 workgroups of an XCD walk their
tile list 32 tiles at a time and stream the K slices of their operands in step; an LRU cache of 4 MB sees, per K step, one 32 KB
slice (256 rows x 64 elements x 2 B) of every distinct row panel and of every distinct column slice of the round.  It reproduces
rocprofv3's FETCH_SIZE per launch at M = 409 600 (profiles/r04_x_serpentine.log: FFN-up 2.83 GB forward-only, 2.17 with the
serpentine K walk; QKV 2.47 / 2.17) as 2.83 / 1.99 and 2.58 / 2.19, and was used to check that, within "groups of G row panels,
column-major, direction = f(column)", G = 8 with blocks of four column slices is the cheapest for 9 and for 12 slices
(DESIGN.md §3 "Serpentine K walk").

    python tools/l2_round_sim.py            # the two production shapes, forward-only and with the rule in the tree
    python tools/l2_round_sim.py sweep      # group sizes x direction block sizes
"""
import sys
from collections import OrderedDict


def simulate(tiles_n, nk, dirs, group=8, panels_per_xcd=200, cache_bytes=4 << 20, slice_bytes=32 << 10, cus=32):
    """GB fetched beyond L2 by one launch of 1 600 row panels (8 XCDs x panels_per_xcd); dirs[tn] = 1: column slice tn walks K backwards."""
    tiles = []
    for g0 in range(0, panels_per_xcd, group):                      # gemm_kernel_hp's list: groups of `group` panels, column-major inside
        rows = min(group, panels_per_xcd - g0)
        for tn in range(tiles_n):
            tiles += [(g0 + r, tn) for r in range(rows)]
    cache, cap, miss = OrderedDict(), cache_bytes // slice_bytes, 0
    for r0 in range(0, len(tiles), cus):
        rnd = tiles[r0:r0 + cus]
        for step in range(nk):
            need = []
            for tm, tn in rnd:
                k = nk - 1 - step if dirs[tn] else step
                need += [("A", tm, k), ("W", tn, k)]
            for key in dict.fromkeys(need):
                if key in cache:
                    cache.move_to_end(key)
                else:
                    miss += 1
                    cache[key] = 1
                    if len(cache) > cap:
                        cache.popitem(last=False)
    return miss * slice_bytes * 8 / 1e9 * (1600 / (panels_per_xcd * 8))


if __name__ == "__main__":
    nk = 12
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        for tiles_n in (9, 12):
            res = sorted((simulate(tiles_n, nk, [(c // b) & 1 for c in range(tiles_n)], group=g), g, b)
                         for g in (4, 6, 8, 10, 11, 12, 16, 32) for b in (1, 2, 3, 4, 5, 6, 12))
            print(f"{tiles_n} column slices, K = {64 * nk}:")
            for v, g, b in res[:6]:
                print(f"   {v:.2f} GB per launch   groups of {g} panels, direction blocks of {b} slices")
    else:
        for name, tiles_n in (("FFN-up (12 column slices)", 12), ("QKV (9 column slices)", 9)):
            print(f"{name}: forward-only {simulate(tiles_n, nk, [0] * tiles_n):.2f} GB per launch, "
                  f"serpentine (blocks of four) {simulate(tiles_n, nk, [(c >> 2) & 1 for c in range(tiles_n)]):.2f} GB")
