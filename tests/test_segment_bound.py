"""The ray kernels cut a ray of l1 = dx + dy + dz steps every w = floor(dmax * K / l1) pops of its dominant axis
(scan_kernels.h: k_cast, fast_kernels.h: k_fcast) and must be sure that a round's segments fit their LDS queue without
counting them first: K >= total / room + 3 (total = the round's steps, room = queue entries minus rays) guarantees it,
because a ray then has at most l1 / (K - 3) + 1 segments. Checked here on the integers themselves -- every (dx, dy, dz) of a
small cube exhaustively, random long rays, adversarial rounds."""
import numpy as np


def _nseg(dmax, l1, K):
    w = np.maximum(1, (dmax.astype(np.uint64) * np.uint64(K)) // l1.astype(np.uint64))
    return (dmax.astype(np.uint64) + w - 1) // w


def test_per_ray_bound_exhaustive_and_random():
    r = np.arange(0, 41)
    dx, dy, dz = [a.reshape(-1) for a in np.meshgrid(r, r, r, indexing="ij")]
    keep = (dx + dy + dz) > 0
    dx, dy, dz = dx[keep], dy[keep], dz[keep]
    rng = np.random.default_rng(0)
    big = rng.integers(0, 1023, (200000, 3))
    big = big[big.sum(1) > 0]
    for d in (np.stack([dx, dy, dz], 1), big):
        dmax, l1 = d.max(1), d.sum(1)
        for K in (4, 5, 8, 9, 16, 31, 32, 33, 64, 87, 165, 1000, 4096):
            n = _nseg(dmax, l1, K)
            assert np.all(n >= 1)
            assert np.all(n * (K - 3) <= l1 + (K - 3)), f"K={K}: a ray has more than l1/(K-3)+1 segments"


def test_round_fits_the_queue():
    rng = np.random.default_rng(1)
    for qcap, batch in ((1024, 256), (512, 128), (768, 192), (2048, 512)):
        for _ in range(300):
            nr = int(rng.integers(1, batch + 1))
            scale = int(rng.choice([3, 30, 300, 1022]))
            d = rng.integers(0, scale + 1, (nr, 3))
            d[d.sum(1) == 0, 0] = 1
            dmax, l1 = d.max(1), d.sum(1)
            total, room = int(l1.sum()), qcap - nr
            for k_min in (8, 32):
                K = max(k_min, (total + room - 1) // room + 3)
                assert int(_nseg(dmax, l1, K).sum()) <= qcap, (qcap, nr, scale, K)
