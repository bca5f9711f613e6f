"""Host-side random draws for centroid initialisation and empty-cluster reseeds.

The reference draws them from rand 0.9's ``StdRng`` (ChaCha12) via ``choose_multiple`` /
``choose`` (src/core/vector.rs:412-413, 448-452).  That stream cannot be reproduced or
verified without a Rust toolchain, so it is NOT claimed here: this is the package's own
documented generator (SplitMix64 + Floyd sampling).  Downstream of the draws the assignment
codes are bit-identical to the reference's; the centroids are bit-identical only with
``exact_update=True`` (sums in the reference's row order) and otherwise within
``1e-5 * max(1, |c|)`` per Lloyd step (f64-combined chunk sums), so the
``< 1e-6`` convergence test -- and with it the iteration count -- can differ from the crate's
at a boundary.  Every entry point also accepts the draws from the caller (``init_rows=`` /
``reseed_rows=``), which is how a Rust shim keeps ``StdRng`` on its side.
"""
from __future__ import annotations

_MASK = (1 << 64) - 1


class HostRng:
    def __init__(self, seed: int):
        self.state = int(seed) & _MASK

    def next_u64(self) -> int:
        self.state = (self.state + 0x9E3779B97F4A7C15) & _MASK
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        return z ^ (z >> 31)

    def below(self, n: int) -> int:
        """uniform integer in [0, n) (Lemire's multiply-shift with rejection)"""
        if n <= 0:
            raise ValueError("below(n) needs n > 0")
        threshold = ((1 << 64) - n) % n
        while True:
            x = self.next_u64()
            m = x * n
            if (m & _MASK) >= threshold:
                return m >> 64

    def choose(self, n: int) -> int:
        """stand-in for ``data.choose(&mut rng)`` (vector.rs:450): one row id"""
        return self.below(n)

    def choose_multiple(self, n: int, k: int) -> list[int]:
        """stand-in for ``data.choose_multiple(&mut rng, k)`` (vector.rs:413): k distinct row
        ids (Floyd's algorithm; order = draw order)"""
        if k > n:
            raise ValueError("cannot choose more rows than exist")
        chosen: set[int] = set()
        out: list[int] = []
        for j in range(n - k, n):
            t = self.below(j + 1)
            pick = j if t in chosen else t
            chosen.add(pick)
            out.append(pick)
        return out
