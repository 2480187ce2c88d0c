"""The PNG encoder's code-length rule (ist_png_deflate.hip, phase C) restated on the CPU (tools/sim_code_lengths.py): on any
histogram a 16 KiB chunk can have it must end with a COMPLETE prefix code of at most 15 bits, within 16 rounds, and stay
close to Huffman's cost.  The GPU tests decode what the kernel wrote; this one checks the rule itself over many more
histograms than a GPU test can afford."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import sim_code_lengths as S  # noqa: E402


def _histograms(n, seed):
    rng = np.random.default_rng(seed)
    for trial in range(n):
        ns = int(rng.integers(2, 287))
        kind = trial % 5
        if kind == 0:
            c = rng.integers(1, 2000, ns)
        elif kind == 1:
            c = np.maximum(1, (16000 * rng.dirichlet(np.ones(ns) * 0.1)).astype(int))
        elif kind == 2:
            c = np.maximum(1, (2.0 ** rng.uniform(0, 14, ns)).astype(int))
        elif kind == 3:
            c = np.ones(ns, int)
            c[0] = int(rng.integers(1, 16000))
        else:
            c = np.maximum(1, (2.0 ** rng.integers(0, 14, ns)).astype(int))      # exact powers of two
        counts = [0] * 286
        for k, i in enumerate(rng.permutation(286)[:ns]):
            counts[i] = int(c[k])
        while sum(counts) > 16500:
            counts = [max(1, x // 2) if x else 0 for x in counts]
        yield counts


def test_rule_gives_a_complete_code_of_at_most_15_bits():
    worst_rounds, worst_ratio = 0, 0.0
    for counts in _histograms(1500, 5):
        length, rounds = S.shannon_complete(counts)
        assert length is not None, counts
        assert set(length) == {i for i, c in enumerate(counts) if c}
        assert 1 <= min(length.values()) and max(length.values()) <= 15
        assert sum(2 ** (15 - l) for l in length.values()) == 2 ** 15          # Kraft sum exactly 1: complete, not over-subscribed
        worst_rounds = max(worst_rounds, rounds)
        opt = S.huffman_lengths(counts)
        if max(opt.values()) <= 15:
            ours = sum(counts[i] * l for i, l in length.items())
            best = sum(counts[i] * l for i, l in opt.items())
            worst_ratio = max(worst_ratio, ours / best)
    assert worst_rounds <= 16
    assert worst_ratio < 1.15


def test_two_symbols_get_one_bit_each():
    counts = [0] * 286
    counts[256] = 1
    counts[7] = 16000
    length, _ = S.shannon_complete(counts)
    assert length == {7: 1, 256: 1}
