#!/usr/bin/env python3
"""What the PNG encoder's code construction costs against Huffman's algorithm (CPU only, no GPU).

ist_png_deflate.hip builds its per-chunk literal/length code from Shannon lengths (smallest l with count * 2^l >= total)
and then hands the Kraft slack back - one bit off per symbol, shortest codes first, symbol order inside a length, round
after round until the code is complete.  This script restates that rule and compares the coded size with optimal lengths
on (a) Paeth-filtered rows of the bench's photo-like content and (b) random histograms, and reports how many rounds the
completion needs."""
import heapq
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def paeth_residuals(img):
    a = img.astype(np.int32)
    left = np.zeros_like(a)
    left[:, 1:] = a[:, :-1]
    up = np.zeros_like(a)
    up[1:] = a[:-1]
    ul = np.zeros_like(a)
    ul[1:, 1:] = a[:-1, :-1]
    p = left + up - ul
    pa, pb, pc = abs(p - left), abs(p - up), abs(p - ul)
    pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
    return ((a - pred) & 255).astype(np.uint8)


def huffman_lengths(counts):
    items = [(c, i) for i, c in enumerate(counts) if c]
    if len(items) == 1:
        return {items[0][1]: 1}
    heap = [(c, n, (i,)) for n, (c, i) in enumerate(items)]
    heapq.heapify(heap)
    length = {i: 0 for _, i in items}
    n = len(heap)
    while len(heap) > 1:
        a, b = heapq.heappop(heap), heapq.heappop(heap)
        for i in a[2] + b[2]:
            length[i] += 1
        n += 1
        heapq.heappush(heap, (a[0] + b[0], n, a[2] + b[2]))
    return length


def shannon_complete(counts, max_rounds=16):
    """the kernel's rule; returns (lengths or None if not complete after max_rounds, rounds used)"""
    total = sum(counts)
    length = {i: min(15, max(1, int(np.ceil(np.log2(total / c))))) for i, c in enumerate(counts) if c}
    slack = 2 ** 15 - sum(2 ** (15 - l) for l in length.values())
    rounds = 0
    while slack > 0 and rounds < max_rounds:
        rounds += 1
        for l in range(2, 16):
            cls = [i for i in sorted(length) if length[i] == l]
            cost = 2 ** (15 - l)
            take = min(len(cls), slack // cost)
            for i in cls[:take]:
                length[i] = l - 1
            slack -= take * cost
    return (length if slack == 0 else None), rounds


def tokens(row):
    """a row's bytes as the encoder's symbols: literals, a run of equal bytes as one literal + length symbols"""
    sym, i, r = [], 0, row.tolist()
    while i < len(r):
        j = i
        while j + 1 < len(r) and r[j + 1] == r[i] and j - i < 65:
            j += 1
        sym.append(r[i])
        if j - i + 1 >= 4:
            sym.append(257 + min(28, (j - i - 3) // 4))
            i = j + 1
        else:
            i += 1
    sym.append(256)
    return sym


def main():
    import bench
    from PIL import Image
    photo = np.asarray(Image.open(io.BytesIO(bench.photo_jpeg(0, 4032, 128))).convert("RGBA"))
    noise = np.random.default_rng(0).integers(0, 256, (64, 4032, 4), dtype=np.uint8)
    smooth = bench.synth_np(0, 4032, 64)
    if smooth.shape[-1] == 3:
        smooth = np.concatenate([smooth, np.full(smooth.shape[:2] + (1,), 255, np.uint8)], -1)
    for name, im in (("photo-like (decoded JPEG)", photo), ("bench synthetic", smooth), ("uniform noise", noise)):
        res = paeth_residuals(im)
        ours = best = 0
        worst_rounds = 0
        for y in range(min(64, im.shape[0])):
            counts = np.bincount(tokens(np.concatenate([[4], res[y].reshape(-1)])), minlength=286).tolist()
            length, rounds = shannon_complete(counts)
            worst_rounds = max(worst_rounds, rounds)
            opt = huffman_lengths(counts)
            ours += sum(counts[i] * l for i, l in length.items())
            best += sum(counts[i] * l for i, l in opt.items())
        print("%-28s token bits vs Huffman %.4f   rounds <= %d" % (name, ours / best, worst_rounds))
    rng = np.random.default_rng(1)
    ratios, worst, fails = [], 0, 0
    for trial in range(3000):
        ns = int(rng.integers(2, 287))
        kind = trial % 4
        if kind == 0:
            c = rng.integers(1, 2000, ns)
        elif kind == 1:
            c = np.maximum(1, (16000 * rng.dirichlet(np.ones(ns) * 0.1)).astype(int))
        elif kind == 2:
            c = np.maximum(1, (2.0 ** rng.uniform(0, 14, ns)).astype(int))
        else:
            c = np.ones(ns, int)
            c[0] = int(rng.integers(1, 16000))
        counts = [0] * 286
        for k, i in enumerate(rng.permutation(286)[:ns]):
            counts[i] = int(c[k])
        while sum(counts) > 16500:
            counts = [max(1, x // 2) if x else 0 for x in counts]
        length, rounds = shannon_complete(counts)
        worst = max(worst, rounds)
        if length is None:
            fails += 1
            continue
        assert sum(2 ** (15 - l) for l in length.values()) == 2 ** 15 and max(length.values()) <= 15
        opt = huffman_lengths(counts)
        if max(opt.values()) <= 15:
            ratios.append(sum(counts[i] * l for i, l in length.items()) / sum(counts[i] * l for i, l in opt.items()))
    print("3000 random histograms: complete after <= %d rounds, %d not complete; token bits vs Huffman mean %.4f max %.4f"
          % (worst, fails, np.mean(ratios), np.max(ratios)))


if __name__ == "__main__":
    main()
