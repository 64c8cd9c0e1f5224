"""Restatement of numpy's arg-introselect for the one call the reference makes:
``np.argpartition(row_fp16, -1)[-1:]`` (scripts/sampling/feature_extraction.py:293).

Third-party algorithm (numpy 2.x, numpy/_core/src/npysort/selection.cpp, BSD-3; not part of
/root/reference).  For npy_half the "kth == num-1 -> linear max scan" shortcut is NOT taken
(npy_half is a uint16 typedef, so ``inexact<type>()`` is false) and no SIMD arg-select exists
for 16-bit keys, so the scalar median-of-3 introselect runs and its swap sequence decides
which of several tied maxima ends up in the last slot.  The port below reproduces that
sequence; tests/test_oracle_analysis.py checks it against numpy itself on tie-heavy rows.
Test infrastructure only.
"""
import numpy as np


def _msb(n):
    d = 0
    while n > 1:
        n >>= 1
        d += 1
    return d


def _median5(v, t, o):
    def lt(a, b):
        return v[t[o + a]] < v[t[o + b]]

    def sw(a, b):
        t[o + a], t[o + b] = t[o + b], t[o + a]
    if lt(1, 0):
        sw(1, 0)
    if lt(4, 3):
        sw(4, 3)
    if lt(3, 0):
        sw(3, 0)
    if lt(4, 1):
        sw(4, 1)
    if lt(2, 1):
        sw(2, 1)
    if lt(3, 2):
        return 1 if lt(3, 1) else 3
    return 2


def _introselect(v, t, off, num, kth):
    """arg-introselect on the index window t[off:off+num] (kth relative to the window)."""
    low, high = 0, num - 1
    if kth - low < 3:                                           # dumb_select
        for i in range(kth + 1):
            minidx, minval = i, v[t[off + i]]
            for k in range(i + 1, num):
                if v[t[off + k]] < minval:
                    minidx, minval = k, v[t[off + k]]
            t[off + i], t[off + minidx] = t[off + minidx], t[off + i]
        return
    depth_limit = _msb(num) * 2
    while low + 1 < high:
        ll, hh = low + 1, high
        if depth_limit > 0 or hh - ll < 5:
            mid = low + (high - low) // 2
            a = lambda i: v[t[off + i]]                          # noqa: E731
            if a(high) < a(mid):
                t[off + high], t[off + mid] = t[off + mid], t[off + high]
            if a(high) < a(low):
                t[off + high], t[off + low] = t[off + low], t[off + high]
            if a(low) < a(mid):
                t[off + low], t[off + mid] = t[off + mid], t[off + low]
            t[off + mid], t[off + low + 1] = t[off + low + 1], t[off + mid]
        else:                                                     # median of medians of 5
            n2 = hh - ll
            nmed = n2 // 5
            sub = 0
            for i in range(nmed):
                m = _median5(v, t, off + ll + sub)
                t[off + ll + sub + m], t[off + ll + i] = t[off + ll + i], t[off + ll + sub + m]
                sub += 5
            if nmed > 2:
                _introselect(v, t, off + ll, nmed, nmed // 2)
            mid = ll + nmed // 2
            t[off + mid], t[off + low] = t[off + low], t[off + mid]
            ll -= 1
            hh += 1
        depth_limit -= 1
        pivot = v[t[off + low]]
        while True:                                               # unguarded partition
            ll += 1
            while v[t[off + ll]] < pivot:
                ll += 1
            hh -= 1
            while pivot < v[t[off + hh]]:
                hh -= 1
            if hh < ll:
                break
            t[off + ll], t[off + hh] = t[off + hh], t[off + ll]
        t[off + low], t[off + hh] = t[off + hh], t[off + low]
        if hh >= kth:
            high = hh - 1
        if hh <= kth:
            low = ll
    if high == low + 1:
        if v[t[off + high]] < v[t[off + low]]:
            t[off + high], t[off + low] = t[off + low], t[off + high]


def argpartition_last(row):
    """Index numpy returns for np.argpartition(row, -1)[-1:][0] on an fp16 row (NaN-free)."""
    v = [float(x) for x in np.asarray(row)]
    n = len(v)
    t = list(range(n))
    _introselect(v, t, 0, n, n - 1)
    return t[n - 1]
