#!/usr/bin/env python3
"""What sorting queued hits into cost classes could save, from the histograms of tools/experiments/hist.py.

A wave of a launch over the queue waits for its longest ray, so its node steps cost E[max of 64] of
the steps of the rays in it. For a way of filing hits into classes (each class its own waves) this
prints that expectation per wave, next to the mean (what perfectly sorted rays would cost), and the
same for a walk cut into slices by a step budget (every slice its own launch over rays packed
densely, state carried across): the model behind two experiments of this build.

    python tools/experiments/cost_classes.py <dir with hist_<workload>.npy>
"""
import glob
import os
import sys

import numpy as np


def emax(pmf, n=64):
    F = np.cumsum(pmf)
    F = F / F[-1]
    return float(np.sum(1 - F[:-1] ** n))  # E[max of n draws] for values 0..len-1


def classes_cost(h, classes):
    tot = h.sum()
    c = 0.0
    for m in classes:
        pm = (h * m[:, :, None]).sum(axis=(0, 1))
        n = pm.sum()
        if n > 0:
            c += n / 64 * emax(pm)
    return c / (tot / 64)


def slices_cost(h1, cuts):
    N = h1.sum()
    cost, lo = 0.0, 0
    for e in list(cuts) + [len(h1)]:
        rem = h1.copy()
        if lo > 0:
            rem[: lo + 1] = 0
        n = rem.sum()
        if n <= 0:
            break
        pm = np.zeros(len(h1))
        for s in range(len(h1)):
            if rem[s] > 0:
                pm[min(s, e) - lo] += rem[s]
        cost += n / 64 * emax(pm)
        lo = e
    return cost / (N / 64)


for f in sorted(glob.glob(os.path.join(sys.argv[1], "hist_*.npy"))):
    h = np.load(f)
    wl = os.path.basename(f)[5:-4]
    steps = np.arange(64)
    mean = (h.sum(axis=(0, 1)) * steps).sum() / h.sum()
    obj = np.arange(8)[:, None] * np.ones((1, 64))
    prev = np.ones((8, 1)) * np.arange(64)[None, :]
    print(f"== {wl}: {h.sum():.0f} scattered rays, mean {mean:.1f} node steps (the histogram clips at 63)")
    print(f"   node steps a wave issues (E[max of 64]): one class {classes_cost(h, [np.ones((8, 64), bool)]):.1f}; "
          f"two classes, previous ray under/over 8 steps (what the kernels do) {classes_cost(h, [prev < 8, prev >= 8]):.1f}")
    for th in [(4, 8, 16), (2, 6, 12), (8, 16, 32)]:
        cl = [prev < th[0], (prev >= th[0]) & (prev < th[1]), (prev >= th[1]) & (prev < th[2]), prev >= th[2]]
        print(f"   four classes by the previous ray's steps {th}: {classes_cost(h, cl):.1f}")
    print(f"   eight classes by the hit's object: {classes_cost(h, [(obj == o) for o in range(8)]):.1f}; "
          f"sixteen by object and previous steps: {classes_cost(h, [(obj == o) & (prev < 8) for o in range(8)] + [(obj == o) & (prev >= 8) for o in range(8)]):.1f}; "
          f"512 by object and every previous step count: {classes_cost(h, [(prev == p) & (obj == o) for p in range(64) for o in range(8)]):.1f}")
    h1 = h.sum(axis=(0, 1))
    print("   walk cut into slices (state carried over, rays repacked): " + "; ".join(f"cuts {c}: {slices_cost(h1, c):.1f}" for c in [(8,), (8, 20), (4, 10, 24)]))
