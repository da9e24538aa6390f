"""The lower bound behind the filter kernel's early exit (csrc/filter.hip, filter_kernel):

    sum over a 256-pixel group of dE00(cell pixel, library pixel)
        >= sqrt( (|sum dL|)^2 / S_L^2 + rho ((|sum da|)^2 + (|sum db|)^2) / S_C^2 )

with S_L, S_C, rho the worst constants the group's lightness range and largest chroma allow.
A candidate is dropped when (partial sum + bound of the unvisited groups) passes the cell's
limit, so the bound must NEVER exceed the true sum: checked here against the oracle's
CIEDE2000 on random and adversarial groups (the numpy restatement below is the kernel's
formula in float64; the kernel itself is held to the unpruned path in test_gpu_filter.py)."""
import numpy as np
import pytest


def sqrt_ratio7(x):
    u = (x / 25.0) ** 7
    return np.sqrt(u / (u + 1.0))


def group_bound(cell, lib):
    """cell, lib: [..., n, 3] Lab pixels of one group -> lower bound of sum_p dE00."""
    dev = 0.5 * (np.abs(cell[..., 0] - 50).max(-1) + np.abs(lib[..., 0] - 50).max(-1))
    s_l = 1.0 + 0.015 * dev ** 2 / np.sqrt(20.0 + dev ** 2)
    cbar = 0.5 * (np.hypot(cell[..., 1], cell[..., 2]).max(-1) + np.hypot(lib[..., 1], lib[..., 2]).max(-1))
    gc = (1.5 - 0.5 * sqrt_ratio7(cbar)) * cbar
    s_c = 1.0 + 0.045 * gc
    rho = 1.0 - 0.86603 * sqrt_ratio7(gc)
    d = np.abs(cell.sum(-2) - lib.sum(-2))
    return np.sqrt((d[..., 0] / s_l) ** 2 + rho * (d[..., 1] ** 2 + d[..., 2] ** 2) / s_c ** 2)


def groups(rng, kind, n_groups, n=256):
    lo, hi = np.array([0, -128, -128.0]), np.array([100, 127, 127.0])
    if kind == "uniform":
        return rng.uniform(lo, hi, (n_groups, n, 3)), rng.uniform(lo, hi, (n_groups, n, 3))
    if kind == "patches":                 # what photos and the synthetic library look like: a base colour + noise
        base = lambda: rng.uniform(lo, hi, (n_groups, 1, 3))
        return base() + rng.normal(0, 6, (n_groups, n, 3)), base() + rng.normal(0, 6, (n_groups, n, 3))
    if kind == "close-patches":           # contenders: nearly the same base colour
        b = rng.uniform(lo, hi, (n_groups, 1, 3))
        return b + rng.normal(0, 4, (n_groups, n, 3)), b + rng.normal(0, 1.5, (n_groups, 1, 3)) + rng.normal(0, 4, (n_groups, n, 3))
    if kind == "blue":                    # hue ~275 deg, high chroma: the rotation term R_T at its largest
        def blue():
            th = np.deg2rad(rng.uniform(255, 295, (n_groups, 1)) + rng.normal(0, 4, (n_groups, n)))
            r = rng.uniform(30, 120, (n_groups, 1)) + rng.normal(0, 5, (n_groups, n))
            return np.stack([rng.uniform(10, 90, (n_groups, 1)) + rng.normal(0, 4, (n_groups, n)), r * np.cos(th), r * np.sin(th)], -1)
        return blue(), blue()
    if kind == "blue-chroma-shift":       # same hue and lightness, different chroma: only tC (and R_T) speak
        th = np.deg2rad(275.0)
        def ray(r):
            return np.stack([np.full_like(r, 50.0), r * np.cos(th), r * np.sin(th)], -1)
        r1 = rng.uniform(20, 100, (n_groups, 1)) + rng.normal(0, 1, (n_groups, n))
        return ray(r1), ray(r1 + rng.uniform(-15, 15, (n_groups, 1)))
    if kind == "near-grey":
        a, b = rng.uniform(lo, hi, (n_groups, n, 3)), rng.uniform(lo, hi, (n_groups, n, 3))
        a[..., 1:] *= 0.02; b[..., 1:] *= 0.02
        return a, b
    if kind == "dark-vs-bright":
        a, b = rng.uniform(lo, hi, (n_groups, n, 3)), rng.uniform(lo, hi, (n_groups, n, 3))
        a[..., 0] = rng.uniform(0, 15, (n_groups, n)); b[..., 0] = rng.uniform(85, 100, (n_groups, n))
        return a, b
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["uniform", "patches", "close-patches", "blue", "blue-chroma-shift", "near-grey", "dark-vs-bright"])
def test_group_bound_never_exceeds_the_true_sum(oracle, kind):
    rng = np.random.default_rng(sum(map(ord, kind)))
    cell, lib = groups(rng, kind, 1500)
    cell, lib = cell.astype(np.float32), lib.astype(np.float32)        # what the engine holds
    true = oracle.colour_diff_batch(2, cell.reshape(-1, 3), lib.reshape(-1, 3)).reshape(cell.shape[:2]).sum(1)
    bound = group_bound(cell.astype(np.float64), lib.astype(np.float64))
    assert (bound <= true * (1 + 1e-9) + 1e-9).all(), (kind, (bound / np.maximum(true, 1e-30)).max())
    if kind in ("patches", "dark-vs-bright"):
        assert np.median(bound / true) > 0.25           # and it is worth having


def test_the_constants_are_worst_cases(oracle):
    """g(C) = (1.5 - 0.5 sqrt(C^7 / (C^7 + 25^7))) C grows with C (so the largest chroma of the two
    groups bounds every pair's C'), and T stays below 1.93 (so S_H <= S_C)."""
    c = np.linspace(0, 200, 200001)
    g = (1.5 - 0.5 * sqrt_ratio7(c)) * c
    assert (np.diff(g) > 0).all()
    h = np.deg2rad(np.linspace(0, 360, 360001))
    t = 1 - 0.17 * np.cos(h - np.deg2rad(30)) + 0.24 * np.cos(2 * h) + 0.32 * np.cos(3 * h + np.deg2rad(6)) - 0.20 * np.cos(4 * h - np.deg2rad(63))
    assert 0 < t.min() and t.max() < 1.93 and 0.015 * t.max() < 0.045


@pytest.mark.parametrize("type_id,kind", [(0, "uniform"), (0, "patches"), (1, "close-patches"), (1, "blue")])
def test_euclid_block_bound_never_exceeds_the_true_sum(oracle, type_id, kind):
    """RGB-Euclidean / CIE76: sum_block |x_p - y_p| >= |sum_block x_p - sum_block y_p| (triangle
    inequality of the Euclidean norm), the bound of filter_lds_kernel's early exit -- no constants."""
    rng = np.random.default_rng(7 + sum(map(ord, kind)))
    cell, lib = groups(rng, kind, 2000, n=64)
    if type_id == 0:                                       # what RGB inputs look like: integer-valued 0..255
        cell = np.clip(np.rint(cell * 2.0 + 60), 0, 255); lib = np.clip(np.rint(lib * 2.0 + 60), 0, 255)
    cell, lib = cell.astype(np.float32), lib.astype(np.float32)
    true = oracle.colour_diff_batch(type_id, cell.reshape(-1, 3), lib.reshape(-1, 3)).reshape(cell.shape[:2]).sum(1)
    d = np.abs(cell.astype(np.float64).sum(1) - lib.astype(np.float64).sum(1))
    bound = np.sqrt((d ** 2).sum(1))
    assert (bound <= true * (1 + 1e-12) + 1e-9).all()
    if kind == "patches":
        assert np.median(bound / true) > 0.5
