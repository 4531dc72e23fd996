"""CPU tests of oracle/match_oracle.cpp (SearchByProjection restatement): known answers + an independent brute-force model."""
import math

import numpy as np

from tests import match_scenes as ms
from tests import oracle_lib


def _popcount_dist(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def test_descriptor_distance_is_hamming(oracle):
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (200, 32), dtype=np.uint8); b = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    b[:5] = a[:5]; b[5] = ~a[5]
    d = oracle_lib.descriptor_distance(a, b)
    assert list(d) == [_popcount_dist(a[i], b[i]) for i in range(200)] and d[0] == 0 and d[5] == 256


def _f32(x):
    return np.float32(x)


def brute_force(p, cur, last, Tc, Tl):
    """An independent, literal model of src/ORBmatcher.cc:547-678 + src/Frame.cc:332-381 in numpy float32 / python."""
    fx, fy, cx, cy, bf = (_f32(p[k][0]) for k in ("fx", "fy", "cx", "cy", "bf"))
    minX, maxX, minY, maxY, th = (_f32(p[k][0]) for k in ("minX", "maxX", "minY", "maxY", "th"))
    sf = p["scale_factors"][0]
    wInv = _f32(64) / (maxX - minX); hInv = _f32(48) / (maxY - minY)
    n = len(cur["kps"])
    grid = {}
    for i in range(n):
        c = int(cur["grid_cell"][i])
        if c >= 0:
            grid.setdefault(c, []).append(i)
    R, t = Tc[:3, :3].astype(np.float64), Tc[:3, 3].astype(np.float64)
    twc = (-(R.T @ t)).astype(np.float32)
    tlc = (Tl[:3, :3].astype(np.float64) @ twc.astype(np.float64) + Tl[:3, 3].astype(np.float64)).astype(np.float32)
    mb = bf / fx
    fwd, bwd = tlc[2] > mb, -tlc[2] > mb
    held = [-1] * n
    nm = 0
    hist = [[] for _ in range(30)]
    for i in range(len(last["xyz"])):
        if not (last["flags"][i] & 1):
            continue
        xc3 = (R @ last["xyz"][i].astype(np.float64) + t).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            invz = _f32(np.float64(1.0) / np.float64(xc3[2]))
            if invz < 0:
                continue
            u = fx * xc3[0] * invz + cx; v = fy * xc3[1] * invz + cy
        if not (u >= minX and u <= maxX and v >= minY and v <= maxY):
            continue
        o = int(last["octave"][i])
        r = th * sf[o]
        lo, hi = (o, -1) if fwd else ((0, o) if bwd else (o - 1, o + 1))
        x0 = max(0, int(math.floor((u - minX - r) * wInv))); x1 = min(63, int(math.ceil((u - minX + r) * wInv)))
        y0 = max(0, int(math.floor((v - minY - r) * hInv))); y1 = min(47, int(math.ceil((v - minY + r) * hInv)))
        if x0 >= 64 or x1 < 0 or y0 >= 48 or y1 < 0:
            continue
        best, bi = 256, -1
        for ix in range(x0, x1 + 1):
            for iy in range(y0, y1 + 1):
                for j in grid.get(ix * 48 + iy, []):
                    oc = int(cur["kps"]["octave"][j])
                    if (lo > 0 or hi >= 0) and (oc < lo or (hi >= 0 and oc > hi)):
                        continue
                    if not (abs(cur["un_xy"][j, 0] - u) < r and abs(cur["un_xy"][j, 1] - v) < r):
                        continue
                    if held[j] >= 0 and (last["flags"][held[j]] & 2):
                        continue
                    if cur["uright"][j] > 0 and abs((u - bf * invz) - cur["uright"][j]) > r:
                        continue
                    d = _popcount_dist(last["desc"][i], cur["desc"][j])
                    if d < best:
                        best, bi = d, j
        if best <= 100:
            held[bi] = i
            nm += 1
            if p["check_orientation"][0]:
                rot = _f32(last["angle"][i]) - _f32(cur["kps"]["angle"][bi])
                if rot < 0:
                    rot = rot + _f32(360)
                x = float(rot * _f32(_f32(1.0) / _f32(30)))
                b = int(math.floor(x + 0.5)) if x >= 0 else -int(math.floor(-x + 0.5))
                hist[0 if b == 30 else b].append(bi)
    if p["check_orientation"][0]:
        m = [0, 0, 0]; ind = [-1, -1, -1]
        for i in range(30):
            s = len(hist[i])
            if s > m[0]:
                m = [s, m[0], m[1]]; ind = [i, ind[0], ind[1]]
            elif s > m[1]:
                m = [m[0], s, m[1]]; ind = [ind[0], i, ind[1]]
            elif s > m[2]:
                m[2] = s; ind[2] = i
        if m[1] < _f32(0.1) * _f32(m[0]):
            ind[1] = ind[2] = -1
        elif m[2] < _f32(0.1) * _f32(m[0]):
            ind[2] = -1
        for i in range(30):
            if i not in ind:
                for j in hist[i]:
                    held[j] = -1
                    nm -= 1
    return np.array(held, np.int32), nm


def test_oracle_matches_brute_force_model(oracle):
    for seed, th, tz, chk, cluster in ((1, 15, 0.0, True, False), (2, 15, 0.3, True, False), (3, 7, -0.3, False, False), (4, 60, 0.3, True, True)):
        p = ms.params(None, th, chk, dtype=oracle_lib.MATCH_PARAMS_DTYPE)
        cur, last, Tc, Tl = ms.random_pair(seed, p, n_cur=300, n_last=280, tz=tz, cluster=cluster)
        mo, no = oracle_lib.search_by_projection(p, cur, last, Tc, Tl)
        mb, nb = brute_force(p, cur, last, Tc, Tl)
        assert no == nb and np.array_equal(mo, mb), (seed, no, nb)
        assert no > 30


def test_greedy_blocking_and_overwrite_known_answer(oracle):
    """Two last-frame points compete for one current keypoint: a holder with Observations() > 0 blocks the later point (which then
    takes its second-best candidate); a holder without observations is overwritten (nmatches still counts both assignments)."""
    p = ms.params(None, 15, False, dtype=oracle_lib.MATCH_PARAMS_DTYPE)
    kps = np.zeros(2, ms.KEYPOINT_DTYPE)
    kps["x"], kps["y"], kps["octave"] = [100, 104], [100, 100], [0, 0]
    xy = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
    dA = np.zeros(32, np.uint8); dB = np.zeros(32, np.uint8); dB[0] = 0xFF   # 8 bits apart
    cur = dict(kps=kps, un_xy=xy, uright=np.full(2, -1, np.float32), grid_cell=ms.grid_cells(xy, p), desc=np.stack([dA, dB]))
    fx, fy, cx, cy = (float(p[k][0]) for k in ("fx", "fy", "cx", "cy"))
    pt = np.array([[(101 - cx) * 2 / fx, (100 - cy) * 2 / fy, 2.0]], np.float32)
    for obs0, want, nm in ((2, [0, 1], 2), (0, [1, -1], 2)):
        last = dict(xyz=np.repeat(pt, 2, 0), desc=np.stack([dA, dA]), flags=np.array([1 | obs0, 1 | 2], np.uint8), octave=np.zeros(2, np.int32),
                    angle=np.zeros(2, np.float32))
        m, n = oracle_lib.search_by_projection(p, cur, last, np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32))
        assert list(m) == want and n == nm


def test_golden_match(oracle):
    """Committed golden vectors of SearchByProjection (tests/golden/make_golden.py)."""
    import hashlib
    import os
    from tests import oracle_lib
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "match_pairs.npz"))
    p = ms.params(None, float(g["th"]), True, dtype=oracle_lib.MATCH_PARAMS_DTYPE)
    for name in ("a", "b"):
        seed, nc, nl, cluster = (int(v) for v in g[f"{name}_spec"])
        c, l, Tc, Tl = ms.random_pair(seed, p, n_cur=nc, n_last=nl, tz=float(g[f"{name}_tz"]), cluster=bool(cluster))
        assert hashlib.sha256(c["desc"].tobytes() + l["desc"].tobytes() + l["xyz"].tobytes()).hexdigest() == str(g[f"{name}_input_sha256"])
        out, nm = oracle_lib.search_by_projection(p, c, l, Tc, Tl)
        assert nm == int(g[f"{name}_nmatches"]) and np.array_equal(out, g[f"{name}_matches"])
    assert int(g["a_nmatches"]) > 300 and int(g["b_nmatches"]) > 300
