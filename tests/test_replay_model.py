"""Host model of the deferred compaction's symbolic replay (manhattanslam_amd/csrc/msl_sf_map.hip, k_replay) against the literal per-keyframe
algorithm (new surfel k -> k-th largest hole else appended; back-to-front refill of the leftover holes: reference src/SurfelMapping.cpp:366-391).

Inside a window of keyframes the device moves nothing: elements keep their physical slot (base elements 0 .. n0 - 1, the k-th new surfel of keyframe
f = ext[f] + k), deleted slots are logged, and one wave replays the window's placements and tail moves on (virtual position <-> element) tables that
only hold what differs from the identity, plus one run per keyframe for the new surfels that were appended (clipped when the array shrinks; where a run
and an explicit entry cover the same position the later one wins).  `deferred()` below is that scheme statement for statement; the GPU parity tests
check the kernel itself."""
import random, sys

def literal_step(arr, deleted_pos, new_elems):
    """arr: list of element ids; deleted_pos: positions (any order) of holes; new_elems: ordered list of new ids."""
    d = sorted(deleted_pos)
    D, K, n = len(d), len(new_elems), len(arr)
    arr = list(arr)
    for k, e in enumerate(new_elems):
        if k < D: arr[d[D - 1 - k]] = e
        else: arr.append(e)
    if D > K:
        R = D - K
        for i in range(1, R + 1):
            hole, src = d[R - i], n - i
            if src != hole: arr[hole] = arr[src]
        arr = arr[:n - R]
    return arr

def deferred(n0, steps):
    """steps: list of (deleted_ids, K).  ids are physical: base 0..n0-1, new (f,k) -> ext[f]+k.  returns final virtual array as ids."""
    ext = [n0]
    for (_, K) in steps: ext.append(ext[-1] + K)
    LOC, VPOS = {}, {}          # vpos -> (id, stamp) ; id -> vpos
    runs = []                   # dicts: f, k0, vstart, cnt, stamp
    n = n0
    def vpos_of(i):
        if i in VPOS: return VPOS[i]
        if i < n0: return i
        for r in runs:
            base = ext[r['f']] + r['k0']
            if base <= i < base + r['cnt0']:
                return r['vstart'] + (i - base)
        raise AssertionError('no vpos for %d' % i)
    def loc_of(p, ):
        best = None
        if p in LOC: best = LOC[p]
        for r in reversed(runs):
            if r['vstart'] <= p < r['vstart'] + r['cnt']:
                if best is None or r['stamp'] > best[1]:
                    best = (ext[r['f']] + r['k0'] + (p - r['vstart']), r['stamp'])
                break
        if best is None:
            assert p < n0
            return p
        return best[0]
    for f, (dels, K) in enumerate(steps):
        stamp = f + 1
        d = sorted(vpos_of(i) for i in dels)
        D = len(d)
        for k in range(min(K, D)):
            t = d[D - 1 - k]; e = ext[f] + k
            LOC[t] = (e, stamp); VPOS[e] = t
        if K > D:
            runs.append(dict(f=f, k0=D, vstart=n, cnt=K - D, cnt0=K - D, stamp=stamp))
            n += K - D
        elif D > K:
            R = D - K; nFinal = n - R
            low = d[:R]
            import bisect
            cntLow = bisect.bisect_left(low, nFinal)
            srcs = []
            for a in range(cntLow):
                p = nFinal + a
                while True:
                    lb = bisect.bisect_left(low, p)
                    if lb < R and low[lb] == p: p = n - (R - lb)
                    else: break
                srcs.append(loc_of(p))
            for a in range(cntLow):
                LOC[low[a]] = (srcs[a], stamp); VPOS[srcs[a]] = low[a]
            for r in runs:
                if r['vstart'] + r['cnt'] > nFinal: r['cnt'] = max(0, nFinal - r['vstart'])
            n = nFinal
    return [loc_of(p) for p in range(n)], ext

def run_case(rng, n0, F, maxD, maxK, pre_holes=0):
    arr = list(range(n0))
    ext = n0
    steps = []
    live = set(arr)
    holes0 = set(rng.sample(range(n0), min(pre_holes, n0)))
    live -= holes0
    for f in range(F):
        K = rng.randint(0, maxK)
        D = rng.randint(0, min(maxD, len(live)))
        dels = set(rng.sample(sorted(live), D))
        if f == 0: dels |= holes0
        live -= dels
        new = [ext + k for k in range(K)]
        pos = [i for i, e in enumerate(arr) if e in dels]
        assert len(pos) == len(dels), (len(pos), len(dels))
        arr = literal_step(arr, pos, new)
        live |= set(new)
        ext += K
        steps.append((list(dels), K))
        assert set(arr) == live
    got, _ = deferred(n0, steps)
    assert got == arr, (n0, F, steps)



import pytest


@pytest.mark.parametrize("seed", [5, 6, 7])
def test_replay_equals_the_literal_loop(seed):
    rng = random.Random(seed)
    for it in range(1500):
        n0 = rng.choice([0, 1, 3, 10, 40, 200])
        run_case(rng, n0, rng.randint(1, 8), rng.choice([0, 2, 5, 30]), rng.choice([0, 2, 5, 30]), pre_holes=rng.choice([0, 0, 3]))


def test_replay_long_windows_and_heavy_churn():
    rng = random.Random(11)
    for it in range(60):
        run_case(rng, rng.choice([50, 300, 1000]), 32, rng.choice([3, 40, 200]), rng.choice([0, 3, 40, 200]), pre_holes=rng.choice([0, 10]))
