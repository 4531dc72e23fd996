"""The checker half of bench.py's parity gate (tools/cpu_baseline.py --parity-gate; SURVEY.md 8(d) "parity gates in the same run") on CPU:
fed with results that ARE the oracle's it must say ok, fed with a wrong keypoint / a moved surfel / a changed counter it must say where."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--parity-gate", path], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_checker_accepts_the_oracles_results_and_names_every_kind_of_mismatch(oracle, tmp_path):
    import importlib.util
    from tests import oracle_lib
    spec = importlib.util.spec_from_file_location("msl_synth", os.path.join(ROOT, "manhattanslam_amd", "synth.py"))
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    I = synth.TUM1
    W, H, n, nk = 640, 480, 2, 2
    grays = np.stack([synth.orb_frame(synth.ORB_SEED + k) for k in range(n)])
    ex = oracle.orb_create(1000, 1.2, 8, 20, 7)
    cap = 1200
    kps = np.zeros((n, cap * 28), np.uint8); desc = np.zeros((n, cap * 32), np.uint8); cnt = np.zeros(n, np.int32)
    for f in range(n):
        k, d = ex.extract(grays[f])
        cnt[f] = len(k); kps[f, :len(k) * 28] = np.frombuffer(k.tobytes(), np.uint8); desc[f, :len(k) * 32] = d.reshape(-1)
    m0 = synth.surfel_map(30000, ref=0).astype(oracle_lib.SURFEL_DTYPE)
    frames = [synth.surfel_frame(k) for k in range(nk)]
    sf = oracle_lib.OracleSurfel(W, H, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
    sf.map_set(m0)
    for k in range(nk):
        sf.fuse_map(k, *frames[k])
    m1 = sf.map_get()
    base = dict(size=np.array([W, H], np.int32), intr=np.array([I["fx"], I["fy"], I["cx"], I["cy"]], np.float64), orb_gray=grays, orb_n=cnt, orb_kps=kps, orb_desc=desc,
                sf_map0=m0.view(np.uint8).reshape(len(m0), -1), sf_map_gpu=m1.view(np.uint8).reshape(len(m1), -1), sf_refs=np.arange(nk),
                sf_gray=np.stack([f[0] for f in frames]), sf_depth=np.stack([f[1] for f in frames]), sf_member=frames[0][2][None],
                sf_poses=np.stack([np.asarray(f[3], np.float32).reshape(16) for f in frames]))
    p = str(tmp_path / "gate.npz")
    np.savez(p, **base)
    res = _run(p)
    assert res["ok"] and res["orb_frames"] == n and res["keyframes"] == nk and res["max_abs"] == 0.0 and res["bit_identical"], res

    def broken(**change):
        d = dict(base); d.update(change)
        np.savez(p, **d)
        return _run(p)

    bad = desc.copy(); bad[1, 5 * 32 + 3] ^= 1
    r = broken(orb_desc=bad)
    assert not r["ok"] and "ORB frame 1" in r["failures"][0], r
    r = broken(orb_n=cnt - np.array([0, 1], np.int32))
    assert not r["ok"] and "keypoints" in r["failures"][0], r
    mm = m1.copy(); mm["pz"][123] += 3e-4
    r = broken(sf_map_gpu=mm.view(np.uint8).reshape(len(mm), -1))
    assert not r["ok"] and "pz" in r["failures"][0] and r["max_abs"] > 2e-4, r
    mm = m1.copy(); mm["px"][77] += 5e-5                      # inside the tolerance: accepted, reported
    r = broken(sf_map_gpu=mm.view(np.uint8).reshape(len(mm), -1))
    assert r["ok"] and 4e-5 < r["max_abs"] < 1e-4 and not r["bit_identical"], r
    mm = m1.copy(); mm["updateTimes"][5] += 1
    r = broken(sf_map_gpu=mm.view(np.uint8).reshape(len(mm), -1))
    assert not r["ok"] and "updateTimes" in r["failures"][0], r
    r = broken(sf_map_gpu=m1[:-1].view(np.uint8).reshape(len(m1) - 1, -1))
    assert not r["ok"] and "surfels" in r["failures"][0], r
