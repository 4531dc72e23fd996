#!/usr/bin/env python3
"""Regenerates the golden vectors under tests/golden/ from the CPU oracle.

The reference ships no golden vectors and cannot be built in this image (OpenCV/Eigen absent), so these
fixtures pin the *oracle's* outputs on seeded synthetic inputs (inputs are regenerated from the seed; only
a SHA-256 of the input is stored).  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from manhattanslam_amd import synth  # noqa: E402
from tests import oracle_lib  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
o = oracle_lib.load()

seed = synth.ORB_SEED + 5
img = synth.orb_frame(seed)
k, d = o.orb_create().extract(img)
np.savez_compressed(os.path.join(OUT, "orb_640x480.npz"), seed=seed, image_sha256=hashlib.sha256(img.tobytes()).hexdigest(),
                    keypoints=k, descriptors=d)
seed = 42
img = synth.orb_frame(seed, 400, 304)
k, d = o.orb_create(500, 1.2, 6, 25, 9).extract(img)
np.savez_compressed(os.path.join(OUT, "orb_400x304.npz"), seed=seed, image_sha256=hashlib.sha256(img.tobytes()).hexdigest(),
                    keypoints=k, descriptors=d)

I = synth.TUM1
sf = oracle_lib.OracleSurfel(640, 480, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
local = synth.surfel_map(30000, ref=2).astype(oracle_lib.SURFEL_DTYPE)
gray, depth, member, pose = synth.surfel_frame(2, variant="B")
lo, no = sf.fuse(2, gray, depth, member, pose, local)
changed = np.flatnonzero((lo.view(np.uint8).reshape(len(lo), -1) != local.view(np.uint8).reshape(len(lo), -1)).any(1))
np.savez_compressed(os.path.join(OUT, "surfel_640x480_B.npz"), frame=2, n_local=len(local),
                    depth_sha256=hashlib.sha256(depth.tobytes()).hexdigest(), new_surfels=no, changed_index=changed.astype(np.int32),
                    changed_surfels=lo[changed], seeds=sf.seeds(), index_sha256=hashlib.sha256(sf.index().tobytes()).hexdigest())
# organised cloud + initial PEAC block statistics of the same frame (16-bit depth, factor 1/5000)
d16 = np.clip(np.round(depth * 5000.0), 0, 65535).astype(np.uint16)
d16[200:320, 300:420] += 4000          # a box 0.8 m further away: depth discontinuities
cloud, stats = oracle_lib.peac_block_stats(d16, I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1.0 / 5000.0))
np.savez_compressed(os.path.join(OUT, "peac_640x480.npz"), frame=2, depth16_sha256=hashlib.sha256(d16.tobytes()).hexdigest(),
                    cloud_sha256=hashlib.sha256(cloud.tobytes()).hexdigest(), stats=stats)
print("golden vectors written:", os.listdir(OUT))
# the whole plane extractor (block PCA, clustering, erosion, region growing): ICL intrinsics, light dropout
I4 = synth.ICL
frame4, drop4 = 100, 0.001
_, depth4, _, _ = synth.surfel_frame(frame4, intr=I4, dropout=drop4)
mem4, npl4, blk4 = oracle_lib.peac_run(synth.depth_u16(depth4), I4["fx"], I4["fy"], I4["cx"], I4["cy"], np.float32(1.0 / 5000.0))
np.savez_compressed(os.path.join(OUT, "peac_membership_640x480.npz"), frame=frame4, dropout=drop4, nplanes=npl4, membership=mem4, blocks=blk4)
print("peac membership golden:", npl4, "planes")
# SearchByProjection on two seeded synthetic pairs (forward motion, clustered keypoints; no motion, sparse)
from tests import match_scenes as ms  # noqa: E402
mp = ms.params(None, 15.0, True, dtype=oracle_lib.MATCH_PARAMS_DTYPE)
gm = {}
for name, (seed, nc, nl, tz, cluster) in {"a": (101, 1000, 950, 0.3, True), "b": (102, 900, 850, 0.0, False)}.items():
    c, l, Tc, Tl = ms.random_pair(seed, mp, n_cur=nc, n_last=nl, tz=tz, cluster=cluster)
    out, nm = oracle_lib.search_by_projection(mp, c, l, Tc, Tl)
    gm[f"{name}_spec"] = np.array([seed, nc, nl, int(cluster)], np.int64); gm[f"{name}_tz"] = np.float64(tz)
    gm[f"{name}_matches"] = out; gm[f"{name}_nmatches"] = np.int32(nm)
    gm[f"{name}_input_sha256"] = hashlib.sha256(c["desc"].tobytes() + l["desc"].tobytes() + l["xyz"].tobytes()).hexdigest()
np.savez_compressed(os.path.join(OUT, "match_pairs.npz"), th=15.0, **gm)
print("match golden:", int(gm["a_nmatches"]), int(gm["b_nmatches"]), "matches")
# Round 4: the furnished room (curved objects, depth edges, z^2 noise, dropout) -- SurfelFusion on a mostly-in-view map, and the plane extractor
Ic = synth.TUM1
scn = synth.clutter_scene()
kc = 40
grayc, depthc, memberc, posec, _ = synth.clutter_frame(kc, scene=scn)
localc = synth.surfel_map_dense(10000, ref=kc, scene=scn, k_lo=kc - 25, k_hi=kc + 35, flip=0.05, floating=0.02, min_update_times=1).astype(oracle_lib.SURFEL_DTYPE)
sfc = oracle_lib.OracleSurfel(640, 480, Ic["fx"], Ic["fy"], Ic["cx"], Ic["cy"], 30.0, 0.5)
loc, noc = sfc.fuse(kc, grayc, depthc, memberc, posec, localc)
chg = np.flatnonzero((loc.view(np.uint8).reshape(len(loc), -1) != localc.view(np.uint8).reshape(len(loc), -1)).any(1))
memc, nplc, _ = oracle_lib.peac_run(synth.depth_u16(depthc), Ic["fx"], Ic["fy"], Ic["cx"], Ic["cy"], np.float32(1.0 / 5000.0))
np.savez_compressed(os.path.join(OUT, "clutter_640x480.npz"), frame=kc, n_local=len(localc), depth_sha256=hashlib.sha256(depthc.tobytes()).hexdigest(),
                    gray_sha256=hashlib.sha256(grayc.tobytes()).hexdigest(), map_sha256=hashlib.sha256(localc.tobytes()).hexdigest(),
                    new_surfels=noc, changed_index=chg.astype(np.int32), changed_surfels=loc[chg], seeds=sfc.seeds(),
                    index_sha256=hashlib.sha256(sfc.index().tobytes()).hexdigest(), peac_membership=memc.astype(np.int8), peac_nplanes=nplc)
print("clutter golden:", len(noc), "new,", len(chg), "changed surfels,", nplc, "planes, membership range", memc.min(), memc.max())
