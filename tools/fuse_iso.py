"""k_fuse / SurfelFusion micro benchmark on the dense-in-view map of bench.py, for kernel-variant A/B (MSL_LIB selects the library).
  python tools/fuse_iso.py [passes]      -> one JSON line: isolated k_fuse time (single stream), event time and keyframes/s with both streams
Run on the GPU box through gpurun; not part of the judged bench."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manhattanslam_amd import SurfelFusion, synth, SURFEL_DTYPE

P = int(sys.argv[1]) if len(sys.argv) > 1 else 3
W, H, I, Dn, B = 640, 480, synth.TUM1, 64, 32
grays, depths, member, poses, smap = synth.bench_inputs(0, Dn, 1000000, W, H, I)
smap = smap.astype(SURFEL_DTYPE)
dg, dd, dm = torch.from_numpy(grays).cuda(), torch.from_numpy(depths).cuda(), torch.from_numpy(member).cuda()


def run(one_stream):
    sf = SurfelFusion(W, H, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
    sf.set_batch_capacity(B); sf.map_reserve(2200000)
    sf.map_upload(smap); sf.map_snapshot()
    if one_stream:
        sf.set_stream(torch.cuda.current_stream().cuda_stream)
    names = sf.kernel_names()

    def one_pass():
        sf.map_restore()
        for sb in range(256 // B):
            f0 = (sb * B) % Dn
            sf.fuse_resident_batch(np.arange(sb * B, sb * B + B), dg[f0:], dd[f0:], dm, [poses[(f0 + k) % Dn] for k in range(B)], device=True, member_shared=True)
    one_pass(); sf.sync()
    sf.profile_enable((1 << names.index("k_fuse")) | (1 << names.index("k_empty")) | (1 << names.index("k_compact")))
    sf.profile_stride(5)
    t0 = time.perf_counter()
    for _ in range(P):
        one_pass()
    sf.sync()
    dt = time.perf_counter() - t0
    pr = sf.profile_read()
    out = {"kf_per_s": round(P * 256 / dt, 1)}
    for k in ("k_fuse", "k_empty", "k_compact"):
        ms, nl = pr[k]
        out[k + "_us"] = round(ms * 1e3 / max(nl, 1), 2)
    out["ctr"] = {k: int(v) for k, v in sf.counters().items()}
    sf.close()
    return out


only = os.environ.get("FUSE_ISO_ONLY")   # "one" / "two": just that run (counter passes)
res = {"lib": os.path.basename(os.environ.get("MSL_LIB", "default")), "defer": os.environ.get("MSL_SF_DEFER", "1")}
if only != "two": res["one_stream"] = run(True)
if only != "one": res["two_streams"] = run(False)
print(json.dumps(res))
