import sys, time, os, threading
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from manhattanslam_amd import peac, synth
I = synth.ICL
W, H = 640, 480
d = np.stack([synth.depth_u16(synth.surfel_frame(k, intr=I, dropout=0.001)[1]) for k in range(32)])
d16 = torch.from_numpy(np.concatenate([d, d]).view(np.int16)).cuda().contiguous()
prm = peac.default_params()
hm = np.zeros((64, H // 2, W // 2), np.int32); hn = np.zeros(64, np.int32)
def call():
    t = time.perf_counter()
    peac.plane_membership_device(d16, 1, 64, W, H, I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1 / 5000.0), prm, hm, hn)
    return (time.perf_counter() - t) * 1e3
call()
print("alone      ", " ".join(f"{call():.0f}" for _ in range(30)))
# with a host->device copy of the membership after every call (what bench does)
ts = []
for _ in range(30):
    ts.append(call()); x = torch.from_numpy(hm).cuda()
print("with h2d   ", " ".join(f"{t:.0f}" for t in ts))
# with sleeps between calls (threads idle ~20 ms)
ts = []
for _ in range(30):
    ts.append(call()); time.sleep(0.02)
print("with sleeps", " ".join(f"{t:.0f}" for t in ts))
# with a busy GPU: big matmuls enqueued from another thread
stop = False
def gpu_load():
    a = torch.randn(4096, 4096, device="cuda")
    while not stop:
        for _ in range(20): a = (a @ a).clamp(-1, 1)
        torch.cuda.synchronize()
th = threading.Thread(target=gpu_load); th.start()
ts = [call() for _ in range(30)]
stop = True; th.join()
print("gpu busy   ", " ".join(f"{t:.0f}" for t in ts))
