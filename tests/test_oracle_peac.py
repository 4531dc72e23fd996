"""CPU: the PEAC-front oracle (oracle/peac_oracle.cpp) against closed forms -- SURVEY.md 8(f) rank 2."""
import numpy as np

from tests.oracle_lib import peac_block_stats

FX, FY, CX, CY, FACTOR = 517.3, 516.5, 318.6, 255.3, 1.0 / 5000.0


def test_constant_depth_block_sums_closed_form():
    H, W = 40, 60
    d = np.full((H, W), 5000, np.uint16)                     # 1 m everywhere (up to the float factor)
    cloud, st = peac_block_stats(d, FX, FY, CX, CY, FACTOR)
    assert cloud.shape == (20 * 30, 3) and len(st) == 2 * 3
    z = float(np.float64(5000) * np.float32(FACTOR))
    assert np.all(cloud[:, 2] == z)
    # vertex (r, c): x = ((double)(2c) - cx) * z / fx in double with float intrinsics
    r, c = 7, 11
    x = (np.float64(2 * c) - np.float64(np.float32(CX))) * z / np.float64(np.float32(FX))
    y = (np.float64(2 * r) - np.float64(np.float32(CY))) * z / np.float64(np.float32(FY))
    assert cloud[r * 30 + c, 0] == x and cloud[r * 30 + c, 1] == y
    assert np.all(st["N"] == 100) and np.all(st["nouse"] == 0)
    b = st[1 * 3 + 2]                                        # block row 1, column 2: rows 10..19, cols 20..29
    blk = cloud.reshape(20, 30, 3)[10:20, 20:30].reshape(-1, 3)
    # sequential sums in window raster order
    acc = np.zeros(9)
    for x, y, zz in blk:
        acc += [x, y, zz, x * x, y * y, zz * zz, x * y, y * zz, x * zz]
    got = [b[k] for k in ("sx", "sy", "sz", "sxx", "syy", "szz", "sxy", "syz", "sxz")]
    assert list(acc) == got


def test_missing_data_and_discontinuity_reject_blocks():
    H, W = 40, 40
    d = np.full((H, W), 10000, np.uint16)                    # 2 m
    d[4, 6] = 0                                              # missing point in block (0, 0): vertex (2, 3)
    d[24:, 20:] = 20000                                      # 4 m region starting at vertex row 12: step inside block row 1
    _, st = peac_block_stats(d, FX, FY, CX, CY, FACTOR)
    st = st.reshape(2, 2)
    assert st["nouse"][0, 0] == 1 and st["N"][0, 0] == 0 and st["sx"][0, 0] == 0          # INIT_STRICT: one missing point rejects
    assert st["nouse"][0, 1] == 0 and st["N"][0, 1] == 100
    assert st["nouse"][1, 1] == 1                                                         # 2 m -> 4 m step: |dz| = 2 > 0.04 * 2 + 0.02
    assert st["nouse"][1, 0] == 1     # the right neighbour of its last column lies across the step: neighbours outside the block count
    _, loose = peac_block_stats(d, FX, FY, CX, CY, FACTOR, init_loose=True)
    loose = loose.reshape(2, 2)
    assert loose["nouse"][0, 0] == 0 and loose["N"][0, 0] == 99                            # INIT_LOOSE tolerates < half missing


def test_blocks_that_do_not_fit_are_dropped():
    d = np.full((50, 70), 5000, np.uint16)                   # cloud 25 x 35 -> 2 x 3 blocks of 10 x 10
    cloud, st = peac_block_stats(d, FX, FY, CX, CY, FACTOR)
    assert cloud.shape[0] == 25 * 35 and len(st) == 6


def test_golden_peac():
    """Committed golden vector (tests/golden/make_golden.py) pins the oracle's behaviour."""
    import hashlib
    import os
    from manhattanslam_amd import synth
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "peac_640x480.npz"))
    _, depth, _, _ = synth.surfel_frame(int(g["frame"]), variant="B")
    d16 = np.clip(np.round(depth * 5000.0), 0, 65535).astype(np.uint16)
    d16[200:320, 300:420] += 4000
    assert hashlib.sha256(d16.tobytes()).hexdigest() == str(g["depth16_sha256"])
    I = synth.TUM1
    cloud, st = peac_block_stats(d16, I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1.0 / 5000.0))
    assert hashlib.sha256(cloud.tobytes()).hexdigest() == str(g["cloud_sha256"])
    assert st.tobytes() == g["stats"].tobytes()
