"""Would k_lba_schur's chunk skipping (k_lba_occ, used by the full BA) help the bench's local windows?  Counts, for the
r3 bench windows (bench.py: 40 fixed key frames, 10 free -- every 4th window 25 free), the 16-landmark chunks each
64-row tile of BB touches.  CPU only.  Round 4: every chunk of every tile row is occupied (a landmark of these windows
has 5.9 / 13.2 free observers out of 10 / 25, in no particular order), so the skip removes nothing there; the
useful / dense ratio of 0.24 (bench `roofline_mfma.useful_over_dense`) is sparsity INSIDE the chunks."""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vieo_slam_amd import synth_ba

for seed, nl in ((503, 25), (500, 10)):
    w = synth_ba.make_lba_vio_problem(seed, n_local=nl, n_fixed=40, n_points=2000)[:6]
    kfs, obs = w[1], w[4]
    free = np.where(kfs["fixed"] == 0)[0]
    slot = -np.ones(len(kfs), int)
    slot[free] = np.arange(len(free))
    n_mp = int(obs["mp"].max()) + 1
    nf = len(free)
    RB, nch = (6 * nf + 63) // 64, (n_mp + 15) // 16
    o = obs[slot[obs["kf"]] >= 0]
    a, ch = slot[o["kf"]], o["mp"] // 16
    occ = np.zeros((RB, nch), bool)
    for r in range(RB):
        sel = (a >= r * 64 // 6) & (a <= min((r * 64 + 63) // 6, nf - 1))
        occ[r, np.unique(ch[sel])] = True
    pairs = sum((occ[i] & occ[j]).sum() for i in range(RB) for j in range(i, RB)) / (nch * RB * (RB + 1) / 2)
    k = np.bincount(o["mp"], minlength=n_mp)
    print("free key frames %d: %d tile rows x %d chunks, occupied per tile row %s, tile pairs with work %.3f, "
          "free observers per landmark %.1f" % (nf, RB, nch, np.round(occ.mean(1), 3), pairs, k.mean()))
