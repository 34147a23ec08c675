"""Phase times of k_pose_opt_vio<256> on the chained sequential replay from s_memtime probes inside the kernel.
Needs a library built with the probes: VIEO_EXTRA_HIPCC_FLAGS=-DVIEO_POSE_PROBE (touch csrc/pose_opt_vio.hip first: the
build only looks at modification times).  Output of round 2: profiles/r2g_pose_probe.txt."""
import sys, ctypes, numpy as np
sys.path.insert(0, "/root/repo")
from vieo_slam_amd import replay, _lib
n = 40
L = _lib.lib()
f = ctypes.CDLL(_lib.LIB_PATH).vieo_debug_pose_probe
out = (ctypes.c_ulonglong * 24)()
if len(sys.argv) > 1 and sys.argv[1] == "rig":  # the rig instance: 20 calls of the one-call rig tracker
    sys.argv = [sys.argv[0]] + sys.argv[2:]
    sys.path.insert(0, "/root/repo/tools")
    import run_rig_tracker
    f(out)  # (reset)
    run_rig_tracker.main()
    n = int(sys.argv[4]) + 1
else:
    seq = replay.Sequence(1, n)
    Rc = replay.ChainedReplay(seq, replay.HipStages())
    Rc.run(n)
f(out)
v = np.array(list(out), float)
names16 = ["", "", "", "", "", "", "", "", "", "", "", "(11) top of trial: sync + backup", "(12) solve", "(13) generic_errors: tid 0's edge",
           "(14) generic_errors: wait for the other edges", "(15) generic_errors: information products",
           "(16) wait for the other lanes' Jacobians", "[17] imu_error rotation rows (own clock)",
           "[18] imu_error p, v rows", "[19] prior_error", "[20] imu_linearize p, v rows", "[21] imu_linearize rotation rows",
           "[22] prior_linearize"]
names = ["loop/bookkeeping", "generic_errors", "visual linearize loop", "block_sum", "publish + imu/prior linearize", "H assembly", "backup + ldlt", "ns_inc", "(unused)", "trial: errors of all edges", "iteration tail"]
tot = v[:17].sum()
for i, nm in enumerate(names): print("%-46s %6.1f %%  %9.0f cycles per pose call" % (nm, 100 * v[i] / tot, v[i] / (2 * (n - 1))))
for i in (11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22):
    print("%-46s %6.1f %%  %9.0f cycles per pose call" % (names16[i], 100 * v[i] / tot, v[i] / (2 * (n - 1))))
print("total cycles per pose call %.0f = %.0f us at 2.4 GHz" % (tot / (2 * (n - 1)), tot / (2 * (n - 1)) / 2400))
