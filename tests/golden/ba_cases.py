"""The seeded cases behind tests/golden/ba_golden.npz: one function per path that maps a callable set (the CPU
oracle's, or the HIP product's) to the arrays that are pinned.  Shared by make_ba_golden.py (writes the file from
the oracle) and tests/test_golden_ba.py (re-runs the oracle on CPU, the HIP path on the GPU box)."""
import numpy as np

from vieo_slam_amd import synth_ba, synth_fisheye, tri_search


def nav_vec(nav):
    nav = np.atleast_1d(nav)
    return np.concatenate([nav["p"], nav["q"], nav["v"], nav["dbg"], nav["dba"]], axis=1)


def cases(api):
    """api: dict of callables with the oracle's signatures.  Yields (name, array, abs_tol_for_float_compare)."""
    fr, obs, _ = synth_ba.make_pose_problem(3, n_obs=240)
    res, outl = api["pose"](fr, obs)
    yield "pose_nav", nav_vec(res["nav"])[:, :7], 1e-4
    yield "pose_outliers", outl.astype(np.uint8), 0
    rig = synth_ba.camera_rig("kb8")
    fr, obs, _ = synth_ba.make_pose_problem(4, n_obs=240, rig=rig)
    res, outl = api["pose"](fr, obs)
    yield "pose_rig_nav", nav_vec(res["nav"])[:, :7], 1e-4
    yield "pose_rig_outliers", outl.astype(np.uint8), 0
    F, obs, _ = synth_ba.make_vio_problem(5, n_obs=260, compute_marg=True)
    res, outl = api["pose_vio"](F, obs)
    yield "vio_nav", nav_vec(res["base"]["nav"]), 1e-4
    yield "vio_outliers", outl.astype(np.uint8), 0
    yield "vio_marg_diag", np.diag(res["H_marg"].reshape(15, 15)) / np.abs(res["H_marg"]).max(), 1e-5
    w = synth_ba.make_lba_problem(6, n_local=5, n_fixed=3, n_points=500)
    navs, pts, erase, r = api["lba"](*w[:4])
    yield "lba_nav", nav_vec(navs)[:, :7], 1e-4
    yield "lba_n_erase", np.array([int(erase.sum())]), 4
    w = synth_ba.make_lba_vio_problem(7, n_local=5, n_fixed=3, n_points=500)
    navs, pts, erase, r = api["lba_vio"](*w[:6])
    yield "lba_vio_nav", nav_vec(navs), 1e-4
    w = synth_ba.make_lba_vio_problem(8, n_local=14, n_fixed=1, n_points=900, anchors=5, span=4)
    navs, pts, r = api["gba_vio"](w[0], w[1], w[2], w[4], w[5], 4, True)
    yield "gba_vio_nav", nav_vec(navs), 1e-4
    c = synth_fisheye.make_fisheye_case(9, rig="kb8", n_points=250)
    o = api["fisheye"](c["params"], c["keys"], c["descs"], c["num_mono"])
    yield "fisheye_groups", o["group_idx"].astype(np.int32), 0
    yield "fisheye_good", o["group_good"].astype(np.uint8), 0
    yield "fisheye_depth", o["depth"].astype(np.float32), 1e-4
    # encoder edges (a15-a18) and the triangulation search (8f-2), added after the first fixture set
    F, obs, _ = synth_ba.make_vio_problem(15, n_obs=200, compute_marg=True, enc=True)
    res, outl = api["pose_vio"](F, obs)
    yield "vio_enc_nav", nav_vec(res["base"]["nav"]), 1e-4
    yield "vio_enc_marg_diag", np.diag(res["H_marg"].reshape(15, 15)) / np.abs(res["H_marg"]).max(), 1e-5
    w = synth_ba.make_lba_problem(16, n_local=5, n_fixed=3, n_points=400)
    enc, edges = synth_ba.make_lba_enc(16, w[4], synth_ba.lba_enc_pairs(5, 8))
    navs, pts, erase, r = api["lba"](*w[:4], enc=enc)
    yield "lba_enc_nav", nav_vec(navs)[:, :7], 1e-4
    kf1, kf2s, _ = tri_search.make_tri_scene(17, n_points=400, n_neighbours=2, dup_frac=0.2)
    for p, (rows, nm) in enumerate(api["tri"](kf1, kf2s)):
        yield "tri_pairs_%d" % p, rows.astype(np.int32), 0
    kf1, kf2s, _ = tri_search.make_tri_scene(18, n_points=300, n_neighbours=1, rig="kb8")
    rows, nm = api["tri"](kf1, kf2s)[0]
    yield "tri_rig_rows", rows.astype(np.int32), 0
    yield "tri_rig_nmatches", np.array([nm], np.int32), 0
    # System::FinalGBA's form of the full BA (bScaleOpt = true: VertexScale + EdgeReprojectPRS / PRSStereo), round 3
    w = synth_ba.make_lba_vio_problem(19, n_local=12, n_fixed=1, n_points=800, anchors=4, span=4)
    navs, pts, r, scale = api["gba_vio"](w[0], w[1], (w[2] / np.float32(1.04)).astype(np.float32), w[4], w[5], 5, True,
                                         None, True)
    yield "gba_scale_nav", nav_vec(navs), 1e-4
    yield "gba_scale_scale", np.array([scale]), 1e-6
