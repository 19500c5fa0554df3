"""GPU: the BASELINE.json configurations other than the bench workload, as parity-test cases (each against the
oracle with the same tolerances as tests/test_gpu_parity.py), driven through the renderer adapter the trainer uses.

  C1  10k random Gaussians, 256x256, single camera
  C3  HUGS human-only: SMPL-sized (6 890) and subdivided (110 210) Gaussian sets, [P,16,3] SH at active degree 0,
      512x512, the canonical rotating-camera rig (dist 5, fov 0.4)
  C4  HUGS joint human+scene at 1080p, degree 0 taken from the human model, two renders per step
      (full set on a random background + human only on its own background), one backward through both
  C5  one frame of the 300k-Gaussian / 1080p batch from a yawed camera (frames are independent; sharding is
      covered by tests/test_sharding_gloo.py)
"""
import math
import os

import numpy as np
import pytest
import torch

from hugs_amd import synthetic as syn
from oracle import hgs_oracle as ho
from scenes import make_scene
from test_gpu_parity import GRAD_REL_TOL, check_image, rel_l2, run_gpu, to_dev

pytestmark = pytest.mark.gpu


def human_gaussians(P, seed):
    """A person-sized blob at the origin (y up/down extent ~1.7 m), small isotropic-ish splats, non-unit quats
    (the HUGS models feed un-normalised quaternions, hugs_trimlp.py:517-518)."""
    rng = np.random.default_rng(seed)
    means = (rng.standard_normal((P, 3)) * np.array([0.22, 0.55, 0.14])).astype(np.float32)
    s0 = 0.035 / math.sqrt(P / 6890.0)
    scales = (s0 * np.exp(0.3 * rng.standard_normal((P, 3)))).astype(np.float32)
    q = rng.standard_normal((P, 4))
    q = q / np.linalg.norm(q, axis=1, keepdims=True) * rng.uniform(0.8, 1.2, (P, 1))
    shs = np.zeros((P, 16, 3), np.float32)
    shs[:, 0] = rng.standard_normal((P, 3))
    shs[:, 1:] = 0.1 * rng.standard_normal((P, 15, 3))
    opac = rng.uniform(0.05, 1.0, (P, 1)).astype(np.float32)
    return {"xyz": means, "scales": scales, "rotq": q.astype(np.float32), "shs": shs, "opacity": opac}


def scene_model(P, cam0, seed, sigma_px=4.0):
    g = syn.scene_gaussians(P, cam0, seed=seed, sigma_px=sigma_px)
    return {"xyz": g["means3D"], "scales": g["scales"], "rotq": g["rotations"], "shs": g["shs"], "opacity": g["opacities"]}


def as_model(m, device, degree):
    out = {k: to_dev(v, device, True) for k, v in m.items()}
    out["active_sh_degree"] = degree
    return out


def cam_data(cam, device):
    return {k: (to_dev(v, device) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}


def oracle_run(m, cam, bg, degree, dL):
    inp = ho.Inputs(m["xyz"], m["opacity"], cam["world_view_transform"], cam["full_proj_transform"], cam["camera_center"],
                    math.tan(cam["fovx"] * 0.5), math.tan(cam["fovy"] * 0.5), cam["image_height"], cam["image_width"],
                    np.asarray(bg, np.float32), shs=m["shs"], scales=m["scales"], rotations=m["rotq"], sh_degree=degree)
    ho.set_threads(ho.usable_cpus())
    f = ho.forward(inp)
    return f, ho.backward(inp, f, dL)


def check_grads(model, ref_g, what, sl=slice(None)):
    for k, rk in (("xyz", "means3D"), ("opacity", "opacities"), ("shs", "shs"), ("scales", "scales"), ("rotq", "rotations")):
        r = ref_g[rk][sl]
        g = model[k].grad.cpu().numpy().reshape(r.shape)
        assert rel_l2(g, r) <= GRAD_REL_TOL, f"{what}: grad {k} rel L2 {rel_l2(g, r):.2e}"


def test_c1_random_10k_256(device):
    from hugs_amd.renderer import render
    cam = syn.pinhole_camera(256, 256)
    m = scene_model(10_000, cam, seed=11, sigma_px=2.0)
    dL = np.random.default_rng(0).standard_normal((3, 256, 256)).astype(np.float32)
    ref_f, ref_g = oracle_run(m, cam, (0, 0, 0), 3, dL)
    mod = as_model(m, device, 3)
    pkg = render(mod["xyz"], mod["shs"], mod["opacity"], mod["scales"], mod["rotq"], cam_data(cam, device),
                 bg_color=None, active_sh_degree=3)
    assert np.array_equal(pkg["radii"].cpu().numpy(), ref_f["radii"])
    check_image(pkg["render"].detach().cpu().numpy(), np.clip(ref_f["color"], 0, 1), "C1")
    # backward through the clamp of render(): pass-through where the unclamped colour is inside (0,1)
    inside = ((ref_f["color"] >= 0) & (ref_f["color"] <= 1)).astype(np.float32)
    _, ref_g = oracle_run(m, cam, (0, 0, 0), 3, dL * inside)
    pkg["render"].backward(to_dev(dL, device))
    check_grads(mod, ref_g, "C1")
    assert rel_l2(pkg["viewspace_points"].grad.cpu().numpy(), ref_g["means2D"]) <= GRAD_REL_TOL


@pytest.mark.parametrize("P", [6890, 110_210])
def test_c3_human_only_512(P, device):
    from hugs_amd.renderer import render_human_scene
    cam = syn.rotating_camera(3, 10, dist=5.0, fov=0.4, img_size=512)
    m = human_gaussians(P, seed=5)
    dL = (np.random.default_rng(1).standard_normal((3, 512, 512)) * 1e-3).astype(np.float32)
    bg = (1.0, 1.0, 1.0)
    ref_f, _ = oracle_run(m, cam, bg, 0, dL)
    assert (ref_f["radii"] > 0).mean() > 0.9 and ref_f["N"] > P  # the rig really looks at the blob
    inside = ((ref_f["color"] >= 0) & (ref_f["color"] <= 1)).astype(np.float32)
    _, ref_g = oracle_run(m, cam, bg, 0, dL * inside)
    human = as_model(m, device, 0)
    pkg = render_human_scene(cam_data(cam, device), human, None, bg_color=torch.ones(3, device=device), render_mode="human")
    assert np.array_equal(pkg["human_radii"].cpu().numpy(), ref_f["radii"])
    assert np.array_equal(pkg["human_visibility_filter"].cpu().numpy(), ref_f["radii"] > 0)
    check_image(pkg["render"].detach().cpu().numpy(), np.clip(ref_f["color"], 0, 1), f"C3 P={P}")
    pkg["render"].backward(to_dev(dL, device))
    check_grads(human, ref_g, f"C3 P={P}")
    # degree 0 on 16-coefficient storage: coefficients above the active degree get exactly zero gradient
    assert float(human["shs"].grad[:, 1:].abs().max()) == 0.0


def _matrix_to_quaternion(m):
    """[n,3,3] -> (w,x,y,z), the branch-free half of the standard conversion (trace > -1 for every matrix used here);
    plain torch so that the same function runs on the GPU tensors and on the CPU reference chain"""
    w = 0.5 * torch.sqrt(torch.clamp(1.0 + m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2], min=1e-8))
    return torch.stack([w, (m[:, 2, 1] - m[:, 1, 2]) / (4 * w), (m[:, 0, 2] - m[:, 2, 0]) / (4 * w), (m[:, 1, 0] - m[:, 0, 1]) / (4 * w)], 1)


@pytest.mark.parametrize("P", [6890, 110_210])
def test_c3_lbs_posed_human_through_the_rasterizer(P, device):
    """BASELINE configs[2] is the "LBS -> rasterizer path": canonical Gaussians are posed by the learned-LBS skinning
    (hugs_trimlp.py:477-489, 517 -- hugs_amd.lbs.lbs_skin, fused HIP) and rendered by the rasterizer, and the loss
    gradient flows back through both to the canonical means, the LBS weights and the joint transforms.  Checked against
    the CPU chain: oracle rasterizer backward, then the LBS oracle's backward."""
    from hugs_amd.lbs import lbs_skin
    from hugs_amd.renderer import render_human_scene
    from oracle import lbs_oracle as lo
    cam = syn.rotating_camera(2, 10, dist=5.0, fov=0.4, img_size=512)
    m = human_gaussians(P, seed=9)
    rng = np.random.default_rng(4)
    J = 24
    A = np.tile(np.eye(4, dtype=np.float32), (J, 1, 1))
    for j in range(J):      # small rigid motions per joint (a pose), joints spread over the body
        A[j, :3, :3] = lo.batch_rodrigues(0.25 * rng.standard_normal((1, 3)).astype(np.float32))[0]
        A[j, :3, 3] = 0.05 * rng.standard_normal(3)
    centres = (rng.standard_normal((J, 3)) * np.array([0.22, 0.55, 0.14])).astype(np.float32)
    logit = -np.linalg.norm(m["xyz"][:, None] - centres[None], axis=-1) / 0.08
    W = np.exp(logit - logit.max(1, keepdims=True))
    W = (W / W.sum(1, keepdims=True)).astype(np.float32)
    R0 = lo.batch_rodrigues(rng.standard_normal((P, 3)).astype(np.float32))
    dL = (rng.standard_normal((3, 512, 512)) * 1e-3).astype(np.float32)

    tA, tW, tx, tR = (to_dev(a, device, True) for a in (A, W, m["xyz"], R0))
    xyz, T, rot = lbs_skin(tA, tW, tx, tR)
    rotq = _matrix_to_quaternion(rot)
    human = {"xyz": xyz, "rotq": rotq, "scales": to_dev(m["scales"], device, True), "shs": to_dev(m["shs"], device, True),
             "opacity": to_dev(m["opacity"], device, True), "active_sh_degree": 0}
    pkg = render_human_scene(cam_data(cam, device), human, None, bg_color=torch.ones(3, device=device), render_mode="human")
    pkg["render"].backward(to_dev(dL, device))

    # ---- CPU chain on the posed inputs the GPU produced
    posed = dict(m, xyz=xyz.detach().cpu().numpy(), rotq=rotq.detach().cpu().numpy())
    ref_f, _ = oracle_run(posed, cam, (1.0, 1.0, 1.0), 0, dL)
    assert np.array_equal(pkg["radii"].cpu().numpy(), ref_f["radii"]) and (ref_f["radii"] > 0).mean() > 0.9
    check_image(pkg["render"].detach().cpu().numpy(), np.clip(ref_f["color"], 0, 1), f"C3 LBS P={P}")
    inside = ((ref_f["color"] >= 0) & (ref_f["color"] <= 1)).astype(np.float32)
    _, ref_g = oracle_run(posed, cam, (1.0, 1.0, 1.0), 0, dL * inside)
    # posed skinning output of the HIP kernel against the LBS oracle
    o_xyz, o_T, o_rot = lo.skin(A, W, m["xyz"], R0)
    np.testing.assert_allclose(posed["xyz"], o_xyz, rtol=2e-5, atol=2e-6)
    # rasterizer gradients w.r.t. (posed xyz, rotq) -> through the quaternion conversion (torch, CPU) -> LBS oracle backward
    c_rot = torch.from_numpy(o_rot.astype(np.float64)).requires_grad_(True)
    _matrix_to_quaternion(c_rot).backward(torch.from_numpy(ref_g["rotations"].astype(np.float64)))
    ref = lo.skin_backward(A, W, m["xyz"], R0, ref_g["means3D"], None, c_rot.grad.numpy())
    for t, k in ((tA, "A"), (tW, "weights"), (tx, "v"), (tR, "rotmat")):
        assert rel_l2(t.grad.cpu().numpy().reshape(ref[k].shape), ref[k]) <= 2 * GRAD_REL_TOL, k


# (the statement-by-statement adapter -- HGS_FRAME_CALL=0, what this test runs for comparison -- renders the human-only frame on a side stream as
#  an autograd node of its own: torch then warns that the human tensors' AccumulateGrad nodes sit on another stream than that node.  Expected
#  there, harmless (the adapter fences the streams), and gone from the default path since round 5's render_pair: one node, one stream.)
@pytest.mark.filterwarnings("ignore:The AccumulateGrad node's stream does not match")
@pytest.mark.parametrize("joint", ["second_segment", "concat"])
@pytest.mark.parametrize("n_human,n_scene", [(30_000, 100_000), (110_210, 200_000)])
def test_c4_joint_human_scene_1080p(n_human, n_scene, joint, device, monkeypatch):
    """(110 210, 200 000) is the full BASELINE configs[3] size (SMPL subdivided twice, hugs_human.yaml:28 + the 200k scene;
    what tools/bench_c4.py times): both renders through render_human_scene with the side stream on, the sparse-frame
    backward on the human-only render, radii exact, images and every gradient against the oracle.  The joint render goes
    through the two-segment form of the C ABI (the scene as hgs_segment: nothing concatenated, gradients written in place)
    and, "concat", through the reference's own torch.cat form."""
    from hugs_amd.renderer import gs_renderer, render_human_scene
    monkeypatch.setattr(gs_renderer, "_JOINT_CONCAT", joint == "concat")
    H, W = 1080, 1920
    cam0 = syn.pinhole_camera(H, W)
    hm = human_gaussians(n_human, seed=7)
    hm["xyz"] = (hm["xyz"] + np.array([0.0, 0.0, 4.0], np.float32)).astype(np.float32)  # stand 4 m in front of the camera
    sm = scene_model(n_scene, cam0, seed=8)
    rng = np.random.default_rng(2)
    bg, hbg = rng.uniform(0, 1, 3).astype(np.float32), rng.uniform(0, 1, 3).astype(np.float32)
    dL1 = (rng.standard_normal((3, H, W)) * 1e-3).astype(np.float32)
    dL2 = (rng.standard_normal((3, H, W)) * 1e-3).astype(np.float32)
    joint = {k: np.concatenate([hm[k], sm[k]], 0) for k in hm}  # human first, scene second (gs_renderer.py:33-37)
    f1, _ = oracle_run(joint, cam0, bg, 0, dL1)
    f2, _ = oracle_run(hm, cam0, hbg, 0, dL2)
    in1 = ((f1["color"] >= 0) & (f1["color"] <= 1)).astype(np.float32)
    in2 = ((f2["color"] >= 0) & (f2["color"] <= 1)).astype(np.float32)
    _, g1 = oracle_run(joint, cam0, bg, 0, dL1 * in1)
    _, g2 = oracle_run(hm, cam0, hbg, 0, dL2 * in2)

    human, scene = as_model(hm, device, 0), as_model(sm, device, 3)  # the joint render uses the HUMAN's degree
    pkg = render_human_scene(cam_data(cam0, device), human, scene, bg_color=to_dev(bg, device),
                             human_bg_color=to_dev(hbg, device), render_mode="human_scene", render_human_separate=True)
    nh = hm["xyz"].shape[0]
    assert np.array_equal(pkg["radii"].cpu().numpy(), f1["radii"])
    assert np.array_equal(pkg["human_radii"].cpu().numpy(), f2["radii"])
    assert np.array_equal(pkg["scene_radii"].cpu().numpy(), f1["radii"][nh:])
    check_image(pkg["render"].detach().cpu().numpy(), np.clip(f1["color"], 0, 1), "C4 joint")
    check_image(pkg["human_img"].detach().cpu().numpy(), np.clip(f2["color"], 0, 1), "C4 human-only")
    (pkg["render"] * to_dev(dL1, device)).sum().backward(retain_graph=True)
    (pkg["human_img"] * to_dev(dL2, device)).sum().backward()
    # the human's parameters receive the sum of both renders' gradients; the scene's only the joint render's
    for k, rk in (("xyz", "means3D"), ("opacity", "opacities"), ("scales", "scales"), ("rotq", "rotations"), ("shs", "shs")):
        r = g1[rk][:nh] + g2[rk]
        assert rel_l2(human[k].grad.cpu().numpy().reshape(r.shape), r) <= GRAD_REL_TOL, k
        r = g1[rk][nh:]
        assert rel_l2(scene[k].grad.cpu().numpy().reshape(r.shape), r) <= GRAD_REL_TOL, k
    # only the FIRST render's screen-space gradient is what the trainer consumes (gs_trainer.py:316-342)
    assert rel_l2(pkg["viewspace_points"].grad.cpu().numpy(), g1["means2D"]) <= GRAD_REL_TOL


def test_c5_one_frame_of_the_300k_batch(device):
    from hugs_amd.renderer import render
    H, W, P = 1080, 1920, 300_000
    cam0 = syn.pinhole_camera(H, W)
    m = scene_model(P, cam0, seed=0)
    yaw = math.radians(1.5) * 5  # the camera rank 5 of an 8-GPU run renders
    w2c = np.eye(4)
    w2c[0, 0], w2c[0, 2], w2c[2, 0], w2c[2, 2] = math.cos(yaw), math.sin(yaw), -math.sin(yaw), math.cos(yaw)
    cam = syn.camera_from_w2c(w2c, cam0["fovx"], cam0["fovy"], H, W)
    ref_f, _ = oracle_run(m, cam, (1, 1, 1), 3, np.zeros((3, H, W), np.float32))
    mod = as_model(m, device, 3)
    with torch.no_grad():
        pkg = render(mod["xyz"], mod["shs"], mod["opacity"], mod["scales"], mod["rotq"], cam_data(cam, device),
                     bg_color=torch.ones(3, device=device), active_sh_degree=3)
    assert np.array_equal(pkg["radii"].cpu().numpy(), ref_f["radii"])
    check_image(pkg["render"].cpu().numpy(), np.clip(ref_f["color"], 0, 1), "C5 frame")


def test_storage_order_changes_nothing_but_speed(device):
    """hugs_amd.spatial.morton_order: the same Gaussians stored in Morton order render the same image -- bit for bit when no
    two of them share a depth (the tie-break is the index) -- and the same gradients once un-permuted (float atomics:
    summation order).  Long runs per (binning group, tile) are what the reordering buys."""
    from hugs_amd.spatial import morton_order, permute_model
    sc = make_scene(P=5000, H=270, W=480, seed=21, D=2, with_culled=True)
    order = morton_order(sc["means3D"])
    assert sorted(order.tolist()) == list(range(5000)) and not np.array_equal(order, np.arange(5000))
    assert np.array_equal(order, morton_order(torch.from_numpy(sc["means3D"]).to(device)).cpu().numpy())
    t0, c0, r0 = run_gpu(sc, device)
    c0.backward(to_dev(sc["dL_dpix"], device))
    sc1 = dict(sc)
    sc1.update(permute_model({k: sc[k] for k in ("means3D", "opacities", "shs", "scales", "rotations")}, order))
    t1, c1, r1 = run_gpu(sc1, device)
    c1.backward(to_dev(sc["dL_dpix"], device))
    assert torch.equal(c0, c1)
    assert torch.equal(r0[torch.from_numpy(order).to(device)], r1)
    o = torch.from_numpy(order).to(device)
    for k in ("means3D", "opacities", "shs", "scales", "rotations", "means2D"):
        a, b = t0[k].grad[o], t1[k].grad
        assert float((a - b).norm() / a.norm().clamp_min(1e-30)) <= (5e-5 if k == "rotations" else 1e-5), k   # (float atomics in another order)


@pytest.mark.gpu
def test_rccl_branch_of_the_bench_runs_as_a_process_group_of_one(device):
    """the RCCL branch of bench.py / hugs_amd.sharding (backend "nccl": process-group init with a device id,
    broadcast of the Gaussians, all_gather / all_reduce of the per-frame scalars on device tensors, barrier) had never executed
    on any box -- every recorded multi-rank run used gloo, because two ranks cannot share one GPU over RCCL.  A process group
    of ONE rank with the collectives forced on runs that code on a one-GPU lease: RCCL initialises, the collectives complete,
    the line carries the fields the scaling run is read by.  (configs[4]'s frames, gs_trainer.py:463,551,616.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HGS_BENCH_FORCE_PG="1", HGS_SHARDING_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--gaussians", "300000",
                        "--no-cpu-baseline", "--no-two-streams"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["backend"].startswith("nccl") and out["ranks_seen"] == 1 and out["expected_ranks"] == 1
    assert out["per_rank_N"] == out["config"]["N_per_frame_all_ranks"] and out["per_rank_N"][0] > 0 and out["value"] > 100.0
    assert "world_size 1 (nccl" in r.stderr


@pytest.mark.gpu
def test_the_wait_for_n_survives_sleeps_that_overshoot(device):
    """The library sleeps through a wait for N that the shape's record expects to be long (one rank = half a core instead of two: DESIGN 4.1 / 6).
    On a box whose sleeps overshoot (a loaded host: round 6's last 8-rank run) a fixed wake-up margin made every frame late AND taught the
    record a longer wait, which lengthened the next sleep: 400 us of overshoot took the bench workload from 2 000 to 77 frames/s.  The margin
    now follows the overshoot and an overslept wait is recorded as the time the sleep was MEANT to end.  HGS_WAIT_TEST_OVERSLEEP_US injects the
    overshoot; the frame rate must not care (a loose bound: boxes differ)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HGS_WAIT_SLEEP")}
    fps = {}
    for inject in ("0", "400"):
        env = dict(base, HGS_WAIT_TEST_OVERSLEEP_US=inject)
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "300", "--warmup", "30", "--no-cpu-baseline", "--no-two-streams"],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        fps[inject] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["value"]
    assert fps["400"] >= 0.8 * fps["0"], fps
