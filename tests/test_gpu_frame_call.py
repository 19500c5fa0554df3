"""GPU: the one-call-per-frame entry points of the C++ binding (csrc_torch/hgs_torch.cpp: render / render_pair, round 5) against
the statement-by-statement adapter they replace (hugs_amd/renderer/gs_renderer.py with HGS_FRAME_CALL=0) -- the mirror of
/root/reference/hugs/renderer/gs_renderer.py:20-161.  Same kernels on the same inputs: images, radii and visibility are equal
bit for bit; gradients agree up to the order of the backward's float atomics (test_gpu_parity.order_tol)."""
import numpy as np
import pytest
import torch

from hugs_amd import synthetic as syn
from test_gpu_configs import as_model, cam_data, human_gaussians, scene_model
from test_gpu_parity import order_tol, rel_l2, to_dev

pytestmark = pytest.mark.gpu

MODEL_KEYS = ("xyz", "opacity", "scales", "rotq", "shs")


def _models(device, n_human=4000, n_scene=9000, H=272, W=400, seed=3):
    cam = syn.pinhole_camera(H, W)
    hm = human_gaussians(n_human, seed=seed)
    hm["xyz"] = (hm["xyz"] + np.array([0.0, 0.0, 3.0], np.float32)).astype(np.float32)
    sm = scene_model(n_scene, cam, seed=seed + 1)
    rng = np.random.default_rng(seed)
    dL = [to_dev((rng.standard_normal((3, H, W)) * 1e-2).astype(np.float32), device) for _ in range(2)]
    return cam, hm, sm, dL


def _grads(model):
    return {k: (None if model[k].grad is None else model[k].grad.detach().clone()) for k in MODEL_KEYS}


def _need_cpp():
    import diff_gaussian_rasterization as dgr
    if dgr._load_cpp() is None:
        pytest.skip("the C++ binding is not built / not selected")


@pytest.mark.parametrize("feats", ["shs", "rgb"])
def test_render_as_one_call_equals_the_statement_path(feats, device, monkeypatch):
    _need_cpp()
    from hugs_amd.renderer import gs_renderer
    cam, hm, _, dL = _models(device)
    if feats == "rgb":
        hm["shs"] = np.ascontiguousarray(hm["shs"][:, 0])   # [P,3]: precomputed colours (gs_renderer.py:119-123)
    out = {}
    for frame_call in (True, False):
        monkeypatch.setattr(gs_renderer, "_FRAME_CALL", frame_call)
        m = as_model(hm, device, 0)
        pkg = gs_renderer.render(m["xyz"], m["shs"], m["opacity"], m["scales"], m["rotq"], cam_data(cam, device),
                                 bg_color=torch.full((3,), 0.25, device=device), active_sh_degree=0)
        assert (pkg["render"].grad_fn.name() == "HgsRasterizeBackward") and pkg["viewspace_points"].is_leaf
        assert float(pkg["viewspace_points"].detach().abs().max()) == 0.0 and pkg["viewspace_points"].requires_grad
        assert pkg["visibility_filter"].dtype == torch.bool and pkg["radii"].dtype == torch.int32
        pkg["render"].backward(dL[0])
        out[frame_call] = (pkg, _grads(m), pkg["viewspace_points"].grad.clone())
    a, b = out[True], out[False]
    for k in ("render", "radii", "visibility_filter"):
        assert torch.equal(a[0][k], b[0][k]), k
    assert torch.equal(a[0]["visibility_filter"], a[0]["radii"] > 0)
    for k in MODEL_KEYS:
        assert rel_l2(a[1][k].cpu().numpy(), b[1][k].cpu().numpy()) <= order_tol("rotations" if k == "rotq" else k), k
    assert rel_l2(a[2].cpu().numpy(), b[2].cpu().numpy()) <= order_tol("means2D")


def test_viewspace_leaves_share_zero_storage_but_not_gradients(device):
    """Every frame's viewspace tensor is a fresh leaf over one zero-filled buffer (no fill kernel per frame): its .grad is its own."""
    _need_cpp()
    from hugs_amd.renderer import gs_renderer
    cam, hm, _, dL = _models(device)
    m = as_model(hm, device, 0)
    pk = [gs_renderer.render(m["xyz"], m["shs"], m["opacity"], m["scales"], m["rotq"], cam_data(cam, device), active_sh_degree=0) for _ in range(2)]
    v0, v1 = pk[0]["viewspace_points"], pk[1]["viewspace_points"]
    assert v0 is not v1 and v0.data_ptr() == v1.data_ptr() and v0.shape == (hm["xyz"].shape[0], 3)
    pk[0]["render"].backward(dL[0])
    assert v0.grad is not None and v1.grad is None
    pk[1]["render"].backward(2.0 * dL[0])
    assert rel_l2(v1.grad.cpu().numpy(), 2.0 * v0.grad.cpu().numpy()) <= order_tol("means2D")
    assert float(v0.detach().abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        v0.add_(1.0)   # autograd refuses in-place operations on a leaf that requires grad: the shared zeros stay zeros


@pytest.mark.parametrize("loss", ["both", "joint_only", "human_only", "two_backwards"])
def test_both_renders_of_a_step_as_one_node_equal_two_renders(loss, device, monkeypatch):
    """render_human_scene(render_human_separate=True): ONE call, one node, the human-only frame on the library's side stream and its
    gradients of the human tensors summed inside the joint frame's per-Gaussian kernel -- against the two-render statement path
    (autograd sums the two nodes' gradients).  `loss`: which images take part, and whether through one backward or two."""
    _need_cpp()
    from hugs_amd.renderer import gs_renderer, render_human_scene
    cam, hm, sm, dL = _models(device)
    bg, hbg = torch.tensor([0.1, 0.6, 0.9], device=device), torch.tensor([0.7, 0.2, 0.3], device=device)
    out = {}
    for frame_call in (True, False):
        monkeypatch.setattr(gs_renderer, "_FRAME_CALL", frame_call)
        human, scene = as_model(hm, device, 0), as_model(sm, device, 3)
        pkg = render_human_scene(cam_data(cam, device), human, scene, bg_color=bg, human_bg_color=hbg, render_mode="human_scene",
                                 render_human_separate=True)
        if frame_call:
            assert pkg["render"].grad_fn is pkg["human_img"].grad_fn and pkg["render"].grad_fn.name() == "HgsRasterizeBackward"
        if loss == "both":
            torch.autograd.backward([pkg["render"], pkg["human_img"]], dL)
        elif loss == "joint_only":
            pkg["render"].backward(dL[0])
        elif loss == "human_only":
            pkg["human_img"].backward(dL[1])
        else:
            pkg["render"].backward(dL[0], retain_graph=True)
            pkg["human_img"].backward(dL[1])
        torch.cuda.synchronize()
        vs = pkg["viewspace_points"].grad
        out[frame_call] = (pkg, _grads(human), _grads(scene), None if vs is None else vs.clone())
    a, b = out[True], out[False]
    nh = hm["xyz"].shape[0]
    for k in ("render", "radii", "visibility_filter", "human_img", "human_radii", "human_visibility_filter", "scene_radii",
              "scene_visibility_filter"):
        assert torch.equal(a[0][k], b[0][k]), k
    assert torch.equal(a[0]["scene_radii"], a[0]["radii"][nh:]) and a[0]["human_radii"].shape[0] == nh
    for k in MODEL_KEYS:
        tol = order_tol("rotations" if k == "rotq" else k)
        assert rel_l2(a[1][k].cpu().numpy(), b[1][k].cpu().numpy()) <= tol, f"human {k}"
        if loss == "human_only":
            assert a[2][k] is None or float(a[2][k].abs().max()) == 0.0
        else:
            assert rel_l2(a[2][k].cpu().numpy(), b[2][k].cpu().numpy()) <= tol, f"scene {k}"
    if loss == "human_only":   # the joint render's screen-space gradient is the one the trainer reads (gs_trainer.py:316-342): none here
        assert a[3] is None or float(a[3].abs().max()) == 0.0
    else:
        assert rel_l2(a[3].cpu().numpy(), b[3].cpu().numpy()) <= order_tol("means2D")


def test_one_node_backward_twice_and_after_release(device):
    _need_cpp()
    from hugs_amd.renderer import render_human_scene
    cam, hm, sm, dL = _models(device, n_human=1500, n_scene=3000)
    human, scene = as_model(hm, device, 0), as_model(sm, device, 3)
    pkg = render_human_scene(cam_data(cam, device), human, scene, bg_color=torch.ones(3, device=device), render_mode="human_scene",
                             render_human_separate=True)
    torch.autograd.backward([pkg["render"], pkg["human_img"]], dL, retain_graph=True)
    first = _grads(human)
    for v in list(human.values()) + list(scene.values()):
        if torch.is_tensor(v):
            v.grad = None
    torch.autograd.backward([pkg["render"], pkg["human_img"]], dL)   # (fresh slabs with zeroed accumulators)
    for k in MODEL_KEYS:
        assert rel_l2(human[k].grad.cpu().numpy(), first[k].cpu().numpy()) <= order_tol("rotations" if k == "rotq" else k), k
    with pytest.raises(RuntimeError, match="retain_graph"):
        pkg["render"].backward(dL[0])


def test_input_modified_in_place_after_the_forward_is_caught(device):
    _need_cpp()
    from hugs_amd.renderer import render
    cam, hm, _, dL = _models(device, n_human=800)
    m = as_model(hm, device, 0)
    x = m["xyz"] * 1.0   # (a non-leaf the test may write into)
    pkg = render(x, m["shs"], m["opacity"], m["scales"], m["rotq"], cam_data(cam, device), active_sh_degree=0)
    x.mul_(2.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        pkg["render"].backward(dL[0])


def test_frame_calls_under_no_grad_build_no_graph_and_feed_the_deferred_path(device):
    _need_cpp()
    import diff_gaussian_rasterization as dgr
    from hugs_amd.renderer import render, render_batch, render_human_scene
    cam, hm, sm, _ = _models(device, n_human=1200, n_scene=2500)
    human, scene = as_model(hm, device, 0), as_model(sm, device, 3)
    with torch.no_grad():
        pkg = render_human_scene(cam_data(cam, device), human, scene, bg_color=torch.ones(3, device=device), render_mode="human_scene",
                                 render_human_separate=True)
        one = render(human["xyz"], human["shs"], human["opacity"], human["scales"], human["rotq"], cam_data(cam, device),
                     bg_color=torch.ones(3, device=device), active_sh_degree=0)
    assert pkg["render"].grad_fn is None and pkg["human_img"].grad_fn is None and one["render"].grad_fn is None
    assert torch.equal(one["render"], pkg["human_img"])
    # the shape's record lives in the C++ binding; the deferred (forward-only, pipelined) path finds it there
    key = (device.index, hm["xyz"].shape[0], cam["image_height"], cam["image_width"])
    assert dgr._cpp.get_hint(*key) is not None
    fr = {"means3D": human["xyz"].detach(), "feats": human["shs"].detach(), "opacity": human["opacity"].detach(),
          "scales": human["scales"].detach(), "rotations": human["rotq"].detach(), "data": cam_data(cam, device),
          "bg_color": torch.ones(3, device=device), "active_sh_degree": 0}
    batch = render_batch([fr, fr])
    assert torch.equal(batch[0]["render"], one["render"]) and torch.equal(batch[1]["radii"], one["radii"])


@pytest.mark.parametrize("entry", ["module", "render", "pair"])
def test_a_dropped_frame_takes_its_node_and_scratch_with_it(entry, device):
    """The image owns the node, the node must not own the image: frames rendered with a graph and dropped -- with or without a
    backward -- leave no memory behind (a reference cycle image -> node -> image kept every frame's scratch alive)."""
    _need_cpp()
    import gc
    from hugs_amd.renderer import render, render_human_scene
    from test_gpu_parity import run_gpu
    from scenes import CASES, make_scene
    cam, hm, sm, dL = _models(device, n_human=1500, n_scene=3000)
    human, scene = as_model(hm, device, 0), as_model(sm, device, 3)
    sc = make_scene(**CASES["basic_d3"])

    def frame(backward):
        if entry == "module":
            _, color, _ = run_gpu(sc, device)
            g = to_dev(sc["dL_dpix"], device)
        elif entry == "render":
            color = render(human["xyz"], human["shs"], human["opacity"], human["scales"], human["rotq"], cam_data(cam, device), active_sh_degree=0)["render"]
            g = dL[0]
        else:
            color = render_human_scene(cam_data(cam, device), human, scene, bg_color=torch.ones(3, device=device), render_mode="human_scene",
                                       render_human_separate=True)["render"]
            g = dL[0]
        if backward:
            color.backward(g)
            for v in list(human.values()) + list(scene.values()):
                if torch.is_tensor(v):
                    v.grad = None

    for backward in (False, True):
        for _ in range(3):
            frame(backward)
        gc.collect()
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated(device)
        for _ in range(12):
            frame(backward)
        gc.collect()
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated(device) <= before + (1 << 16), (entry, backward, torch.cuda.memory_allocated(device) - before)


def test_a_training_loop_of_step_pairs_stays_put(device):
    """Frame after frame of the same step: from the third on, the human-only frame is enqueued without a wait for N and both frames'
    scratch (binning entries, checkpoint slots) follows what the frames before used -- guesses that N's drift from camera to camera
    can break, which the scan kernel must catch.  Every step's images and gradients equal the first step's from that camera."""
    _need_cpp()
    import math
    from hugs_amd.renderer import render_human_scene
    cam, hm, sm, dL = _models(device, n_human=6000, n_scene=12000, H=400, W=560)
    human, scene = as_model(hm, device, 0), as_model(sm, device, 3)
    bg = torch.ones(3, device=device)

    def camera(k):   # a small orbit: N moves by a few per cent between neighbours
        yaw = math.radians(1.5) * (k % 4)
        w2c = np.eye(4)
        w2c[0, 0], w2c[0, 2], w2c[2, 0], w2c[2, 2] = math.cos(yaw), math.sin(yaw), -math.sin(yaw), math.cos(yaw)
        return cam_data(syn.camera_from_w2c(w2c, cam["fovx"], cam["fovy"], cam["image_height"], cam["image_width"]), device)

    def step(k):
        pkg = render_human_scene(camera(k), human, scene, bg_color=bg, render_mode="human_scene", render_human_separate=True)
        torch.autograd.backward([pkg["render"], pkg["human_img"]], dL)
        out = (pkg["render"].detach().clone(), pkg["human_img"].detach().clone(), pkg["radii"].clone(), _grads(human), _grads(scene),
               pkg["viewspace_points"].grad.clone())
        for v in list(human.values()) + list(scene.values()):
            if torch.is_tensor(v):
                v.grad = None
        return out

    first = {}
    for k in range(24):
        got = step(k)
        ref = first.setdefault(k % 4, got)
        if ref is got:
            continue
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]), k
        for mine, theirs in ((got[3], ref[3]), (got[4], ref[4])):
            for key in MODEL_KEYS:
                assert rel_l2(mine[key].cpu().numpy(), theirs[key].cpu().numpy()) <= order_tol("rotations" if key == "rotq" else key), (k, key)
        assert rel_l2(got[5].cpu().numpy(), ref[5].cpu().numpy()) <= order_tol("means2D"), k


def test_an_argument_error_in_the_step_pair_leaves_the_streams_usable(device):
    """The human-only frame is already running on the library's side stream when the joint frame's arguments are checked: an error
    there must fence the caller's stream behind the side stream before the buffers go (and the next step must be right)."""
    _need_cpp()
    from hugs_amd.renderer import render_human_scene
    cam, hm, sm, dL = _models(device, n_human=1500, n_scene=3000)
    human, scene = as_model(hm, device, 0), as_model(sm, device, 3)
    bg = torch.ones(3, device=device)
    good = render_human_scene(cam_data(cam, device), human, scene, bg_color=bg, render_mode="human_scene", render_human_separate=True)
    bad_scene = dict(scene)
    bad_scene["opacity"] = scene["opacity"][:-5]      # rows do not match means3D: the second set is validated before it reaches the kernels
    for _ in range(3):
        with pytest.raises(RuntimeError, match="rows"):
            render_human_scene(cam_data(cam, device), human, bad_scene, bg_color=bg, render_mode="human_scene", render_human_separate=True)
    again = render_human_scene(cam_data(cam, device), human, scene, bg_color=bg, render_mode="human_scene", render_human_separate=True)
    torch.cuda.synchronize()
    for k in ("render", "human_img", "radii", "human_radii"):
        assert torch.equal(good[k], again[k]), k
