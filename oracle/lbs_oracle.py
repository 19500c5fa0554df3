"""CPU restatement (numpy) of the skinning step of the reference's learned LBS -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/hugs/models/modules/lbs.py:19-73 (`lbs_extra`): rot_mats / pose_feature / pose_offsets
(:32-53), v_posed (:55-58), T = W @ A (:60-66), verts = (T @ [v_posed, 1])[:3] (:68-73) -- and the rotation product
that follows it in the human model, deformed_gs_rotmat = lbs_T[:, :3, :3] @ gs_rotmat
(/root/reference/hugs/models/hugs_trimlp.py:517).  Pinned against golden vectors produced by the reference's own
statements (tests/golden/make_golden.py compiles lbs_extra from its source file).  `batch_rodrigues` comes from the
third-party package smplx (pip dependency, absent from /root/reference, version unpinned by the reference's
scripts/conda_setup.sh): its published formula is restated below -- parity unpinned for that one function; it only
matters when posedirs are enabled (every release config sets disable_posedirs: true).
"""
import numpy as np


def batch_rodrigues(rot_vecs, dtype=np.float32):
    """smplx.lbs.batch_rodrigues: axis-angle [N,3] -> rotation matrices [N,3,3]:
    angle = ||r + 1e-8||, k = r / angle, R = I + sin(angle) K + (1 - cos(angle)) K K."""
    r = np.asarray(rot_vecs, dtype)
    angle = np.linalg.norm(r + dtype(1e-8), axis=1, keepdims=True).astype(dtype)
    k = (r / angle).astype(dtype)
    cos, sin = np.cos(angle)[:, None], np.sin(angle)[:, None]
    rx, ry, rz = k[:, 0], k[:, 1], k[:, 2]
    z = np.zeros_like(rx)
    K = np.stack([z, -rz, ry, rz, z, -rx, -ry, rx, z], 1).reshape(-1, 3, 3).astype(dtype)
    return (np.eye(3, dtype=dtype)[None] + sin * K + (1 - cos) * (K @ K)).astype(dtype)


def lbs_extra(A, v_shaped, posedirs, lbs_weights, pose, disable_posedirs=False, pose2rot=True, dtype=np.float32):
    """A [B,J,4,4], v_shaped [B,V,3], posedirs [P, V*3] or None, lbs_weights [V,J], pose [B,(J)*3] (or rotation matrices
    [B,J,3,3] with pose2rot=False) -> (verts [B,V,3], A, T [B,V,4,4], v_posed, v_shaped)."""
    A = np.asarray(A, dtype)
    v_shaped = np.asarray(v_shaped, dtype)
    B = A.shape[0]
    ident = np.eye(3, dtype=dtype)
    if disable_posedirs:
        v_posed = v_shaped
    else:
        if pose2rot:
            rot = batch_rodrigues(np.asarray(pose, dtype).reshape(-1, 3), dtype).reshape(B, -1, 3, 3)
        else:
            rot = np.asarray(pose, dtype).reshape(B, -1, 3, 3)
        feat = (rot[:, 1:] - ident).reshape(B, -1)
        v_posed = (feat @ np.asarray(posedirs, dtype)).reshape(B, -1, 3).astype(dtype) + v_shaped
    W = np.asarray(lbs_weights, dtype)
    J = A.shape[1]
    T = (W[None] @ A.reshape(B, J, 16)).reshape(B, -1, 4, 4).astype(dtype)
    homo = np.concatenate([v_posed, np.ones_like(v_posed[..., :1])], -1)
    verts = (T @ homo[..., None])[:, :, :3, 0].astype(dtype)
    return verts, A, T, v_posed, v_shaped


def skin(A, weights, v, rotmat=None, dtype=np.float32):
    """single batch element: A [J,16|4,4], weights [n,J], v [n,3], rotmat [n,3,3] -> verts, T [n,4,4], T[:, :3,:3] @ rotmat"""
    verts, _, T, _, _ = lbs_extra(np.asarray(A, dtype).reshape(1, -1, 4, 4), np.asarray(v, dtype)[None], None, weights, None,
                                  disable_posedirs=True, dtype=dtype)
    rot = None if rotmat is None else (T[0][:, :3, :3] @ np.asarray(rotmat, dtype)).astype(dtype)
    return verts[0], T[0], rot


def skin_backward(A, weights, v, rotmat, dL_dverts, dL_dT=None, dL_drot=None, dtype=np.float64):
    """gradients of skin() w.r.t. A [J,4,4], weights [n,J], v [n,3], rotmat [n,3,3] (chain rule, written out)."""
    A = np.asarray(A, dtype).reshape(-1, 16)
    W, v = np.asarray(weights, dtype), np.asarray(v, dtype)
    n = v.shape[0]
    T = (W @ A).reshape(n, 4, 4)
    G = np.zeros((n, 4, 4), dtype) if dL_dT is None else np.asarray(dL_dT, dtype).reshape(n, 4, 4).copy()
    homo = np.concatenate([v, np.ones((n, 1), dtype)], 1)
    gv = np.asarray(dL_dverts, dtype)
    G[:, :3, :] += gv[:, :, None] * homo[:, None, :]                 # verts = T[:3, :] @ [v, 1]
    d_rotmat = None
    if rotmat is not None and dL_drot is not None:
        R, gR = np.asarray(rotmat, dtype), np.asarray(dL_drot, dtype)
        G[:, :3, :3] += gR @ np.transpose(R, (0, 2, 1))              # rot = T[:3,:3] @ R
        d_rotmat = np.transpose(T[:, :3, :3], (0, 2, 1)) @ gR
    Gf = G.reshape(n, 16)
    return {"A": (W.T @ Gf).reshape(-1, 4, 4), "weights": Gf @ A.T,
            "v": np.einsum("nrc,nr->nc", T[:, :3, :3], gv), "rotmat": d_rotmat}
