#!/usr/bin/env python3
"""Generates tests/golden/reference_scene_forward.npz: raw parameters, the dict SceneGS.forward returns for them and the
autograd gradients of its entries -- `setup_functions`, `get_features` and `forward` are compiled from
/root/reference/hugs/models/scene.py (read-only) in THIS container and run on CPU on a stand-in `self` that carries the seven
attributes they touch.  Only these vectors travel.      python tests/golden/make_golden_scene.py"""
import ast
import os
import types

import numpy as np
import torch

REF = "/root/reference/hugs/models/scene.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_scene_forward.npz")


def main():
    tree = ast.parse(open(REF).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "SceneGS")
    ns = {"torch": torch, "build_scaling_rotation": None, "strip_symmetric": None, "inverse_sigmoid": None}
    methods = {}
    for name in ("setup_functions", "get_features", "forward"):
        node = next(f for f in cls.body if isinstance(f, ast.FunctionDef) and f.name == name)
        node.decorator_list = []                                      # (get_features is a property there: called directly here)
        exec(compile(ast.Module(body=[node], type_ignores=[]), REF, "exec"), ns)
        methods[name] = ns[name]
    r = np.random.default_rng(23)
    P, M = 257, 16
    raw = {"_scaling": (r.standard_normal((P, 3)) - 3.0), "_rotation": r.standard_normal((P, 4)) * r.uniform(0.2, 3.0, (P, 1)),
           "_opacity": 2.0 * r.standard_normal((P, 1)), "_features_dc": r.standard_normal((P, 1, 3)),
           "_features_rest": 0.2 * r.standard_normal((P, M - 1, 3)), "_xyz": r.standard_normal((P, 3))}
    raw = {k: v.astype(np.float32) for k, v in raw.items()}
    raw["_rotation"][0] = 0.0                                         # the eps branch of normalize
    me = types.SimpleNamespace(only_rgb=False, active_sh_degree=2, **{k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in raw.items()})
    methods["setup_functions"](me)
    me_get = methods["get_features"]

    class Self(types.SimpleNamespace):
        @property
        def get_features(self):
            return me_get(self)
    me = Self(**vars(me))
    out = methods["forward"](me)
    g = {k: r.standard_normal(tuple(out[k].shape)).astype(np.float32) for k in ("scales", "rotq", "opacity", "shs")}
    torch.autograd.backward([out[k] for k in g], [torch.from_numpy(v) for v in g.values()])
    arrays = {f"raw{k}": v for k, v in raw.items()}
    arrays.update({f"out_{k}": out[k].detach().numpy() for k in ("xyz", "scales", "rotq", "opacity", "shs")})
    arrays.update({f"g_{k}": v for k, v in g.items()})
    arrays.update({f"grad{k}": getattr(me, k).grad.numpy() for k in ("_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")})
    arrays["out_active_sh_degree"] = np.int32(out["active_sh_degree"])
    arrays["keys_json"] = np.frombuffer(",".join(out.keys()).encode(), dtype=np.uint8)
    np.savez_compressed(OUT, **arrays)
    print(f"wrote {OUT}: {len(arrays)} arrays, {os.path.getsize(OUT) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
