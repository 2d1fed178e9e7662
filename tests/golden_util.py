import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_cases(name):
    """npz with keys 'case/field' -> {case: {field: array}}"""
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    cases = {}
    for k in z.files:
        case, field = k.split("/")
        cases.setdefault(case, {})[field] = z[k]
    return cases


def projection_cases():
    """yield (name, dtype, inputs dict (cast to dtype), expected dict)"""
    cases = load_cases("projection.npz")
    for name in sorted(cases):
        if not name.endswith("_f64"):
            continue
        base = cases[name]
        for tag, dt in (("f64", np.float64), ("f32", np.float32)):
            exp = cases[name[:-3] + tag]
            ins = {k[3:]: base[k].astype(dt) for k in base if k.startswith("in_")}
            meta = dict(image_size=tuple(int(x) for x in base["image_size"]),
                        depth_range=tuple(float(x) for x in base["depth_range"]), blur_cov=float(base["blur_cov"]))
            yield name[:-3] + tag, dt, ins, exp, meta


def sh_cases():
    cases = load_cases("sh.npz")
    for name in sorted(cases):
        if not name.endswith("_f64"):
            continue
        base = cases[name]
        for tag, dt in (("f64", np.float64), ("f32", np.float32)):
            exp = cases[name[:-3] + tag]
            ins = {k[3:]: base[k].astype(dt) for k in base if k.startswith("in_")}
            yield name[:-3] + tag, dt, ins, base["indexes"], exp
