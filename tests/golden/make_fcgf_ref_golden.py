"""Reference-pinned fixtures for the FCGF_APR pieces whose modules cannot be imported here (their files import
open3d / MinkowskiEngine / tensorboardX at the top, all absent) but whose functions are pure torch / numpy.

Run in the build container only (needs /root/reference):   python tests/golden/make_fcgf_ref_golden.py

How the reference is run: each source file is parsed with `ast`; the function definitions (or, for code that sits
inline in a training loop / data loader, the individual statements) named below are cut out of the parse tree,
compiled unchanged and executed here - the code that produces the expected values IS the reference's text, nothing
is re-typed.  Nothing is copied into the repo either: only inputs and outputs are stored (fcgf_ref.npz).

  est_quad_linear_robust + helpers   FCGF_APR/util/transform_estimation.py:5-116
  _hash                              FCGF_APR/util/misc.py:6-18
  find_nn_gpu                        FCGF_APR/lib/eval.py:18-48         (with the importable lib/metrics.py:pdist)
  contrastive_hardest_negative_loss  FCGF_APR/lib/trainer.py:400-452    (method body, run with a stub `self`)
  NPR regulariser + point assembly   FCGF_APR/lib/complement_trainer.py:432-442 (statements of _train_epoch)
  APG transform + crop               FCGF_APR/lib/complement_data_loader.py:65-70 (method), :620-628 (statements)

Not pinned here: the Chamfer call itself (`chamferdist`, third party, absent), MinkowskiEngine, open3d.
"""
import ast
import importlib.util
import os
import sys
import textwrap
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/FCGF_APR"


def _tree(rel):
    src = open(os.path.join(REF, rel), encoding="utf-8").read()
    return src, ast.parse(src)       # the `coding: future_fstrings` cookie is a comment to ast.parse(str)


def _defs(rel, names, ns, cls=None):
    """exec the named top-level functions (or methods of class `cls`) of a reference file into `ns`."""
    src, tree = _tree(rel)
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    found = {}
    for node in body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            found[node.name] = textwrap.dedent("\n".join(src.splitlines()[node.lineno - 1:node.end_lineno]))
    missing = set(names) - set(found)
    assert not missing, f"{rel}: {missing} not found"
    for name in names:
        exec(compile(found[name], f"{REF}/{rel}:{name}", "exec"), ns)
    return ns


def _statements(rel, picks):
    """Source text of single statements anywhere in a reference file.  `picks`: list of predicates on ast nodes, each
    must match exactly one statement; returned dedented, in the given order."""
    src, tree = _tree(rel)
    lines = src.splitlines()
    out = []
    for pred in picks:
        hits = sorted((n for n in ast.walk(tree) if isinstance(n, ast.stmt) and pred(n)), key=lambda n: n.lineno)
        assert len(hits) >= 1, f"{rel}: statement not found"
        n = hits[0]                  # first in source order
        out.append(textwrap.dedent("\n".join(lines[n.lineno - 1:n.end_lineno])))
    return out


def _assigns(name):
    return lambda n: isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name) \
        and n.targets[0].id == name


def ref_pdist():
    spec = importlib.util.spec_from_file_location("ref_metrics", os.path.join(REF, "lib/metrics.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.pdist


def main():
    rng = np.random.default_rng(2024)
    out = {}
    pdist = ref_pdist()

    # ---- IRLS pose (est_quad_linear_robust) -------------------------------------------------------------------
    te = _defs("util/transform_estimation.py",
               ["rot_x", "rot_y", "rot_z", "get_trans", "update_pcd", "build_linear_system", "solve_linear_system",
                "compute_weights", "est_quad_linear_robust"], {"torch": torch})
    n = 3000
    p0 = rng.uniform(-25, 25, (n, 3)).astype(np.float32)
    Tg = te["get_trans"](torch.tensor([[0.03], [-0.02], [0.08], [0.6], [-0.4], [0.15]]))
    p1 = (torch.from_numpy(p0) @ Tg[:3, :3].t() + Tg[:3, 3]).numpy() + rng.normal(0, 0.02, (n, 3)).astype(np.float32)
    bad = rng.random(n) < 0.3                      # 30 % gross outliers: what the re-weighting is there for
    p1[bad] = rng.uniform(-25, 25, (int(bad.sum()), 3)).astype(np.float32)
    w = rng.uniform(0.2, 1.0, (n, 1)).astype(np.float32)
    out["irls_p0"], out["irls_p1"], out["irls_w"], out["irls_T_gt"] = p0, p1, w, Tg.numpy()
    out["irls_T"] = te["est_quad_linear_robust"](torch.from_numpy(p0), torch.from_numpy(p1)).numpy()
    out["irls_T_weighted"] = te["est_quad_linear_robust"](torch.from_numpy(p0), torch.from_numpy(p1),
                                                          torch.from_numpy(w)).numpy()

    # ---- feature NN (find_nn_gpu), chunked and unchunked, both distance types --------------------------------
    ev = _defs("lib/eval.py", ["find_nn_gpu"], {"torch": torch, "np": np, "pdist": pdist})
    F0 = rng.standard_normal((1203, 32)).astype(np.float32)
    F1 = rng.standard_normal((1000, 32)).astype(np.float32)
    F0 /= np.linalg.norm(F0, axis=1, keepdims=True)
    F1 /= np.linalg.norm(F1, axis=1, keepdims=True)
    F1[17] = F1[3]                                  # an exact tie: torch.min keeps the first index
    out["nn_F0"], out["nn_F1"] = F0, F1
    i_a, d_a = ev["find_nn_gpu"](torch.from_numpy(F0), torch.from_numpy(F1), nn_max_n=500, return_distance=True)
    i_b, d_b = ev["find_nn_gpu"](torch.from_numpy(F0), torch.from_numpy(F1), nn_max_n=-1, return_distance=True,
                                 dist_type='L2')
    assert torch.equal(i_a, i_b)
    out["nn_inds"], out["nn_d2"], out["nn_d"] = i_a.numpy(), d_a.numpy(), d_b.numpy()

    # ---- hardest-contrastive loss ---------------------------------------------------------------------------------
    misc = _defs("util/misc.py", ["_hash"], {"np": np})
    tr = _defs("lib/trainer.py", ["contrastive_hardest_negative_loss"],
               {"np": np, "torch": torch, "F": F, "pdist": pdist, "_hash": misc["_hash"]},
               cls="HardestContrastiveLossTrainer")
    N0, N1, NP = 3000, 2800, 6000
    G0 = rng.standard_normal((N0, 32)).astype(np.float32)
    G1 = rng.standard_normal((N1, 32)).astype(np.float32)
    pos = np.stack([rng.integers(0, N0, NP), rng.integers(0, N1, NP)], 1).astype(np.int64)
    pos = np.unique(pos, axis=0)
    G1[pos[:2000, 1]] = G0[pos[:2000, 0]] + 0.05 * rng.standard_normal((2000, 32)).astype(np.float32)
    G0 /= np.linalg.norm(G0, axis=1, keepdims=True)
    G1 /= np.linalg.norm(G1, axis=1, keepdims=True)
    stub = types.SimpleNamespace(pos_thresh=0.1, neg_thresh=1.4)      # config.py:34-35
    for tag, num_pos, num_hn in (("a", 1024, 256), ("b", 100000, 4096)):   # b: no positive sub-sampling, sel = all
        np.random.seed(77)
        pl, nl = tr["contrastive_hardest_negative_loss"](stub, torch.from_numpy(G0), torch.from_numpy(G1),
                                                         torch.from_numpy(pos), num_pos=num_pos,
                                                         num_hn_samples=num_hn)
        np.random.seed(77)                           # the same three draws, in the reference's order (:413-420)
        sel0 = np.random.choice(N0, min(N0, num_hn), replace=False)
        sel1 = np.random.choice(N1, min(N1, num_hn), replace=False)
        pos_sel = np.random.choice(len(pos), num_pos, replace=False) if len(pos) > num_pos else np.arange(len(pos))
        out[f"hc_{tag}_sel0"], out[f"hc_{tag}_sel1"], out[f"hc_{tag}_pos_sel"] = sel0, sel1, pos_sel
        out[f"hc_{tag}_loss"] = np.array([float(pl), float(nl)])
        out[f"hc_{tag}_args"] = np.array([num_pos, num_hn])
    out["hc_F0"], out["hc_F1"], out["hc_pos"] = G0, G1, pos
    out["hash_keys"] = misc["_hash"](pos, max(N0, N1))

    # ---- NPR regulariser + generated-point assembly (statements inside GenerativePairTrainer._train_epoch) -----
    reg_if, mod_assign = _statements("lib/complement_trainer.py", [
        lambda n: isinstance(n, ast.If) and isinstance(n.test, ast.Compare) and "regularization_type" in
        ast.unparse(n.test) and "'L2'" in ast.unparse(n.test) and "Repel" not in ast.unparse(n.test)
        and "generated.reshape" in ast.unparse(n) and "raw_reg_loss" not in ast.unparse(n)
        and "regularize_loss" in ast.unparse(n),
        lambda n: _assigns("mod_generated")(n) and "batch_enc_coords" in ast.unparse(n)])
    gen = (rng.standard_normal((900, 12)) * 0.4).astype(np.float32)           # [N, 3 * ratio], ratio 4
    coords = rng.integers(-60, 60, (900, 3)).astype(np.int32)
    out["npr_generated"], out["npr_coords"] = gen, coords
    for kind in ("L2", "RepelL2", "RepelL1"):
        ns = {"torch": torch, "generated": torch.from_numpy(gen), "i": 0,
              "batch_enc_coords": [torch.from_numpy(coords)],
              "self": types.SimpleNamespace(regularization_type=kind, alpha=0.1, voxel_size=0.3,
                                            point_generation_ratio=4)}
        exec(compile(reg_if, f"{REF}/lib/complement_trainer.py:regulariser", "exec"), ns)
        exec(compile(mod_assign, f"{REF}/lib/complement_trainer.py:mod_generated", "exec"), ns)
        out[f"npr_reg_{kind}"] = np.array(float(ns["regularize_loss"]))
        out["npr_mod_generated"] = ns["mod_generated"].numpy()

    # ---- APG: transform the complement frames, crop to the key frame's radius ---------------------------------------
    dl = _defs("lib/complement_data_loader.py", ["apply_transform"], {"np": np}, cls="PointDataset")
    crop = _statements("lib/complement_data_loader.py", [_assigns("max_dist_square_0"), _assigns("xyz_cmpl_0"),
                                                         _assigns("xyz_nghb_0")])
    # (the first `xyz_cmpl_0 = ...` assignment in the file is the list comprehension that applies the transforms,
    #  :576-577; the concatenate at :624 is the second: pick both explicitly)
    src, tree = _tree("lib/complement_data_loader.py")
    cm = [n for n in ast.walk(tree) if _assigns("xyz_cmpl_0")(n)]
    texts = [textwrap.dedent("\n".join(src.splitlines()[n.lineno - 1:n.end_lineno])) for n in cm]
    move = next(t for t in texts if "apply_transform(xyz_k, M_k)" in t)
    cat = next(t for t in texts if "np.concatenate" in t)
    key = rng.uniform(-40, 40, (5000, 3)).astype(np.float32)
    frames = [rng.uniform(-60, 60, (4000, 3)).astype(np.float32) for _ in range(4)]
    poses = []
    for k in range(4):
        a = np.deg2rad(rng.uniform(-12, 12))
        M = np.eye(4)
        M[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
        M[:3, 3] = [6.0 * (k - 1.5), rng.uniform(-1, 1), rng.uniform(-0.2, 0.2)]
        poses.append(M)
    ns = {"np": np, "self": types.SimpleNamespace(apply_transform=lambda p, t: dl["apply_transform"](None, p, t)),
          "xyz_0": key, "xyz_cmpl_0": list(frames), "list_M_0": poses}
    for text in (move, crop[0], cat, crop[2]):
        exec(compile(text, f"{REF}/lib/complement_data_loader.py:apg", "exec"), ns)
    out["apg_key"], out["apg_frames"], out["apg_poses"] = key, np.stack(frames), np.stack(poses)
    out["apg_nghb"] = ns["xyz_nghb_0"]

    np.savez_compressed(os.path.join(HERE, "fcgf_ref.npz"), **out)
    print("wrote fcgf_ref.npz:", {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    sys.exit(main())
