"""Reference-pinned fixtures, second batch (round 5): everything on the Predator_APR side -- and the last FCGF_APR
closes -- that can be executed in the build container although its module cannot be imported (top-level `import
open3d` / `chamferdist` / `MinkowskiEngine` ...).

Run in the build container only (needs /root/reference and `make -C oracle ref`):
    python tests/golden/make_predator_ref_golden.py

Same rule as make_fcgf_ref_golden.py: a reference file is either imported as it is (models/mlp.py, lib/timer.py,
FCGF_APR/lib/metrics.py: plain torch / numpy) or parsed with `ast`, the named function definitions (or single
statements of a training / test loop) are cut out, compiled unchanged and executed on seeded inputs.  Only inputs and
outputs are stored (predator_ref.npz: numeric arrays).

  GenerativeMLP_54 / _4 forward, state_dict, gradients   Predator_APR/models/mlp.py:103-179          (imported)
  NPR loss statements                                     Predator_APR/lib/trainer.py:176-183         (statements)
  Trainer.chamfer_distance                                Predator_APR/lib/trainer.py:131-140         (method; the
        third-party `ChamferDistance` it instantiates is absent: a brute-force torch stand-in with the published
        semantics -- sum over the points of array1 of the squared distance to the nearest point of array2 -- is bound
        to that name, so the Chamfer VALUE stays "unpinned", the arithmetic around it is the reference's)
  collate_fn_descriptor + batch_*_kpconv                  Predator_APR/datasets/dataloader.py:15-198  (functions; the two
        `cpp_wrappers` module names bound to the reference's own C++ in oracle/_ref)
  calibrate_neighbors                                     Predator_APR/datasets/dataloader.py:200-232 (function)
  score sampling                                          Predator_APR/lib/tester.py:80-92            (statements)
  get_angle_deviation                                     Predator_APR/lib/benchmark_utils.py:170-185 (function)
  corr_dist                                               FCGF_APR/lib/metrics.py:13-19               (imported)
  evaluate_hit_ratio + apply_transform                    FCGF_APR/lib/trainer.py:392-395, :373-377   (methods)
  find_corr                                               FCGF_APR/scripts/test_apr.py:43-57          (function, with
        find_nn_gpu cut from lib/eval.py:18-48 and the importable pdist)
"""
import ast
import importlib.util
import os
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
PRED = "/root/reference/Predator_APR"
FCGF = "/root/reference/FCGF_APR"


def _tree(path):
    src = open(path, encoding="utf-8").read()
    return src, ast.parse(src)


def _defs(path, names, ns, cls=None):
    src, tree = _tree(path)
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    found = {}
    for node in body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            found[node.name] = textwrap.dedent("\n".join(src.splitlines()[node.lineno - 1:node.end_lineno]))
    missing = set(names) - set(found)
    assert not missing, f"{path}: {missing} not found"
    for name in names:
        exec(compile(found[name], f"{path}:{name}", "exec"), ns)
    return ns


def _first_statement(path, pred, within=None):
    """Dedented source of the first statement (in source order) matching `pred`; `within` = (class, function) names."""
    src, tree = _tree(path)
    scope = tree
    if within is not None:
        for name in within:
            scope = next(n for n in ast.walk(scope) if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name == name)
    hits = sorted((n for n in ast.walk(scope) if isinstance(n, ast.stmt) and pred(n)), key=lambda n: n.lineno)
    assert hits, f"{path}: statement not found"
    n = hits[0]
    return textwrap.dedent("\n".join(src.splitlines()[n.lineno - 1:n.end_lineno])), n.lineno


def _assigns(name):
    return lambda n: isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name) \
        and n.targets[0].id == name


def _import(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class BruteChamfer(torch.nn.Module):
    """Stand-in for chamferdist.ChamferDistance()(a [1,n,3], b [1,m,3]): sum_i min_j |a_i - b_j|^2 (float32)."""

    def forward(self, a, b):
        d = (a[0].unsqueeze(1) - b[0].unsqueeze(0)).pow(2).sum(2)
        return d.min(1)[0].sum()


def npr_fixtures(out, rng):
    mlp = _import(os.path.join(PRED, "models/mlp.py"), "ref_predator_mlp")
    cfg = types.SimpleNamespace(generative_model="GenerativeMLP_54", final_feats_dim=32, point_generation_ratio=4,
                                batch_norm_momentum=0.02)
    torch.manual_seed(5)
    gen = mlp.get_GenerativeMLP(cfg, radius=None)
    for k, v in gen.state_dict().items():
        out[f"mlp54_sd/{k}"] = v.numpy().copy()
    n = 700
    feats = rng.standard_normal((n, 32)).astype(np.float32)
    feats /= np.linalg.norm(feats, axis=1, keepdims=True)
    pcd = rng.uniform(-20, 20, (n, 3)).astype(np.float32)
    nghb = (pcd[rng.integers(0, n, 3000)] + rng.normal(0, 0.4, (3000, 3))).astype(np.float32)
    out["npr_feats"], out["npr_pcd"], out["npr_nghb"] = feats, pcd, nghb
    # the four statements of Trainer.inference_one_batch (train branch), frame 0
    tr = os.path.join(PRED, "lib/trainer.py")
    scope = ("Trainer", "inference_one_batch")
    s_reg, l0 = _first_statement(tr, _assigns("regularize_loss"), scope)
    s_mod, _ = _first_statement(tr, _assigns("mod_generated"), scope)
    s_ch, _ = _first_statement(tr, _assigns("chamfer_loss_raw"), scope)
    s_gl, l1 = _first_statement(tr, lambda n: isinstance(n, ast.AugAssign) and isinstance(n.target, ast.Name)
                                and n.target.id == "generative_loss", scope)
    assert 170 <= l0 <= l1 <= 190, (l0, l1)
    cd = _defs(tr, ["chamfer_distance"], {"torch": torch, "ChamferDistance": BruteChamfer}, cls="Trainer")
    stub = types.SimpleNamespace(device="cpu", point_generation_ratio=4, regularization_strength=0.01, loss_ratio=0.5)
    stub.chamfer_distance = lambda a, b: cd["chamfer_distance"](stub, a, b)
    gen.train()
    x = torch.from_numpy(feats).requires_grad_(True)
    generated = gen(x)                                        # radius None -> a tensor (models/mlp.py:140-143)
    out["mlp54_train_out"] = generated.detach().numpy().copy()
    ns = {"torch": torch, "self": stub, "generated": generated, "src_pcd": torch.from_numpy(pcd),
          "inputs": {"src_nghb": torch.from_numpy(nghb)}, "generative_loss": 0}
    for text in (s_reg, s_mod, s_ch, s_gl):
        exec(compile(text, f"{tr}:npr", "exec"), ns)
    out["npr_regularize_loss"] = np.array(float(ns["regularize_loss"]))
    out["npr_chamfer_loss_raw"] = np.array(float(ns["chamfer_loss_raw"]))
    out["npr_generative_loss"] = np.array(float(ns["generative_loss"]))
    out["npr_mod_generated"] = ns["mod_generated"].detach().numpy().copy()
    ns["generative_loss"].backward()
    out["npr_grad_feats"] = x.grad.numpy().copy()
    for k, p in gen.named_parameters():
        out[f"npr_grad/{k}"] = p.grad.numpy().copy()
    for k, v in gen.state_dict().items():                     # running statistics after ONE training forward
        if "running" in k or "num_batches" in k:
            out[f"mlp54_sd_after/{k}"] = v.numpy().copy()
    gen.eval()
    with torch.no_grad():
        out["mlp54_eval_out"] = gen(torch.from_numpy(feats)).numpy().copy()
    # (x, radius) return and the one-hidden-layer variant
    cfg4 = types.SimpleNamespace(generative_model="GenerativeMLP_4", final_feats_dim=32, point_generation_ratio=6,
                                 batch_norm_momentum=0.1)
    torch.manual_seed(6)
    g4 = mlp.get_GenerativeMLP(cfg4, radius=2.5, in_channels=32).eval()
    for k, v in g4.state_dict().items():
        out[f"mlp4_sd/{k}"] = v.numpy().copy()
    with torch.no_grad():
        y, radius = g4(torch.from_numpy(feats))
    out["mlp4_eval_out"], out["mlp4_radius"] = y.numpy().copy(), np.array(radius)


def collate_fixtures(out):
    from apr_amd import synth
    from apr_amd.predator.configs.models import kitti_config
    from oracle import predator_points_oracle as PREF
    assert PREF.available(), "run `make -C oracle ref` first"
    timer = _import(os.path.join(PRED, "lib/timer.py"), "ref_predator_timer")
    wrap_sub = types.SimpleNamespace(subsample_batch=lambda points, batches, sampleDl=0.1, max_p=0, verbose=0:
                                     PREF.subsample_batch(np.asarray(points), np.asarray(batches), sampleDl, max_p))
    wrap_nb = types.SimpleNamespace(batch_query=lambda q, s, qb, sb, radius=1.0:
                                    PREF.batch_query(np.asarray(q), np.asarray(s), np.asarray(qb), np.asarray(sb), radius))
    ns = {"np": np, "torch": torch, "cpp_subsampling": wrap_sub, "cpp_neighbors": wrap_nb, "Timer": timer.Timer}
    _defs(os.path.join(PRED, "datasets/dataloader.py"),
          ["batch_grid_subsampling_kpconv", "batch_neighbors_kpconv", "collate_fn_descriptor", "calibrate_neighbors"], ns)
    cfg = kitti_config()

    def item(seed, beams, az):
        a, b, T = synth.make_pair(seed, n_beams=beams, n_azimuth=az)
        lens = np.array([len(a), len(b)], np.int32)
        pts, ln = PREF.subsample_batch(np.concatenate([a, b]), lens, sampleDl=cfg.first_subsampling_dl)
        src, tgt = pts[:ln[0]], pts[ln[0]:]
        one = lambda p: np.ones((len(p), 1), np.float32)
        return (src, tgt, one(src), one(tgt), T[:3, :3].astype(np.float32), T[:3, 3:].astype(np.float32),
                torch.zeros((0, 2), dtype=torch.int64), src, tgt, src, tgt, None)

    data = [item(100 + s, 16, 400) for s in range(10)]
    caps = ns["calibrate_neighbors"](data, cfg, ns["collate_fn_descriptor"], keep_ratio=0.8, samples_threshold=2000)
    out["calib_caps"] = np.asarray(caps).astype(np.int64)
    out["calib_seeds"] = np.arange(100, 110)
    it = item(7, 16, 400)
    batch = ns["collate_fn_descriptor"]([it], cfg, list(caps) + [int(caps[-1])])
    out["collate_src"], out["collate_tgt"] = it[0], it[1]
    for l in range(len(batch["points"])):
        out[f"collate_points_{l}"] = batch["points"][l].numpy()
        out[f"collate_neighbors_{l}"] = batch["neighbors"][l].numpy().astype(np.int32)
        out[f"collate_pools_{l}"] = batch["pools"][l].numpy().astype(np.int32)
        out[f"collate_upsamples_{l}"] = batch["upsamples"][l].numpy().astype(np.int32)
        out[f"collate_lengths_{l}"] = batch["stack_lengths"][l].numpy().astype(np.int32)
    out["collate_levels"] = np.array(len(batch["points"]))
    assert batch["neighbors"][0].dtype == torch.int64 and batch["features"].shape == (len(it[0]) + len(it[1]), 1)


def tester_fixtures(out, rng):
    te = os.path.join(PRED, "lib/tester.py")
    scope = ("KITTITester", "test")
    texts = [_first_statement(te, _assigns("n_points"), scope)[0],
             _first_statement(te, _assigns("src_scores"), scope)[0],
             _first_statement(te, _assigns("tgt_scores"), scope)[0],
             _first_statement(te, lambda n: isinstance(n, ast.If) and "src_pcd.size(0) > n_points" in ast.unparse(n.test),
                              scope)[0],
             _first_statement(te, lambda n: isinstance(n, ast.If) and "tgt_pcd.size(0) > n_points" in ast.unparse(n.test),
                              scope)[0]]
    n0, n1 = 5600, 3000                                      # one side above the 5000 cap, one below
    ns = {"np": np, "torch": torch,
          "src_pcd": torch.from_numpy(rng.uniform(-30, 30, (n0, 3)).astype(np.float32)),
          "tgt_pcd": torch.from_numpy(rng.uniform(-30, 30, (n1, 3)).astype(np.float32)),
          "src_feats": torch.from_numpy(rng.standard_normal((n0, 4)).astype(np.float32)),
          "tgt_feats": torch.from_numpy(rng.standard_normal((n1, 4)).astype(np.float32)),
          "src_overlap": torch.from_numpy(rng.random(n0).astype(np.float32)),
          "src_saliency": torch.from_numpy(rng.random(n0).astype(np.float32)),
          "tgt_overlap": torch.from_numpy(rng.random(n1).astype(np.float32)),
          "tgt_saliency": torch.from_numpy(rng.random(n1).astype(np.float32))}
    for k in ("src_pcd", "tgt_pcd", "src_feats", "tgt_feats", "src_overlap", "src_saliency", "tgt_overlap",
              "tgt_saliency"):
        out[f"tester_{k}"] = ns[k].numpy().copy()
    np.random.seed(123)
    for t in texts:
        exec(compile(t, f"{te}:sampling", "exec"), ns)
    out["tester_idx_src"] = np.asarray(ns["idx_src"]).astype(np.int64)
    assert "idx_tgt" not in ns
    out["tester_src_pcd_out"], out["tester_src_feats_out"] = ns["src_pcd"].numpy(), ns["src_feats"].numpy()
    out["tester_next_uniform"] = np.array(np.random.random_sample())     # position of the legacy stream afterwards


def metric_fixtures(out, rng):
    bu = _defs(os.path.join(PRED, "lib/benchmark_utils.py"), ["get_angle_deviation"], {"np": np})
    from scipy.spatial.transform import Rotation
    Rp = Rotation.from_rotvec(rng.normal(0, 0.6, (12, 3))).as_matrix()
    Rg = Rotation.from_rotvec(rng.normal(0, 0.6, (12, 3))).as_matrix()
    Rp[0] = Rg[0]                                             # zero deviation; clip at the domain edge
    Rp[1] = Rg[1] @ Rotation.from_rotvec([0, 0, np.pi]).as_matrix()
    out["angle_R_pred"], out["angle_R_gt"] = Rp, Rg
    out["angle_degs"] = bu["get_angle_deviation"](Rp, Rg)
    # corr_dist: imported as it is
    metrics = _import(os.path.join(FCGF, "lib/metrics.py"), "ref_fcgf_metrics")
    xyz0 = rng.uniform(-20, 20, (4000, 3)).astype(np.float32)
    a = np.deg2rad(7.0)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
    T[:3, 3] = [1.0, -2.0, 0.5]
    est = T.copy()
    est[:3, 3] += np.array([0.3, 0.0, -0.2], np.float32)
    w = rng.random(len(xyz0)).astype(np.float32)
    out["cd_xyz0"], out["cd_T"], out["cd_est"], out["cd_w"] = xyz0, T, est, w
    t = torch.from_numpy
    out["cd_plain"] = np.array(float(metrics.corr_dist(t(est), t(T), t(xyz0), None, max_dist=0.25)))
    out["cd_weighted"] = np.array(float(metrics.corr_dist(t(est), t(T), t(xyz0), None, weight=t(w), max_dist=0.25)))
    out["cd_default"] = np.array(float(metrics.corr_dist(t(est), t(T), t(xyz0), None)))
    # evaluate_hit_ratio (a method of ContrastiveLossTrainer that calls self.apply_transform)
    tr = _defs(os.path.join(FCGF, "lib/trainer.py"), ["apply_transform", "evaluate_hit_ratio"], {"np": np, "torch": torch},
               cls="ContrastiveLossTrainer")
    stub = types.SimpleNamespace()
    stub.apply_transform = lambda pts, trans: tr["apply_transform"](stub, pts, trans)
    xyz1 = (xyz0 @ T[:3, :3].T + T[:3, 3] + rng.normal(0, 0.06, xyz0.shape)).astype(np.float32)
    out["hr_xyz1"] = xyz1
    for thr in (0.1, 0.05):
        out[f"hr_{thr}"] = np.array(tr["evaluate_hit_ratio"](stub, t(xyz0), t(xyz1), t(T), thresh=thr))
    # find_corr (scripts/test_apr.py) on top of find_nn_gpu (lib/eval.py) and the importable pdist
    ev = _defs(os.path.join(FCGF, "lib/eval.py"), ["find_nn_gpu"], {"torch": torch, "np": np, "pdist": metrics.pdist})
    fc = _defs(os.path.join(FCGF, "scripts/test_apr.py"), ["find_corr"], {"np": np, "torch": torch,
                                                                          "find_nn_gpu": ev["find_nn_gpu"]})
    F0 = rng.standard_normal((2300, 16)).astype(np.float32)
    F1 = rng.standard_normal((2100, 16)).astype(np.float32)
    p0 = rng.uniform(-5, 5, (2300, 3)).astype(np.float32)
    p1 = rng.uniform(-5, 5, (2100, 3)).astype(np.float32)
    out["fc_F0"], out["fc_F1"], out["fc_xyz0"], out["fc_xyz1"] = F0, F1, p0, p1
    np.random.seed(11)
    a0, a1 = fc["find_corr"](p0, p1, t(F0), t(F1), subsample_size=1500)
    out["fc_sub_xyz0"], out["fc_sub_xyz1"] = np.asarray(a0), np.asarray(a1)
    b0, b1 = fc["find_corr"](p0, p1, t(F0), t(F1), subsample_size=-1)
    out["fc_all_xyz0"], out["fc_all_xyz1"] = np.asarray(b0), np.asarray(b1)


def main():
    rng = np.random.default_rng(2025)
    out = {}
    npr_fixtures(out, rng)
    tester_fixtures(out, rng)
    metric_fixtures(out, rng)
    collate_fixtures(out)
    np.savez_compressed(os.path.join(HERE, "predator_ref.npz"), **out)
    size = os.path.getsize(os.path.join(HERE, "predator_ref.npz"))
    print(f"wrote predator_ref.npz ({size / 1e6:.2f} MB):", {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    sys.exit(main())
