"""The low-resolution side of the dataset items: host mirrors against fixtures generated from the reference's
own functions (g31), the cv2.resize(INTER_CUBIC) kernel against its numpy restatement (oracle/cv2_cubic.py; cv2 is not
in this image: parity unpinned against cv2 itself), and the device pipeline that assembles the batch keys."""
import os
import sys

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_host_mirrors_equal_the_reference_functions():
    """interpolate_torch / simulate_low_res / per-colour weights / per-pixel lookup / np_blur / np_prod_binary_noise /
    np_add_gaussian_noise (dataset_dpsr.py:592-645,684-744,1037-1180): bit for bit what the reference's functions
    returned on the same inputs and seeds."""
    from dlib.datasets import lowres as L
    g = np.load(os.path.join(G, "g31_lowres.npz"))
    hr = g["hr"]
    for sc in (2, 4, 8):
        lo = L.interpolate_torch(hr, 1. / sc)
        assert np.array_equal(lo, g[f"interp_x{sc}"])
        assert np.array_equal(L.simulate_low_res(np.copy(lo), 3 + sc, 7., 6.), g[f"sim_x{sc}"])
    w = L.per_color_weights(list(g["ppiw_tiles"]), 0, 255, 0.1)
    assert np.array_equal(w, g["ppiw_weights"])
    pw = L.per_pixel_weight(torch.from_numpy(g["ppiw_patch_u8"]).float()[None] / 255., torch.from_numpy(g["ppiw_weights"]))
    assert torch.equal(pw[0], torch.from_numpy(g["ppiw_patch_w"]))
    for name, fn, kw in (("blur", L.da_blur, dict(prob=1.0, area=0.4, sigma=1.3)),
                         ("dot", L.da_dot_bin_noise, dict(prob=1.0, area=0.5, p=0.3)),
                         ("gaus", L.da_add_gaus_noise, dict(prob=1.0, area=0.5, std=0.05))):
        for seed in range(8):
            np.random.seed(1000 + seed)
            assert np.array_equal(fn(np.copy(g["da_base"]), **kw), g[f"da_{name}_{seed}"]), (name, seed)


def test_cubic_restatement_is_the_a075_bicubic():
    """oracle/cv2_cubic.py restates OpenCV's INTER_CUBIC: its float path must coincide with any other implementation of
    the same published kernel (A = -0.75, half-pixel centres, replicate border) -- torch's bicubic is one -- and its uint8
    path can only differ from the rounded float result by the fixed-point coefficient rounding (one grey level)."""
    import torch.nn.functional as F
    from cv2_cubic import resize_cubic
    rng = np.random.default_rng(0)
    for (H, W, Ho, Wo) in [(16, 20, 128, 160), (64, 64, 512, 512), (37, 41, 74, 82), (128, 128, 64, 64)]:
        x = rng.random((H, W), dtype=np.float32)
        t = F.interpolate(torch.from_numpy(x)[None, None], size=(Ho, Wo), mode="bicubic", align_corners=False)[0, 0].numpy()
        assert np.abs(resize_cubic(x, (Wo, Ho)) - t).max() < 1e-6
        xu = (x * 255).astype(np.uint8)
        tu = F.interpolate(torch.from_numpy(xu.astype(np.float32))[None, None], size=(Ho, Wo), mode="bicubic",
                           align_corners=False)[0, 0].numpy()
        d = np.abs(resize_cubic(xu, (Wo, Ho)).astype(np.int64) - np.clip(np.rint(tu), 0, 255).astype(np.int64))
        assert d.max() <= 1 and (d > 0).mean() < 0.05
        assert np.array_equal(resize_cubic(xu, (W, H)), xu)        # same size: a copy, as cv2


@pytest.mark.gpu
def test_resize_cubic_kernel_matches_the_restatement():
    from srhip import ops
    from cv2_cubic import resize_cubic
    rng = np.random.default_rng(1)
    for (H, W, Ho, Wo) in [(16, 16, 128, 128), (64, 64, 512, 512), (17, 23, 136, 184), (40, 30, 80, 60), (32, 32, 32, 32),
                           (96, 80, 48, 40)]:
        xu = rng.integers(0, 256, (3, H, W), dtype=np.uint8)
        got = ops.resize_cubic(torch.from_numpy(xu).cuda(), (Ho, Wo)).cpu().numpy()
        for b in range(3):
            assert np.array_equal(got[b], resize_cubic(xu[b], (Wo, Ho))), (H, W, Ho, Wo)      # bit-exact
        xf = rng.random((2, H, W), dtype=np.float32)
        gotf = ops.resize_cubic(torch.from_numpy(xf).cuda(), (Ho, Wo)).cpu().numpy()
        for b in range(2):
            assert np.abs(gotf[b] - resize_cubic(xf[b], (Wo, Ho))).max() < 1e-6
    x = torch.tensor([[-0.5, 0.25], [1.5, 1.0]]).cuda()
    assert ops.clip01_(x.clone()).cpu().tolist() == [[0.0, 0.25], [1.0, 1.0]]
    u = torch.arange(256, dtype=torch.uint8).cuda()
    assert torch.equal(ops.u8_to_unit(u).cpu(), torch.from_numpy(np.float32(np.arange(256) / 255.)))


def _fixture_args():
    import yaml
    from dlib.utils.tools import Dict2Obj
    fx = os.path.join(G, "eval_exp")
    a = Dict2Obj(yaml.safe_load(open(os.path.join(fx, "exp", "config_model.yml"))))
    a.train_dsets, a.data_root, a.splits_root = a.test_dsets, os.path.join(fx, "data"), os.path.join(fx, "folds")
    a.h_size, a.batch_size, a.myseed = 64, 2, 3
    return a, fx


@pytest.mark.gpu
def test_resident_train_set_low_res_side():
    """Batch keys around the hot path: l_to_h_img (cv2.resize of the -- augmented -- LR patch, clipped), --ppiw weights,
    LR-only augmentations through the reference's numpy stream, 'edt' sampling, synthesised LR tiles."""
    from PIL import Image
    from cv2_cubic import resize_cubic
    from dlib.datasets import lowres as L
    from dlib.utils.utils_dataloaders import get_train_set
    a, fx = _fixture_args()
    raw_h = [np.asarray(Image.open(os.path.join(fx, "data", "caco2", "t", f"h_{i}.tif"))) for i in range(3)]
    a.sample_tr_patch, a.sample_tr_patch_th_style, a.sample_tr_patch_th = "edt", "fix_threshold", 12
    a.ppiw, a.ppiw_min_per_col_w = True, 0.1
    ts = get_train_set(a, "cuda")
    table = L.per_color_weights(raw_h, 0, 255, 0.1)
    n = 0
    for b in ts.epoch(0):
        assert b["l_to_h_img"].shape == b["h_im"].shape == (2, 1, 64, 64) and b["l_to_h_img_aug"] is b["l_to_h_img"]
        for k in range(2):
            lo = b["l_im"][k, 0].cpu().numpy()
            want = np.clip(resize_cubic(lo, (64, 64)), 0., 1.)
            assert np.abs(b["l_to_h_img"][k, 0].cpu().numpy() - want).max() < 1e-6
            hp = (b["h_im"][k, 0].cpu() * 255.).to(torch.uint8).numpy()
            assert np.array_equal(b["h_per_pixel_weight"][k, 0].cpu().numpy(), np.float32(table)[hp])
        n += 1
    assert n == 1
    # LR-only augmentations: the batch's l_im is what the reference's functions make of the un-augmented crop, with the
    # same numpy stream
    a.ppiw = False
    a.sample_tr_patch = "uniform"
    a.da_add_gaus_noise, a.da_add_gaus_noise_prob, a.da_add_gaus_noise_area, a.da_add_gaus_noise_std = True, 1.0, 0.5, 0.05
    a.da_dot_bin_noise, a.da_dot_bin_noise_prob, a.da_dot_bin_noise_area, a.da_dot_bin_noise_p = True, 1.0, 0.4, 0.3
    plain = get_train_set(Dict2ObjCopy(a, da_add_gaus_noise=False, da_dot_bin_noise=False), "cuda")
    noisy = get_train_set(a, "cuda")
    pb = next(iter(plain.epoch(0)))
    np.random.seed(77)
    nb = next(iter(noisy.epoch(0)))
    assert pb["origin"] == nb["origin"] and pb["mode"] == nb["mode"]
    np.random.seed(77)
    for k in range(2):
        base = np.ascontiguousarray(pb["l_im"][k].permute(1, 2, 0).cpu().numpy())
        want = L.apply_lr_augmentations(base, a)
        assert np.array_equal(nb["l_im"][k].permute(1, 2, 0).cpu().numpy(), np.float32(want))
    assert not torch.equal(pb["l_im"], nb["l_im"])


def Dict2ObjCopy(a, **kw):
    from dlib.utils.tools import Dict2Obj
    b = Dict2Obj(dict(a))            # Dict2Obj IS a dict
    for k, v in kw.items():
        b[k] = v
    return b


@pytest.mark.gpu
def test_eval_pairs_l_to_h_and_synthesised_low_resolution(tmp_path):
    """EvalPairs: 'l_to_h_img' = cv2.resize(LR tile, HR size) / 255 on the device; a pair without a true LR tile gets the
    reference's synthesis (bicubic down-scaling + seeded noise in the cells' region, seeded by the item index)."""
    from cv2_cubic import resize_cubic
    from dlib.datasets import lowres as L
    from dlib.utils.utils_dataloaders import EvalPairs, imread_gray_uint8
    a, fx = _fixture_args()
    hp = os.path.join(fx, "data", "caco2", "t", "h_0.tif")
    lp = os.path.join(fx, "data", "caco2", "t", "l_0.tif")
    pairs_h = {"t/h_0.tif": {"abs_path": hp, "low_path_key": "t/l_0.tif"}}
    ev = EvalPairs(a, pairs_h, {"t/l_0.tif": {"abs_path": lp}})
    it = ev[0]
    l8 = imread_gray_uint8(lp)[:, :, 0]
    H, W = it["h_im"].shape[1:]
    assert np.array_equal(np.rint(it["l_to_h_img"][0].cpu().numpy() * 255).astype(np.uint8), resize_cubic(l8, (W, H)))
    # no true LR tile: synthesis (the path must look like a CACO-2 tile path: utils_image.py:202-208)
    d = tmp_path / "caco2" / "CELL0"
    d.mkdir(parents=True)
    import shutil
    shutil.copy(hp, d / "h_0.tif")
    pairs_h = {"CELL0/h_0.tif": {"abs_path": str(d / "h_0.tif"), "low_path_key": "None_0"}}
    ev = EvalPairs(a, pairs_h, {"None_0": {"abs_path": str(d / "missing.tif")}})
    it = ev[0]
    h8 = imread_gray_uint8(hp)
    want = L.simulate_low_res(np.clip(L.interpolate_torch(h8, 1. / a.scale), 0, 255), seed=0, th=7., sigma=6.)
    assert np.array_equal(np.rint(it["l_im"].permute(1, 2, 0).numpy() * 255).astype(np.uint8), want)
    assert it["l_path"] == str(d / "h_0.tif")


@pytest.mark.gpu
def test_srcnn_runs_through_the_cli():
    """`main.py --net_type SRCNN` needs the batch key 'l_to_h_img' -- now produced by synth_batch (and by the
    dataset classes); two training iterations + the evaluation sweep run."""
    import main as M
    rc = M.main(["--net_type", "SRCNN", "--method", "SRCNN", "--scale", "2", "--h_size", "64", "--batch_size", "2",
                 "--max_iters", "2", "--outd", os.path.join(os.environ.get("TMPDIR", "/tmp"), "srcnn_cli")])
    assert rc in (0, None)
