"""Training-crop sampler (reference dlib/datasets/dataset_dpsr.py:293-507) against g21,
which holds what the REFERENCE's PatchSampler did: the probability vector it handed to
np.random.multinomial, its seeded 'roi' and 'uniform' draws, its ROI mask.

CPU: the host mirror reproduces the reference's draws from the same seeds; the oracle's probability
map equals the reference's vector; the inverse-CDF form has those probabilities as its CDF steps.
GPU: srhip_roi_sample picks, for every uniform, exactly the origin of the oracle's inverse CDF, and
its empirical distribution over 200k draws is the reference's."""
import os
import random

import numpy as np
import pytest
import torch

import sr_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    z = np.load(os.path.join(G, "g21_patch_sampler.npz"))
    return {k: z[k] for k in z.files}


def test_host_sampler_reproduces_reference_draws():
    from dlib.datasets.dataset_dpsr import PatchSampler, roi_probabilities, SAMPLE_ROI, SAMPLE_UNIF, TH_FIX, TH_AUTO
    g = load()
    for name in "abc":
        img = g[f"{name}/img"]
        P, th, seed_np, seed_py = [int(v) for v in g[f"{name}/cfg"]]
        pm = O.roi_origin_pmf(img, th, P)
        assert np.array_equal(pm.reshape(-1), g[f"{name}/pvals"])
        assert np.array_equal(roi_probabilities(img, float(th), P), pm)
        s = PatchSampler(SAMPLE_ROI, P, 256, TH_FIX, float(th))
        np.random.seed(seed_np)
        draws = np.array([s(img, False)[:2] for _ in range(40)])
        assert np.array_equal(draws, g[f"{name}/roi_draws"])
        assert np.array_equal(s(img, True)[2], g[f"{name}/roi_u8"])
        u = PatchSampler(SAMPLE_UNIF, P, 256, TH_FIX, float(th))
        random.seed(seed_py)
        assert np.array_equal(np.array([u(img, False)[:2] for _ in range(40)]), g[f"{name}/uniform_draws"])
        # inverse CDF: origin i is chosen exactly for u in [cdf(i-1), cdf(i))
        cdf = np.cumsum(pm.reshape(-1))
        for uu in np.random.RandomState(1).rand(200):
            r0, c0 = O.roi_origin_from_uniform(img, th, P, float(uu))
            i = r0 * (img.shape[1] - P) + c0
            assert (cdf[i - 1] if i else 0.0) - 1e-12 <= uu <= cdf[i] + 1e-12


def test_host_sampler_edt_styles_reproduce_reference_draws_and_otsu():
    """'edt' / 'edt*roi' (dataset_dpsr.py:371-457): the probabilities equal the ones the reference handed to
    np.random.multinomial (captured in g32) and the seeded draws are the reference's.  'automatic_threshold': Otsu's
    threshold restated from skimage's published algorithm (skimage is not in this image: parity unpinned) separates a
    bimodal image where it should."""
    from dlib.datasets.lowres import otsu_threshold
    from dlib.datasets.dataset_dpsr import PatchSampler, SAMPLE_ROI, TH_FIX, TH_AUTO
    g = np.load(os.path.join(G, "g32_patch_sampler_edt.npz"))
    for name in ("a", "b"):
        img = g[f"{name}/img"]
        P, th = (int(v) for v in g[f"{name}/cfg"])
        for style, tag in (("edt", "edt"), ("edt*roi", "edtxroi")):
            s = PatchSampler(style, P, 256, TH_FIX, float(th))
            pm = s.origin_probabilities(img, float(th))
            assert np.array_equal(pm.reshape(-1), g[f"{name}/{tag}_pvals"])
            np.random.seed(int(g[f"{name}/{tag}_seed"]))
            draws = np.array([s(img, False)[:2] for _ in range(30)])
            assert np.array_equal(draws, g[f"{name}/{tag}_draws"])
    rng = np.random.RandomState(0)
    img = np.where(rng.rand(64, 64) < 0.3, rng.randint(150, 200, (64, 64)), rng.randint(5, 40, (64, 64))).astype(np.uint8)
    th = otsu_threshold(img, 256)
    assert 39 <= th < 150          # the bin centre that closes the low mode (skimage: foreground = image > threshold)
    r0, c0, roi = PatchSampler(SAMPLE_ROI, 8, 256, TH_AUTO, 0.)(img, True)
    assert roi.dtype == np.uint8 and roi.mean() == (img >= th).mean() and 0 <= r0 <= 56 and 0 <= c0 <= 56


@pytest.mark.gpu
def test_device_roi_sampler_matches_oracle_and_reference_distribution():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from srhip import ops
    from dlib.datasets.dataset_dpsr import DeviceRoiSampler
    g = load()
    tiles = [torch.from_numpy(g[f"{n}/img"]).cuda() for n in "abc"]
    gen = torch.Generator(device="cuda").manual_seed(5)
    for k, name in enumerate("abc"):
        img = g[f"{name}/img"]
        P, th = int(g[f"{name}/cfg"][0]), int(g[f"{name}/cfg"][1])
        u = torch.rand(4096, dtype=torch.float64, device="cuda", generator=gen)
        u[:4] = torch.tensor([0.0, 0.5, 1.0 - 2.0 ** -53, 0.25], dtype=torch.float64)
        org = ops.roi_sample(tiles, [k] * 4096, P, th, u).cpu().numpy()
        want = np.array([O.roi_origin_from_uniform(img, th, P, float(v)) for v in u.cpu().numpy()])
        assert np.array_equal(org, want), name                      # bit-exact against the oracle's inverse CDF
    # mixed batch through the sampler object, origins stay on the device and feed the gather
    s = DeviceRoiSampler(tiles, psize=8, threshold=7, seed=3)
    ids = [0, 2, 1, 0, 2, 2, 1, 0]
    org, u = s.sample(ids)
    for b, t in enumerate(ids):
        assert tuple(org[b].tolist()) == O.roi_origin_from_uniform(g[f"{'abc'[t]}/img"], 7, 8, float(u[b]))
    o = org.cpu().tolist()
    patch = ops.patch_gather(tiles, ids, [v[0] for v in o], [v[1] for v in o], [0] * 8, 8)
    assert patch.shape == (8, 1, 8, 8)
    # distribution: 200k draws on the small tile against the reference's probability vector
    img, P, th = g["b/img"], int(g["b/cfg"][0]), int(g["b/cfg"][1])
    n = 200000
    u = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen)
    org = ops.roi_sample(tiles, [1] * n, P, th, u).cpu().numpy()
    Wc = img.shape[1] - P
    freq = np.bincount(org[:, 0] * Wc + org[:, 1], minlength=g["b/pvals"].size) / n
    p = g["b/pvals"]
    assert np.abs(freq - p).max() <= 6.0 * np.sqrt(p.max() / n)      # 6 sigma of the largest cell
    roi_mass = p[(O.roi_origin_pmf(img, th, P).reshape(-1) > p.min() * 2)].sum()
    got = freq[(O.roi_origin_pmf(img, th, P).reshape(-1) > p.min() * 2)].sum()
    assert abs(got - roi_mass) <= 0.005


@pytest.mark.gpu
def test_resident_train_set_batches_match_numpy_crops():
    """ResidentTrainSet (TRAIN phase of DatasetDPSR on the device): fold files -> resident uint8 tiles -> per batch
    origins ('uniform' and 'roi'), LR origin = HR origin // scale, one of the 8 augmentations, uint8 -> float;
    every returned patch equals the numpy crop + augment_img + /255 of the raw tile (dataset_dpsr.py:866-894,
    914-915; the oracle's patch_batch is pinned against the reference's functions in g12)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import yaml
    from PIL import Image
    from dlib.utils.tools import Dict2Obj
    from dlib.utils.utils_dataloaders import get_train_set
    fx = os.path.join(G, "eval_exp")
    a = Dict2Obj(yaml.safe_load(open(os.path.join(fx, "exp", "config_model.yml"))))
    ds = a.test_dsets
    a.train_dsets, a.data_root, a.splits_root = ds, os.path.join(fx, "data"), os.path.join(fx, "folds")
    a.h_size, a.batch_size, a.myseed = 64, 2, 3
    raw_h = [np.asarray(Image.open(os.path.join(fx, "data", "caco2", "t", f"h_{i}.tif"))) for i in range(3)]
    raw_l = [np.asarray(Image.open(os.path.join(fx, "data", "caco2", "t", f"l_{i}.tif"))) for i in range(3)]
    for style in ("uniform", "roi"):
        a.sample_tr_patch, a.sample_tr_patch_th_style, a.sample_tr_patch_th = style, "fix_threshold", 12
        ts = get_train_set(a, "cuda")
        assert len(ts) == 1 and len(ts.hr) == 3 and ts.hr[0].dtype == torch.uint8 and ts.hr[0].is_cuda
        seen = set()
        for epoch in range(4):
            for b in ts.epoch(epoch):
                assert b["h_im"].shape == (2, 1, 64, 64) and b["l_im"].shape == (2, 1, 8, 8)
                for k in range(2):
                    i = int(b["h_id"][k].split("_")[1].split(".")[0])
                    assert b["l_id"][k] == f"t/l_{i}.tif"
                    (r0, c0), mode = b["origin"][k], b["mode"][k]
                    assert 0 <= r0 <= 128 - 64 and 0 <= c0 <= 136 - 64 and 0 <= mode <= 7
                    want_h = O.patch_batch([torch.from_numpy(raw_h[i].copy())], [0], [r0], [c0], [mode], 64)
                    want_l = O.patch_batch([torch.from_numpy(raw_l[i].copy())], [0], [r0 // 8], [c0 // 8], [mode], 8)
                    assert torch.equal(b["h_im"][k:k + 1].cpu(), want_h) and torch.equal(b["l_im"][k:k + 1].cpu(), want_l)
                    seen.add(i)
        assert seen == {0, 1, 2}          # shuffling re-seeded by set_epoch: the dropped sample rotates
