"""eval.py on the libsrhip path against a reference-format experiment folder.

tests/golden/eval_exp/ was written by oracle/make_goldens.py::g_eval_fixture with the REFERENCE's
own code: config_model.yml from its get_config(), best-models/G-model.pth from its SwinIR, and
expected/ = what its evaluate_single_ds / save_tracker wrote (details_*.yml, <ds>.yaml, roi-*.yaml,
tracker.pkl, roi_tracker.pkl) for the model and for the bicubic baseline row.

CPU: host logic (fold files, TIFF reading, config -> registry, tracker arithmetic and file
formats).  GPU: the whole eval.py run, file by file."""
import os
import pickle
import shutil
import sys

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "eval_exp")
DS = "caco2_test_X_8_in_64_out_512_cell_CELL0"
MTRS = ("psnr", "mse", "nrmse", "ssim", "psnr_y")


def _args():
    from dlib.utils.tools import Dict2Obj
    with open(os.path.join(FX, "exp", "config_model.yml")) as f:
        a = Dict2Obj(yaml.safe_load(f))
    a.data_root, a.splits_root = os.path.join(FX, "data"), os.path.join(FX, "folds")
    return a


def test_config_folds_and_images_load():
    from dlib.utils import constants
    from dlib.utils.utils_dataloaders import get_all_eval_loaders, get_pairs
    from dlib.models.select_network import define_G
    a = _args()
    assert a.test_dsets == DS and a.scale == 8 and a.eval_over_roi_also and a.model_select_mtr == constants.PSNR_MTR
    net = define_G(a)
    sd = torch.load(os.path.join(FX, "exp", "best-models", "G-model.pth"), map_location="cpu")
    assert list(sd.keys()) == list(net.state_dict().keys())
    net.load_state_dict(sd, strict=True)
    pairs = get_pairs(os.path.join(FX, "folds", DS, "l_h.txt"))
    assert list(pairs.items())[0] == ("t/l_0.tif", "t/h_0.tif")
    loaders = get_all_eval_loaders(a, a.test_dsets)
    assert list(loaders) == [DS] and len(loaders[DS]) == 2 and len(loaders[DS].dataset) == 3
    batches = list(loaders[DS])
    assert batches[0]["l_im"].shape == (2, 1, 16, 17) and batches[0]["h_im"].shape == (2, 1, 128, 136)
    assert batches[1]["h_id"] == ["t/h_2.tif"] and batches[0]["l_im"].dtype == torch.float32
    from PIL import Image
    raw = np.asarray(Image.open(os.path.join(FX, "data", "caco2", "t", "h_1.tif")))
    assert torch.equal(batches[0]["h_im"][1, 0], torch.from_numpy(np.float32(raw / 255.)))


def test_tracker_arithmetic_and_files_match_reference(tmp_path):
    """Feeding the dataset means the reference measured through THIS build's tracker functions gives the
    reference's tracker.pkl / roi_tracker.pkl and <ds>.yaml / roi-<ds>.yaml, key for key."""
    from dlib.utils import constants
    from dlib.utils import utils_tracker as T
    from dlib.utils.utils_trainer import _fast_update_tracker
    a = _args()
    exp = os.path.join(FX, "expected")
    want_t = pickle.load(open(os.path.join(exp, "tracker.pkl"), "rb"))
    want_r = pickle.load(open(os.path.join(exp, "roi_tracker.pkl"), "rb"))
    tracker, roi = T.init_tracker(a), T.init_tracker(a)
    assert {k: list(v) for k, v in tracker.items()} == {k: list(v) for k, v in want_t.items()}
    for name in (DS, f"{DS}_{a.basic_interpolation}"):
        T.reset_tracker_eval(tracker, constants.TESTSET, name)
        T.reset_tracker_eval(roi, constants.TESTSET, name)
        full = yaml.safe_load(open(os.path.join(exp, "best-models", f"{name}.yaml")))
        part = yaml.safe_load(open(os.path.join(exp, "best-models", f"roi-{name}.yaml")))
        tracker, best = _fast_update_tracker(a, tracker, {m: full[f"last_{m}"] for m in MTRS}, constants.TESTSET, name)
        roi, _ = _fast_update_tracker(a, roi, {m: part[f"last_{m}"] for m in MTRS}, constants.TESTSET, name, best)
        got = T.write_current_perf_eval(tracker, constants.TESTSET, name, str(tmp_path), f"{name}.yaml", -1, -1)
        assert got == full and yaml.safe_load(open(tmp_path / f"{name}.yaml")) == full
        assert T.write_current_perf_eval(roi, constants.TESTSET, name, None, "x", -1, -1) == part
        assert T.is_last_perf_best_perf(tracker, roi, True, False, constants.TESTSET, name, constants.PSNR_MTR)
    assert tracker == want_t and roi == want_r
    T.save_tracker(str(tmp_path), tracker, roi)
    t2, r2 = T.find_last_tracker(str(tmp_path), a)
    assert t2 == want_t and r2 == want_r
    msg = T.current_perf_to_str(full, part, a.model_select_mtr, False)
    assert "MASTER" in msg and "ROI:" in msg
    # validation-style history: the master metric picks the index, the others follow it
    tr = T.init_tracker(a)
    vs = a.valid_dsets
    for p, m in ((30.0, 5.0), (32.0, 7.0), (31.0, 1.0)):
        tr, best = T.update_tracker_eval(tr, constants.VALIDSET, vs, constants.PSNR_MTR, torch.tensor(p))
        tr, none = T.update_tracker_eval(tr, constants.VALIDSET, vs, constants.MSE_MTR, np.array(m), best)
        assert none is None
    assert tr[constants.VALIDSET][vs]["psnr"] == {"vals": [30.0, 32.0, 31.0], "best_val": 32.0}
    assert tr[constants.VALIDSET][vs]["mse"]["best_val"] == 7.0 and best == 1
    tr = T.update_tracker_train(tr, ["master_loss", "l1"], [0.5, 0.25], constants.PR_ITER)
    tr = T.update_tracker_train(tr, ["master_loss", "l1"], [0.75, 0.125], constants.PR_ITER)
    assert tr[constants.TRAINSET][constants.PR_ITER]["l1"] == {"vals": [0.25, 0.125], "best_val": 0.125}


@pytest.mark.gpu
def test_eval_py_reproduces_the_reference_evaluation(tmp_path):
    """sr-caco-2_amd/eval.py on a copy of the experiment folder: per-image details (model row and bicubic
    row, full image and ROI-averaged), the summary yamls and both trackers against what the reference wrote.
    Tolerances: PSNR / PSNR_Y 0.01 dB (north_star), MSE / NRMSE 1e-4 relative, SSIM 5e-5 (fp32 SSIM maps)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import importlib.util
    exp = tmp_path / "exp"
    shutil.copytree(os.path.join(FX, "exp"), exp)
    spec = importlib.util.spec_from_file_location("srhip_eval", os.path.join(ROOT, "sr-caco-2_amd", "eval.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    tracker, roi = ev.evaluate_pretrained(["--cudaid", "0", "--exp_path", str(exp), "--data_root",
                                           os.path.join(FX, "data"), "--splits_root", os.path.join(FX, "folds")])
    want = os.path.join(FX, "expected")

    def close(name, a, b):
        if name in ("psnr", "psnr_y"):
            assert abs(a - b) <= 0.01, (name, a, b)
        elif name == "ssim":
            assert abs(a - b) <= 5e-5, (name, a, b)
        else:
            assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (name, a, b)

    out = exp / f"eval_test_{DS}"
    for f in ("log.txt", "log.json", "tracker.pkl", "roi_tracker.pkl"):
        assert (out / f).is_file(), f
    for name in (DS, f"{DS}_bicubic"):
        for pre in ("details_", "roi_details_"):
            got = yaml.safe_load(open(exp / "best-models" / f"{pre}{name}.yml"))
            ref = yaml.safe_load(open(os.path.join(want, "best-models", f"{pre}{name}.yml")))
            assert list(got) == list(ref) == ["t/h_0.tif", "t/h_1.tif", "t/h_2.tif"]
            for img in ref:
                assert set(got[img]) == set(ref[img]) == set(MTRS)
                for m in MTRS:
                    close(m, got[img][m], ref[img][m])
        for pre in ("", "roi-"):
            got = yaml.safe_load(open(exp / "best-models" / f"{pre}{name}.yaml"))
            ref = yaml.safe_load(open(os.path.join(want, "best-models", f"{pre}{name}.yaml")))
            assert set(got) == set(ref)
            for k, v in ref.items():
                if isinstance(v, float):
                    close(k.split("_", 1)[1], got[k], v)
                else:
                    assert got[k] == v
    for got, file in ((tracker, "tracker.pkl"), (roi, "roi_tracker.pkl")):
        ref = pickle.load(open(os.path.join(want, file), "rb"))
        assert pickle.load(open(out / file, "rb")) == got
        assert {k: list(v) for k, v in got.items()} == {k: list(v) for k, v in ref.items()}
        for name, node in ref["test"].items():
            for m, rec in node.items():
                assert len(got["test"][name][m]["vals"]) == len(rec["vals"])
                for a, b in zip(got["test"][name][m]["vals"] + [got["test"][name][m]["best_val"]],
                                rec["vals"] + [rec["best_val"]]):
                    close(m, a, b)
    # the current weights came back after the best-model evaluation; predictions of the first images exist
    assert (exp / "best-models" / "G-current_model.pth").is_file()
    assert (exp / "images" / "test" / DS / "t_h_0.tif.png").is_file()
