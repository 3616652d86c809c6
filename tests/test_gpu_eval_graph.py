"""ModelPlain.test() with the evaluation forward replayed from a hipGraph (--eval_graph True / SRHIP_EVAL_GRAPH=1): the same
bits as the eager forward, for networks of each engine family; a weight change drops the graph."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))

CASES = [("swinir", "SWINIR", 4, ["--swinir_depths", "2+2", "--swinir_embed_dim", "60", "--swinir_num_heads", "6+6"]),
         ("EDSR_LIIF", "EDSR_LIIF", 4, []), ("DBPN", "DBPN", 2, []), ("OmniSR", "OmniSR", 4, ["--OmniSR_res_num", "1"]),
         ("ACT", "ACT", 2, ["--ACT_n_resblocks", "2"]), ("DFCAN", "DFCAN", 2, []), ("ENLCN", "ENLCN", 2, ["--ENLCN_n_resblock", "8"]),
         # MemNet on fp16 storage checks its output for finiteness on the host: not inside a capture
         ("MemNet", "MemNet", 2, ["--amp", "True", "--MemNet_num_memory_blocks", "2", "--MemNet_num_residual_blocks", "2"])]


@pytest.mark.parametrize("net_type,method,scale,extra", CASES)
def test_eval_graph_replays_the_eager_forward(net_type, method, scale, extra):
    import main as M
    from dlib.models.select_model import define_model
    argv = ["--net_type", net_type, "--method", method, "--task", "super-resolution", "--scale", str(scale), "--n_channels", "1",
            "--h_size", "96", "--batch_size", "2"] + extra
    torch.manual_seed(0)
    model = define_model(M.parse_input(argv + ["--eval_graph", "False"]))
    model.init_train()
    batch = M.synth_batch(2, scale, 96, model.device, 3)
    model.feed_data(batch)
    default_on = bool(getattr(model.netG, "eval_graph_default", False))     # OmniSR asks for the replay itself
    if default_on:
        os.environ["SRHIP_EVAL_GRAPH"] = "0"                               # the eager reference
    model.test()
    ref = model.E.clone()
    os.environ.pop("SRHIP_EVAL_GRAPH", None)
    assert not model._eval_graphs
    model.args.eval_graph = True
    outs = []
    for _ in range(3):                      # eager (creates buffers), capture + replay, replay
        model.test()
        outs.append(model.E.clone())
    key = next(iter(model._eval_graphs))
    assert model._eval_graphs[key]["g"] is not None
    for o in outs:
        assert torch.equal(o, ref)
    # another input through the same graph
    batch2 = M.synth_batch(2, scale, 96, model.device, 4)
    model.feed_data(batch2)
    model.test()
    got = model.E.clone()
    model.args.eval_graph = False
    if default_on:
        os.environ["SRHIP_EVAL_GRAPH"] = "0"
    model.test()
    os.environ.pop("SRHIP_EVAL_GRAPH", None)
    assert torch.equal(got, model.E)


@pytest.mark.parametrize("net_type,method,extra", [("DFCAN", "DFCAN", []), ("NLSN", "NLSN", ["--NLSN_n_resblocks", "4", "--NLSN_n_feats", "64"]),
                                                   ("swinir", "SWINIR", ["--swinir_depths", "2+2", "--swinir_embed_dim", "60",
                                                                         "--swinir_num_heads", "6+6"])])
def test_captured_training_step_survives_a_larger_validation_forward(net_type, method, extra):
    """the captured training step holds raw addresses of grow-only scratch buffers ('fft2_ws', 'gate_ws':
    DFCAN; 'nlsa_sort', 'nlsa_ret': NLSN) and of the engines' named buffers; a validation forward on a LARGER image between
    two replays re-allocates them.  The buffers carry a generation count now: the step notices, runs eagerly once and
    re-captures.  Train (graph) -> test() on a larger image -> train again == the same sequence with eager steps."""
    import main as M
    from dlib.models.select_model import define_model
    from srhip import ops
    argv = ["--net_type", net_type, "--method", method, "--task", "super-resolution", "--scale", "2", "--n_channels", "1",
            "--h_size", "64", "--batch_size", "2", "--G_optimizer_type", "sgd", "--G_optimizer_lr", "0.01"] + extra
    results = []
    for graph in (True, False):
        torch.manual_seed(0)
        model = define_model(M.parse_input(argv + ["--train_graph", str(graph)]))
        model.init_train()
        for b in getattr(model.netG, "swin_blocks", lambda: [])():
            b.drop_prob = 0.0
        small = M.synth_batch(2, 2, 64, model.device, 3)
        big = M.synth_batch(2, 2, 192, model.device, 5)
        outs = []
        for it in range(3):                  # eager (buffers), capture + replay, replay
            torch.manual_seed(100 + it)      # NLSN draws its LSH rotations from the global generator
            model.feed_data(small)
            model.optimize_parameters(0, it)
        if graph:
            assert model.step_fn._graph is not None and model.step_fn._graph["g"] is not None
            gen0 = ops.realloc_generation()
        torch.manual_seed(7)
        model.feed_data(big)
        model.test()                         # grows the scratch buffers / re-makes the engine's evaluation buffers
        outs.append(model.E.clone())
        if graph and net_type == "DFCAN":     # (the fused SwinIR engine keeps evaluation buffers of their own and NLSN's sort
            # scratch is a fixed 16 bytes since the in-tree counting sort: their larger forward replaces nothing -- then there
            # is nothing to invalidate and the replays must simply stay right)
            assert ops.realloc_generation() != gen0, "the larger forward replaced no buffer: the test does not test"
        for it in range(3, 6):
            torch.manual_seed(100 + it)
            model.feed_data(small)
            model.optimize_parameters(0, it)
        torch.cuda.synchronize()
        assert model.check_finite()
        if graph:
            assert model.step_fn._graph["g"] is not None         # re-captured
        results.append((outs[0], model.step_fn.fp.flat.clone()))
    if net_type == "NLSN":
        # NLSN's backward scatters with index_add_ (float atomics: two eager runs differ in the last bits too): a replay on
        # freed scratch memory shows up as garbage, not as rounding
        assert torch.allclose(results[0][0], results[1][0], rtol=1e-3, atol=1e-4)
        assert torch.allclose(results[0][1], results[1][1], rtol=1e-3, atol=1e-5), (results[0][1] - results[1][1]).abs().max().item()
        return
    assert torch.equal(results[0][0], results[1][0])
    assert torch.equal(results[0][1], results[1][1]), (results[0][1] - results[1][1]).abs().max().item()
