"""the training log / checkpoint tree (reference key paths, stage/count bookkeeping), extract_NN's arg-min
selection over the LAST stage, and resuming the optimiser from an extracted file (data_writing.jl:4-78, data_extraction.jl:1-149,
train_NDE_args.jl:124-147).  Host-side only."""
import numpy as np
import pytest

from colnde import checkpoint as ck
from colnde.flux_compat import ADAM
from colnde.wind_mixing import train_NDE
from tests.test_training_loops import Quadratic


def _net(seed):
    return ck.network_record(np.random.default_rng(seed).standard_normal(6521).astype(np.float32), (96, 50, 20, 31), ("mish", "mish", "identity"))


def _losses(total):
    return dict(u=0.4 * total, v=0.2 * total, T=0.1 * total, dudz=0.15 * total, dvdz=0.1 * total, dTdz=0.05 * total)


def test_nde_log_tree_counts_stages_and_extracts_the_argmin_of_the_last_stage(tmp_path):
    path, out = str(tmp_path / "train.jld2tree"), str(tmp_path / "extracted.jld2tree")
    opts = [[ADAM(1e-3)], [ADAM(2e-4)]]
    ck.write_metadata_NDE_training(path, ["-1e-3"], [1, 1], [range(1, 10), range(1, 20)], {"ν₀": 1e-4, "ν₋": 0.1, "ΔRi": 1.0, "Riᶜ": 0.25, "Pr": 1.0},
                                   opts, _net(0), _net(1), _net(2))
    sc = dict(u=1.0, v=1.0, T=1.0, dudz=5e-3, dvdz=5e-3, dTdz=5e-3)
    totals = {1: [3.0, 0.5, 2.0], 2: [1.5, 0.9, 0.7, 0.7, 1.1]}          # stage 1 holds the global minimum: extract_NN must NOT pick it
    for stage, vals in totals.items():
        opt = opts[stage - 1][0]
        for i, t in enumerate(vals):
            opt.update(np.zeros(4, np.float32), np.full(4, 0.1 * (i + 1)))
            ck.write_data_NDE_training(path, _losses(t), sc, _net(10 * stage + i), _net(100 + i), _net(200 + i), stage, opt)
    with ck.GroupFile(path) as f:
        assert f.keys("training_data/loss/total") == ["1", "2"] and f.keys("training_data/loss/total/2") == ["1", "2", "3", "4", "5"]
        assert set(f.keys("training_data/loss")) == {"total", "profile", "gradient", "u", "v", "T", "∂u∂z", "∂v∂z", "∂T∂z"}
        assert np.isclose(f["training_data/loss/profile/1/2"], 0.7 * 0.5) and np.isclose(f["training_data/loss/gradient/1/2"], 0.3 * 0.5)
        assert f["training_info/loss_scalings"]["∂u∂z"] == 5e-3 and f.keys("training_data/optimizer") == ["state", "β", "η"]
    idx = ck.extract_NN(path, out, "NDE")
    assert idx == 3                                                          # the FIRST of the two equal minima of stage 2 (Julia argmin)
    with ck.GroupFile(out) as f:
        np.testing.assert_array_equal(f["neural_network/uw"]["theta"], _net(22)["theta"])
        np.testing.assert_allclose(f["losses/total"], totals[2])
        assert f["optimizer/η"] == 2e-4 and list(f["optimizer/β"]) == [0.9, 0.999] and f["optimizer/state"]["m"].shape == (4,)
        assert f["training_info/parameters"]["Riᶜ"] == 0.25
    with pytest.raises(KeyError):                                            # JLD2 refuses to overwrite a dataset; so does the tree
        with ck.GroupFile(path, "a") as f:
            f["training_data/loss/total/2/1"] = np.float32(0)


def test_flux_nn_log_and_extract(tmp_path):
    path, out = str(tmp_path / "uw.jld2tree"), str(tmp_path / "uw_extracted.jld2tree")
    ck.write_metadata_NN_training(path, ["-1e-3"], {"ν₀": 1e-4}, [2], [ADAM(1e-3)], _net(0), "uw")
    for i, loss in enumerate([0.3, 0.1, 0.2]):
        ck.write_data_NN_training(path, loss, _net(i))
    assert ck.extract_NN(path, out, "NN") == 2
    with ck.GroupFile(out) as f:
        np.testing.assert_array_equal(f["neural_network"]["theta"], _net(1)["theta"])
        np.testing.assert_allclose(f["losses"], [0.3, 0.1, 0.2], rtol=1e-6)
    ck.write_data_NN(str(tmp_path / "nets"), _net(1), _net(2), _net(3))
    assert ck.GroupFile(str(tmp_path / "nets")).keys("neural_network") == ["uw", "vw", "wT"]


def test_resume_from_extracted_file_continues_the_optimiser(tmp_path):
    """An interrupted run resumed through the checkpoint equals the uninterrupted one: θ, ADAM moments and running powers survive."""
    prob = Quadratic(12)
    w0 = np.zeros(12, np.float32)
    full = train_NDE(prob, w0, [ADAM(0.05)], epochs=2, maxiters=10, continue_state=True)
    opt = ADAM(0.05)
    first = train_NDE(prob, w0, [opt], epochs=1, maxiters=10, continue_state=True)
    path, out = str(tmp_path / "t"), str(tmp_path / "x")
    thirds = np.array_split(first.weights, 3)
    recs = [ck.network_record(t, (1, len(t)), ("identity",)) for t in thirds]
    ck.write_metadata_NDE_training(path, [], [1], [], {"ν₀": 1e-4}, [[opt]], *recs)
    ck.write_data_NDE_training(path, _losses(first.history[-1]["total"]), dict(u=1, v=1, T=1, dudz=0, dvdz=0, dTdz=0), *recs, 1, opt)
    ck.extract_NN(path, out, "NDE")
    weights, nets, params, opt2 = ck.load_extracted_NDE(out, rate=0.05)
    np.testing.assert_array_equal(weights, first.weights)
    second = train_NDE(prob, weights, [opt2], epochs=1, maxiters=10, continue_state=True)
    np.testing.assert_allclose(second.weights, full.weights, rtol=1e-6, atol=1e-7)


# What the reference's READER asks a training log for (wind_mixing/src/data_extraction.jl:4-86, type == "NDE"; :88-104 otherwise), as key
# templates — data, not source — with $S = "$N_stages" and $I = the entry index:
READER_KEYS_NDE = ["training_info/train_files", "training_info/parameters", "training_info/loss_scalings",
                   "training_data/neural_network/uw/$S", "training_data/loss/total/$S/$I", "training_data/loss/profile/$S/$I",
                   "training_data/loss/gradient/$S/$I", "training_data/loss/u/$S/$I", "training_data/loss/v/$S/$I", "training_data/loss/T/$S/$I",
                   "training_data/loss/∂u∂z/$S/$I", "training_data/loss/∂v∂z/$S/$I", "training_data/loss/∂T∂z/$S/$I",
                   "training_data/neural_network/uw/$S/$I", "training_data/neural_network/vw/$S/$I", "training_data/neural_network/wT/$S/$I",
                   "training_data/optimizer/η/$S/$I", "training_data/optimizer/β/$S/$I", "training_data/optimizer/state/$S/$I"]
READER_KEYS_NN = ["training_info/train_files", "training_info/parameters", "training_data/loss/$I", "training_data/neural_network/$I"]


def test_written_tree_has_exactly_the_keys_the_reference_reader_asks_for(tmp_path):
    """no JLD2 bytes, but the key list of the tree `write_*_training` produces must be the key list
    `extract_NN` of the reference reads — every reader key present for every stage and entry, and no leaf the reader does not know."""
    import os
    import re
    path = str(tmp_path / "log.tree")
    opt = ADAM(1e-3)
    ck.write_metadata_NDE_training(path, ["f"], [1], [range(1, 5)], {"Pr": 1.0}, [[opt]], _net(0), _net(1), _net(2))
    sc = dict(u=1.0, v=1.0, T=1.0, dudz=5e-3, dvdz=5e-3, dTdz=5e-3)
    for stage, n in ((1, 2), (2, 3)):
        for i in range(n):
            opt.update(np.zeros(3, np.float32), np.ones(3))
            ck.write_data_NDE_training(path, _losses(1.0 + i), sc, _net(3), _net(4), _net(5), stage, opt)
    with ck.GroupFile(path) as f:
        leaves = set()

        def walk(g):
            for k in f.keys(g):
                q = (g + "/" + k) if g else k
                if os.path.isdir(f._p(q)):
                    walk(q)
                else:
                    leaves.add(q)
        walk("training_data")
        for stage, n in (("1", 2), ("2", 3)):
            for i in range(1, n + 1):
                for t in READER_KEYS_NDE:
                    assert f.haskey(t.replace("$S", stage).replace("$I", str(i))), t
        templ = {re.sub(r"/\d+/\d+$", "/$S/$I", q) for q in leaves}
        assert templ == {t for t in READER_KEYS_NDE if t.endswith("$S/$I")}, templ ^ {t for t in READER_KEYS_NDE if t.endswith("$S/$I")}
    # the flux-NN log (type != "NDE")
    p2 = str(tmp_path / "nn.tree")
    ck.write_metadata_NN_training(p2, ["f"], {"Pr": 1.0}, [1], [ADAM(1e-3)], _net(0), "uw")
    for i in range(3):
        ck.write_data_NN_training(p2, 1.0 / (i + 1), _net(i))
    with ck.GroupFile(p2) as f:
        for i in (1, 2, 3):
            for t in READER_KEYS_NN:
                assert f.haskey(t.replace("$I", str(i))), t
    # in the build container (where the reference is readable) the templates above are checked against the reader's own text
    ref = "/root/reference/wind_mixing/src/data_extraction.jl"
    if os.path.exists(ref):
        src = open(ref, encoding="utf-8").read()
        asked = set(re.findall(r'file\["([^"]+)"\]', src[:src.index("@info \"Writing file\"")]))
        norm = {a.replace("$(N_stages)", "$S").replace("$N_stages", "$S").replace("$NN_index", "$I").replace("$i", "$I") for a in asked}
        norm.discard("training_data/loss/$S/$I")          # the legacy single-loss layout (:47-59), not written by data_writing.jl any more
        for group in ("training_info", "training_data", "training_data/loss", "training_data/neural_network/uw"):
            norm.discard(group)                            # groups the reader only counts the entries of (`keys(file[...])`)
        assert norm == set(READER_KEYS_NDE) | set(READER_KEYS_NN), norm ^ (set(READER_KEYS_NDE) | set(READER_KEYS_NN))


def test_extract_nn_indexes_the_last_stage_by_count_like_the_reference(tmp_path):
    path = str(tmp_path / "log.tree")
    opt = ADAM(1e-3)
    ck.write_metadata_NDE_training(path, ["f"], [1], [range(1, 5)], {"Pr": 1.0}, [[opt]], _net(0), _net(1), _net(2))
    sc = dict(u=1.0, v=1.0, T=1.0, dudz=5e-3, dvdz=5e-3, dTdz=5e-3)
    opt.update(np.zeros(3, np.float32), np.ones(3))
    ck.write_data_NDE_training(path, _losses(1.0), sc, _net(3), _net(4), _net(5), 3, opt)       # one stage group, named "3"
    with pytest.raises(KeyError, match="N_stages"):
        ck.extract_NN(path, str(tmp_path / "out.tree"), "NDE")
