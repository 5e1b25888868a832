"""cmd/run_perturb.sh's entry point end to end on the GPU (synthetic data): same stdout lines and output files as the
reference's main_perturb.py, checkpoints in the reference's layout, --resume works, the reference-layout state_dict
loads into the CPU oracle model."""
import os
import pickle
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(args, cwd):
    env = dict(os.environ, PYTHONUNBUFFERED="1")
    r = subprocess.run([sys.executable, "-u", "main_perturb.py"] + args, cwd=cwd, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def _losses(out):
    import re
    return [float(v) for v in re.findall(r"Loss (\S+) \(", out)]


def test_run_perturb_entry_point(tmp_path, orc):
    cwd = os.path.join(ROOT, "cv_a-fan_amd")          # the reference runs from Classification/, we from the package dir
    line = open(os.path.join(cwd, "cmd", "run_perturb.sh")).read().strip().splitlines()[-1]
    assert line.startswith("python -u main_perturb.py --seed 3") and "--gamma 0.5" in line     # reference command line
    save = str(tmp_path / "run")
    common = ["--seed", "3", "--save_dir", save, "--gamma", "0.5", "--arch", "resnet20s", "--perturb_idx", "7",
              "--synthetic", "512", "--batch_size", "64", "--print_freq", "2", "--steps", "2"]
    out = _run(common + ["--epochs", "2"], cwd)
    assert "Epoch: [0][0/8]\tLoss" in out and "Epoch: [1][" in out
    # the entry point runs the product path: every 16+-channel convolution on the library's kernels, one hipGraph
    assert "convolutions outside the library's kernels: 0" in out
    # ... and the captured step survives the eager validation passes between epochs (regression: a graph holding vendor
    # convolutions read freed memory -> NaN from the first iteration of epoch 1)
    import math
    ls = _losses(out)
    assert len(ls) >= 8 and all(math.isfinite(v) for v in ls), ls
    assert "l2 mean = " in out and "linf mean = " in out and "train_accuracy" in out and "valid_accuracy" in out
    for f in ("checkpoint.pt", "best_model.pt", "result.pkl", "result_norm.pkl"):
        assert os.path.exists(os.path.join(save, f)), f
    ck = torch.load(os.path.join(save, "checkpoint.pt"), map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "state_dict", "best_prec1", "optimizer", "scheduler"} and ck["epoch"] == 2
    # reference layout: loads into the plain-torch oracle model (what main_inference.py:50-51 does with resnet56)
    ref = orc.resnet20s()
    ref.load_state_dict(ck["state_dict"])
    # optimizer state in torch.optim.SGD's layout: keyed by index in model.parameters(); index 0 is the unused `w`
    n_params = len(list(ref.parameters()))
    assert ck["optimizer"]["param_groups"][0]["params"] == list(range(n_params))
    assert 0 not in ck["optimizer"]["state"] and 1 in ck["optimizer"]["state"]
    assert ck["optimizer"]["state"][1]["momentum_buffer"].shape == ref.sequential_model[1].weight.shape
    norms = pickle.load(open(os.path.join(save, "result_norm.pkl"), "rb"))
    assert set(norms) == {"l2", "linf"} and set(norms["l2"]) == {1, 2}
    assert abs(float(norms["linf"][1]) - 2 * 0.5 / 255) < 1e-4        # K=2 unclipped steps of gamma = 0.5/255
    out2 = _run(common + ["--epochs", "3", "--resume"], cwd)
    assert "resume from checkpoint" in out2 and "Epoch: [2][" in out2 and "Epoch: [1][" not in out2


def test_main_learnable_entry_point(tmp_path, orc):
    """main_learnable.py end to end (synthetic data): the reference's stdout lines (weightK = ..., two learning rates,
    Epoch/Loss/Accuracy, l2/linf means), checkpoint keys incl. `optimizer_w`, `w` on the simplex, --resume."""
    cwd = os.path.join(ROOT, "cv_a-fan_amd")
    save = str(tmp_path / "learn")
    env = dict(os.environ, PYTHONUNBUFFERED="1")
    common = ["--seed", "3", "--save_dir", save, "--synthetic", "256", "--batch_size", "64", "--print_freq", "1",
              "--steps", "1", "--gamma", "0.5"]

    def run(extra):
        r = subprocess.run([sys.executable, "-u", "main_learnable.py"] + common + extra, cwd=cwd, env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        return r.stdout

    out = run(["--epochs", "1"])
    assert "weight1 = " in out and "weight9 = " in out and "Epoch: [0][0/4]\tLoss" in out
    assert "convolutions outside the library's kernels: 0" in out
    import math
    assert all(math.isfinite(v) for v in _losses(out))
    assert "l2 mean = " in out and "linf mean = " in out and "valid_accuracy" in out
    ck = torch.load(os.path.join(save, "checkpoint.pt"), map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "state_dict", "best_prec1", "optimizer", "optimizer_w", "scheduler"} and ck["epoch"] == 1
    w = ck["state_dict"]["w"]
    assert w.shape == (9,) and abs(float(w.sum()) - 1.0) < 1e-5
    ref = orc.resnet56s(init_weight_eta=1 / 9)
    ref.load_state_dict(ck["state_dict"])                      # reference layout (335 tensors incl. w, mean, std)
    out = run(["--epochs", "2", "--resume"])
    assert "resume from checkpoint" in out and "Epoch: [1][0/4]" in out and "Epoch: [0][" not in out
