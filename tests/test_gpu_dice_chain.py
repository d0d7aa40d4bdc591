"""North-star output #2: "Dice on labelled val ... within +-0.3" -- the chain pre-train -> fine-tune -> ``EvalEpocher`` ->
``UniversalDice`` on the HIP path (f32 storage and the benchmarked bf16 storage) against the CPU oracle doing the same
steps from the same seeds (tests/_dice_chain.py; reference: main_pretrain_encoder.py:21-38, val.py:45-66,
semi_seg/epochers/new_epocher.py:56-97,241-289, contrastyou/meters/general_dice_meter.py:19-175)."""
import numpy as np
import pytest
import torch

from tests import _dice_chain as DC

pytestmark = pytest.mark.gpu

_ORACLE = {}


def _oracle():
    if "r" not in _ORACLE:
        _ORACLE["d"] = DC.make_data()
        _ORACLE["r"] = DC.run_oracle(_ORACLE["d"])
    return _ORACLE["d"], _ORACLE["r"]


# Dice points = percent: the north star's +-0.3 is 0.003 on the [0, 1] scale of UniversalDice
DICE_BAR = 0.003


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_pretrain_finetune_val_dice_matches_the_oracle(dt):
    data, ref = _oracle()
    h = DC.HYPER
    got = DC.run_hip(data, torch.float32 if dt == "fp32" else torch.bfloat16)
    # the task is really learned on both sides (a chain that learns nothing would agree trivially at the Dice of a constant map)
    assert ref["dsc"]["DSC_mean"] > 0.9 and got["dsc"]["DSC_mean"] > 0.9, (ref["dsc"], got["dsc"])
    assert ref["ft_curve"][-1] < 0.5 * ref["ft_curve"][0] and got["ft_curve"][-1] < 0.5 * got["ft_curve"][0]
    # the trainer ran the schedule its config names
    np.testing.assert_allclose(got["lrs"], [DC.epoch_lr(h, e) for e in range(1, h["max_epoch"] + 1)], rtol=1e-6)
    # per-class Dice of the validation scans after the last epoch, and the best validation score the trainer kept: the north
    # star's own bar
    for c in ("DSC1", "DSC2", "DSC3", "DSC_mean"):
        assert abs(got["dsc"][c] - ref["dsc"][c]) <= DICE_BAR, (dt, c, got["dsc"], ref["dsc"])
    assert abs(got["score"] - ref["dsc"]["DSC_mean"]) <= DICE_BAR
    assert abs(got["best_score"] - ref["best_score"]) <= DICE_BAR
    # the curves: contrastive loss of the pre-train steps, supervised loss of the fine-tune steps, validation loss and Dice per
    # epoch once the annealing has started to settle them (earlier epochs move by whole points per step on both sides)
    np.testing.assert_allclose(got["pre_curve"], ref["pre_curve"], rtol=2e-3 if dt == "fp32" else 1e-2)
    ft_tol = 2e-2 if dt == "fp32" else 6e-2
    rel = np.abs(np.array(got["ft_curve"]) - np.array(ref["ft_curve"])) / np.array(ref["ft_curve"])
    assert rel.max() <= ft_tol, (dt, float(rel.max()), int(rel.argmax()))
    assert abs(got["val_loss"] - ref["val_loss"]) <= ft_tol * ref["val_loss"]
    tail = [abs(a["DSC_mean"] - b["DSC_mean"]) for a, b in zip(got["dice_curve"][-10:], ref["dice_curve"][-10:])]
    assert max(tail) <= 3 * DICE_BAR, (dt, tail)
    print(f"\n[dice chain {dt}] oracle {ref['dsc']}  hip {got['dsc']}  ft loss {ref['ft_curve'][0]:.4f} -> "
          f"{ref['ft_curve'][-1]:.4f} (hip {got['ft_curve'][-1]:.4f}), max rel curve distance {rel.max():.2e}, "
          f"Dice tail distance {max(tail):.2e}")
