"""GPU parity of the learners (IPPOLearner.train, Prediction_policy.learn, Behavior_policy.learn) through
the reference's API against fixtures recorded from the real reference (tests/golden)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["ippo_train", "ippo_train_mpe", "ippo_train_tanh"])
def test_ippo_train_matches_reference(golden, tag):
    from tests.test_emu_learners import check_ippo_train
    check_ippo_train(golden(tag), "cuda")


def test_prediction_learn_matches_reference(golden):
    from tests.test_emu_learners import check_prediction_learn
    check_prediction_learn(golden("prediction_learn"), "cuda")


def test_behavior_learn_matches_reference(golden):
    from tests.test_emu_learners import check_behavior_learn
    check_behavior_learn(golden("behavior_learn"), "cuda")


@pytest.mark.parametrize("tag", ["small", "wide", "hwy"])
def test_gat_backward_matches_reference(golden, tag):
    from tests.test_emu_gat_backward import check_gat_backward
    check_gat_backward(golden("gat_" + tag), "cuda")


def test_behavior_hard_learn_matches_reference(golden):
    from tests.test_emu_learners import check_behavior_hard_learn
    check_behavior_hard_learn(golden("behavior_hard_learn"), "cuda")


def test_behavior_fc_learn_matches_reference(golden):
    from tests.test_emu_learners import check_behavior_fc_learn
    check_behavior_fc_learn(golden("behavior_fc_learn"), "cuda")
