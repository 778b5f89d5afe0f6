"""GPU parity of the learners (IPPOLearner.train, Prediction_policy.learn, Behavior_policy.learn) through
the reference's API against fixtures recorded from the real reference (tests/golden)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["ippo_train", "ippo_train_mpe"])
def test_ippo_train_matches_reference(golden, tag):
    from tests.test_emu_learners import check_ippo_train
    check_ippo_train(golden(tag), "cuda")
