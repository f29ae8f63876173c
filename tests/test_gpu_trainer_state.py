"""Trainer state: the 10-minute synchronization file (annonet_train_main.cpp:403), resume, and the learning-rate schedule.

  * SetSynchronizationFile is called BEFORE SetClassCount in the reference (annonet_train_main.cpp:400-405): the resume must
    wait until the net structure is final, for any class count;
  * save -> new trainer -> load -> continue equals an uninterrupted run bit for bit (parameters, momentum, running statistics
    incl. the bn update counters, learning rate, the losses still ahead of the schedule's lag);
  * the step at which a loss enters the schedule is fixed (lag of anh_trainer::kLossLag steps), so the learning-rate history
    does not depend on how often the host synchronises or polls."""
import numpy as np
import pytest

import annonet_amd as aa

pytestmark = pytest.mark.gpu

D = 35   # valid for 2 levels


def batch(seed, n=3, classes=4):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (n, D, D, 3), dtype=np.uint8)
    lab = rng.integers(0, classes, (n, D, D)).astype(np.uint16)
    lab[rng.random((n, D, D)) < 0.1] = aa.LABEL_IGNORE
    return list(img), [aa.set_weights(l, 0.5, 0.5) for l in lab]


def make_trainer(classes=4, sync=None, precision=aa.ANH_FP32, seed=5, lr=0.05):
    """the reference's call order: Initialize, SetNetWidth, SetSynchronizationFile, BeVerbose, SetClassCount, rates (annonet_train_main.cpp:400-410)"""
    t = aa.TrainingNet(2, 3, precision, seed=seed)
    t.Initialize()
    t.SetNetWidth(0.25, 4)
    if sync:
        t.SetSynchronizationFile(sync, 600)
    t.SetClassCount(classes)
    t.SetLearningRate(lr)
    t.SetLearningRateShrinkFactor(0.5)
    t.SetIterationsWithoutProgressThreshold(6)
    t.SetPreviousLossValuesDumpAmount(2)
    t.SetAllBatchNormalizationRunningStatsWindowSizes(3)   # small window: the bn update counters saturate inside the test
    return t


def state_of(t):
    p, r = t.get_params()
    return p, r, t.get_momentum(), t.GetLearningRate(), t.step_count()


def test_sync_file_resume_with_the_reference_call_order(tmp_path):
    path = str(tmp_path / "annonet_trainer_state_file.dat")
    a = make_trainer(classes=4)
    for i in range(5):
        a.StartTraining(*batch(i))
    a.save_state(path)
    want = state_of(a)
    b = make_trainer(classes=4, sync=path)     # names the file while the class count is still the default 3
    got = state_of(b)                          # first use of the net: resumes
    for w, g in zip(want, got):
        np.testing.assert_array_equal(g, w)
    c = make_trainer(classes=3, sync=path)     # a different net: the file must be refused, not half-loaded
    with pytest.raises(aa.AnnonetHipError, match="different net"):
        c.get_params()


@pytest.mark.parametrize("precision", [aa.ANH_FP32, aa.ANH_BF16])
def test_save_load_continue_equals_uninterrupted_run(tmp_path, precision):
    path = str(tmp_path / "state.dat")
    ref = make_trainer(precision=precision)
    lr_ref = []
    for i in range(14):
        ref.StartTraining(*batch(i))
        lr_ref.append(ref.GetLearningRate())
    a = make_trainer(precision=precision)
    for i in range(7):
        a.StartTraining(*batch(i))
    a.save_state(path)
    b = make_trainer(precision=precision, seed=99)   # different init: everything must come from the file
    b.load_state(path)
    lr_b = lr_ref[:7]
    for i in range(7, 14):
        b.StartTraining(*batch(i))
        lr_b.append(b.GetLearningRate())
    assert lr_b == lr_ref
    for w, g in zip(state_of(ref), state_of(b)):
        np.testing.assert_array_equal(g, w)
    assert abs(ref.get_last_loss() - b.get_last_loss()) == 0


def test_learning_rate_history_does_not_depend_on_host_timing():
    runs = []
    for polling in (False, True):
        t = make_trainer(lr=1e-7)             # a rate too small to move the loss: the history is a plateau from the first step
        lrs = []
        for i in range(30):
            t.StartTraining(*batch(i % 4))    # a repeating batch sequence: the loss plateaus and the rate shrinks
            if polling:
                t.synchronize()
                t.get_last_loss()
            lrs.append(t.GetLearningRate())
        runs.append((lrs, state_of(t)))
    assert runs[0][0] == runs[1][0]
    assert runs[0][0][-1] < 1e-7, "the schedule never shrank the rate: the test does not exercise the lag"
    for w, g in zip(runs[0][1], runs[1][1]):
        np.testing.assert_array_equal(g, w)


@pytest.mark.parametrize("damage", ["other_net", "old_magic", "truncated"])
def test_a_state_file_that_cannot_be_loaded_is_an_error_not_a_fresh_start(tmp_path, damage):
    """The reference's host loop asks GetLearningRate() before anything else (annonet_train_main.cpp:583): a synchronization file
    of another net, of the older format or cut short must still surface as an error on StartTraining — never as a silent restart
    from step 0 whose periodic save would overwrite the user's state — and the file must be left as it was."""
    path = tmp_path / "annonet_trainer_state_file.dat"
    a = make_trainer(classes=4)
    for i in range(3):
        a.StartTraining(*batch(i))
    a.save_state(str(path))
    blob = path.read_bytes()
    classes = 4
    if damage == "other_net":
        classes = 3
    elif damage == "old_magic":
        blob = blob.replace(b"ANHTS002", b"ANHTS001", 1)
    else:
        blob = blob[:len(blob) // 2]
    path.write_bytes(blob)
    b = make_trainer(classes=classes, sync=str(path))
    lr = b.GetLearningRate()                  # the getter cannot report the failure ...
    assert lr == 0.05
    for _ in range(2):                        # ... every call with a status does, for as long as the file is there
        with pytest.raises(aa.AnnonetHipError, match="different net|older format|truncated|corrupt"):
            b.StartTraining(*batch(0, classes=classes))
    assert b.step_count() == 0
    assert path.read_bytes() == blob
    path.unlink()                             # the user removes the file: training starts from scratch
    b.StartTraining(*batch(0, classes=classes))
    assert b.step_count() == 1


def test_posted_losses_survive_the_ring_wrapping_around():
    """The update kernel posts each step's loss into a pinned ring of 256 tagged slots (api.cpp: loss_ring); the host reads a slot
    kLossLag steps later, or at once when asked.  300 steps: a trainer that asks after every step and one that never asks until the
    end see the same last loss, the same learning-rate history and the same parameters — the slots are reused without mixing steps."""
    data = [batch(100 + i % 7) for i in range(300)]
    a, b = make_trainer(lr=0.02), make_trainer(lr=0.02)
    losses, rates = [], []
    for d in data:
        a.StartTraining(*d)
        losses.append(a.get_last_loss())     # forces the newest slot to be read right away
        rates.append(a.GetLearningRate())
    for d in data:
        b.StartTraining(*d)                  # losses reach the schedule through the lagged reads only
    assert b.step_count() == a.step_count() == 300
    assert b.get_last_loss() == losses[-1]
    assert b.GetLearningRate() == rates[-1]
    assert len(set(rates)) > 1, "the schedule never moved: the test would not see a mixed-up loss"
    assert all(np.isfinite(losses)) and len(set(losses)) > 100   # (the rate shrinks to nothing on this random data: late losses repeat)
    pa, ra = a.get_params()
    pb, rb = b.get_params()
    np.testing.assert_array_equal(pb, pa)
    np.testing.assert_array_equal(rb, ra)
