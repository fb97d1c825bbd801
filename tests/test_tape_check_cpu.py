"""The launch tape's recorder refuses a step that contains framework (ATen) kernels: a replay would skip them (mgsv_amd/tape.py)."""
import pytest
import torch

from mgsv_amd import tape as T


def test_views_and_allocations_pass_the_watcher():
    x = torch.arange(12.0).reshape(3, 4)
    with T.LaunchTape.record(check=True, _every_device=True) as tp:
        y = x[:, 1:3].transpose(0, 1)
        z = torch.empty_like(x).view(-1)
        assert y.shape == (2, 3) and z.numel() == 12
    assert tp.foreign_ops == []
    assert tp.counts() == (0, 0, 0)
    tp.close()
    assert tp.handle.value == 0


def test_a_framework_kernel_inside_the_recording_is_refused_by_name():
    x = torch.ones(8)
    with pytest.raises(T.ForeignKernelError, match=r"aten::add x1.*aten::mul x2"):
        with T.LaunchTape.record(check=True, _every_device=True):
            y = x + 1
            y = y * 2
            y = y * 3
    # check=False: recorded, reported, not refused
    with T.LaunchTape.record(check=False, _every_device=True) as tp:
        x.zero_()
    assert tp.foreign_ops == ["zero_"]
    assert not T.recording()


def test_host_callbacks_are_replayed_in_issue_order_and_are_not_counted_as_framework_kernels():
    """made_tape_callback (the data-parallel step's gradient all-reduces ride the tape this way): the action runs once while recording and
    once per replay, in order; what it does to tensors is not flagged by the watcher; an exception inside a replayed callback surfaces
    from replay(); made_tape_interleave leaves a tape of callbacks as it is."""
    log = []
    x = torch.zeros(3)
    with T.LaunchTape.record(check=True, _every_device=True) as tp:
        T.callback(lambda: (log.append("a"), x.add_(1)))
        T.callback(lambda: log.append("b"))
    assert log == ["a", "b"] and tp.foreign_ops == [] and float(x[0]) == 1.0
    assert tp.counts() == (0, 0, 2)
    tp.interleave(2)
    tp.replay(); tp.replay()
    assert log == ["a", "b"] * 3 and float(x[0]) == 3.0
    boom = {"on": False}

    def maybe():
        if boom["on"]:
            raise ValueError("inside the replayed callback")
    with T.LaunchTape.record() as tp2:
        T.callback(maybe)
    tp2.replay()
    boom["on"] = True
    with pytest.raises(ValueError, match="inside the replayed callback"):
        tp2.replay()
    T.callback(lambda: log.append("c"))                      # outside a recording: just runs
    assert log[-1] == "c"
