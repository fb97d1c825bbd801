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
