"""`torch.device('cuda')` without an index is the rank's CURRENT device (verdict round 4, item 6): every handle-creating call site resolves it through
iris_amd._lib.device_index; CPU devices are refused (no fallback)."""
import pytest
import torch


def test_device_index_resolution(monkeypatch):
    from iris_amd import _lib as L
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 3)
    assert L.device_index(torch.device("cuda")) == 3
    assert L.device_index("cuda") == 3
    assert L.device_index(torch.device("cuda", 5)) == 5
    assert L.device_index("cuda:0") == 0
    with pytest.raises(L.IrisError):
        L.device_index("cpu")


def test_no_call_site_defaults_to_device_zero():
    """the `device.index or 0` idiom (None -> 0: rank 3's tables on GPU 0) must not come back"""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "iris_amd")
    bad = []
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                for i, line in enumerate(open(os.path.join(d, f)), 1):
                    if re.search(r"\.index\s+or\s+0", line):
                        bad.append((f, i))
    assert not bad, bad
