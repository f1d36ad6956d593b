"""CPU: the NPR MLP mirror has the reference's module / parameter layout (FCGF_APR/model/mlp.py)."""
import importlib.util
import os

import pytest
import torch

from apr_amd.fcgf.lib import apg


@pytest.mark.skipif(not os.path.exists("/root/reference/FCGF_APR/model/mlp.py"), reason="reference not present")
def test_generative_mlp_state_dict_equals_reference():
    spec = importlib.util.spec_from_file_location("ref_mlp", "/root/reference/FCGF_APR/model/mlp.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    for name in ("GenerativeMLP", "GenerativeMLP_98", "GenerativeMLP_54"):
        torch.manual_seed(0)
        a = getattr(ref, name)(in_channel=32, out_points=4)
        torch.manual_seed(0)
        b = getattr(apg, name)(in_channel=32, out_points=4)
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys())
        assert all(torch.equal(sa[k], sb[k]) for k in sa)


def test_layout_without_reference():
    m = apg.GenerativeMLP_98(in_channel=128, out_points=4)
    sd = m.state_dict()
    assert sd["mlp.0.weight"].shape == (512, 128) and sd["mlp.3.weight"].shape == (256, 512)
    assert sd["mlp.6.weight"].shape == (12, 256) and "mlp.2.running_var" in sd
