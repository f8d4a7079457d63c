"""Round-4 additions on the GPU: sub-module flag switches, float-runner dropout stream in checkpoints, frozen-BatchNorm training
(frostnet_features.py:354-359 `_freeze_stages`), the fp32-gradient parity mode, the fused-block inference kernels."""
import numpy as np
import pytest
import torch

from oracle import frost_oracle as O

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def relerr(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def F():
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import frostnet
    assert torch.cuda.is_available()
    return frostnet


def test_submodule_flag_switches_reach_the_runner(F):
    """`model.layer3.apply(disable_fake_quant)` (a sub-module, not the root) must be refused like the root-level call, and in eval a
    site re-enabled through a sub-module `.apply` must observe again although the host summary said "all observers off"."""
    import torch.ao.quantization as aoq
    torch.manual_seed(3)
    model = F.MODEL_REGISTRY["frostnet_quant_small_1_0"](drop_rate=0.0)
    F.qat_prepare(model, version=0)
    model.cuda().train()
    x = torch.randn(4, 3, 64, 64, device="cuda")
    with torch.no_grad():
        model(x)
    model.layer3.apply(aoq.disable_fake_quant)
    with pytest.raises(NotImplementedError, match="fake_quant_enabled"):
        with torch.no_grad():
            model(x)
    model.layer3.apply(aoq.enable_fake_quant)
    with torch.no_grad():
        model(x)
    model.eval()
    model.apply(aoq.disable_observer)
    site = model.layer2[0].conv1.conv[0].activation_post_process
    with torch.no_grad():
        model(x)
    before = site.activation_post_process.max_val.clone()
    with torch.no_grad():
        model(3.0 * x)
    assert torch.equal(before, site.activation_post_process.max_val)          # every observer is off: nothing moves
    model.layer2[0].conv1.apply(aoq.enable_observer)                          # ONE site back on, through a sub-module
    with torch.no_grad():
        model(3.0 * x)
    assert not torch.equal(before, site.activation_post_process.max_val), "the re-enabled observer did not run"
    # a flag written straight into the buffer (no apply at all) is seen by the next eval forward too
    model.layer2[0].conv1.apply(aoq.disable_observer)
    other = model.layer1[1].conv1.conv[0].activation_post_process
    b2 = other.activation_post_process.max_val.clone()
    other.observer_enabled[0] = 1
    with torch.no_grad():
        model(5.0 * x)
    assert not torch.equal(b2, other.activation_post_process.max_val)


def test_float_runner_dropout_stream_is_checkpointed(F, tmp_path):
    """the float (StatAssist warm-up) runner draws dropout from the shared device Philox stream; harness.save_checkpoint / load_checkpoint
    must carry it, and restoring must write the draw counter in place."""
    from frostnet_amd import harness
    from frostnet_amd.optimizer import QSGD
    torch.manual_seed(5)
    model = F.MODEL_REGISTRY["frostnet_quant_small_1_0"](drop_rate=0.2).cuda().train()
    opt = QSGD([{"params": [p]} for p in model.parameters()], lr=1e-3, momentum=0.9)
    x = torch.randn(4, 3, 64, 64, device="cuda")
    for _ in range(2):
        model(x).sum().backward()
        opt.zero_grad()
    r = model.hip_runner()
    assert type(r).__name__ == "FloatRunner" and r.rng_state()["draws"] == 2
    path = str(tmp_path / "ck.pth.tar")
    harness.save_checkpoint(harness.checkpoint_state(model, opt, epoch=0), path)
    ctr = r._drop_ctr
    model(x).sum().backward()
    assert r.rng_state()["draws"] == 3
    ck = harness.load_checkpoint(model, opt, path)
    assert "hip_rng" in ck and r.rng_state()["draws"] == 2
    assert model.hip_runner()._drop_ctr.data_ptr() == ctr.data_ptr()          # restored in place
