"""Config c2: bf16 inference of the float model on the HIP kernels vs the fp32 definition of the same module (the stock-module
CPU path, which tests/test_oracle_golden.py pins to the reference on G5).  Tolerance: bf16 storage of weights and of every layer
output (8 mantissa bits, ~70 roundings deep) -> norm-wise relative error of the logits <= 3e-2, arg-max agreement on >= 7/8 images."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _randomize_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.rand(m.num_features, generator=g) * 0.8 + 0.6
            m.bias.data = torch.rand(m.num_features, generator=g) * 0.2 - 0.1
            m.running_mean.data = torch.randn(m.num_features, generator=g) * 0.1
            m.running_var.data = torch.rand(m.num_features, generator=g) * 0.5 + 0.5


@pytest.mark.parametrize("name,res,batch", [("frostnet_small_1_0", 64, 4), ("frostnet_large_1_0", 224, 8), ("frostnet_base_0_75", 96, 3)])
def test_bf16_inference_vs_fp32_definition(name, res, batch):
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import frostnet as F
    torch.manual_seed(7)
    model = F.MODEL_REGISTRY[name]()
    _randomize_bn(model, 11)
    model.eval()
    x = torch.randn(batch, 3, res, res)
    with torch.no_grad():
        ref = model(x)                                  # CPU: stock torch modules = the reference's definition
    model.cuda()
    out = model.hip_infer_bf16(x.cuda()).cpu()
    out_cl = model.hip_infer_bf16(x.cuda().contiguous(memory_format=torch.channels_last)).cpu()
    assert torch.equal(out, out_cl)                     # layout of the input must not matter
    rel = float((out - ref).norm() / ref.norm())
    agree = int((out.argmax(1) == ref.argmax(1)).sum())
    assert rel <= 3e-2, rel
    assert agree >= batch - max(1, batch // 8), (agree, batch)
    with pytest.raises(RuntimeError):
        model.train(); model.hip_infer_bf16(x.cuda())
