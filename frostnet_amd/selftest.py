"""smoke(): one tiny QAT step on cuda:0 checked against the CPU oracle (filled in below by the model code)."""


def smoke():
    import torch
    from . import load_library
    load_library()
    assert torch.cuda.is_available(), "smoke() needs a GPU"
