"""Autograd bridge for the HIP hot path (training / integrated gradients).  Placeholder until the
backward kernels land: requesting gradients fails loudly instead of silently using torch ops."""


def _nyi(*a, **k):
    raise NotImplementedError("xnrs_amd: backward kernels are not built yet; run under torch.no_grad()")


mha = additive = text_encoder = user_encoder = embedding_linear = _nyi
