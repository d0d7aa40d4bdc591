"""How far the fp32 CPU oracle's own gradients move under fp32 rounding noise.

The oracle's gradients are not continuous in the inputs: every ReLU and every 2x2 max-pool is a decision, and in the small
networks of the decoder tests (batch statistics over 32 values, a few thousand decisions) some of them always sit within
1e-6 (relative) of a tie.  Round 5, tools/diag/dense_hook_partial2.py: multiplying the first block's output by 1 + 1e-7 noise
leaves the oracle-vs-device gradient differences at 7e-6, 1 + 1e-6 noise moves them to 6.7e-2 and 1 + 1e-5 noise to 6.4e-2 --
one decision falling the other way, not an amplified error; perturbing the INPUT images by 2e-6 moved the oracle's own
gradients by 1e-2 .. 1.4e-1 on every one of twenty seeds.  The device's exact-f32 convolutions (v_mfma_f32_16x16x4_f32) follow
the CPU's rounding closely enough to make the same decisions, and those tests hold them to 5e-3; a correct fp32 implementation
with another rounding -- the split-bf16 products, 4e-7 .. 1e-6 from fp64 per layer like the exact path -- cannot be held to
more than the oracle's own sensitivity, which this measures."""
import numpy as np
import torch


def relmax(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def oracle_sensitivity(inputs, oracle_grads, eps=2e-6, seed=1000):
    """``oracle_grads(*inputs)`` -> {name: numpy gradient}; the largest relative (to the tensor's largest entry) change of any
    of them when every input (images in [0, 1]) is perturbed by ``eps`` x standard normal noise."""
    g0 = oracle_grads(*inputs)
    gen = torch.Generator().manual_seed(seed)
    g1 = oracle_grads(*tuple(t + eps * torch.randn(t.shape, generator=gen) for t in inputs))
    return max(relmax(g1[k], g0[k]) for k in g0)
