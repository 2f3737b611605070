"""TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing else).

CPU restatement, in numpy float32, of the optimizer update on the training-step path: the
reference builds ``torch.optim.AdamW(model.parameters(), lr=..., weight_decay=...)`` (main.py:87-91)
and calls ``optimizer.step()`` once per step (src/trainer.py:75,137).  The arithmetic lives in torch
(pinned torch==2.1.0 in the reference's pyproject.toml:16), not in the reference tree; this file
restates torch's documented single-tensor AdamW algorithm (decoupled weight decay, bias-corrected
moments, amsgrad off) and tests/test_oracle_golden.py pins it against torch.optim.AdamW itself.
"""
import numpy as np


def adamw_step(p, g, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
    """One AdamW update; all arrays float32, `step` counts from 1.  Returns (p, m, v)."""
    f = np.float32
    g = (g * f(grad_scale)).astype(f)
    p = (p * f(1.0 - lr * weight_decay)).astype(f)
    m = (m + (g - m) * f(1.0 - beta1)).astype(f)
    v = (v * f(beta2) + f(1.0 - beta2) * g * g).astype(f)
    bc1 = f(1.0 - beta1 ** step)
    bc2_sqrt = f(np.sqrt(1.0 - beta2 ** step))
    denom = (np.sqrt(v) / bc2_sqrt + f(eps)).astype(f)
    p = (p - f(lr) / bc1 * (m / denom)).astype(f)
    return p, m, v
