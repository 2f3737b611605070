"""Debug: per-parameter error of the bf16 HIP step vs the fp32 and bf16 reference fixtures."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_golden, rel_err
from test_gpu_step import _build, _InjectedDraw
from src.loss import variational_lower_bound_loss
dev = torch.device("cuda:0")
for name in ["c1_tiny", "hd96", "nomask_att"]:
    g = load_golden(f"step_{name}.npz")
    model = _build(g, dev, torch.bfloat16)
    ids = torch.from_numpy(g["input_ids"]).to(dev); att = torch.from_numpy(g["attention_mask"]).to(dev)
    model.masking_strategy = _InjectedDraw(model.masking_strategy, torch.from_numpy(g["draw"]).to(dev))
    logits, masked, mask = model(input_ids=ids, attention_mask=att, t=torch.tensor([float(g["t"][0])]))
    loss = variational_lower_bound_loss(logits=logits, targets=ids, masked_input_ids=masked, attention_mask=att, mask_token_id=103)
    loss.backward()
    print(name, "loss", loss.item(), float(g["loss_fp32"]), float(g["loss_bf16"]))
    rows = []
    for n, p in model.transformer.named_parameters():
        rows.append((rel_err(p.grad, g["grad_fp32/" + n]), rel_err(p.grad, g["grad_bf16/" + n]), rel_err(g["grad_bf16/" + n], g["grad_fp32/" + n]), n))
    rows.sort(reverse=True)
    for r in rows[:8]:
        print("   mine-vs-fp32 %.4f  mine-vs-refbf16 %.4f  refbf16-vs-fp32 %.4f  %s" % r)
    print("   median mine-vs-fp32 %.4f  ref %.4f" % (np.median([r[0] for r in rows]), np.median([r[2] for r in rows])))
