"""Achieved HBM GB/s of the bandwidth-bound kernels of the training step (default-model shapes,
65,536 tokens) through the C ABI: algorithmic bytes (what the kernel must read and write once)
divided by the event-timed launch duration, against the 8 TB/s HBM3E peak.  Prints a table and
writes gpurun_out/hbm_kernels.json."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from src import _native as N

dev = torch.device("cuda:0")
PEAK = 8000.0
B, S, d, F, H, V = 128, 512, 768, 1024, 8, 30522
hd = d // H
T = B * S
Vp = (V + 63) // 64 * 64
ROWS = int(os.environ.get("CE_ROWS", 24576))      # logits rows the loss reads (about 0.38 * T on average)
lib = N.lib()
st = N.stream
torch.manual_seed(0)
bf = torch.bfloat16


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


results = []


def report(name, nbytes, fn, note=""):
    s = timeit(fn)
    gbs = nbytes / s / 1e9
    results.append({"kernel": name, "bytes": nbytes, "us": s * 1e6, "GB/s": gbs, "frac_of_peak": gbs / PEAK, "note": note})
    print(f"{name:28s} {nbytes / 1e6:9.1f} MB  {s * 1e6:8.1f} us  {gbs:7.0f} GB/s  {100 * gbs / PEAK:5.1f} %  {note}", flush=True)


# ---- forward masking: 17 B/token (in-kernel Philox draw), 21 B/token with the injected draw
ids = torch.randint(1000, V, (B, S), device=dev, dtype=torch.int64)
ids[:, 0], ids[:, -1] = 101, 102
draw = torch.rand(B, S, device=dev)
mo = torch.empty_like(ids)
mb = torch.empty(B, S, dtype=torch.uint8, device=dev)
for tokens_b, tokens_s in ((B, S), (16 * B, S)):
    ids_x = ids.repeat(tokens_b // B, 1)
    draw_x = draw.repeat(tokens_b // B, 1)
    mo_x, mb_x = torch.empty_like(ids_x), torch.empty(tokens_b, tokens_s, dtype=torch.uint8, device=dev)
    n = tokens_b * tokens_s
    report(f"mask_fwd philox T={n}", 17 * n,
           lambda: N.check(lib.ldm_mask_fwd(ids_x.data_ptr(), None, 0, 0.5, None, 0.0, tokens_b, tokens_s, 101, 102, 0, 103,
                                            7, 0, mo_x.data_ptr(), mb_x.data_ptr(), st()), "mask"))
    report(f"mask_fwd draw   T={n}", 21 * n,
           lambda: N.check(lib.ldm_mask_fwd(ids_x.data_ptr(), draw_x.data_ptr(), 0, 0.5, None, 0.0, tokens_b, tokens_s, 101,
                                            102, 0, 103, 7, 0, mo_x.data_ptr(), mb_x.data_ptr(), st()), "mask"))

# ---- RMSNorm
x = torch.randn(T, d, device=dev).to(bf)
w = torch.ones(d, device=dev)
y = torch.empty_like(x)
rstd = torch.empty(T, device=dev)
report("rmsnorm_fwd", 2 * T * d * 2 + 4 * T,
       lambda: N.check(lib.ldm_rmsnorm_fwd(x.data_ptr(), w.data_ptr(), T, d, 1e-12, 1, y.data_ptr(), rstd.data_ptr(), st()), "n"))
dy = torch.randn(T, d, device=dev).to(bf)
dres = torch.randn(T, d, device=dev).to(bf)
dx = torch.empty_like(x)
dw = torch.empty(d, device=dev)
ws = torch.empty(lib.ldm_rmsnorm_bwd_ws_floats(T, d), device=dev)
report("rmsnorm_bwd (+residual grad)", 4 * T * d * 2 + 4 * T,
       lambda: N.check(lib.ldm_rmsnorm_bwd(x.data_ptr(), w.data_ptr(), rstd.data_ptr(), dy.data_ptr(), dres.data_ptr(), T, d,
                                           1, dx.data_ptr(), dw.data_ptr(), ws.data_ptr(), st()), "nb"),
       "x, dy, dres read; dx written; includes the dw second stage")

# ---- RoPE on q and k in place
qkv = torch.randn(T, 3 * d, device=dev).to(bf)
inv = 1.0 / (10000 ** (torch.arange(0, hd, 2, device=dev).float() / hd))
fr = torch.outer(torch.arange(S, device=dev).float(), inv)
cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
report("rope_qk (in place)", 2 * T * 2 * d * 2,
       lambda: N.check(lib.ldm_rope_qk(qkv.data_ptr(), 1, B, S, H, hd, 3 * d, cos.data_ptr(), sin.data_ptr(),
                                       1.0 / math.sqrt(hd), 0, st()), "rope"))

# ---- SwiGLU
ag = torch.randn(T, 2 * F, device=dev).to(bf)
h = torch.empty(T, F, device=dev, dtype=bf)
report("swiglu_fwd", 3 * T * F * 2,
       lambda: N.check(lib.ldm_swiglu_fwd(ag.data_ptr(), T, F, 1, h.data_ptr(), st()), "sw"))
dh = torch.randn(T, F, device=dev).to(bf)
dag = torch.empty_like(ag)
report("swiglu_bwd", 5 * T * F * 2,
       lambda: N.check(lib.ldm_swiglu_bwd(ag.data_ptr(), dh.data_ptr(), T, F, 1, dag.data_ptr(), st()), "swb"))

# ---- embedding
table = torch.randn(V, d, device=dev)
emb = torch.empty(T, d, device=dev, dtype=bf)
mids = torch.where(draw < 0.5, torch.full_like(ids, 103), ids)
report("embed_fwd", T * 8 + T * d * 2 + min(T, V) * d * 4,
       lambda: N.check(lib.ldm_embed_fwd(mids.data_ptr(), table.data_ptr(), T, d, V, emb.data_ptr(), 1, st()), "e"),
       "ids + output + each touched table row once")
dtab = torch.zeros(V, d, device=dev)
report("embed_bwd", T * 8 + T * d * 2,
       lambda: N.check(lib.ldm_embed_bwd(mids.data_ptr(), emb.data_ptr(), 1, T, d, V, 103, dtab.data_ptr(), st()), "eb"),
       "ids + dout read; float atomics into the table not counted")

# ---- masked cross-entropy over ROWS participating rows of bf16 logits
logits = torch.randn(ROWS, Vp, device=dev).to(bf)
tg = torch.randint(0, V, (ROWS,), device=dev, dtype=torch.int64)
allrows = torch.ones(ROWS, dtype=torch.uint8, device=dev)
lse = torch.empty(ROWS, device=dev)
tmp = torch.empty(ROWS, device=dev)
out = torch.empty(2, device=dev)
report("ce_fwd", ROWS * V * 2 + 8 * ROWS,
       lambda: N.check(lib.ldm_ce_fwd(logits.data_ptr(), 1, Vp, ROWS, V, 1, tg.data_ptr(), None, 0, allrows.data_ptr(), None,
                                      None, 0, lse.data_ptr(), tmp.data_ptr(), out[0:1].data_ptr(), out[1:2].data_ptr(),
                                      st()), "ce"))
ident = torch.arange(ROWS, dtype=torch.int32, device=dev)
cnt = torch.tensor([ROWS], dtype=torch.int32, device=dev)
g1 = torch.ones(1, device=dev)
dl = torch.empty(ROWS, Vp, device=dev, dtype=bf)
report("ce_bwd_compact", 2 * ROWS * V * 2 + 8 * ROWS,
       lambda: N.check(lib.ldm_ce_bwd_compact(logits.data_ptr(), 1, Vp, ROWS, V, 1, tg.data_ptr(), None, 0, lse.data_ptr(),
                                              out[1:2].data_ptr(), g1.data_ptr(), ident.data_ptr(), cnt.data_ptr(),
                                              dl.data_ptr(), Vp, st()), "ceb"))

os.makedirs("gpurun_out", exist_ok=True)
json.dump({"peak_GB/s": PEAK, "shapes": {"B": B, "S": S, "d": d, "F": F, "V": V, "ce_rows": ROWS},
           "method": "algorithmic bytes / event-timed average of 20 launches", "kernels": results},
          open("gpurun_out/hbm_kernels.json", "w"), indent=1)

# ---- one-pass loss + gradient over compact rows (read once, written once)
logits2 = torch.randn(ROWS, Vp, device=dev).to(bf)
cnt_i = torch.tensor([ROWS], dtype=torch.int32, device=dev)
report("ce_fwd_grad (one pass)", 2 * ROWS * V * 2 + 8 * ROWS,
       lambda: N.check(lib.ldm_ce_fwd_grad(logits2.data_ptr(), 1, Vp, ROWS, V, tg.data_ptr(), cnt_i.data_ptr(), 0, 0.0,
                                           lse.data_ptr(), tmp.data_ptr(), out[0:1].data_ptr(), out[1:2].data_ptr(), st()),
                       "cefg"))
json.dump({"peak_GB/s": PEAK, "shapes": {"B": B, "S": S, "d": d, "F": F, "V": V, "ce_rows": ROWS},
           "method": "algorithmic bytes / event-timed average of 20 launches", "kernels": results},
          open("gpurun_out/hbm_kernels.json", "w"), indent=1)
