"""config.hip_graph in the validation loop for the model variants: every eval forward must be captured and replay bit-identically to
the eager forward (run on the GPU box)."""
import sys, warnings; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from types import SimpleNamespace
import torch
from uncrtaints_amd.src.backbones.base_model import BaseModel
dev = "cuda"
base = dict(model="uncrtaints", use_sar=True, encoder_widths=[128], decoder_widths=[128] * 5, out_conv=[26],
            mean_nonLinearity=True, var_nonLinearity="softplus", agg_mode="att_group", encoder_norm="group",
            decoder_norm="batch", n_head=16, d_model=256, d_k=4, pad_value=0, padding_mode="reflect",
            positional_encoding=True, covmode="diag", scale_by=1.0, separate_out=False, use_v=False,
            block_type="mbconv", pretrain=False, loss="MGNLL", lr=1e-3, gamma=1.0, device=dev, chunk_size=None, hip_graph=True)
variants = {"default": {}, "use_v": dict(use_v=True), "residual": dict(block_type="residual"), "att_mean": dict(agg_mode="att_mean"),
            "mean": dict(agg_mode="mean"), "separate_out": dict(separate_out=True), "iso": dict(covmode="iso", out_conv=[14]),
            "w64": dict(encoder_widths=[64], decoder_widths=[64] * 3, n_head=8)}
g = torch.Generator().manual_seed(0)
for name, kw in variants.items():
    cfg = SimpleNamespace(**{**base, **kw})
    torch.manual_seed(1)
    try:
        m = BaseModel(cfg).to(dev).eval()
    except Exception as e:
        print(name, "construction:", type(e).__name__, str(e)[:80]); continue
    ok = True
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for it in range(4):
            x = torch.rand(2, 3, 15, 64, 64, generator=g); y = torch.rand(2, 1, 13, 64, 64, generator=g)
            dates = torch.tensor([[0, 10, 25], [3, 14, 40]])
            with torch.no_grad():
                m.set_input({"A": x, "B": y, "dates": dates, "masks": None}); m.forward()
                want = m.netG(m.real_A, batch_positions=m.dates)
            ok = ok and torch.equal(m.fake_B, want)
    st = [("graph" if v["graph"] not in (None, False) else str(v["graph"])) for k, v in m._graphs.items() if k[0] == "eval_forward"]
    print(name, "bit-identical" if ok else "MISMATCH", st, [str(x.message)[:90] for x in w if "hip_graph" in str(x.message)])
