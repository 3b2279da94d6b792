"""Semantics of gfx950's ds_read_b64_tr_b16 (LDS transpose read), measured: which LDS elements does lane l receive?
Run on the GPU box: python tools/probe_tr_b16.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uncrtaints_amd import engine as E, hip_backend as hb


def run(offs):
    o = torch.tensor(offs, dtype=torch.int32, device="cuda")
    out = torch.zeros(256, dtype=torch.int32, device="cuda")
    hb.call("uncr_debug_tr_b16_probe", o, out, E._stream())
    torch.cuda.synchronize()
    return out.view(64, 4).cpu().tolist()


def show(title, offs, lanes=range(0, 64)):
    r = run(offs)
    print("==", title)
    for l in lanes:
        print(f"  lane {l:2d} (offset {offs[l]:4d}): {r[l]}")


# (1) every lane reads its own 4 consecutive elements: offs = 4*l
show("offs = 4*lane", [4 * l for l in range(64)], range(0, 20))
# (2) all lanes the same address
show("offs = 0 for all", [0] * 64, range(0, 6))
# (3) rows of a [K][N] bf16 matrix with N = 16 elements per row: lane l -> row (l % 16), 4-element column block (l / 16)
show("offs = 16*(l%16) + 4*(l/16)", [16 * (l % 16) + 4 * (l // 16) for l in range(64)], range(0, 64))
# (4) rows of 64 elements: lane -> row l%16, column block 4*(l/16)
show("offs = 64*(l%16) + 4*(l/16)", [64 * (l % 16) + 4 * (l // 16) for l in range(64)], range(0, 20))
