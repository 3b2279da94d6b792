"""PositionalEncoder with the reference's surface (model/src/backbones/positional_encoding.py:5-31).

Only the denominators are computed here (once, on the host, exactly as the reference __init__ does);
the sin/cos table itself is produced inside the L-TAE position-bias HIP kernel (csrc/ltae.hip)."""
import torch
import torch.nn as nn


class PositionalEncoder(nn.Module):
    def __init__(self, d, T=1000, repeat=None, offset=0):
        super().__init__()
        self.d = d
        self.T = T
        self.repeat = repeat
        self.denom = torch.pow(T, 2 * (torch.arange(offset, offset + d).float() // 2) / d)
        self.updated_location = False

    def denom_on(self, device):
        if self.denom.device != device:
            self.denom = self.denom.to(device)
            self.updated_location = True
        return self.denom

    def forward(self, batch_positions):
        """[B,T] -> [B,T,d*repeat] sinusoid table (HIP kernel; bias-free call of the position-bias op)."""
        from ... import engine as E
        from ... import hip_backend as hb
        B, T = batch_positions.shape
        D = self.d * (self.repeat or 1)
        out = torch.empty((B * T, D), device=batch_positions.device, dtype=torch.float32)
        hb.call("uncr_ltae_posbias", batch_positions.reshape(-1).contiguous().float(),
                self.denom_on(batch_positions.device), self.d, None, out, B * T, D, 1, E._stream())
        return out.view(B, T, D)
