"""CPU: checkpoint I/O keeps the reference's file layout and load semantics."""
from types import SimpleNamespace

import torch

from uncrtaints_amd.src import model_utils as MU
from uncrtaints_amd.src.backbones.base_model import BaseModel


def _cfg(tmp, covmode="diag", **kw):
    oc = 26 if covmode == "diag" else 14
    c = SimpleNamespace(model="uncrtaints", use_sar=True, encoder_widths=[128], decoder_widths=[128] * 5, out_conv=[oc],
                        mean_nonLinearity=True, var_nonLinearity="softplus", agg_mode="att_group", encoder_norm="group",
                        decoder_norm="batch", n_head=16, d_model=256, d_k=4, pad_value=0, padding_mode="reflect",
                        positional_encoding=True, covmode=covmode, scale_by=1.0, separate_out=False, use_v=False,
                        block_type="mbconv", pretrain=False, loss="MGNLL", lr=1e-3, gamma=0.9, device="cpu",
                        chunk_size=None, res_dir=str(tmp), experiment_name="exp", resume_from=False, trained_checkp="")
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_save_then_load_checkpoint_roundtrip(tmp_path):
    cfg = _cfg(tmp_path)
    torch.manual_seed(0)
    m1 = BaseModel(cfg)
    MU.save_model(cfg, 3, m1, "model_epoch_3")
    blob = torch.load(tmp_path / "exp" / "model_epoch_3.pth.tar")
    assert set(blob) == {"epoch", "state_dict", "state_dict_G", "optimizer_G", "scheduler_G"} and blob["epoch"] == 3
    torch.manual_seed(1)
    m2 = BaseModel(cfg)
    MU.load_checkpoint(cfg, str(tmp_path), m2, "model_epoch_3")
    for (k1, v1), (k2, v2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_load_checkpoint_legacy_block_names(tmp_path):
    cfg = _cfg(tmp_path)
    torch.manual_seed(0)
    m1 = BaseModel(cfg)
    legacy = {}
    for k, v in m1.state_dict().items():
        parts = k.split(".")
        if parts[1] in ("in_block", "out_block"):          # netG.in_block.0.x -> netG.in_block1.x
            parts = [parts[0], parts[1] + str(int(parts[2]) + 1)] + parts[3:]
        legacy[".".join(parts)] = v
    (tmp_path / "exp").mkdir()
    torch.save({"state_dict": legacy}, tmp_path / "exp" / "old.pth.tar")
    torch.manual_seed(2)
    m2 = BaseModel(cfg)
    MU.load_checkpoint(cfg, str(tmp_path), m2, "old")
    for (k1, v1), (k2, v2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(v1, v2), k1


def test_load_model_partial_and_freeze(tmp_path):
    # pre-trained iso model (14 outputs) -> diag model (26 outputs): everything but the output layer loads and
    # is frozen; the first 13 output kernels are copied; the output layer stays trainable
    cfg_iso, cfg_diag = _cfg(tmp_path, "iso"), _cfg(tmp_path, "diag")
    torch.manual_seed(0)
    src = BaseModel(cfg_iso)
    MU.save_model(cfg_iso, 1, src, "pre")
    cfg_diag.trained_checkp = str(tmp_path / "exp" / "pre.pth.tar")
    torch.manual_seed(5)
    dst = BaseModel(cfg_diag)
    MU.load_model(cfg_diag, dst, train_out_layer=True, load_out_partly=True)
    sd_s, sd_d = src.netG.state_dict(), dst.netG.state_dict()
    for k in sd_s:
        if "out_conv.conv.conv.0" not in k:
            assert torch.equal(sd_s[k], sd_d[k]), k
    assert torch.equal(sd_d["out_conv.conv.conv.0.weight"][:13], sd_s["out_conv.conv.conv.0.weight"][:13])
    assert dst.frozen
    for k, p in dst.netG.named_parameters():
        assert p.requires_grad == ("out_conv.conv.conv.0" in k), k
    # strict path: same architecture, keep everything trainable, restore optimizer/scheduler when resuming
    cfg_iso.trained_checkp, cfg_iso.resume_from = cfg_diag.trained_checkp, True
    torch.manual_seed(7)
    again = BaseModel(cfg_iso)
    MU.load_model(cfg_iso, again, train_out_layer=False)
    assert not again.frozen and all(p.requires_grad for p in again.netG.parameters())
    assert torch.equal(again.netG.state_dict()["in_conv.conv.conv.0.weight"], sd_s["in_conv.conv.conv.0.weight"])


def _ref_ckpt_cfg(tmp_path):
    import json
    import os
    import shutil
    from conftest import GOLDEN, load_golden
    g = load_golden("g17_refcheckpoint")
    meta = json.loads(str(g["meta"]))
    cfg = _cfg(tmp_path, "diag", encoder_widths=meta["encoder_widths"], decoder_widths=meta["decoder_widths"],
               d_model=meta["d_model"], out_conv=meta["out_conv"], lr=meta["lr"], gamma=meta["gamma"])
    os.makedirs(tmp_path / "exp", exist_ok=True)
    shutil.copy(os.path.join(GOLDEN, "g17_refcheckpoint.pth.tar"), tmp_path / "exp" / "model_epoch_7.pth.tar")
    return g, meta, cfg


def test_load_checkpoint_written_by_the_reference(tmp_path):
    """A `.pth.tar` written by the REFERENCE's save_model (model_utils.py:117-125; fixture g17 from make_golden.py) loads
    into the build's BaseModel through load_checkpoint: same keys, same numbers (checksums recorded at generation time)."""
    from conftest import checksum
    g, meta, cfg = _ref_ckpt_cfg(tmp_path)
    blob = torch.load(tmp_path / "exp" / "model_epoch_7.pth.tar")
    assert set(blob) == {"epoch", "state_dict", "state_dict_G", "optimizer_G", "scheduler_G"} and blob["epoch"] == meta["epoch"]
    torch.manual_seed(3)
    m = BaseModel(cfg)
    assert list(blob["state_dict_G"].keys()) == list(m.netG.state_dict().keys())
    assert list(blob["state_dict"].keys()) == list(m.state_dict().keys())
    MU.load_checkpoint(cfg, str(tmp_path), m, "model_epoch_7")
    for k, v in m.netG.state_dict().items():
        ref = g["sum/" + k]
        got = checksum(v.double().numpy())
        assert abs(got - ref).max() <= 1e-9 * max(1.0, abs(ref).max()), k
    # the optimizer / scheduler states are in the reference's format as well: they load into torch's Adam / ExponentialLR
    m.optimizer_G.load_state_dict(blob["optimizer_G"])
    m.scheduler_G.load_state_dict(blob["scheduler_G"])
    st = m.optimizer_G.state_dict()["state"]
    assert [float(v["step"]) for v in st.values()] == list(g["adam_step"])


def test_load_model_from_a_reference_checkpoint(tmp_path):
    """load_model (model_utils.py:128-231): pre-trained weights out of the reference-written file into a fresh model."""
    from conftest import checksum
    g, meta, cfg = _ref_ckpt_cfg(tmp_path)
    cfg.trained_checkp = str(tmp_path / "exp" / "model_epoch_7.pth.tar")
    torch.manual_seed(4)
    m = BaseModel(cfg)
    fresh = {k: v.clone() for k, v in m.netG.state_dict().items()}
    MU.load_model(cfg, m)
    blob = torch.load(cfg.trained_checkp)["state_dict_G"]
    for k, v in m.netG.state_dict().items():
        if k.startswith("out_conv.conv.conv.0"):
            # reference semantics (model_utils.py:147-156): the first 13 output kernels are copied, the rest keeps its own
            # initialisation, and the layer stays trainable
            assert torch.equal(v[:13], blob[k][:13]) and torch.equal(v[13:], fresh[k][13:]), k
            continue
        ref = g["sum/" + k]
        assert abs(checksum(v.double().numpy()) - ref).max() <= 1e-9 * max(1.0, abs(ref).max()), k
    trainable = [k for k, p in m.netG.named_parameters() if p.requires_grad]
    assert trainable == ["out_conv.conv.conv.0.weight", "out_conv.conv.conv.0.bias"] and m.frozen


def test_weight_init_reproduces_the_reference_stream():
    """the build's classes + weight_init consume torch's RNG exactly like the reference's
    (`netG.apply(weight_init)`, train_reconstruct.py:627): under the same seed every parameter is bit-identical
    (fixture g16: checksums recorded from the reference, two constructor variants)."""
    import numpy as np
    from conftest import checksum, load_golden
    from uncrtaints_amd.src.backbones import uncrtaints as U
    from uncrtaints_amd.src.learning.weight_init import weight_init
    g = load_golden("g16_weightinit")
    for tag, kw in (("diag", dict(covmode="diag", out_conv=[26])), ("iso_usev", dict(covmode="iso", out_conv=[14], use_v=True))):
        torch.manual_seed(7)
        m = U.UNCRTAINTS(**{**dict(input_dim=15, out_nonlin_mean=True, out_nonlin_var="softplus", scale_by=1.0), **kw})
        m.apply(weight_init)
        sd = m.state_dict()
        keys = [k[len(tag) + 5:] for k in g.files if k.startswith(tag + "/sum/")]
        assert sorted(keys) == sorted(k for k, v in sd.items() if v.dtype.is_floating_point)
        for k in keys:
            assert np.array_equal(checksum(sd[k].numpy()), g[f"{tag}/sum/{k}"]), (tag, k)
        assert np.array_equal(m.temporal_encoder.attention_heads.Q.detach().numpy(), g[f"{tag}/Q"])
        assert np.array_equal(m.in_conv.conv.conv[0].bias.detach().numpy(), g[f"{tag}/in_conv_bias"])
