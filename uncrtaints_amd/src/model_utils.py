"""Factory with the reference's flag -> constructor mapping (model/src/model_utils.py:85-108)."""
from .backbones import uncrtaints

S1_BANDS = 2
S2_BANDS = 13


def get_generator(config):
    if config.model != "uncrtaints":
        raise NotImplementedError(f"model '{config.model}' is outside the MI355X hot path (uncrtaints only)")
    net = uncrtaints.UNCRTAINTS(
        input_dim=S1_BANDS * config.use_sar + S2_BANDS,
        encoder_widths=config.encoder_widths,
        decoder_widths=config.decoder_widths,
        out_conv=config.out_conv,
        out_nonlin_mean=config.mean_nonLinearity,
        out_nonlin_var=config.var_nonLinearity,
        agg_mode=config.agg_mode,
        encoder_norm=config.encoder_norm,
        decoder_norm=config.decoder_norm,
        n_head=config.n_head,
        d_model=config.d_model,
        d_k=config.d_k,
        pad_value=config.pad_value,
        padding_mode=config.padding_mode,
        positional_encoding=config.positional_encoding,
        covmode=config.covmode,
        scale_by=config.scale_by,
        separate_out=config.separate_out,
        use_v=config.use_v,
        block_type=config.block_type,
        is_mono=config.pretrain,
    )
    # not a reference flag: config.act_dtype = 'bf16' stores the activations as bf16 (BASELINE config 3); default fp32
    act = getattr(config, "act_dtype", None)
    if act is not None:
        net.set_act_dtype(act)
    return net


def get_model(config):
    from .backbones.base_model import BaseModel
    return BaseModel(config)


# ------------------------------------------------------------------------------------------------
# checkpoint I/O with the reference's file layout and semantics (model/src/model_utils.py:117-231)
# ------------------------------------------------------------------------------------------------
import os  # noqa: E402

import torch  # noqa: E402


def save_model(config, epoch, model, name):
    """`<res_dir>/<experiment_name>/<name>.pth.tar` holding epoch, the BaseModel and netG state dicts and the
    optimizer / scheduler states (model_utils.py:117-125)."""
    payload = {"epoch": epoch,
               "state_dict": model.state_dict(),
               "state_dict_G": model.netG.state_dict(),
               "optimizer_G": model.optimizer_G.state_dict(),
               "scheduler_G": model.scheduler_G.state_dict()}
    out_dir = os.path.join(config.res_dir, config.experiment_name)
    os.makedirs(out_dir, exist_ok=True)
    torch.save(payload, os.path.join(out_dir, f"{name}.pth.tar"))


def freeze_layers(net, apply_to=None, grad=False):
    """Set requires_grad on every (non-integer) parameter, or only on those named in `apply_to` with a matching
    shape (model_utils.py:221-231)."""
    if net is None:
        return
    for k, v in net.named_parameters():
        if v.dtype == torch.int64:
            continue
        if apply_to is None or (k in apply_to and v.size() == apply_to[k].size()):
            v.requires_grad_(grad)


def load_model(config, model, train_out_layer=True, load_out_partly=True):
    """Load `state_dict_G` from `config.trained_checkp` into model.netG (model_utils.py:128-196): strictly if the
    architectures match and the output layer is not to be re-trained; otherwise the shape-compatible subset,
    optionally copying the first 13 (mean) kernels of the output layer, freezing what was loaded (all but the
    output layer when `train_out_layer`).  With `config.resume_from` the optimizer/scheduler states follow."""
    ckpt = torch.load(config.trained_checkp, map_location=config.device)
    pretrained = dict(ckpt["state_dict_G"])
    own = model.netG.state_dict()
    same = pretrained.keys() == own.keys()
    print(f"The new and the (pre-)trained model architectures are {'' if same else 'not '}identical.\n")
    S2 = S2_BANDS
    strict_ok = False
    if not train_out_layer:
        try:
            model.netG.load_state_dict(pretrained, strict=True)
            strict_ok = True
        except Exception:
            strict_ok = False
    if strict_ok:
        freeze_layers(model.netG, grad=True)
        model.frozen, frozen = False, []
    else:
        wk, bk = "out_conv.conv.conv.0.weight", "out_conv.conv.conv.0.bias"
        if load_out_partly and wk in pretrained and wk in own:
            # the first 13 (mean) kernels are copied into the new output layer; the entries then no longer match
            # the full layer's shape and are dropped by the size filter below (reference behaviour)
            w, b = own[wk], own[bk]
            with torch.no_grad():
                w[:S2, ...] = pretrained[wk][:S2, ...]
                b[:S2, ...] = pretrained[bk][:S2, ...]
            pretrained[wk], pretrained[bk] = w[:S2, ...], b[:S2, ...]
        pretrained = {k: v for k, v in pretrained.items() if k in own and v.size() == own[k].size()}
        own.update(pretrained)
        model.netG.load_state_dict(own, strict=False)
        model.frozen = True
        freeze_layers(model.netG, grad=True)
        if train_out_layer:
            loaded = {k: v for k, v in pretrained.items() if "out_conv.conv.conv.0" not in k}
        else:
            loaded = pretrained
        freeze_layers(model.netG, apply_to=loaded, grad=False)
        frozen = list(loaded.keys())
    train_these = [k for k in own.keys() if k not in frozen]
    print(f"\nFroze these layers: {frozen}")
    print(f"\nTrain these layers: {train_these}")
    if getattr(config, "resume_from", False):
        model.optimizer_G.load_state_dict(ckpt["optimizer_G"])
        model.scheduler_G.load_state_dict(ckpt["scheduler_G"])


def load_checkpoint(config, checkp_dir, model, name):
    """Load `<checkp_dir>/<experiment_name>/<name>.pth.tar`["state_dict"] into `model`; if the strict load fails,
    retry with the legacy key rename in_block<k> / out_block<k> -> in_block.<k-1> / out_block.<k-1>
    (model_utils.py:201-219)."""
    path = os.path.join(checkp_dir, config.experiment_name, f"{name}.pth.tar")
    print(f"Loading checkpoint {path}")
    state = torch.load(path, map_location=config.device)["state_dict"]
    try:
        model.load_state_dict(state, strict=True)
        return
    except Exception:
        pass
    renamed = {}
    for key, val in state.items():
        parts = key.split(".")
        if len(parts) > 1 and ("in_block" in key or "out_block" in key) and parts[1][-1:].isdigit():
            parts[1] = parts[1][:-1] + "." + str(int(parts[1][-1]) - 1)
            key = ".".join(parts)
        renamed[key] = val
    model.load_state_dict(renamed, strict=False)
