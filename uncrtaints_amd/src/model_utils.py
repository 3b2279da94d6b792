"""Factory with the reference's flag -> constructor mapping (model/src/model_utils.py:85-108)."""
from .backbones import uncrtaints

S1_BANDS = 2
S2_BANDS = 13


def get_generator(config):
    if config.model != "uncrtaints":
        raise NotImplementedError(f"model '{config.model}' is outside the MI355X hot path (uncrtaints only)")
    return uncrtaints.UNCRTAINTS(
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


def get_model(config):
    from .backbones.base_model import BaseModel
    return BaseModel(config)
