"""Reference-style initialisation (model/src/learning/weight_init.py:4-74), restated for the module types
that occur on the uncrtaints path: Conv1d N(0,1); Conv2d / Linear xavier-normal with N(0,1) bias;
BatchNorm weight N(0,1), bias 0; GroupNorm and the attention queries untouched."""
import torch.nn as nn
import torch.nn.init as init


def weight_init(m, spread=1.0):
    if isinstance(m, nn.Conv1d):
        init.normal_(m.weight.data, mean=0, std=spread)
        if m.bias is not None:
            init.normal_(m.bias.data, mean=0, std=spread)
    elif isinstance(m, (nn.Conv2d, nn.Linear)):
        init.xavier_normal_(m.weight.data, gain=spread)
        if m.bias is not None:
            init.normal_(m.bias.data, mean=0, std=spread)
    elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
        init.normal_(m.weight.data, mean=0, std=spread)
        init.constant_(m.bias.data, 0)
