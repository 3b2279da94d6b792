"""Training wrapper with the reference's surface (model/src/backbones/base_model.py:10-131): holds netG,
the criterion, Adam and ExponentialLR; `optimize_parameters` runs forward -> zero_grad -> loss/backward
-> step -> rescale in the reference's order.  Device->host copies of the predictions (base_model.py:130-131)
are opt-in (`config.export_to_host`), they are a per-step sync used only for logging."""
import torch
import torch.nn as nn

from .. import losses, model_utils

S2_BANDS = 13


class BaseModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.frozen = False
        self.len_epoch = 0
        self.scale_by = config.scale_by
        self.netG = model_utils.get_generator(self.config)
        self.criterion = losses.get_loss(self.config)
        self.log_vars = None
        self.optimizer_G = torch.optim.Adam([{'params': self.netG.parameters()}], lr=config.lr)
        self.scheduler_G = torch.optim.lr_scheduler.ExponentialLR(self.optimizer_G, gamma=self.config.gamma)
        self.real_A = self.fake_B = self.real_B = self.dates = self.masks = None
        self.netG.variance = None
        self.data_parallel = None      # set to a uncrtaints_amd.parallel.BucketedDataParallel(self.netG) for multi-GPU training

    def forward(self):
        self.fake_B = self.netG(self.real_A, batch_positions=self.dates)
        self.netG.variance = None

    def backward_G(self):
        self.get_loss_G()
        self.loss_G.backward()

    def get_loss_G(self):
        mi, vi = self.netG.mean_idx, self.netG.vars_idx
        if vi > mi and self.fake_B.shape[2] == vi:
            # the slices of base_model.py:72-85 as one autograd node: the loss kernels read them in place and their backward hands
            # ONE gradient buffer back to the head (losses.split_prediction)
            mean, var = losses.split_prediction(self.fake_B, mi, vi)
        else:
            mean, var = self.fake_B[:, :, :mi, ...], self.fake_B[:, :, mi:vi, ...]
        self.loss_G, self.netG.variance = losses.calc_loss(self.criterion, self.config, mean, self.real_B, var=var)

    def set_input(self, input):
        dev = self.config.device
        self.real_A = self.scale_by * input['A'].to(dev)
        self.real_B = self.scale_by * input['B'].to(dev)
        self.dates = None if input['dates'] is None else input['dates'].to(dev)
        self.masks = input['masks'].to(dev) if input.get('masks') is not None else None

    def reset_input(self):
        self.real_A = self.real_B = self.dates = self.masks = None

    def rescale(self):
        if getattr(self, 'real_A', None) is not None:
            self.real_A = 1 / self.scale_by * self.real_A
        self.real_B = 1 / self.scale_by * self.real_B
        self.fake_B = 1 / self.scale_by * self.fake_B[:, :, :S2_BANDS, ...]
        if getattr(self.netG, 'variance', None) is not None:
            self.netG.variance = 1 / self.scale_by ** 2 * self.netG.variance

    def optimize_parameters(self):
        self.forward()
        self.real_A = None
        # The reference's plain zero_grad() (base_model.py:117, set_to_none=True): a parameter without a gradient in a step is
        # skipped by Adam, and autograd hands every parameter its fresh gradient tensor instead of accumulating into an old one.
        # Under BucketedDataParallel (`self.data_parallel`, optional) the wrapper resets its bucket bookkeeping as well, packs and
        # all-reduces the buckets from its hooks, and is waited on before the optimizer step.
        dp = getattr(self, "data_parallel", None)
        if dp is not None:
            dp.zero_grad()
        else:
            self.optimizer_G.zero_grad()
        self.backward_G()
        if dp is not None:
            dp.finish()
        self.optimizer_G.step()
        self.rescale()
        self.reset_input()
        if self.netG.training and getattr(self.config, "export_to_host", False):
            self.fake_B = self.fake_B.cpu()
            if self.netG.variance is not None:
                self.netG.variance = self.netG.variance.cpu()
