import torch


class ModelMixin(torch.nn.Module):
    """The part of diffusers.ModelMixin the reference's UNet relies on: nn.Module + dtype / device."""
    _supports_gradient_checkpointing = False

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device
