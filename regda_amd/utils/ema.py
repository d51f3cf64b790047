"""ExponentialMovingAverage with the reference's public surface (regda/utils/ema.py:34-65): `register()`, `update()`,
`apply_shadow()`, `restore()`, and the `shadow` / `backup` dicts keyed by parameter name.  Only parameters that require
gradients are averaged; BatchNorm buffers are not (they stay the live model's).

This class is the drop-in for scripts that drive the EMA themselves.  The fused training step (regda_amd/ssl.py) does
not use it: there the average is one more output of the optimizer kernel (rgda_sgd_step)."""
import torch


class ExponentialMovingAverage:
    def __init__(self, model, decay):
        self.model, self.decay = model, decay
        self.shadow, self.backup = {}, {}

    def _trainable(self):
        return ((n, p) for n, p in self.model.named_parameters() if p.requires_grad)

    def _resync(self):
        sync = getattr(self.model, 'sync_weights', None)       # regda_amd models mirror their weights in bf16
        if sync is not None:
            sync()

    def register(self):
        self.shadow = {n: p.detach().clone() for n, p in self._trainable()}

    @torch.no_grad()
    def update(self):
        d = self.decay
        for n, p in self._trainable():
            if n not in self.shadow:
                raise AssertionError(f'{n} was not registered')
            self.shadow[n] = (1.0 - d) * p.detach() + d * self.shadow[n]

    @torch.no_grad()
    def apply_shadow(self):
        """Swap the averaged weights in (the live ones are kept in `backup`)."""
        for n, p in self._trainable():
            if n not in self.shadow:
                raise AssertionError(f'{n} was not registered')
            self.backup[n] = p.detach().clone()
            p.copy_(self.shadow[n])             # in place: regda_amd parameters are views of one flat buffer
        self._resync()

    @torch.no_grad()
    def restore(self):
        for n, p in self._trainable():
            if n not in self.backup:
                raise AssertionError(f'{n} has no backup: apply_shadow() first')
            p.copy_(self.backup[n])
        self.backup = {}
        self._resync()
