"""ExponentialMovingAverage -- mirror of regda/utils/ema.py:34-65 (shadow/backup swap over
parameters with requires_grad; BN buffers are not averaged)."""


class ExponentialMovingAverage:
    def __init__(self, model, decay):
        self.model = model
        self.decay = decay
        self.shadow = {}
        self.backup = {}

    def register(self):
        for name, param in self.model.named_parameters():
            if param.requires_grad:
                self.shadow[name] = param.data.clone()

    def update(self):
        for name, param in self.model.named_parameters():
            if param.requires_grad:
                assert name in self.shadow
                self.shadow[name] = ((1.0 - self.decay) * param.data + self.decay * self.shadow[name]).clone()

    def apply_shadow(self):
        for name, param in self.model.named_parameters():
            if param.requires_grad:
                assert name in self.shadow
                self.backup[name] = param.data.clone()
                param.data.copy_(self.shadow[name])     # in place: parameters are views of the flat buffers
        if hasattr(self.model, 'sync_weights'):
            self.model.sync_weights()

    def restore(self):
        for name, param in self.model.named_parameters():
            if param.requires_grad:
                assert name in self.backup
                param.data.copy_(self.backup[name])
        self.backup = {}
        if hasattr(self.model, 'sync_weights'):
            self.model.sync_weights()
