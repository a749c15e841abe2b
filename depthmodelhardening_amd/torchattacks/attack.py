"""Base class of the attacks: same contract as the reference's ``torchattacks/attack.py``.

Reference: Attack.__init__ attack.py:14-35 (device from the model's first parameter, mode flags),
Attack.__call__ attack.py:296-320 (model.eval() -- or selective train -- around forward(), training
mode restored afterwards, optional uint8 return).
"""
import torch

from ..ops import frozen_weights


class Attack(object):
    def __init__(self, name, model):
        self.attack = name
        self.model = model
        self.model_name = str(model).split("(")[0]
        self.device = next(model.parameters()).device
        self._attack_mode = 'default'
        self._targeted = False
        self._return_type = 'float'
        self._supported_mode = ['default']
        self._model_training = False
        self._batchnorm_training = False
        self._dropout_training = False

    def forward(self, *input):
        raise NotImplementedError

    def get_mode(self):
        return self._attack_mode

    def set_mode_default(self):
        self._attack_mode = 'default'
        self._targeted = False

    def set_return_type(self, type):
        if type not in ('float', 'int'):
            raise ValueError(type + " is not a valid type. [Options: float, int]")
        self._return_type = type

    def set_training_mode(self, model_training=False, batchnorm_training=False, dropout_training=False):
        self._model_training = model_training
        self._batchnorm_training = batchnorm_training
        self._dropout_training = dropout_training

    def _to_uint(self, images):
        return (images * 255).type(torch.uint8)

    def __str__(self):
        info = {k: v for k, v in self.__dict__.items() if k not in ('model', 'attack') and k[0] != "_"}
        info['attack_mode'] = self._attack_mode
        info['return_type'] = self._return_type
        return self.attack + "(" + ', '.join('{}={}'.format(k, v) for k, v in info.items() if not torch.is_tensor(v)) + ")"

    def __call__(self, *input, **kwargs):
        given_training = self.model.training
        if self._model_training:
            self.model.train()
            for _, m in self.model.named_modules():
                if not self._batchnorm_training and 'BatchNorm' in m.__class__.__name__:
                    m.eval()
                if not self._dropout_training and 'Dropout' in m.__class__.__name__:
                    m.eval()
        else:
            self.model.eval()
        # the attack only updates the perturbation: weight-derived tensors (Winograd-transformed filters) are
        # computed once for all its steps
        with frozen_weights():
            images = self.forward(*input, **kwargs)
        if given_training:
            self.model.train()
        if self._return_type == 'int':
            images = self._to_uint(images)
        return images
