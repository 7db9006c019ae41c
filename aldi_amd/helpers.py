"""Hook objects and small helpers with the reference's names and behaviour (aldi/helpers.py:7-63)."""
import random

import torch


class SaveIO:
    """Stash of a sub-module's input / output (reference: a torch forward hook; here filled by the engine)."""
    def __init__(self):
        self.input = None
        self.output = None

    def __call__(self, module, module_in, module_out):
        self.input = module_in
        self.output = module_out


class ManualSeed:
    """Forward pre-hook that re-seeds the GLOBAL torch RNG (aldi/helpers.py:17-26); seed drawn from Python's `random`."""
    def __init__(self):
        self.reset_seed()

    def reset_seed(self):
        self.seed = random.randint(0, 2**32 - 1)

    def __call__(self, module, args):
        torch.manual_seed(self.seed)


class ReplaceProposalsOnce:
    """Swap the `proposals` argument of roi_heads once, training only (aldi/helpers.py:28-42)."""
    def __init__(self):
        self.proposals = None

    def set_proposals(self, proposals):
        self.proposals = proposals

    def __call__(self, module, args):
        ret = None
        if self.proposals is not None and module.training:
            images, features, proposals, gt_instances = args
            ret = (images, features, self.proposals, gt_instances)
            self.proposals = None
        return ret


def set_attributes(obj, params):
    if params:
        for k, v in params.items():
            if k != "self" and not k.startswith("_"):
                setattr(obj, k, v)


class HookPoint:
    """Stands where the reference has an nn.Module it registers hooks on (roi_heads, rpn_head, ...)."""
    def __init__(self, owner, name):
        self.owner, self.name = owner, name
        self.pre_hooks, self.hooks = [], []

    @property
    def training(self):
        return self.owner.training

    def register_forward_pre_hook(self, hook):
        self.pre_hooks.append(hook)
        return hook

    def register_forward_hook(self, hook):
        self.hooks.append(hook)
        return hook

    def fire_pre(self, args=()):
        for h in self.pre_hooks:
            r = h(self, args)
            if r is not None:
                args = r
        return args

    def fire(self, inp, out):
        for h in self.hooks:
            h(self, inp, out)
