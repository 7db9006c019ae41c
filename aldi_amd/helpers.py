"""Hook objects and small helpers with the reference's names and behaviour (aldi/helpers.py:7-63)."""
import random

import torch


class SaveIO:
    """Forward hook that remembers the most recent (input, output) pair of the module it is registered on -- how the distiller
    reads the RPN / box-head tensors of student and teacher (reference aldi/helpers.py:7-15; here the engine fires it)."""
    input = None
    output = None

    def __call__(self, module, module_in, module_out):
        self.input, self.output = module_in, module_out


class ManualSeed:
    """Forward pre-hook: `torch.manual_seed(self.seed)` on the GLOBAL generator every time the hooked module runs, so that student
    and teacher draw the same ROI samples (aldi/helpers.py:17-26).  `reset_seed` takes a fresh 32-bit seed from Python's `random`."""
    SEED_RANGE = (0, 2**32 - 1)

    def __init__(self):
        self.seed = None
        self.reset_seed()

    def reset_seed(self):
        self.seed = random.randint(*self.SEED_RANGE)

    def __call__(self, module, args):
        torch.manual_seed(self.seed)


class ReplaceProposalsOnce:
    """Forward pre-hook on roi_heads: the next TRAINING-mode call sees `proposals` in place of its own third positional argument;
    the replacement is consumed by that call (aldi/helpers.py:28-42).  Eval-mode calls pass through and leave it pending."""
    def __init__(self):
        self.proposals = None

    def set_proposals(self, proposals):
        self.proposals = proposals

    def __call__(self, module, args):
        pending = self.proposals
        if pending is None or not module.training:
            return None
        self.proposals = None
        images, features, _own, gt_instances = args
        return images, features, pending, gt_instances


def set_attributes(obj, params):
    """`set_attributes(self, locals())` in a constructor: every public local becomes an attribute (aldi/helpers.py:44-49)"""
    for name, value in (params or {}).items():
        if name == "self" or name.startswith("_"):
            continue
        setattr(obj, name, value)


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


def hook_points(model):
    """every HookPoint of a detector (the places the reference registers module hooks on), nested ones included"""
    seen, todo = [], [v for v in vars(model).values() if isinstance(v, HookPoint)]
    while todo:
        p = todo.pop()
        if any(p is q for q in seen):
            continue
        seen.append(p)
        todo += [v for v in vars(p).values() if isinstance(v, HookPoint)]
    return seen


def foreign_hooks(model, own) -> bool:
    """True when a hook point of `model` carries a hook that is not one of `own` (identity): somebody outside the distiller is listening"""
    for p in hook_points(model):
        for h in list(p.pre_hooks) + list(p.hooks):
            if not any(h is o for o in own):
                return True
    return False
