"""EMA teacher with the reference's interface (aldi/ema.py:8-60).  The update over the whole
state (parameters AND buffers) is one fused HIP stream over the flat fp32 state instead of a
Python loop + load_state_dict."""
import copy


class EMA:
    def __init__(self, model, alpha, start_iter=0):
        self.model = copy.deepcopy(model)
        self.alpha = alpha
        self.start_iter = start_iter
        self.exclude_keys = ['query_embed']      # DETR query embeddings are copied, not averaged (aldi/ema.py:17,39-41)

    def _student(self, model):
        return model.module if hasattr(model, "module") else model

    def _check(self, student):
        skeys = set(student.layout.state_dict_keys())
        for key in self.model.layout.state_dict_keys():
            if key not in skeys:
                raise Exception("{} is not found in student model".format(key))
            if any(k in key for k in self.exclude_keys) and not hasattr(self.model.weights, "ranges"):
                raise NotImplementedError("excluded (copied) keys are not part of the R50-FPN layout")

    def _init_ema_weights(self, model):
        s = self._student(model)
        self._check(s)
        self.model.weights.ema_from(s.weights, self.alpha, copy_only=True)

    def _update_ema(self, model, iter):
        s = self._student(model)
        self._check(s)
        self.model.weights.ema_from(s.weights, self.alpha, copy_only=False)
        self._copy_excluded(s)

    def _copy_excluded(self, student):
        """keys matching `exclude_keys` are copied from the student instead of averaged (reference aldi/ema.py:39-41: DETR's
        query embeddings); flat-container models address them by state_dict name"""
        W = self.model.weights
        names = [k for k in self.model.layout.state_dict_keys() if any(x in k for x in self.exclude_keys)]
        if not names:
            return
        for lo, hi in W.ranges(names):
            W.master[lo:hi].copy_(student.weights.master[lo:hi])
        W.refresh()

    def update_weights(self, model, iter):
        if iter <= self.start_iter:
            self._init_ema_weights(model)
        else:
            self._update_ema(model, iter)

    def inference(self, data, **kwargs):
        return self.model.inference(data, **kwargs)

    def state_dict(self):
        return {"model." + k: v for k, v in self.model.state_dict().items()}

    def load_state_dict(self, sd):
        self.model.load_state_dict({k[len("model."):] if k.startswith("model.") else k: v for k, v in sd.items()})
